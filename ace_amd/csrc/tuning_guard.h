// The kernels' compile-time tuning / ablation / trace macros (each `#ifndef X / #define X <shipped value>` in the .hip files) exist for
// same-box A/B builds (tools/mkvar.sh NAME -DX=...).  The SHIPPED library is compiled with none of them set: one instruction stream
// per kernel, the one the test suite runs.  A build that sets any of them must say so with -DACE_MEASUREMENT_SWITCHES (mkvar.sh adds
// it), which also enables the historical engine switches of capi.hip; such a library never lands at ace_amd/libace_sfno.so
// (ace_amd/build.py refuses extra flags for the default output).
#pragma once
#if !defined(ACE_MEASUREMENT_SWITCHES)
#if defined(ACE_WL_ABL) || defined(ACE_WL_D) || defined(ACE_WL_WAVES) || defined(ACE_WS_ABL) || defined(ACE_WS_ACC2) || defined(ACE_WS_FD4) || \
    defined(ACE_WS_FINE) || defined(ACE_WS_HOLD4) || defined(ACE_WS_VSPAN) || defined(ACE_WS_REARLY) || defined(ACE_WS_MINMAX) || defined(ACE_FFT_ABL) || defined(ACE_FFT_FWD_WAVES) || defined(ACE_FFT_FWD_PLN_WAVES) ||            \
    defined(ACE_FFT_INV_ROWS) || defined(ACE_FFT_INV_WAVES) || defined(ACE_FFT_QROWS) || defined(ACE_FFT_ROWS) || defined(ACE_FFT_XCD) ||       \
    defined(ACE_FFT_XCD_INV) || defined(ACE_G4_PIN) || defined(ACE_EXP_NOLDSREAD) || defined(ACE_EXP_NOBARRIER) || defined(ACE_EXP_NOGLOBAL) || \
    defined(ACE_EXP_NOLDSWRITE) || defined(ACE_G3_W128_PCT) || defined(ACE_G4_PFD) || defined(ACE_G4_REGEPI) || defined(ACE_G4_RTOUCH) ||       \
    defined(ACE_GEMM2_BK) || defined(ACE_LB) || defined(ACE_PF) || defined(ACE_MLP_ABL) || defined(ACE_MLP_FDEPTH) || defined(ACE_X_TRACE) ||    \
    defined(ACE_DH_TRACE) || defined(ACE_DH_TRACE_WG) || defined(ACE_DH_PB) || defined(ACE_DH_WGS) || defined(ACE_DH_STRIPS) || defined(ACE_DH_ORDER) || defined(ACE_LF_TRACE) || defined(ACE_DEBUG_WS)
#error "tuning / ablation / trace macro set without -DACE_MEASUREMENT_SWITCHES: the shipped library is built with the defaults only (tools/mkvar.sh)"
#endif
#endif
