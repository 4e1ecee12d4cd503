/*
 * ace_sfno.h - C ABI of libace_sfno.so: the MI355X-native SFNO forward step.
 *
 * This is the drop-in boundary for the hot path of ai2cm/ace (`fme`): the per-step
 * forward of the Spherical Fourier Neural Operator inside the autoregressive stepper.
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference tree).  The reference is pure Python on torch; its "FFI" for this path is
 * the nn.Module call boundary, so a maintainer binds these functions with ctypes
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer to fp32
 *     unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls enqueue
 *     work and return; nothing synchronises unless documented;
 *   - return value 0 = success; on failure a negative code is returned and
 *     ace_last_error() holds a message (thread-local).  The library never aborts;
 *   - tensors are contiguous, row-major, in the reference's own shapes.
 */
#ifndef ACE_SFNO_H
#define ACE_SFNO_H

#ifdef __cplusplus
extern "C" {
#endif

#define ACE_OK 0
#define ACE_ERR_INVALID (-1)   /* bad argument / unsupported configuration (Python: ValueError) */
#define ACE_ERR_RUNTIME (-2)   /* HIP runtime failure (Python: RuntimeError) */
#define ACE_ERR_STATE (-3)     /* call sequence error, e.g. forward before all weights are set */

const char* ace_last_error(void);
int ace_version(void);

/* ------------------------------------------------------------------------------------------
 * Spherical harmonic transform plans.
 * Replaces RealSHT.__init__ / InverseRealSHT.__init__ (fme/sht_fix.py:69-111, 154-192): quadrature
 * nodes/weights for `grid` in {"legendre-gauss","lobatto","equiangular"}, the orthonormal
 * Condon-Shortley Legendre table (fp64 -> fp32) and the folded longitude DFT matrices.
 * lmax/mmax <= 0 select the reference defaults (lmax = nlat, or nlat-1 for lobatto;
 * mmax = nlon/2+1).  One plan serves both directions.
 * ------------------------------------------------------------------------------------------ */
typedef struct ace_sht_plan ace_sht_plan;

int ace_sht_plan_create(int nlat, int nlon, int lmax, int mmax, const char* grid, ace_sht_plan** plan);
/* Same, with the arithmetic of the Legendre stage chosen: precision 0 = exact fp32 MFMA (ace_sht_plan_create),
 * 1 = "f16x3" (error-compensated fp16 MFMA with dynamic range tracking, fp32-class accuracy; the mode the network
 * runs in by default - DESIGN.md 3.1).  Not part of the reference API (fme/sht_fix.py has one arithmetic). */
int ace_sht_plan_create_ex(int nlat, int nlon, int lmax, int mmax, const char* grid, int precision, ace_sht_plan** plan);
void ace_sht_plan_destroy(ace_sht_plan* plan);
int ace_sht_plan_dims(const ace_sht_plan* plan, int* nlat, int* nlon, int* lmax, int* mmax);
/* Which kernel family the plan's LAST Legendre launch of each direction took (test instrumentation: parity tests assert that the
 * launch they check really ran the kernel they mean to check): 0 tile engine, 1 register-resident strip kernel, 2 equatorially
 * folded strip kernel, 3 its big form (more than 96 folded latitudes: the 0.25-degree grid); -1 = no launch yet. */
int ace_sht_plan_route(const ace_sht_plan* plan, int* forward, int* inverse);

/* RealSHT.forward (fme/sht_fix.py:119-139): x (n, nlat, nlon) f32 -> coeffs (n, lmax, mmax) complex64
 * stored interleaved (re, im) as 2*n*lmax*mmax floats.  May grow plan-owned scratch (hipMalloc) on
 * the first call at a given n; not capture-safe on that first call. */
int ace_sht_forward(ace_sht_plan* plan, const float* x, float* coeffs, int n, void* stream);

/* InverseRealSHT.forward (fme/sht_fix.py:202-226): coeffs (n, lmax, mmax) complex64 -> x (n, nlat, nlon). */
int ace_sht_inverse(ace_sht_plan* plan, const float* coeffs, float* x, int n, void* stream);

/* Host-only (no GPU touched): build the fp32 tables into caller buffers for inspection.
 * which: 0 = forward Legendre*quadrature wt[m][l][k] (dense, mmax*lmax*nlat floats)
 *        1 = inverse Legendre pct[m][l][k]           (dense, mmax*lmax*nlat floats)
 *        2 = quadrature nodes cos(theta) ascending (nlat doubles), 3 = quadrature weights (nlat doubles) */
int ace_sht_tables_host(int nlat, int nlon, int lmax, int mmax, const char* grid, int which, void* out_host);

/* ------------------------------------------------------------------------------------------
 * Building blocks, exported for parity tests and micro-benchmarks.
 * ------------------------------------------------------------------------------------------ */

/* nn.Conv2d(Cin, Cout, 1) [+ activation] on (n, Cin, hw) -> (n, Cout, hw)  (sfnonet.py:229, layers.py:117-124).
 * weight (Cout, Cin) row-major, bias (Cout) or NULL.  act: 0 none, 1 GELU(erf), 2 ReLU, 3 SiLU. */
int ace_conv1x1(const float* x, const float* weight, const float* bias, float* y, int n, int cin, int cout,
                long hw, int act, void* stream);

/* Same contraction on the compensated-fp16 engine ("f16x3": both operands split hi+lo fp16, three exact-product
 * MFMAs, fp32 accumulation).  Test / micro-benchmark entry: prepares the weight planes on every call and synchronises. */
int ace_conv1x1_f16x3(const float* x, const float* weight, const float* bias, float* y, int n, int cin, int cout,
                      long hw, int act, void* stream);

/* MLP.forward (fme/ace/models/modulus/layers.py:97-137): y = W2 act(W1 x + b1) + b2 on (n, cin, hw) -> (n, cout, hw),
 * on the packed-operand f16x3 engine (the hidden activation is produced and consumed as pre-split fp16 planes and
 * never exists in fp32).  Needs cin % 8 == 0, hid % 8 == 0, hw % 4 == 0.  Test / micro-benchmark entry: prepares the
 * weight planes on every call and synchronises. */
int ace_mlp_f16x3(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, int n,
                  int cin, int hid, int cout, long hw, int act, void* stream);

/* _contract_dhconv (fme/ace/models/modulus/contractions.py:183-195): out[b,o,l,m] = sum_i coeffs[b,i,l,m] * weight[i,o,l]
 * on complex values; coeffs / out (n, c, L, Mm) complex64 interleaved, weight (c, c, L, 2).  Runs the network's own kernel
 * (compensated fp16 MFMA, filter streamed once; entries with m > l are not contracted and come out zero, as they are zero
 * in every SHT output).  Needs c % 128 == 0.  Test / micro-benchmark entry: prepares the filter planes on every call and
 * synchronises. */
int ace_dhconv_f16x3(const float* coeffs, const float* weight, float* out, int n, int c, int L, int Mm, void* stream);

/* nn.InstanceNorm2d(C, eps, affine) (sfnonet.py:593-601) on (n, C, hw); gamma/beta may be NULL. */
int ace_instance_norm(const float* x, const float* gamma, const float* beta, float eps, float* y, int n, int c,
                      long hw, void* stream);

/* ConditionalLayerNorm.forward restricted to noise conditioning (fme/core/models/conditional_sfno/layers.py:95-141,
 * 245-318): per-PIXEL layer norm over the c channels (biased variance, eps inside the sqrt, optional elementwise
 * gamma/beta (c)), then y = y_norm * (1 + W_scale noise) + W_bias noise with the 1x1 convolutions W_scale, W_bias
 * (c, noise_dim), noise (n, noise_dim, hw).  w_scale = w_bias = NULL: plain ChannelLayerNorm.  Any hw (16-byte accesses when hw % 4 == 0).
 * Test / building-block entry: allocates its statistics workspace and synchronises. */
int ace_conditional_layer_norm(const float* x, const float* noise, const float* gamma, const float* beta,
                               const float* w_scale, const float* w_bias, float eps, float* y, int n, int c,
                               int noise_dim, long hw, void* stream);

/* The same operator as the f16x3 NoiseConditionedSFNO runs it: ONE pass per 32-pixel tile - fp64 statistics, the two
 * conditioning convolutions on the matrix cores with error-compensated fp16 operands, apply - instead of a statistics and an
 * apply pass.  Needs c % 256 == 0 (c <= 1024), hw % 4 == 0 (c > 512: hw % 32 == 0), noise_dim <= 128; any other shape is ACE_ERR_INVALID (no fallback:
 * the network routes such shapes to the two-pass form itself).  Test / building-block entry: packs the weights, synchronises. */
int ace_conditional_layer_norm_f16x3(const float* x, const float* noise, const float* gamma, const float* beta,
                                     const float* w_scale, const float* w_bias, float eps, float* y, int n, int c,
                                     int noise_dim, long hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * The network.  Replaces SphericalFourierNeuralOperatorNet.__init__/forward
 * (fme/ace/models/modulus/sfnonet.py:341-685, 713-749) as built by
 * SphericalFourierNeuralOperatorBuilder.build (fme/ace/registry/sfno.py:44-61).
 * ------------------------------------------------------------------------------------------ */
typedef struct ace_sfno ace_sfno;

typedef struct ace_sfno_config {
    int in_chans, out_chans;      /* n_in_channels, n_out_channels of ModuleConfig.build */
    int nlat, nlon;               /* dataset_info.img_shape */
    int embed_dim, num_layers;
    int scale_factor;             /* >= 1: the blocks between the first filter's inverse transform and the last one's forward transform
                                     work on the (nlat / sf) x (nlon / sf) Gauss-Legendre grid (sfnonet.py:467-515) */
    float hard_thresholding_fraction;
    int operator_type;            /* 0 = "diagonal", 1 = "dhconv" */
    int normalization_layer;      /* 0 = "none", 1 = "instance_norm", 2 = conditional layer norm (NoiseConditionedSFNO),
                                     3 = "layer_norm": nn.LayerNorm over (nlat, nlon), norm0 / norm1 weight and bias are (nlat, nlon) fields
                                     (sfnonet.py:584-592) */
    int activation_function;      /* 1 = "gelu", 2 = "relu", 3 = "silu" */
    int use_mlp;
    float mlp_ratio;
    int encoder_layers;
    int pos_embed, big_skip;
    int data_grid;                /* 0 = "legendre-gauss", 2 = "equiangular" */
    int max_batch;                /* workspace is sized for this many samples (B = samples x ensemble) */
    int precision;                /* 0 = exact fp32 MFMA everywhere (the reference's arithmetic);
                                     1 = "f16x3": 1x1 convolutions on compensated fp16 MFMA (hi/lo split of both
                                     operands, 3 exact-product MFMAs, fp32 accumulate): fp32-class accuracy */
    /* NoiseConditionedSFNO (fme/ace/registry/stochastic_sfno.py:266-397; normalization_layer == 2) */
    int noise_embed_dim;          /* channels of the conditioning noise */
    int affine_norms;             /* elementwise gamma/beta in the layer norms */
    int normalize_big_skip;       /* conditional layer norm on the big-skip input */
    int filter_num_groups;        /* groups of the spectral filter (weight (G, L, C/G, C/G, 2)); >= 1 */
    int residual_filter_factor;   /* 0 / 1: none; r > 1: the big skip's input is band-limited on the data grid to lmax = nlat / r,
                                     mmax = nlon / r / 2 + 1 (sfnonet.py:473-497, 715-716) */
} ace_sfno_config;

int ace_sfno_create(const ace_sfno_config* cfg, ace_sfno** net);
void ace_sfno_destroy(ace_sfno* net);

/* Upload one parameter by its reference state_dict name: "pos_embed", "encoder.0.weight",
 * "blocks.3.filter.filter.weight", ...  `src` is a device pointer to `numel` contiguous floats in the
 * reference's shape; the library keeps its own copy (re-laid-out where a kernel wants it), so the
 * caller may free or update `src` afterwards and must call this again after an update.
 * Synchronises `stream` before returning. */
int ace_sfno_set_weight(ace_sfno* net, const char* name, const float* src, long numel, void* stream);

/* Monotonic counter bumped by every ace_sfno_set_weight: anything a caller captured around forwards of this handle (its
 * own hipGraph of a whole rollout window) is stale once the value differs from the one seen at capture time. */
long ace_sfno_weights_generation(const ace_sfno* net);

/* SURVEY 8(b) workspace_size(handle, B): device bytes the library owns for this handle when running batches up to
 * `batch` <= max_batch - activations workspace (sized by max_batch at creation), both operand forms of the weights,
 * SHT tables.  Everything else (input, output, caller tensors) is caller-owned.  Returns -1 on a bad argument. */
long ace_sfno_workspace_size(const ace_sfno* net, int batch);

/* Number of parameters / name of parameter i / its numel, in the reference's state_dict order. */
int ace_sfno_num_weights(const ace_sfno* net);
const char* ace_sfno_weight_name(const ace_sfno* net, int i);
long ace_sfno_weight_numel(const ace_sfno* net, int i);

/* Module.__call__ (fme/core/registry/module.py:74-86): in (batch, in_chans, nlat, nlon) ->
 * out (batch, out_chans, nlat, nlon).  No allocation, no host synchronisation: capture-safe. */
int ace_sfno_forward(ace_sfno* net, const float* in, float* out, int batch, void* stream);

/* NoiseConditionedModel.forward -> conditional SphericalFourierNeuralOperatorNet.forward
 * (fme/ace/registry/stochastic_sfno.py:128-172, fme/core/models/conditional_sfno/sfnonet.py:770-824) for a net created
 * with normalization_layer == 2: `noise` is the (batch, noise_embed_dim, nlat, nlon) conditioning field on the device
 * (the host side draws it - gaussian, or isotropic through ace_sht_inverse - as the reference does).  Parameters use
 * the conditional model's state_dict names without the "conditional_model." prefix.  Same stream / allocation rules as
 * ace_sfno_forward; ace_sfno_forward itself fails for such a net. */
int ace_sfno_forward_conditioned(ace_sfno* net, const float* in, const float* noise, float* out, int batch, void* stream);
/* ... with the per-stage hipEvent timing of ace_sfno_forward_timed (synchronises). */
int ace_sfno_forward_conditioned_timed(ace_sfno* net, const float* in, const float* noise, float* out, int batch,
                                       void* stream, float* ms_per_stage, int* calls_per_stage);

/* Measurement: ace_sfno_forward with a hipEvent after every launch group on `stream` (the reference's
 * CUDATimer children, fme/core/benchmark/timer.py:105-168; block children conditional_sfno/sfnonet.py:388-437,
 * filter children s2convolutions.py:372-431).  Synchronises `stream`.  ms_host[ace_sfno_num_stages()] receives the
 * milliseconds per stage summed over blocks, calls_host (optional) the number of launch groups per stage. */
int ace_sfno_num_stages(void);
/* ace_sht_plan_route of the network's internal (legendre-gauss) plan after a forward. */
int ace_sfno_sht_route(const ace_sfno* net, int* forward, int* inverse);
const char* ace_sfno_stage_name(int i);
int ace_sfno_forward_timed(ace_sfno* net, const float* in, float* out, int batch, void* stream, float* ms_host,
                           int* calls_host);

/* Debug/test tap: copy the activation after block `i` (batch, embed_dim, nlat, nlon) of the LAST
 * forward into dst.  i = -1: the encoder output (after pos_embed). Only valid with ace_sfno_set_taps(net, 1). */
int ace_sfno_set_taps(ace_sfno* net, int enable);
int ace_sfno_get_tap(ace_sfno* net, int i, float* dst, int batch, void* stream);

/* hipGraph path: captures ace_sfno_forward(in, out, batch) once per distinct (in, out, batch) and
 * replays it on `stream` afterwards (the autoregressive loop with static buffers). */
int ace_sfno_forward_graph(ace_sfno* net, const float* in, float* out, int batch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stepper glue (fme/core/packer.py:45-52 + fme/core/normalizer.py:213-236, fused).
 * srcs/dsts: DEVICE arrays of nch device pointers; strides: DEVICE array of per-sample strides (floats).
 *   pack:   dst[b][j][:] = (srcs[j][b*strides[j] + :] - mean[j]) / std[j]
 *   unpack: dsts[j][b*strides[j] + :] = src[b][j][:] * std[j] + mean[j]
 * ------------------------------------------------------------------------------------------ */
int ace_pack_normalize(const float* const* srcs, const long* strides, const float* mean, const float* std_,
                       float* dst, int batch, int nch, long hw, void* stream);
int ace_unpack_denormalize(const float* src, const float* mean, const float* std_, float* const* dsts,
                           const long* strides, int batch, int nch, long hw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Post-step physics (fme/core/step/single_module.py:669-716): the AtmosphereCorrector
 * (fme/core/corrector/atmosphere.py:349-398 order, 404-700 corrections), the prescribed-SST Ocean
 * (fme/core/ocean.py:167-222, fme/core/prescriber.py:54-117) and the prescribed prognostics, applied in place on the
 * denormalised output planes of one step, in the reference's order.  Four kernel launches per step (the corrections are
 * a chain of area-weighted global means), deterministic fp64 reductions, no allocation, no host synchronisation:
 * capture-safe.  A plane is a (batch, nlat, nlon) fp32 field: device pointer + per-sample stride in floats; p = NULL
 * means "absent".
 * ------------------------------------------------------------------------------------------ */
#define ACE_PHYS_MAX_LEVELS 16
#define ACE_PHYS_MAX_POSITIVE 32
#define ACE_PHYS_MAX_PRESCRIBED 8

typedef struct ace_phys_config {
    int nlat, nlon;
    int nlev;                          /* vertical layers (ak / bk have nlev + 1 entries) */
    double timestep_seconds;
    int conserve_dry_air;              /* AtmosphereCorrectorConfig.conserve_dry_air */
    int zero_global_mean_moisture_advection;
    int moisture_budget;               /* 0 none, 1 "precipitation", 2 "evaporation", 3 "advection_and_precipitation",
                                          4 "advection_and_evaporation" */
    int clip_frozen_precipitation;
    int energy_budget;                 /* 0 none, 1 "constant_temperature" */
    double unaccounted_heating;        /* EnergyBudgetConfig.constant_unaccounted_heating */
    int ocean;                         /* 0 none, 1 prescribed SST where round(ocean fraction) == 1, 2 interpolate */
    int max_batch;
} ace_phys_config;

typedef struct ace_phys_plane { float* p; long stride; } ace_phys_plane;

typedef struct ace_phys_fields {
    /* output of the step (denormalised), corrected in place; names: fme/core/atmosphere_data.py:18-43 */
    ace_phys_plane ps;                             /* surface_pressure */
    ace_phys_plane wat[ACE_PHYS_MAX_LEVELS];       /* specific_total_water_k */
    ace_phys_plane T[ACE_PHYS_MAX_LEVELS];         /* air_temperature_k */
    ace_phys_plane adv;                            /* tendency_of_total_water_path_due_to_advection */
    ace_phys_plane precip, lhf, shf;               /* precipitation_rate, latent / sensible heat flux */
    ace_phys_plane dswsfc, uswsfc, dlwsfc, ulwsfc, ulwtoa, uswtoa;
    ace_phys_plane frozen;                         /* total_frozen_precipitation_rate, or ... */
    ace_phys_plane frozen_parts[3];                /* ... ICEsfc, GRAUPELsfc, SNOWsfc (summed); all absent: zero */
    ace_phys_plane positive[ACE_PHYS_MAX_POSITIVE];   /* force_positive_names */
    int npositive;
    /* input of the step */
    ace_phys_plane ps_in;
    ace_phys_plane wat_in[ACE_PHYS_MAX_LEVELS];
    ace_phys_plane T_in[ACE_PHYS_MAX_LEVELS];
    ace_phys_plane hgt_in;                         /* surface height (or geopotential) of the input */
    /* next step's forcing / target data */
    ace_phys_plane hgt_next, dswtoa_next;
    float hgt_in_scale, hgt_next_scale;   /* 1, or 1 / 9.80616 when the field is the surface geopotential (PHIS) */
    ace_phys_plane sst, sst_target, ocean_fraction;   /* output SST, next-step SST, next-step ocean fraction */
    ace_phys_plane prescribed_dst[ACE_PHYS_MAX_PRESCRIBED], prescribed_src[ACE_PHYS_MAX_PRESCRIBED];
    int nprescribed;
} ace_phys_fields;

typedef struct ace_physics ace_physics;
const char* ace_physics_last_error(void);
/* area_weights_lat_host: nlat fp32 weights of one grid column (fme/core/metrics.py:14-32, longitudinally uniform);
 * ak_host / bk_host: nlev + 1 fp32 hybrid-sigma interface coefficients (fme/core/coordinates.py:150-280).  Either may be
 * NULL when no configured correction needs it. */
int ace_physics_create(const ace_phys_config* cfg, const float* area_weights_lat_host, const float* ak_host,
                       const float* bk_host, ace_physics** out);
void ace_physics_destroy(ace_physics* phys);
/* A new initial condition: the next ace_physics_apply seeds the dry-air reference mass from its input again
 * (CorrectorState, fme/core/corrector/state.py).  Stream ordered. */
int ace_physics_reset(ace_physics* phys, void* stream);
/* Carry the reference mass across windows: (batch) fp64 on the device.  Both synchronise `stream`. */
int ace_physics_set_reference(ace_physics* phys, const double* ref_dev, int batch, void* stream);
int ace_physics_get_reference(ace_physics* phys, double* ref_dev, int* have_host, int batch, void* stream);
/* One step.  `fields` is a HOST struct of device planes (copied into the kernel arguments). */
int ace_physics_apply(ace_physics* phys, const ace_phys_fields* fields, int batch, void* stream);

/* ------------------------------------------------------------------------------------------
 * HEALPix variant (BASELINE configs[4]): the operators of the reference's HEALPix UNet (fme/ace/models/healpix/) on
 * the 12-face mesh.  Activations of one UNet level are [image = item * 12 + face][channel][row][pitch] fp32 with a row
 * pitch >= the face width, a multiple of 4 (the pitch of that level's padded faces, so that a shifted view of a padded
 * tensor is a plain GEMM operand); gap columns [width, pitch) hold defined values (zeros or finite results).  Every tensor
 * has a "bound slot": 64 unsigned words whose maximum is the bit pattern of a bound on max|x| - zeroed by the caller,
 * written by the producing call, read by the consuming convolution (compensated-fp16 arithmetic, fp32 accumulation: the
 * same fp32-class mode as the SFNO path).  No allocation or host synchronisation outside ace_hpx_weight_create.
 * ------------------------------------------------------------------------------------------ */
#define ACE_HPX_SLACK_FLOATS 16   /* floats a caller keeps allocated behind a padded tensor (zeroed by ace_hpx_pad) */
const char* ace_hpx_last_error(void);
/* HEALPixPadding (healpix_paddings.py:239-611, Karlbauer et al.; "earth2grid" gives the same result): for every cell of
 * the padded mesh [12][nside + 2p][nside + 2p] the two source cells of the unpadded mesh, packed face << 24 | row << 12 |
 * column; padded = 0.5 a + 0.5 b (b == a: plain copy).  Host only. */
int ace_hpx_pad_table_host(int nside, int p, int* idx_a_host, int* idx_b_host);
/* y[item * 12 + face][c0 + ch][m][y_pitch] (m = nside + 2p rows and valid columns, gap columns zero) from
 * x[image][ch][row][x_pitch] by the table (device copies of idx_a / idx_b).  Two calls with different c0 concatenate two
 * sources along the channels (decoder skip connections); the call that writes the last channels also zeroes
 * ACE_HPX_SLACK_FLOATS behind the tensor.  amax (optional): bound slot of y (accumulates over the calls). */
int ace_hpx_pad(const float* x, long x_img_stride, long x_chan_stride, int x_pitch, float* y, int y_chans, int c0, int c,
                const int* idx_a_dev, const int* idx_b_dev, int items, int nside, int p, int y_pitch, unsigned* amax, void* stream);
/* bound slot of a tensor no ace_hpx_* call produced (the network input): amax[64] (zeroed by the caller) <- bits(max|x|). */
int ace_hpx_absmax(const float* x, long n, unsigned* amax, void* stream);
/* A convolution weight prepared for the compensated-fp16 engine: [rows][cols] row-major fp32 on the device ->
 * fp16 hi / lo planes under a power-of-two scale.  Synchronises the stream (once per parameter version). */
typedef struct ace_hpx_weight ace_hpx_weight;
int ace_hpx_weight_create(const float* w_dev, int rows, int cols, void* stream, ace_hpx_weight** out);
void ace_hpx_weight_destroy(ace_hpx_weight* w);
/* nn.Conv2d(k, dilation, padding 0) on already padded faces (+ bias, + residual, activation 0 none / 1 GELU(erf) / 2 ReLU,
 * clamped from above by `cap`: CappedGELU healpix_activations.py:41-85; cap = +inf: none) as ONE contraction over
 * (tap, channel).  x: [imgs][cin][H + (k-1) dil][pitch]; optional second source x2 (channels cin .. cin + cin2 - 1) and
 * residual R with k = 1 only; w: prepared from [cout][(ky k + kx) (cin + cin2) + i]; row_off (k > 1, device): element
 * offset of contraction row (tap, i) = ky dil pitch + kx dil + i (H + (k-1) dil) pitch; R / y: [imgs][cout][H][pitch].
 * xmax / x2max: bound slots of the sources; ymax (optional): bound slot of y. */
int ace_hpx_conv(const float* x, const float* x2, int cin, int cin2, const ace_hpx_weight* w, const long* row_off, const float* bias,
                 const float* R, float* y, int imgs, int cout, int H, int W, int pitch, int k, int dil, int act, float cap,
                 const unsigned* xmax, const unsigned* x2max, unsigned* ymax, void* stream);
/* The k x k (k >= 2) convolutions on the packed-operand engine: ace_hpx_pad_planes does the face padding of x (and, for the skip
 * concatenation, x2 behind it) straight into the engine's activation format - fp16 hi / lo planes [imgs][cpad / 8][(nside + 2 p) x
 * y_pitch cells][8 channels], cpad = channels rounded up to 8, scaled by the bound it publishes to pmax (+ ACE_HPX_SLACK_FLOATS
 * zero ENTRIES of 16 bytes behind each of hi, lo, kept allocated by the caller) - and ace_hpx_conv_packed contracts it with a weight
 * prepared from [cout][(ky k + kx) cpad + i] (zero columns for i >= channels).  Same arithmetic as ace_hpx_conv (compensated fp16,
 * fp32 accumulation), results equal to rounding; both operands stream by LDS-DMA. */
int ace_hpx_pad_planes(const float* x, long x_img_stride, long x_chan_stride, int x_pitch, const float* x2, long x2_img_stride,
                       long x2_chan_stride, int x2_pitch, int cin, int cin2, void* hi, void* lo, const int* idx_a_dev, const int* idx_b_dev,
                       int items, int nside, int p, int y_pitch, const unsigned* xmax, const unsigned* x2max, unsigned* pmax, void* stream);
int ace_hpx_conv_packed(const void* xhi, const void* xlo, int cpad, long x_plane_cells, const ace_hpx_weight* w, const float* bias,
                        float bias_max, float* y, void* yhi, void* ylo, long y_plane_cells, int imgs, int cout, int H, int W, int pitch, int k,
                        int dil, int act, float cap, const unsigned* pmax, unsigned* ymax, void* stream);
/* ... whose result may (also / instead: y NULL) be written in the same format - yhi / ylo: [imgs][cout / 8][H pitch][8], cout % 8 == 0,
 * scaled by the bound winf max|x| + bias_max (bias_max = max |bias|; <= cap for a capped activation) published to ymax - for a 1 x 1
 * convolution that follows (ConvNeXt: 3 x 3 -> GELU -> 1 x 1): ace_hpx_conv1_packed reads it (+ bias, + residual R, activation without a
 * cap), so the widest activation of the block never exists in fp32.
 * x_plane_cells / y_plane_cells (0: the natural sizes): entries per channel-group plane when xhi / yhi point at a shifted origin inside
 * the planes of a larger padded tensor: k = 1 reading the interior of planes padded for a k x k convolution (the ConvNeXt skip
 * convolution shares the block's padded input), or a k x k result written into the interior of the NEXT convolution's padded planes,
 * whose halo ace_hpx_halo_planes then gathers in place from the neighbouring faces' interiors (no second padding pass). */
int ace_hpx_halo_planes(void* hi, void* lo, int cpad, const int* idx_a_dev, const int* idx_b_dev, int items, int nside, int p, int y_pitch,
                        void* stream);
int ace_hpx_conv1_packed(const void* xhi, const void* xlo, int cin, const ace_hpx_weight* w, const float* bias, const float* R, float* y,
                         int imgs, int cout, int H, int W, int pitch, int act, const unsigned* xslot, unsigned* ymax, void* stream);
/* nn.AvgPool2d(2) / nn.MaxPool2d(2) on `planes` = imgs * channels planes (the input's bound also bounds the result). */
int ace_hpx_pool2(const float* x, float* y, long planes, int H, int W, int pitch_in, long plane_stride_in, int pitch_out,
                  long plane_stride_out, int is_max, void* stream);
/* nn.Upsample(scale_factor=2, mode) on `planes` planes (healpix_blocks.py:197-252 "Interpolate", 699-759 SmoothedInterpolate's resize):
 * mode 0 "nearest", 1 "bilinear" (align_corners as torch defines it).  y: [planes][2 H][pitch_out].  The input's bound also bounds the result. */
int ace_hpx_upsample2(const float* x, float* y, long planes, int H, int W, int pitch_in, long plane_stride_in, int pitch_out,
                      long plane_stride_out, int mode, int align_corners, void* stream);
/* nn.ConvTranspose2d(cin, cout, 2, stride 2) + activation (healpix_blocks.py:636-697).  w: prepared from
 * [(dy 2 + dx) cout + o][cin]; tmp: imgs * 4 cout * H * pitch_in floats of scratch; y: [imgs][cout][2 H][pitch_out]. */
int ace_hpx_tconv2(const float* x, const ace_hpx_weight* w, const float* bias, float* tmp, float* y, int imgs, int cin, int cout, int H,
                   int W, int pitch_in, int pitch_out, long plane_stride_out, int act, float cap, const unsigned* xmax, unsigned* ymax,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACE_SFNO_H */
