"""ctypes binding of libace_sfno.so (include/ace_sfno.h).

The product path has no CPU or eager-torch fallback: if the HIP library is
missing this module raises, loudly, at first use."""

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_long, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACE_SFNO_LIB") or os.path.join(HERE, "libace_sfno.so")   # ACE_SFNO_LIB: A/B a variant build

ACE_OK, ACE_ERR_INVALID, ACE_ERR_RUNTIME, ACE_ERR_STATE = 0, -1, -2, -3


class AceSfnoConfig(ctypes.Structure):
    """struct ace_sfno_config (include/ace_sfno.h)."""

    _fields_ = [
        ("in_chans", c_int), ("out_chans", c_int), ("nlat", c_int), ("nlon", c_int),
        ("embed_dim", c_int), ("num_layers", c_int), ("scale_factor", c_int),
        ("hard_thresholding_fraction", c_float), ("operator_type", c_int),
        ("normalization_layer", c_int), ("activation_function", c_int), ("use_mlp", c_int),
        ("mlp_ratio", c_float), ("encoder_layers", c_int), ("pos_embed", c_int), ("big_skip", c_int),
        ("data_grid", c_int), ("max_batch", c_int), ("precision", c_int),
        ("noise_embed_dim", c_int), ("affine_norms", c_int), ("normalize_big_skip", c_int), ("filter_num_groups", c_int),
        ("residual_filter_factor", c_int),
    ]


# ---- post-step physics (ace_physics_*): struct ace_phys_plane / ace_phys_config / ace_phys_fields
MAX_LEVELS, MAX_POSITIVE, MAX_PRESCRIBED = 16, 32, 8


class Plane(ctypes.Structure):
    _fields_ = [("p", c_void_p), ("stride", c_long)]


class PhysConfig(ctypes.Structure):
    """struct ace_phys_config"""
    _fields_ = [("nlat", c_int), ("nlon", c_int), ("nlev", c_int), ("timestep_seconds", c_double),
                ("conserve_dry_air", c_int), ("zero_global_mean_moisture_advection", c_int), ("moisture_budget", c_int),
                ("clip_frozen_precipitation", c_int), ("energy_budget", c_int), ("unaccounted_heating", c_double),
                ("ocean", c_int), ("max_batch", c_int)]


class PhysFields(ctypes.Structure):
    """struct ace_phys_fields"""
    _fields_ = [("ps", Plane), ("wat", Plane * MAX_LEVELS), ("T", Plane * MAX_LEVELS), ("adv", Plane), ("precip", Plane),
                ("lhf", Plane), ("shf", Plane), ("dswsfc", Plane), ("uswsfc", Plane), ("dlwsfc", Plane), ("ulwsfc", Plane),
                ("ulwtoa", Plane), ("uswtoa", Plane), ("frozen", Plane), ("frozen_parts", Plane * 3),
                ("positive", Plane * MAX_POSITIVE), ("npositive", c_int),
                ("ps_in", Plane), ("wat_in", Plane * MAX_LEVELS), ("T_in", Plane * MAX_LEVELS), ("hgt_in", Plane),
                ("hgt_next", Plane), ("dswtoa_next", Plane), ("hgt_in_scale", c_float), ("hgt_next_scale", c_float),
                ("sst", Plane), ("sst_target", Plane), ("ocean_fraction", Plane),
                ("prescribed_dst", Plane * MAX_PRESCRIBED), ("prescribed_src", Plane * MAX_PRESCRIBED), ("nprescribed", c_int)]



# every symbol include/ace_sfno.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ace_last_error": (c_char_p, []),
    "ace_version": (c_int, []),
    "ace_sht_plan_create": (c_int, [c_int, c_int, c_int, c_int, c_char_p, POINTER(c_void_p)]),
    "ace_sht_plan_create_ex": (c_int, [c_int, c_int, c_int, c_int, c_char_p, c_int, POINTER(c_void_p)]),
    "ace_sht_plan_destroy": (None, [c_void_p]),
    "ace_sht_plan_dims": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "ace_sht_plan_route": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "ace_sht_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ace_sht_inverse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ace_sht_tables_host": (c_int, [c_int, c_int, c_int, c_int, c_char_p, c_int, c_void_p]),
    "ace_conv1x1": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_void_p]),
    "ace_conv1x1_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_void_p]),
    "ace_mlp_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_long, c_int, c_void_p]),
    "ace_dhconv_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ace_instance_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_long, c_void_p]),
    "ace_conditional_layer_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                           c_int, c_int, c_int, c_long, c_void_p]),
    "ace_conditional_layer_norm_f16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                                 c_int, c_int, c_int, c_long, c_void_p]),
    "ace_sfno_create": (c_int, [POINTER(AceSfnoConfig), POINTER(c_void_p)]),
    "ace_sfno_destroy": (None, [c_void_p]),
    "ace_sfno_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_long, c_void_p]),
    "ace_sfno_weights_generation": (c_long, [c_void_p]),
    "ace_sfno_workspace_size": (c_long, [c_void_p, c_int]),
    "ace_sfno_num_weights": (c_int, [c_void_p]),
    "ace_sfno_weight_name": (c_char_p, [c_void_p, c_int]),
    "ace_sfno_weight_numel": (c_long, [c_void_p, c_int]),
    "ace_sfno_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ace_sfno_forward_conditioned": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ace_sfno_forward_conditioned_timed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ace_sfno_sht_route": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "ace_sfno_num_stages": (c_int, []),
    "ace_sfno_stage_name": (c_char_p, [c_int]),
    "ace_sfno_forward_timed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, POINTER(c_float), POINTER(c_int)]),
    "ace_sfno_set_taps": (c_int, [c_void_p, c_int]),
    "ace_sfno_get_tap": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "ace_sfno_forward_graph": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ace_pack_normalize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p]),
    "ace_unpack_denormalize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_long, c_void_p]),
    "ace_hpx_last_error": (c_char_p, []),
    "ace_hpx_pad_table_host": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "ace_hpx_pad": (c_int, [c_void_p, c_long, c_long, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            c_void_p, c_void_p]),
    "ace_hpx_absmax": (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    "ace_hpx_weight_create": (c_int, [c_void_p, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "ace_hpx_weight_destroy": (None, [c_void_p]),
    # x, x2, cin, cin2, w, row_off, bias, R, y, imgs, cout, H, W, pitch, k, dil, act, cap, xmax, x2max, ymax, stream
    "ace_hpx_conv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ace_hpx_pad_planes": (c_int, [c_void_p, c_long, c_long, c_int, c_void_p, c_long, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ace_hpx_conv_packed": (c_int, [c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_long, c_int,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ace_hpx_halo_planes": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ace_hpx_conv1_packed": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_void_p, c_void_p]),
    "ace_hpx_pool2": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_long, c_int, c_long, c_int, c_void_p]),
    "ace_hpx_upsample2": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_long, c_int, c_long, c_int, c_int, c_void_p]),
    # x, w, bias, tmp, y, imgs, cin, cout, H, W, pitch_in, pitch_out, plane_stride_out, act, cap, xmax, ymax, stream
    "ace_hpx_tconv2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_long,
                               c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ace_physics_last_error": (c_char_p, []),
    "ace_physics_create": (c_int, [POINTER(PhysConfig), c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "ace_physics_destroy": (None, [c_void_p]),
    "ace_physics_reset": (c_int, [c_void_p, c_void_p]),
    "ace_physics_set_reference": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ace_physics_get_reference": (c_int, [c_void_p, c_void_p, POINTER(c_int), c_int, c_void_p]),
    "ace_physics_apply": (c_int, [c_void_p, POINTER(PhysFields), c_int, c_void_p]),
}

_lib = None
TEST_INSTRUMENTATION = ("ace_sht_plan_route", "ace_sfno_sht_route")   # queries of the parity tests, not used by the product path


class AceLibraryMissing(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """The loaded library; raises AceLibraryMissing if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AceLibraryMissing(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `python -m ace_amd.build`). "
                "ace_amd has no CPU fallback."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            if name in TEST_INSTRUMENTATION and os.environ.get("ACE_SFNO_LIB") and not hasattr(handle, name):
                continue   # a variant build of an older round (same-box A/Bs through ACE_SFNO_LIB) may predate the route queries
            fn = getattr(handle, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(rc: int) -> None:
    """Map the C status to the exception the reference would raise at this boundary."""
    if rc == ACE_OK:
        return
    msg = lib().ace_last_error().decode()
    if rc == ACE_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def ptr(t) -> int:
    """Device pointer of a contiguous CUDA/HIP float32 tensor (or None)."""
    if t is None:
        return None
    import torch

    if not t.is_cuda:
        raise RuntimeError("ace_amd kernels need tensors on an MI355X (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32 and t.dtype != torch.complex64:
        raise TypeError(f"expected float32/complex64 tensor, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    return t.data_ptr()


def current_stream() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream
