"""ctypes binding of libgut_hip.so (C ABI in include/gut_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load this module
raises, and every Tracer call fails loudly.  (The CPU oracle under oracle/ is test infrastructure and
is never imported from here.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GUT_HIP_LIB", os.path.join(HERE, "libgut_hip.so"))  # override: dev experiments only

GUT_ABI_VERSION = 5   # include/gut_hip.h: bumped on every struct / array-length / signature change
GUT_NUM_KERNEL_TIMERS = 11
BWD_RAW_PARAMETER_GRADS = 1
BWD_COMPACT_RADIANCE_GRADS = 2
BWD_SKIP_EPILOGUE = 4
OPT_LAZY_TILE_ORDER = 1
ADAM_CLEAR_CONSUMED_GRADS = 1
GRADIENT_RECORD_FLOATS = 16
OPT_SORTED_REFERENCE_BACKWARD = 2
OPT_EARLY_EXTRA_PERCENT = 3
OPT_FORWARD_TILE_ORDER = 4
OPT_KERNEL_TIMING_SET = 5
OPT_DEBUG_REPLACE_SCRATCH = 100
KERNEL_TIMER_NAMES = ("project", "scan", "expand", "sort", "ranges", "render", "render_bwd", "project_bwd", "optimizer",
                      "optimizer_early", "optimizer_early_2")

SHUTTER_GLOBAL = 4
CAMERA_PINHOLE, CAMERA_FISHEYE = 0, 1

BUF = dict(tiles_count=0, tiles_offset=1, proj_pos=2, conic_opacity=3, extent=4, depth=5, feat=6,
           unsorted_keys=7, unsorted_ids=8, sorted_keys=9, sorted_ids=10, tile_ranges=11, grad_scratch=12,
           tile_traversed_fwd=13, tile_traversed_bwd=14, ordered_ids=15, packed_rows=16)


class GutCamera(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("shutter", C.c_int32),
        ("principal_point", C.c_float * 2), ("focal_length", C.c_float * 2),
        ("radial_coeffs", C.c_float * 6), ("tangential_coeffs", C.c_float * 2), ("thin_prism_coeffs", C.c_float * 4),
        ("max_angle", C.c_float), ("pose_start", C.c_float * 7), ("pose_end", C.c_float * 7),
        ("timestamp_start_us", C.c_int64), ("timestamp_end_us", C.c_int64),
    ]


class GutConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("enable_kernel_timings", C.c_int32), ("particle_radiance_sph_degree", C.c_int32),
        ("particle_kernel_degree", C.c_int32), ("k_buffer_size", C.c_int32), ("global_z_order", C.c_int32),
        ("n_rolling_shutter_iterations", C.c_int32), ("ut_require_all_sigma_points", C.c_int32),
        ("rect_bounding", C.c_int32), ("tight_opacity_bounding", C.c_int32), ("tile_based_culling", C.c_int32),
        ("enable_hitcounts", C.c_int32),
        ("particle_kernel_min_response", C.c_float), ("particle_kernel_min_alpha", C.c_float),
        ("particle_kernel_max_alpha", C.c_float), ("min_transmittance", C.c_float),
        ("ut_alpha", C.c_float), ("ut_beta", C.c_float), ("ut_kappa", C.c_float), ("ut_in_image_margin_factor", C.c_float),
    ]


class GutStats(C.Structure):
    _fields_ = [
        ("num_particles", C.c_uint64), ("num_visible", C.c_uint64), ("num_intersections", C.c_uint64),
        ("num_tiles", C.c_uint64), ("num_pixels", C.c_uint64), ("traversed_fwd", C.c_uint64),
        ("traversed_bwd", C.c_uint64), ("sort_end_bit", C.c_uint32), ("binning_overflows", C.c_uint32),
        ("side_stream_rows", C.c_uint64), ("side_stream_rows_first_launch", C.c_uint64),
    ]


class GutLazyMoments(C.Structure):
    _fields_ = [("d_wave_step", C.c_void_p), ("d_pow_beta1", C.c_void_p), ("d_pow_beta2", C.c_void_p), ("table_len", C.c_uint32),
                ("d_overrun", C.c_void_p)]


EXPORTS = ("gut_default_config", "gut_create", "gut_destroy", "gut_trace", "gut_trace_bwd", "gut_collect_times",
           "gut_get_stats", "gut_debug_buffer", "gut_debug_copy", "gut_kernel_times", "gut_kernel_times_mean", "gut_last_error", "gut_abi_version",
           "gut_ssim_workspace_bytes", "gut_ssim_forward", "gut_ssim_backward",
           "gut_photometric_workspace_bytes", "gut_photometric_loss", "gut_optimize_after_bwd", "gut_set_option",
           "gut_trace_bwd_ex", "gut_optimize_rows_without_gradient", "gut_compact_gradient_rows", "gut_scatter_gradient_records",
           "gut_sh_adam_step_ex", "gut_mark_walked_waves", "gut_adam_unwalked_waves", "gut_activate_pack", "gut_adam_step", "gut_sh_adam_step", "gut_mcmc_relocation",
           "gut_optimize_finish_without_gradient", "gut_scatter_gradient_records_dev", "gut_adam_unwalked_waves_ex", "gut_sync_moments", "gut_trace_fields", "gut_trace_bwd_fields", "gut_selective_adam",
           "gut_trace_model_fields", "gut_trace_bwd_model_fields", "gut_position_gradient_statistics",
           "gut_set_position_gradient_statistics", "gut_mcmc_perturb", "gut_trace_raw_model_fields")

_lib = None


def load():
    """Load libgut_hip.so (raises RuntimeError with build instructions if it is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    # (GUT_HIP_LIBRARY: a developer's diagnostic build of the same sources, e.g. tools/clock_probe.py's stamped library)
    path = os.environ.get("GUT_HIP_LIBRARY") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the 3DGUT HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). There is no CPU fallback in the product path.")
    lib = C.CDLL(path)
    vp, u32, i32, f_p = C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p
    lib.gut_last_error.restype = C.c_char_p
    lib.gut_abi_version.restype = C.c_int
    lib.gut_default_config.argtypes = [C.POINTER(GutConfig)]
    lib.gut_default_config.restype = None
    lib.gut_create.argtypes = [C.POINTER(GutConfig), C.c_int, C.POINTER(vp)]
    lib.gut_destroy.argtypes = [vp]
    lib.gut_destroy.restype = None
    lib.gut_trace.argtypes = [vp, vp, u32, i32, u32, f_p, f_p, i32, i32, f_p, f_p, C.POINTER(GutCamera), f_p, f_p, f_p, f_p]
    lib.gut_trace_bwd.argtypes = [vp, vp, u32, i32, u32, f_p, f_p, i32, i32, f_p, f_p, C.POINTER(GutCamera),
                                  f_p, f_p, f_p, f_p, f_p, f_p]
    lib.gut_trace_fields.argtypes = [vp, vp, u32, i32, u32, f_p, f_p, f_p, f_p, f_p, i32, i32, f_p, f_p, C.POINTER(GutCamera), f_p, f_p, f_p, f_p]
    lib.gut_trace_bwd_fields.argtypes = [vp, vp, u32, i32, u32, f_p, i32, i32, f_p, f_p, C.POINTER(GutCamera), f_p, f_p, f_p, f_p,
                                         f_p, f_p, f_p, f_p, f_p]
    lib.gut_trace_model_fields.argtypes = [vp, vp, u32, i32, u32, f_p, f_p, f_p, f_p, f_p, f_p, i32, i32, f_p, f_p, C.POINTER(GutCamera),
                                           f_p, f_p, f_p, f_p]
    lib.gut_trace_raw_model_fields.argtypes = lib.gut_trace_model_fields.argtypes
    lib.gut_trace_bwd_model_fields.argtypes = [vp, vp, u32, i32, u32, i32, i32, f_p, f_p, C.POINTER(GutCamera), f_p, f_p, f_p, f_p,
                                               f_p, f_p, f_p, f_p, f_p, f_p]
    lib.gut_position_gradient_statistics.argtypes = [vp, u32, f_p, u32, f_p, u32, f_p, f_p, vp]
    lib.gut_set_position_gradient_statistics.argtypes = [vp, f_p, vp]
    lib.gut_mcmc_perturb.argtypes = [vp, u32, f_p, f_p, C.c_float, C.c_uint64, C.c_uint64, f_p]
    lib.gut_collect_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.gut_get_stats.argtypes = [vp, C.POINTER(GutStats)]
    lib.gut_debug_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.gut_debug_copy.argtypes = [vp, i32, vp, C.c_size_t]
    lib.gut_kernel_times.argtypes = [vp, C.POINTER(C.c_float)]
    lib.gut_kernel_times_mean.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    i64 = C.c_int64
    lib.gut_ssim_workspace_bytes.argtypes = [i32, i32, i32]
    lib.gut_ssim_workspace_bytes.restype = C.c_size_t
    lib.gut_ssim_forward.argtypes = [vp, i32, i32, i32, i64, i64, i64, vp, vp, vp, vp]
    lib.gut_ssim_backward.argtypes = [vp, i32, i32, i32, i64, i64, i64, vp, vp, vp, vp, vp]
    lib.gut_photometric_workspace_bytes.argtypes = [i32, i32]
    lib.gut_photometric_workspace_bytes.restype = C.c_size_t
    lib.gut_photometric_loss.argtypes = [vp, i32, i32, vp, vp, C.c_float, C.c_float, C.c_float, vp, vp, vp]
    lib.gut_trace_bwd_ex.argtypes = [vp, vp, u32, i32, u32, f_p, f_p, i32, i32, f_p, f_p, C.POINTER(GutCamera),
                                     f_p, f_p, f_p, f_p, f_p, f_p, u32]
    lib.gut_activate_pack.argtypes = [vp, u32, vp, vp]
    lib.gut_adam_step.argtypes = [vp, C.c_uint64, u32, vp, vp, vp, vp, C.POINTER(C.c_float), C.c_float, C.c_float,
                                  C.c_float, u32, vp]
    fptr = C.POINTER(C.c_float)
    lazy_p = C.POINTER(GutLazyMoments)
    lib.gut_optimize_after_bwd.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, fptr, fptr, C.c_float, C.c_float, C.c_float, u32,
                                           vp, vp, lazy_p]
    lib.gut_optimize_rows_without_gradient.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, fptr, fptr, C.c_float, C.c_float, C.c_float,
                                                       u32, vp, lazy_p]
    lib.gut_sync_moments.argtypes = [vp, u32, vp, vp, vp, vp, lazy_p, u32]
    lib.gut_optimize_finish_without_gradient.argtypes = [vp, vp]
    lib.gut_selective_adam.argtypes = [vp, C.c_uint64, u32, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float]
    lib.gut_set_option.argtypes = [vp, i32, i32]
    lib.gut_mcmc_relocation.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp]
    lib.gut_sh_adam_step.argtypes = [vp, u32, i32, u32, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, fptr, fptr,
                                     C.c_float, C.c_float, C.c_float, u32, vp, vp, u32]
    lib.gut_sh_adam_step_ex.argtypes = lib.gut_sh_adam_step.argtypes + [u32, vp, lazy_p]
    lib.gut_mark_walked_waves.argtypes = [vp, vp, vp]
    lib.gut_adam_unwalked_waves.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp, vp, fptr, fptr, C.c_float, C.c_float, C.c_float, u32, vp]
    lib.gut_adam_unwalked_waves_ex.argtypes = lib.gut_adam_unwalked_waves.argtypes + [lazy_p]
    lib.gut_compact_gradient_rows.argtypes = [vp, vp, vp, vp, u32, vp]
    lib.gut_scatter_gradient_records.argtypes = [vp, vp, u32, u32, vp, vp]
    lib.gut_scatter_gradient_records_dev.argtypes = [vp, vp, vp, u32, u32, vp, vp]
    if lib.gut_abi_version() != GUT_ABI_VERSION:
        raise RuntimeError("libgut_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gut_last_error()
        raise RuntimeError(f"[3dgut] {what}: {msg.decode() if msg else 'unknown error'}")
