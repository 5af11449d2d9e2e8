/*
 * gut_hip.h — C ABI of libgut_hip.so, the MI355X-native replacement of the reference's pybind
 * module `lib3dgut_cc` (threedgut_tracer/bindings.cpp:79-113, include/3dgut/splatRaster.h:47-89).
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer owned by the caller (torch tensors in the Python
 *     shim); the library never retains them past the call.  Scratch (per-Gaussian projection
 *     buffers, tile keys, sort temporaries) is owned by the handle and is grow-only.
 *   - `stream` is a hipStream_t passed as void*; all work of a call is enqueued on it.  The reference blocks in the middle
 *     of its forward on a 4-byte device->host read-back of the intersection count (src/gutRenderer.cu:313-321); here only
 *     the FIRST trace() on a handle does: afterwards the binning buffers are sized from the previous frames' counts, the
 *     whole forward is queued, and the host reads the count back behind it (a frame that needed more is binned and
 *     composited a second time; GutStats.binning_overflows counts those).
 *   - return value 0 = success; anything else is an error and gut_last_error() describes it
 *     (the reference logs and drops its Status codes, splatRaster.cpp:225-237; pybind turns C++
 *     exceptions into RuntimeError — the Python shim raises RuntimeError on non-zero).
 *   - one handle <-> one in-flight view: trace_bwd() must follow the trace() it differentiates, on
 *     the same stream (gutRenderer.cu:413-417).  A handle may be used from different host threads
 *     (autograd worker) but not concurrently.
 */
#ifndef GUT_HIP_H
#define GUT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on every change of a struct layout, an array length or an entry point's signature (2: GutStats grew to 80 bytes and
 * GUT_NUM_KERNEL_TIMERS to 11 in round 2; 3: GutLazyMoments in the gut_optimize_* / gut_sh_adam_step_ex signatures, gut_sync_moments,
 * gut_optimize_finish_without_gradient, gut_scatter_gradient_records_dev, gut_trace_fields / gut_trace_bwd_fields;
 * 4: gut_trace_model_fields / gut_trace_bwd_model_fields, gut_position_gradient_statistics, gut_set_position_gradient_statistics,
 * gut_mcmc_perturb; 5: behaviour, not layout — GUT_OPT_SORTED_REFERENCE_BACKWARD defaults to 1, the reference's own form of the
 * sorted variant's backward; the UT sigma-point spread is rounded from double like the reference's build script does;
 * GutLazyMoments.d_overrun; gut_trace_raw_model_fields; GUT_OPT_FORWARD_TILE_ORDER).  Added under 5 without a bump, nothing that was
 * accepted changed meaning: gut_create takes every value of GutConfig the reference has a kernel for (kernel degree, SH storage degree,
 * rolling-shutter iterations, hit counts), gut_optimize_after_bwd takes a NULL camera position, GUT_OPT_KERNEL_TIMING_SET, and the
 * unsorted backward clamps alpha with the reference's literal 0.99 whatever particle_kernel_max_alpha is. */
#define GUT_ABI_VERSION 5

typedef struct gut_context* gut_handle;

/* sensors/cameraModels.h:34-47 */
enum { GUT_SHUTTER_ROLLING_TOP_TO_BOTTOM = 0, GUT_SHUTTER_ROLLING_LEFT_TO_RIGHT = 1,
       GUT_SHUTTER_ROLLING_BOTTOM_TO_TOP = 2, GUT_SHUTTER_ROLLING_RIGHT_TO_LEFT = 3, GUT_SHUTTER_GLOBAL = 4 };
enum { GUT_CAMERA_OPENCV_PINHOLE = 0, GUT_CAMERA_OPENCV_FISHEYE = 1 };

/* Replaces threedgut::CameraModelParameters + TSensorState (sensors/cameraModels.h:22-58,
 * sensors/sensors.h:33-42) as built by fromOpenCV{Pinhole,Fisheye}CameraModelParameters
 * (bindings.cpp:34-77) and toSensorState (splatRaster.cpp:92-100). */
typedef struct GutCamera {
    int32_t model;              /* GUT_CAMERA_* */
    int32_t shutter;            /* GUT_SHUTTER_* */
    float principal_point[2];
    float focal_length[2];
    float radial_coeffs[6];     /* pinhole: k1..k6 ; fisheye: k1..k4 in [0..3] */
    float tangential_coeffs[2]; /* pinhole only */
    float thin_prism_coeffs[4]; /* pinhole only */
    float max_angle;            /* fisheye only */
    float pose_start[7];        /* world->sensor translation (3) + quaternion (x,y,z,w) */
    float pose_end[7];
    int64_t timestamp_start_us;
    int64_t timestamp_end_us;
} GutCamera;

/* The reference bakes conf.render.* into the kernels as -D defines (setup_3dgut.py:47-70) and its
 * SplatRaster constructor reads only render.enable_kernel_timings (splatRaster.cpp:158-159).  Here
 * the same settings arrive as a struct and reach the kernels as run-time values: every key of
 * render/3dgut.yaml may differ from its default (ut_require_all_sigma_points = true is rejected, as the
 * reference's static_assert does).  Values the reference has no kernel for are rejected by gut_create
 * with a clear message.  The fused optimiser entry points and the model-field traces exist for
 * particle_radiance_sph_degree = 3 only. */
typedef struct GutConfig {
    int32_t abi_version;                  /* GUT_ABI_VERSION */
    int32_t enable_kernel_timings;        /* render.enable_kernel_timings */
    int32_t particle_radiance_sph_degree; /* 3 -> 16 coefficients (default); 0..2: radiance rows and their gradient are [N, 3 (d+1)^2] */
    int32_t particle_kernel_degree;       /* 2 (quadratic, default); 0, 1, 3, 4, 5, 8: the reference's other generalised Gaussians */
    int32_t k_buffer_size;                /* 0 (unsorted) */
    int32_t global_z_order;               /* 1 */
    int32_t n_rolling_shutter_iterations; /* 5 by default; 0..64 */
    int32_t ut_require_all_sigma_points;  /* 0 */
    int32_t rect_bounding, tight_opacity_bounding, tile_based_culling; /* 1,1,1 */
    int32_t enable_hitcounts;             /* 1; 0: the hit-count output is all zeros, as in the reference */
    float particle_kernel_min_response;   /* 0.0113 */
    float particle_kernel_min_alpha;      /* 1/255 */
    float particle_kernel_max_alpha;      /* 0.99 */
    float min_transmittance;              /* 1e-4 */
    float ut_alpha, ut_beta, ut_kappa;    /* 1, 2, 0 */
    float ut_in_image_margin_factor;      /* 0.1 */
} GutConfig;

/* scene/traversal statistics of the most recent trace()/trace_bwd() pair: the quantities the
 * algorithmic-bytes model of SURVEY.md §8d is written in. */
typedef struct GutStats {
    uint64_t num_particles;      /* N */
    uint64_t num_visible;        /* V : Gaussians with tilesCount > 0 */
    uint64_t num_intersections;  /* M */
    uint64_t num_tiles;          /* T */
    uint64_t num_pixels;         /* P */
    uint64_t traversed_fwd;      /* E_f : sum over tiles of list entries fetched before the tile terminated */
    uint64_t traversed_bwd;      /* E_b */
    uint32_t sort_end_bit;       /* 32 + bit_width(T) */
    uint32_t binning_overflows;  /* forwards of this handle whose binning was redone because the frame had more intersections than
                                    the capacity assumed from earlier frames (the forward is queued before the count is known) */
    uint64_t side_stream_rows;   /* Gaussians whose optimiser step the last gut_optimize_rows_without_gradient took (both of its
                                    launches, whole 64-row waves); 0 when the last step did not use it */
    uint64_t side_stream_rows_first_launch; /* ... of which in its first launch (under the forward compositor) */
} GutStats;

/* intermediate buffers exposed to the parity tests (device pointers into handle scratch, valid until
 * the next trace() on the handle) */
enum {
    GUT_BUF_TILES_COUNT = 0,   /* u32 [N]        gutRenderer.cu:166 */
    GUT_BUF_TILES_OFFSET = 1,  /* u32 [N]        inclusive scan */
    GUT_BUF_PROJ_POSITION = 2, /* f32 [N,2] */
    GUT_BUF_CONIC_OPACITY = 3, /* f32 [N,4] */
    GUT_BUF_PROJ_EXTENT = 4,   /* f32 [N,2] */
    GUT_BUF_GLOBAL_DEPTH = 5,  /* f32 [N] */
    GUT_BUF_FEATURES = 6,      /* f32 [N,3] precomputed view-dependent RGB (unclamped) */
    GUT_BUF_UNSORTED_KEYS = 7, /* u64 [M] */
    GUT_BUF_UNSORTED_IDS = 8,  /* u32 [M] */
    GUT_BUF_SORTED_KEYS = 9,   /* u64 [M]  (SORTED_*: the reference's fully sorted lists; built on request, see ORDERED_IDS) */
    GUT_BUF_SORTED_IDS = 10,   /* u32 [M] */
    GUT_BUF_TILE_RANGES = 11,  /* u32 [T,2] */
    GUT_BUF_GRAD_SCRATCH = 12, /* f32 [N,16] per-Gaussian gradient rows of the last trace_bwd */
    GUT_BUF_TILE_TRAVERSED_FWD = 13, /* u32 [T] list entries each tile walked before all its rays terminated */
    GUT_BUF_TILE_TRAVERSED_BWD = 14, /* u32 [T] same, last trace_bwd */
    GUT_BUF_ORDERED_IDS = 15,  /* u32 [M] what the compositors actually walked: per tile, the ids in final order for the chunks the
                                  forward staged (0xFFFFFFFF beyond); equals SORTED_IDS on those positions */
    GUT_BUF_PACKED_ROWS = 16   /* f32 [N,12] the rows the last gut_trace_fields / _model_fields / _raw_model_fields forward packed (and, for
                                  the raw entry point, activated: |quat| in the pad column) — what the kernels actually read */
};

/* fills *cfg with the reference defaults (configs/render/3dgut.yaml + 3dgrt.yaml) */
void gut_default_config(GutConfig* cfg);

/* SplatRaster(json) — splatRaster.cpp:153-169 */
int gut_create(const GutConfig* cfg, int device_index, gut_handle* out);
/* ~SplatRaster */
void gut_destroy(gut_handle h);

/* SplatRaster::trace — splatRaster.cpp:174-245.
 *   d_particle_density  f32 [N,12]  (pos3, density, quat wxyz, scale3, pad)   tracer.py:176-178
 *   d_particle_radiance f32 [N,48]  (16 SH coefficients x RGB; [N, 3 (d+1)^2] on a handle with particle_radiance_sph_degree = d < 3)
 *   d_ray_origin/d_ray_direction f32 [H,W,3] camera-space rays
 * outputs (fully written by the call, no pre-initialisation needed):
 *   d_ray_radiance_density f32 [H,W,4], d_ray_hit_distance f32 [H,W,1] (1e6 for rays that miss the
 *   scene AABB), d_ray_hit_count f32 [H,W,1], d_particle_visibility f32 [N,1] (1.0/0.0)          */
int gut_trace(gut_handle h, void* stream, uint32_t frame_number, int32_t num_active_features /* SH degree 0..3 */,
              uint32_t num_particles, const float* d_particle_density, const float* d_particle_radiance,
              int32_t width, int32_t height, const float* d_ray_origin, const float* d_ray_direction,
              const GutCamera* camera,
              float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
              float* d_particle_visibility);

/* SplatRaster::traceBwd — splatRaster.cpp:247-332.  d_ray_hit_distance_grad may be NULL (treated as zeros; selects
 * the kernel variant without hit-distance gradient terms).  Gradient outputs are fully overwritten:
 *   d_particle_density_grad f32 [N,12], d_particle_radiance_grad f32 [N,48]                       */
int gut_trace_bwd(gut_handle h, void* stream, uint32_t frame_number, int32_t num_active_features,
                  uint32_t num_particles, const float* d_particle_density, const float* d_particle_radiance,
                  int32_t width, int32_t height, const float* d_ray_origin, const float* d_ray_direction,
                  const GutCamera* camera,
                  const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                  const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad,
                  float* d_particle_density_grad, float* d_particle_radiance_grad);

/* Same as gut_trace_bwd with option flags.  GUT_BWD_RAW_PARAMETER_GRADS: d_particle_density must be rows produced by
 * gut_activate_pack (|quat| in the pad column); the [N,12] output is then the gradient w.r.t. the RAW parameters
 * (density logit, un-normalised quaternion, log-scale), i.e. chained through sigmoid / normalise / exp
 * (threedgrut/model/model.py:74-93), ready for gut_adam_step. */
#define GUT_BWD_RAW_PARAMETER_GRADS 1u
/* GUT_BWD_COMPACT_RADIANCE_GRADS (implies raw parameter gradients): d_particle_radiance_grad receives only [N,3] =
 * dL/dRGB masked by (precomputed RGB > 0) instead of the [N,48] SH gradient; the SH gradient of the view is
 * Y_k(dir) (x) that row and is rebuilt by gut_sh_adam_step (also across several views: compact data-parallel exchange). */
#define GUT_BWD_COMPACT_RADIANCE_GRADS 2u
/* GUT_BWD_SKIP_EPILOGUE: stop after the compositing backward; the per-Gaussian gradient rows stay in the handle and the two
 * output pointers may be NULL.  Follow with gut_optimize_after_bwd on the same stream (single-view training step: the
 * epilogue, the SH-gradient rebuild and Adam then run as ONE pass over the Gaussians). */
#define GUT_BWD_SKIP_EPILOGUE 4u
int gut_trace_bwd_ex(gut_handle h, void* stream, uint32_t frame_number, int32_t num_active_features,
                     uint32_t num_particles, const float* d_particle_density, const float* d_particle_radiance,
                     int32_t width, int32_t height, const float* d_ray_origin, const float* d_ray_direction,
                     const GutCamera* camera,
                     const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                     const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad,
                     float* d_particle_density_grad, float* d_particle_radiance_grad, uint32_t flags);

/* gut_trace / gut_trace_bwd with the particle parameters as the FOUR activated tensors the reference's Tracer.render hands to
 * _Autograd.apply (positions [N,3], density [N,1], rotation [N,4] wxyz, scale [N,3]; threedgut_tracer/tracer.py:317-327) instead
 * of the [N,12] concatenation _Autograd.forward builds with torch.cat (:176-178), and with the density gradient returned as the
 * four tensors _Autograd.backward hands back (:268-286) instead of one [N,12] tensor it splits and copies.  The rows are packed
 * into handle scratch by one coalesced kernel and kept for the backward; results are those of gut_trace / gut_trace_bwd on the
 * concatenated rows, bit for bit.  d_rotation / d_rotation_grad must be 16-byte aligned (torch allocations are). */
int gut_trace_fields(gut_handle h, void* stream, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                     const float* d_positions, const float* d_density, const float* d_rotation, const float* d_scale,
                     const float* d_particle_radiance, int32_t width, int32_t height, const float* d_ray_origin,
                     const float* d_ray_direction, const GutCamera* camera, float* d_ray_radiance_density,
                     float* d_ray_hit_distance, float* d_ray_hit_count, float* d_particle_visibility);
int gut_trace_bwd_fields(gut_handle h, void* stream, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                         const float* d_particle_radiance, int32_t width, int32_t height, const float* d_ray_origin,
                         const float* d_ray_direction, const GutCamera* camera, const float* d_ray_radiance_density,
                         const float* d_ray_radiance_density_grad, const float* d_ray_hit_distance,
                         const float* d_ray_hit_distance_grad, float* d_positions_grad, float* d_density_grad, float* d_rotation_grad,
                         float* d_scale_grad, float* d_particle_radiance_grad);

/* gut_trace_fields / gut_trace_bwd_fields with the SH coefficients handed over as the model's TWO feature tensors as well
 * (features_albedo [N,3] = the degree-0 triple, features_specular [N,45]; threedgrut/model/model.py:68-75), instead of the [N,48]
 * tensor `get_features()` builds with torch.cat for every render (threedgut_tracer/tracer.py:322) — a 2.3 GB copy at 6 M
 * Gaussians, plus the split and two copies autograd makes of its gradient.  The projection reads each wave's [64,45] block
 * directly (rows of culled Gaussians are not fetched); the backward writes the two gradient tensors in full (zeros for Gaussians
 * without tiles), coalesced.  Results are those of gut_trace_fields on the concatenation, bit for bit.  d_rotation,
 * d_features_specular and their gradients must be 16-byte aligned.  The backward follows a gut_trace_model_fields (or
 * gut_trace_fields) forward on the same handle and stream. */
int gut_trace_model_fields(gut_handle h, void* stream, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                           const float* d_positions, const float* d_density, const float* d_rotation, const float* d_scale,
                           const float* d_features_albedo, const float* d_features_specular, int32_t width, int32_t height,
                           const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                           float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
                           float* d_particle_visibility);
/* gut_trace_model_fields on the model's PRE-ACTIVATION tensors — what MixtureOfGaussians keeps as nn.Parameters (density logit,
 * un-normalised quaternion, log-scale; threedgrut/model/model.py:74-93 with configs/base_gs.yaml:54-55: sigmoid / exp, rotation
 * always normalize): the three activations and the row packing run as one kernel, and the gut_trace_bwd_model_fields that follows
 * on this handle returns the gradients w.r.t. those raw tensors (chained through sigmoid' / normalize' / exp').  For a trainer that
 * keeps the reference's model and optimiser this removes the model's activation kernels, their backward kernels and the
 * AccumulateGrad copies from every step (6.2 -> 5.2 ms per step on the 6 M stand-in).  The caller (3dgrut_amd/tracer.py: Tracer.render)
 * uses it only for a model whose three activation callables ARE torch.sigmoid / torch.nn.functional.normalize / torch.exp. */
int gut_trace_raw_model_fields(gut_handle h, void* stream, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                               const float* d_positions, const float* d_density_logit, const float* d_rotation_raw, const float* d_log_scale,
                               const float* d_features_albedo, const float* d_features_specular, int32_t width, int32_t height,
                               const float* d_ray_origin, const float* d_ray_direction, const GutCamera* camera,
                               float* d_ray_radiance_density, float* d_ray_hit_distance, float* d_ray_hit_count,
                               float* d_particle_visibility);
int gut_trace_bwd_model_fields(gut_handle h, void* stream, uint32_t frame_number, int32_t num_active_features, uint32_t num_particles,
                               int32_t width, int32_t height, const float* d_ray_origin, const float* d_ray_direction,
                               const GutCamera* camera, const float* d_ray_radiance_density, const float* d_ray_radiance_density_grad,
                               const float* d_ray_hit_distance, const float* d_ray_hit_distance_grad, float* d_positions_grad,
                               float* d_density_grad, float* d_rotation_grad, float* d_scale_grad, float* d_features_albedo_grad,
                               float* d_features_specular_grad);

/* SplatRaster::collectTimes — splatRaster.cpp:334-364: mean ms per tag over the timers recorded
 * since the last call; -1 for a tag with no samples. */
int gut_collect_times(gut_handle h, float* forward_render_ms, float* backward_render_ms);

/* statistics + debug views (parity tests, roofline accounting).  gut_get_stats synchronises the
 * stream of the last call. */
int gut_get_stats(gut_handle h, GutStats* out);
int gut_debug_buffer(gut_handle h, int32_t which, void** d_ptr, size_t* bytes);
/* copies a debug buffer into caller-owned DEVICE memory of at least `bytes` bytes (stream-ordered, then synchronised) */
int gut_debug_copy(gut_handle h, int32_t which, void* d_dst, size_t bytes);

/* Runtime options of a handle (take effect at the next gut_trace).
 * GUT_OPT_LAZY_TILE_ORDER (default 1; unsorted variant only): group the (tile | depth) keys by tile only (2 of the 5 radix
 * passes) and let the forward compositor put each tile's entries in depth order 512 at a time, as far as the tile is
 * actually walked (whole-tile termination leaves most of every list untouched: 86 % on the bench frame).  Images, traversal
 * depths and gradients are identical to the fully sorted path; GUT_BUF_ORDERED_IDS shows what was walked, GUT_BUF_SORTED_*
 * still return the reference's full lists (built on request).  0 restores the full radix sort. */
#define GUT_OPT_LAZY_TILE_ORDER 1
/* GUT_OPT_SORTED_REFERENCE_BACKWARD (default 1 since ABI 5; sorted variant k_buffer_size > 0 only): compute the colour term of
 * d(alpha) in the backward exactly as the reference does — un-doing the back-to-front colour recurrence from the final colour
 * with the UNclamped precomputed colour (gutKBufferRenderer.cuh:127-131, shRadiativeParticles.slang:179-207) although the
 * forward composited max(colour, 0) (:159-161).  0 = the exact derivative of the forward (clamped colour in both passes).  The
 * two are identical wherever no composited colour channel is negative; where one is, the reference's gradient is not the
 * derivative of its own forward — the library follows the reference by default and offers the corrected form as the option. */
#define GUT_OPT_SORTED_REFERENCE_BACKWARD 2
/* GUT_OPT_EARLY_EXTRA_PERCENT (default 100, 0..100; unsorted variant): share of the 256-row blocks in which the second launch of
 * gut_optimize_rows_without_gradient (queued when the backward compositor starts) also takes the 64-row waves that HAVE tiles
 * but hold no Gaussian the forward compositor walked.  The backward compositor is bounded by the forward's per-tile depth, so
 * such waves cannot receive a gradient; their zero-gradient Adam step is the same arithmetic wherever it runs.  0 = only waves
 * without tiles take the side stream. */
#define GUT_OPT_EARLY_EXTRA_PERCENT 3
/* GUT_OPT_DEBUG_REPLACE_SCRATCH (developer probe, not a tuning knob): value = index of one of the handle's scratch buffers
 * (0 tiles_count, 1 tiles_offset, 2 proj_pos, 3 conic_opacity, 4 extent, 5 depth, 6 feat, 7 gradient rows, 8 scan temp, 9 / 10
 * unsorted / grouped keys, 11 / 12 unsorted / grouped ids, 13 sort temp, 14 ordered ids, 15 tile ranges, 16 / 17 traversal
 * depths, 18 / 19 tile launch order / ordered prefix): the buffer moves to a fresh device allocation, contents kept (synchronises
 * the device).  tools/scratch_placement.py uses it to find out whose physical placement the compositing kernels' two speeds
 *  belong to. */
/* GUT_OPT_FORWARD_TILE_ORDER (default -1; unsorted variant): launch order of the forward compositor's tiles.  1 = longest lists
 * first (a one-workgroup counting sort of the list lengths in front of it), 0 = image order, -1 = decided by the library from the
 * share of their lists the last frames walked (longest first above 25 %: then the length of a list says how long its tile will
 * run; read back with the intersection count, no extra synchronisation).  Results do not depend on it. */
#define GUT_OPT_FORWARD_TILE_ORDER 4
/* GUT_OPT_KERNEL_TIMING_SET (default 0; only with enable_kernel_timings): which kernel boundaries of the following frames are
 * bracketed by events.  0 = all of them; 2 = none (as with enable_kernel_timings = 0, switchable at run time: bench.py times its
 * steps this way and collects the per-kernel times in a second pass — the two dozen event records per train step cost 5 - 7 % of
 * a 2.4 ms step on MI355X); 1 = only the optimiser launches on the library's side stream (measured: SLOWER than either, the side
 * stream's second launch loses its head start over the backward compositor to the timing event's barrier packet and takes 1.7
 * instead of 1.1 ms — kept for experiments).  Timers of
 * boundaries that were not bracketed read -1. */
#define GUT_OPT_KERNEL_TIMING_SET 5
#define GUT_OPT_DEBUG_REPLACE_SCRATCH 100
int gut_set_option(gut_handle h, int32_t option, int32_t value);

/* per-kernel hipEvent timings of the last trace / trace_bwd (ms), for bench.py's roofline block.
 * Order: project, scan, expand, sort, ranges, render, render_bwd, project_bwd, optimizer (gut_optimize_after_bwd; -1 when
 * that call was not used), optimizer_early (gut_optimize_rows_without_gradient on its side stream, start of its first to end of
 * its second launch, idle gap included; -1 when not used), optimizer_early_2 (its second launch alone).  Requires
 * enable_kernel_timings; synchronises. */
#define GUT_NUM_KERNEL_TIMERS 11
int gut_kernel_times(gut_handle h, float* ms8);
/* mean per-kernel time over the (at most 64 most recent) trace/trace_bwd calls since the previous call of this
 * function; *count = number of forward calls averaged.  Synchronises. */
int gut_kernel_times_mean(gut_handle h, float* ms8, int32_t* count);

/* ---- "next" row N1: fused SSIM, replaces the external CUDA package `fused_ssim`
 * (threedgrut/model/losses.py:17-33: fused_ssim(img1, img2, padding="valid")).  Mean SSIM of two images with an
 * 11x11 Gaussian window (sigma 1.5), valid padding.  Images are addressed with element strides (channel, row,
 * pixel) so NCHW and NHWC tensors are both consumed in place.  Workspace (derivative maps + partial sums) is
 * caller-owned device memory of gut_ssim_workspace_bytes(); backward consumes the workspace its forward filled.
 * Return 0 on success. */
size_t gut_ssim_workspace_bytes(int32_t channels, int32_t height, int32_t width);
int gut_ssim_forward(void* stream, int32_t channels, int32_t height, int32_t width, int64_t stride_c, int64_t stride_h,
                     int64_t stride_w, const float* d_img1, const float* d_img2, void* d_workspace, float* d_mean_ssim);
int gut_ssim_backward(void* stream, int32_t channels, int32_t height, int32_t width, int64_t stride_c, int64_t stride_h,
                      int64_t stride_w, const float* d_img1, const float* d_img2, const void* d_workspace,
                      const float* d_upstream, float* d_grad_img1);

/* Fused photometric loss of the reference's train step (trainer.py:425-449, configs/base_gs.yaml:111-119), forward AND
 * gradient in three launches, no autograd:  image = rgb + background * (1 - alpha)  (model/background.py:78-93,
 * background 0 = black, 1 = white);  loss = lambda_l1 * mean|image - gt| + lambda_ssim * (1 - SSIM(image, gt)).
 * d_rgba [H,W,4] is gut_trace's ray_radiance_density, d_gt_rgb [H,W,3]; d_loss3 receives {loss, L1, SSIM};
 * d_rgba_grad [H,W,4] receives d(loss)/d(rgba), ready to be passed to gut_trace_bwd. */
size_t gut_photometric_workspace_bytes(int32_t height, int32_t width);
int gut_photometric_loss(void* stream, int32_t height, int32_t width, const float* d_rgba, const float* d_gt_rgb, float background,
                         float lambda_l1, float lambda_ssim, void* d_workspace, float* d_loss3, float* d_rgba_grad);

/* ---- "next" row N2: parameter activation + fused Adam ----
 * gut_activate_pack: raw rows [N,12] (pos3, density logit, quat4, log-scale3, unused) -> activated rows
 *   (pos3, sigmoid, quat/|quat|, exp, |quat|) = the particle_density the tracer consumes (model.py:74-93 +
 *   tracer.py:176-178 in one pass).
 * gut_adam_step: in-place Adam on an [rows, cols] fp32 tensor, cols a multiple of 4 and <= 64, per-column learning
 *   rates (host array).  step >= 1: torch.optim.Adam bias correction; step == 0: none (the reference's SelectiveAdam,
 *   optimizers.cu:47-79).  d_visibility != NULL: rows with visibility == 0 are skipped entirely. */
int gut_activate_pack(void* stream, uint32_t num_particles, const float* d_raw12, float* d_act12);
int gut_adam_step(void* stream, uint64_t rows, uint32_t cols, float* d_param, const float* d_grad, float* d_exp_avg,
                  float* d_exp_avg_sq, const float* lr_per_col, float beta1, float beta2, float eps, uint32_t step,
                  const float* d_visibility);

/* ---- Lazy moment decay of the fused optimiser entry points (gut_optimize_*, gut_adam_unwalked_waves_ex, gut_sh_adam_step_ex) ----
 * Adam with a zero gradient still decays both moments of every parameter (torch.optim.Adam semantics), i.e. reads AND writes
 * 2 x 236 bytes per Gaussian per step for nothing but two multiplications by constants.  With a GutLazyMoments the zero-gradient
 * update of a 64-row wave that cannot receive a gradient in this step — no row of it has a tile, or the forward compositor walked
 * none of its Gaussians (the backward is bounded by the forward's per-tile depth) — READS the moments, brings them up to date in
 * registers, updates the parameters, and does NOT write the moments back: d_wave_step[w] (one uint32 per wave, caller-owned,
 * zero-initialised) holds the step up to which the stored moments of wave w are current, and the next kernel that reads them
 * multiplies by beta^(steps missed), taken from the caller's tables d_pow_beta1/2[k] = (float) beta^k, k < table_len
 * (d_pow[0] = 1).  Every wave that can receive a gradient is brought up to date, updated and written as before.  The rule
 * does not depend on which kernel walks a wave, so one-pass and two-pass steps stay bit-identical.  The result differs from
 * writing the moments every step only by the rounding of beta^k against k successive multiplications (~1e-7 relative).
 * gut_sync_moments brings every stored moment up to `step` (call it before moving rows between waves, reading the moments
 * from outside, or every table_len / 2 steps).  Not available with a visibility mask (SelectiveAdam does not decay at all). */
typedef struct GutLazyMoments {
    uint32_t* d_wave_step;       /* [ceil(N / 64)] */
    const float* d_pow_beta1;    /* [table_len] */
    const float* d_pow_beta2;    /* [table_len] */
    uint32_t table_len;
    uint32_t* d_overrun;         /* may be NULL; one word, caller-zeroed: set to 1 by any kernel that met a wave whose stored moments
                                    had missed table_len or more steps (its decay factor does not exist in the tables: the caller
                                    resumed with a wave_step that does not belong to its step counter, or never called
                                    gut_sync_moments).  The caller checks it where it synchronises anyway (ABI 5) */
} GutLazyMoments;
int gut_sync_moments(void* stream, uint32_t num_particles, float* d_raw_m, float* d_raw_v, float* d_sh_m, float* d_sh_v,
                     const GutLazyMoments* lazy, uint32_t step /* the last optimiser step applied */);

/* lib_optimizers_cc.selective_adam_update (threedgrut/optimizers/optimizers.cu:47-117, bound in optimizers/__init__.py:113-123):
 * the SelectiveAdam update of one [rows, cols] fp32 parameter of ANY width, rows whose visibility byte is 0 untouched, no bias
 * correction.  d_visibility: one byte per row (a torch.bool tensor).  3dgrut_amd/optimizers.py wraps it in the reference's
 * `SelectiveAdam(torch.optim.Adam)` class. */
int gut_selective_adam(void* stream, uint64_t rows, uint32_t cols, float* d_param, const float* d_grad, float* d_exp_avg,
                       float* d_exp_avg_sq, const uint8_t* d_visibility, float lr, float beta1, float beta2, float eps);

/* Fused "SH gradient + Adam" step of the native trainer.  For every Gaussian: Adam on the raw [N,12] row with
 * d_raw_grad12 (already summed over views), then the [N,48] SH row with the gradient
 *     sum_v Y_k(normalize(pos - camera_position[v])) * mrgb[v][i][c]   (k < (sh_degree+1)^2, else 0)
 * rebuilt on the fly from num_views compact rows (gut_trace_bwd_ex(..., GUT_BWD_COMPACT_RADIANCE_GRADS)); both
 * gradients are scaled by grad_scale (1/num_views for a mean over views).  pos is the pre-update position.
 * lr12/lr48: host arrays of per-column learning rates; step/visibility as in gut_adam_step.
 * d_act12_out (may be NULL): if given, every updated row's activation (what gut_activate_pack would compute from the new
 * raw row) is written there, so the next gut_trace needs no separate activation pass.
 * mrgb_view_stride: rows between consecutive views in d_mrgb (0 = num_particles).  All per-Gaussian pointers may be
 * offset to a row range [r0, r1) with num_particles = r1 - r0: that is how the data-parallel trainer pipelines the
 * optimiser over chunks of Gaussians behind the gradient exchange of the following chunk. */
int gut_sh_adam_step(void* stream, uint32_t num_particles, int32_t sh_degree, uint32_t num_views,
                     const float* d_camera_positions /* device [num_views,3] */, const float* d_mrgb /* [num_views,N,3] */,
                     const float* d_raw_grad12, float grad_scale,
                     float* d_raw12, float* d_raw_m, float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v,
                     const float* lr12, const float* lr48, float beta1, float beta2, float eps, uint32_t step,
                     const float* d_visibility, float* d_act12_out, uint32_t mrgb_view_stride);

/* Epilogue of the last gut_trace_bwd_ex(..., GUT_BWD_SKIP_EPILOGUE) + gut_sh_adam_step for ONE view in a single kernel:
 * chains the handle's gradient rows to the raw parameters (sigmoid / normalise / exp recomputed from d_raw12, which must be
 * the rows the forward's gut_activate_pack input was made from), rebuilds the SH gradient from the masked dL/dRGB and
 * applies Adam to d_raw12 / d_sh48.  d_camera_position: device [3] sensor position of the view, or NULL = the sensor position of
 * the cached forward, which the library keeps on the device (the floats its projection evaluated the colours from).  Other arguments as in
 * gut_sh_adam_step.  Consumes the backward context. */
int gut_optimize_after_bwd(gut_handle h, void* stream, int32_t num_active_features, const float* d_camera_position,
                           float* d_raw12, float* d_raw_m, float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v,
                           const float* lr12, const float* lr48, float beta1, float beta2, float eps, uint32_t step,
                           const float* d_visibility, float* d_act12_out, const GutLazyMoments* lazy /* may be NULL */);

/* Optional first half of that optimiser step, to be called BETWEEN gut_trace and gut_trace_bwd_ex(..., GUT_BWD_SKIP_EPILOGUE) on
 * the forward's stream.  Gaussians that cannot receive a gradient from this view get their Adam step (same arithmetic as
 * gut_optimize_after_bwd with an exactly-zero gradient: moments decay, parameters move on their momentum, activation rows
 * rewritten) on a side stream owned by the handle, as pure HBM streaming UNDER and beside the VALU-bound compositing kernels of
 * the same iteration.  They are taken by whole 64-row waves, in two launches of a persistent kernel with a small fixed footprint:
 *   1. right away, ordered behind the binning part of the forward only: waves in which no row has a tile
 *      (tiles_count == 0), in the first 60 % of the row blocks with lazy moments (a 32-register form of the kernel, which
 *      leaves the forward compositor all of its waves), in the first quarter otherwise;
 *   2. when gut_trace_bwd_ex queues the backward compositor: the remaining waves without tiles and — unsorted variant,
 *      GUT_OPT_EARLY_EXTRA_PERCENT — the waves that have tiles but hold no Gaussian among the list entries the forward
 *      compositor walked (the backward compositor is bounded by the forward's per-tile depth).
 * The following gut_optimize_after_bwd (mandatory; same pointers and hyper-parameters; d_visibility must be NULL) then walks only
 * the other waves and orders the caller's stream behind the side stream.  The parameters after the two calls are those of
 * gut_optimize_after_bwd alone; every row of a wave the side stream took is bit-identical to it.  The caller must not touch the
 * parameter tensors on other streams between the two calls. */
int gut_optimize_rows_without_gradient(gut_handle h, void* stream, float* d_raw12, float* d_raw_m, float* d_raw_v, float* d_sh48,
                                       float* d_sh_m, float* d_sh_v, const float* lr12, const float* lr48, float beta1,
                                       float beta2, float eps, uint32_t step, float* d_act12_out,
                                       const GutLazyMoments* lazy /* may be NULL; must match gut_optimize_after_bwd's */);

/* Ends an optimiser step that gut_optimize_rows_without_gradient began but gut_optimize_after_bwd cannot finish (the caller's
 * loss raised, the backward was rejected, ...): every wave the side stream does not own takes the same Adam step with an
 * exactly-zero gradient (whatever the backward compositor may already have accumulated for this view is discarded and the
 * gradient rows are left zero), the caller's stream is ordered behind the side stream, and the handle accepts gut_trace again.
 * Afterwards every row has taken exactly one step of iteration `step`, as if the view had given no gradient at all.  No-op
 * (returns 0) when no step is half applied. */
int gut_optimize_finish_without_gradient(gut_handle h, void* stream);

/* ---- Sparse gradient exchange of the data-parallel trainer (new functionality, the reference is single-GPU) ----
 * A view gives a gradient only to the Gaussians its rays hit.  Instead of dense [N,12] + [N,3] tensors per view, the ranks
 * exchange lists of 64-byte records, one per Gaussian with a non-zero gradient row:
 *     f32[16] = { d pos3, d density logit, d quat4 (un-normalised), d log-scale3, row id (uint32 bits), masked dL/dRGB 3, 0 }
 * (the gradient w.r.t. the RAW parameters, as GUT_BWD_RAW_PARAMETER_GRADS | GUT_BWD_COMPACT_RADIANCE_GRADS would write it).
 *
 * gut_compact_gradient_rows: epilogue of the last gut_trace_bwd_ex(..., GUT_BWD_SKIP_EPILOGUE).  d_particle_density: the
 *   activated rows the forward was given (gut_activate_pack layout).  d_records: [capacity,16] with capacity >= num_particles
 *   (so it cannot overflow); *d_count (device uint32) receives the number of records.  Record order is unspecified; every
 *   row id occurs at most once.  Consumes the backward context (the handle's gradient rows are left zero).
 * gut_scatter_gradient_records: adds `count` records of ONE view into the dense accumulator d_raw_grad12 [N,12] and stores
 *   their dL/dRGB into that view's slab d_mrgb_view [N,3].  Both must be zero wherever no record lands (allocate zeroed once;
 *   gut_sh_adam_step_ex(..., GUT_ADAM_CLEAR_CONSUMED_GRADS) zeroes what it consumed).  Scatter the views one call after the
 *   other in the same order on every rank: the sums are then formed in the same order everywhere and replicas stay
 *   bit-identical.
 * gut_sh_adam_step_ex: gut_sh_adam_step with flags; GUT_ADAM_CLEAR_CONSUMED_GRADS writes zeros over every non-zero row of
 *   d_raw_grad12 / d_mrgb it read. */
#define GUT_ADAM_CLEAR_CONSUMED_GRADS 1u
#define GUT_GRADIENT_RECORD_FLOATS 16
int gut_compact_gradient_rows(gut_handle h, void* stream, const float* d_particle_density, float* d_records, uint32_t capacity,
                              uint32_t* d_count);
int gut_scatter_gradient_records(void* stream, const float* d_records, uint32_t count, uint32_t num_particles, float* d_raw_grad12,
                                 float* d_mrgb_view);
/* the same with the number of valid records read ON THE DEVICE: min(*d_count, max_count) records are taken.  Lets the caller
 * queue the scatter before the host has seen the ranks' record counts (3dgrut_amd/dp.py: RecordExchange). */
int gut_scatter_gradient_records_dev(void* stream, const float* d_records, const uint32_t* d_count, uint32_t max_count,
                                     uint32_t num_particles, float* d_raw_grad12, float* d_mrgb_view);
int gut_sh_adam_step_ex(void* stream, uint32_t num_particles, int32_t sh_degree, uint32_t num_views,
                        const float* d_camera_positions, float* d_mrgb, float* d_raw_grad12, float grad_scale,
                        float* d_raw12, float* d_raw_m, float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v,
                        const float* lr12, const float* lr48, float beta1, float beta2, float eps, uint32_t step,
                        const float* d_visibility, float* d_act12_out, uint32_t mrgb_view_stride, uint32_t flags,
                        const uint8_t* d_wave_flags /* may be NULL; else waves with flag 0 are left alone, see below */,
                        const GutLazyMoments* lazy /* may be NULL */);

/* Data-parallel form of the side-stream optimiser pass (single-view form: gut_optimize_rows_without_gradient).
 * gut_mark_walked_waves: after gut_trace, same stream: d_wave_flags [ceil(N/64)] bytes := 1 for every 64-row wave that holds a
 *   Gaussian among the list entries the forward compositor walked (sorted variant: every wave with a tile), else 0.  The
 *   backward compositor gives gradients to those Gaussians only.  MAX-all-reduce the flags over the ranks (94 KB at 6 M): a
 *   wave whose flag is still 0 receives no gradient from ANY view of the step.
 * gut_adam_unwalked_waves: zero-gradient Adam step (moments decay, parameters move on their momentum, activation rows
 *   rewritten) of every wave whose flag is 0, as a persistent kernel with a small fixed footprint: queue it on a side stream
 *   as soon as the reduced flags are there; it streams under the backward compositor, the gradient exchange and the call below.
 * gut_sh_adam_step_ex(..., d_wave_flags) then skips exactly those waves.  Rows of the skipped waves are bit-identical to what
 *   gut_sh_adam_step_ex without flags computes for them (their gradient is exactly zero). */
int gut_mark_walked_waves(gut_handle h, void* stream, uint8_t* d_wave_flags);
int gut_adam_unwalked_waves(void* stream, uint32_t num_particles, const uint8_t* d_wave_flags, float* d_raw12, float* d_raw_m,
                            float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v, const float* lr12, const float* lr48,
                            float beta1, float beta2, float eps, uint32_t step, float* d_act12_out);
/* ... with lazy moment decay (NULL = as above) */
int gut_adam_unwalked_waves_ex(void* stream, uint32_t num_particles, const uint8_t* d_wave_flags, float* d_raw12, float* d_raw_m,
                               float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v, const float* lr12, const float* lr48,
                               float beta1, float beta2, float eps, uint32_t step, float* d_act12_out, const GutLazyMoments* lazy);

/* ---- "next" row N3: the densification statistics of GSStrategy.update_gradient_buffer (threedgrut/strategy/gs.py:
 * 106-115), which the reference's trainer runs between backward and optimiser in every iteration up to densify.end_iteration
 * (trainer.py:741-743): for every Gaussian whose position gradient of THIS view is non-zero,
 *     norm_accum[i] += | grad_i * |position_i - sensor_position| | / 2,      norm_denom[i] += 1.
 * gut_position_gradient_statistics is that as one kernel (grad / positions: rows of `*_stride` floats, first three columns used;
 * d_sensor_position: device float[3]).  gut_set_position_gradient_statistics folds it into the NEXT gut_optimize_after_bwd on the
 * handle (that kernel holds each row's gradient and pre-update position in registers anyway; the setting is consumed by that one
 * call, so the caller may reallocate its buffers between steps; NULL, NULL clears it): the step keeps its fused one- or two-pass
 * optimiser instead of materialising the [N,3] gradient for a hook (6.5 -> 2.6 ms per step at 6 M Gaussians). */
int gut_position_gradient_statistics(void* stream, uint32_t num_particles, const float* d_position_grad, uint32_t grad_stride,
                                     const float* d_positions, uint32_t position_stride, const float* d_sensor_position,
                                     float* d_norm_accum, int32_t* d_norm_denom);
int gut_set_position_gradient_statistics(gut_handle h, float* d_norm_accum, int32_t* d_norm_denom);

/* ---- "next" row N3: MCMC relocation kernel (threedgrut/strategy/src/gaussian_mcmc.cu:33-73).
 * opacities [n], scales [n,3], ratios [n] (int32, 1..n_max), binoms [n_max,n_max] -> new_opacities [n], new_scales [n,3] */
int gut_mcmc_relocation(void* stream, int32_t n, const float* d_opacities, const float* d_scales, const int32_t* d_ratios,
                        const float* d_binoms, int32_t n_max, float* d_new_opacities, float* d_new_scales);

/* MCMCStrategy.perturb_gaussians (threedgrut/strategy/mcmc.py:147-164), which the reference's trainer runs after EVERY optimiser
 * step of an MCMC run (configs/strategy/mcmc.yaml: perturb.frequency 1): positions += Sigma @ (unit_normal * op_sigmoid(1 - density) *
 * noise_scale) on the raw [N,12] rows (pos3 | density logit | quat wxyz raw | log-scale3 | pad), Sigma = R S S^T R^T, noise_scale =
 * perturb.noise_lr * the current position learning rate.  d_act12 (may be NULL): the trainer's activated rows, whose positions are
 * updated too.  d_unit_normals (may be NULL): [N,3] standard-normal draws to use; NULL = Philox4x32-10 keyed by (seed, step) with the
 * row index as counter (the same draws on every rank).  One pass, 96 bytes per Gaussian, instead of four batched 3x3 matmuls. */
int gut_mcmc_perturb(void* stream, uint32_t num_particles, float* d_raw12, float* d_act12, float noise_scale, uint64_t seed,
                     uint64_t step, const float* d_unit_normals);

const char* gut_last_error(void);
int gut_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GUT_HIP_H */
