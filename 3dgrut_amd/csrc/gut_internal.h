// gut_internal.h — types shared by the HIP translation units of libgut_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/gut_hip.h"

namespace gut {

constexpr int kTile = 16;                 // GUTParameters::Tiling::BlockX/Y  (gutRendererParameters.h:22-31)
constexpr int kBlock = 256;
constexpr uint32_t kInvalid = 0xFFFFFFFFu;

// 3x4 affine transform, row-major: y = R x + t
struct Affine {
    float r[3][3];
    float t[3];
};

// Everything the device code needs to know about the view; built on the host (gut_api.cpp).
struct ViewParams {
    Affine w2s_end;     // world->sensor at the end pose (rolling-shutter fallback, cameraProjections.cuh:162-170)
    float pose_start[7], pose_end[7];  // t(3), q(x,y,z,w): interpolated per sigma point for rolling shutters
    int32_t shutter;
    Affine w2s_start;   // world->sensor at the start pose: projects the sigma points (cameraProjections.cuh:154-157)
    Affine w2s_mid;     // world->sensor at the interpolated mid pose: depth key (gutProjector.cuh:137,317)
    Affine s2w;         // sensor->world: ray transform; s2w.t is the sensor position in world space
    int32_t model;
    int32_t width, height;
    int32_t grid_x, grid_y;
    float principal_point[2];
    float focal_length[2];
    float radial[6];
    float tangential[2];
    float thin_prism[4];
    float max_angle;
    int32_t shutter_iterations;   // GAUSSIAN_N_ROLLING_SHUTTER_ITERATIONS (threedgut.cuh:63), rolling shutters only
};

// Derived render constants (host-computed in fp32 with the same operation order as the oracle).
struct RenderConsts {
    float alpha_threshold, max_alpha, min_response, min_transmittance;
    float min_sensor_z, cov_dilation;
    float ut_delta, ut_w0_mean, ut_wi, ut_w0_cov, ut_margin;
    int32_t rect_bounding, tight_opacity_bounding, tile_culling, global_z_order;
    float max_d2;   // -2 ln(min_response): d2 above this can never be accepted (kernel degree 2; kernel_cutoff_d2 otherwise)
};

// largest d2 at which the kernel's response still reaches r (0 < r < 1), without safety margin
__host__ __device__ inline float kernel_cutoff_d2(int degree, float r) {
    float s, n;
    switch (degree) {
    case 8: s = 0.000685871056241f; n = 8.0f; break;
    case 5: s = 0.0185185185185f; n = 5.0f; break;
    case 4: s = 0.0555555555556f; n = 4.0f; break;
    case 3: s = 0.166666666667f; n = 3.0f; break;
    case 1: s = 1.5f; n = 1.0f; break;
    case 0: { const float t = (1.0f - r) / 0.329630334487f; return t * t; }
    default: return -2.0f * logf(r);
    }
    return powf(-logf(r) / s, 2.0f / n);
}

// device-side counters (one 64-byte line)
struct Counters {
    unsigned long long visible;        // V
    unsigned long long traversed_fwd;  // E_f
    unsigned long long traversed_bwd;  // E_b
    unsigned long long side_stream_rows;
    unsigned long long side_stream_rows_first;
    unsigned long long pad[3];
};

// Lazy moment decay of the fused optimiser kernels (GutLazyMoments of the C ABI + the step being applied).  A 64-row wave that
// cannot receive a gradient in a step — no row has a tile, or the forward walked none of its Gaussians — gets its parameter
// update from m, v READ but NOT WRITTEN BACK: wave_step[w] remembers up to which step the stored moments are current, and
// whoever reads them next first multiplies by beta^(steps missed) from a host-made table (pow1[k] = fl(beta1^k), pow1[0] = 1).
// Saves 472 of the 1488 bytes per row of the zero-gradient update.  wave_step == nullptr: every update writes its moments.
struct LazyMoments {
    uint32_t* wave_step = nullptr;
    const float* pow1 = nullptr;
    const float* pow2 = nullptr;
    uint32_t len = 0;
    uint32_t t = 0;     // the optimiser step being applied (1-based)
    uint32_t* overrun = nullptr;   // GutLazyMoments.d_overrun
};

// what K1 zeroes on its way (the frame's small clears): ranges [tiles] uint2, trav_bwd [tiles], wave_walked [4 * blocks] bytes (or null)
struct FrameClears {
    uint2* ranges = nullptr;
    uint32_t* trav_bwd = nullptr;
    uint8_t* wave_walked = nullptr;
    uint32_t tiles = 0;
    float* cam_pos = nullptr;   // [3]: K1 also leaves the sensor position (ViewParams.s2w.t) on the device
};

// ---- launch wrappers implemented in the .hip files -------------------------------------------------
void launch_project(hipStream_t s, const ViewParams& v, const RenderConsts& c, uint32_t n, int sh_degree,
                    const float* density12, const float* sph48, uint32_t* tiles_count, float* proj_pos,
                    float* conic_opacity, float* extent, float* depth, float* feat, float* visibility,
                    uint32_t* wave_sums /* [4 * blocks]: tile count of every 64-row wave (first level of the scan) */, const float* sph_albedo /* non-null: sph48 is features_specular [N,45], this is features_albedo [N,3] */,
                    const FrameClears& clears);
// second level of the scan of the tile counts: block_prefix[b] = list entries of the Gaussians before 256-row block b, *total = M
void launch_scan_wave_sums(hipStream_t s, uint32_t n, const uint32_t* wave_sums, uint32_t* block_prefix, uint32_t* total,
                           uint32_t* host_out /* device view of pinned host words, or null */, const uint32_t* walk_sums /* or null */);
void launch_expand(hipStream_t s, const ViewParams& v, const RenderConsts& c, uint32_t n, const uint32_t* tiles_count,
                   const uint32_t* wave_sums, const uint32_t* block_prefix, const uint32_t* total /* device word: M; the tail
                   [M, capacity) of keys / ids is padded here */, const float* proj_pos, const float* conic_opacity, const float* extent, const float* depth,
                   uint64_t* keys, uint32_t* ids, uint32_t capacity);
void launch_pad_keys(hipStream_t s, const uint32_t* count, uint32_t sort_n, uint64_t* keys, uint32_t* ids);
// debug view only: ordered_ids[range.x + tile_ordered[t] .. range.y) := padding id for every tile t
void launch_mask_unordered(hipStream_t s, uint32_t tiles, const uint32_t* ranges, const uint32_t* tile_ordered, uint32_t* ordered_ids);
void launch_tile_ranges(hipStream_t s, uint32_t m, const uint64_t* sorted_keys, uint32_t* ranges);
// K8 output as the four tensors of the reference's _Autograd.backward instead of one [N,12] tensor (pos == nullptr: [N,12])
struct GradFields {
    float* pos = nullptr;   // [N,3]
    float* dns = nullptr;   // [N,1]
    float* rot = nullptr;   // [N,4] (16-byte aligned)
    float* scl = nullptr;   // [N,3]
    float* alb = nullptr;   // [N,3]  } with spec: the SH gradient as the model's two feature tensors instead of [N,48]
    float* spec = nullptr;  // [N,45] } (16-byte aligned)
};
void launch_project_bwd(hipStream_t s, const ViewParams& v, uint32_t n, int sh_degree, const float* density12,
                        const uint32_t* tiles_count, const float* feat, float* grad16 /* rows read are left zero */,
                        float* density_grad12, float* sph_grad48, bool raw_grads, const GradFields& fields = GradFields());
void launch_pack_fields(hipStream_t s, uint32_t n, const float* pos, const float* dns, const float* rot, const float* scl,
                        float* density12);
// out[row][c] = c < in_width ? in[row][c] : 0 for c < out_width: widens [N, 3 (d+1)^2] radiance rows to the kernels' 48 columns and
// narrows the [N,48] gradient back (render.particle_radiance_sph_degree < 3)
void launch_resize_sph_rows(hipStream_t s, uint32_t n, uint32_t in_width, uint32_t out_width, const float* in, float* out);
void launch_pack_activate_fields(hipStream_t s, uint32_t n, const float* pos, const float* dns_logit, const float* rot_raw,
                                 const float* log_scl, float* act12);   // gut_train.hip (activate_row)

void launch_render(hipStream_t s, const ViewParams& v, const RenderConsts& c, const float* density12,
                   const float* feat, const float* ray_ori, const float* ray_dir, const uint32_t* ranges,
                   const uint32_t* sorted_ids, const uint32_t* d_num_intersections, float* rgba, float* dist, float* hits,
                   uint32_t* tile_traversed, const uint64_t* tile_keys /* lazy order only */, uint32_t* ordered_ids /* NULL = list is fully sorted */,
                   uint32_t* tile_ordered /* lazy order: entries of each tile's list written to ordered_ids */,
                   const uint32_t* tile_order = nullptr /* launch order of the tiles (longest lists first) or NULL */,
                   int kernel_degree = 2 /* render.particle_kernel_degree; != 2 runs the kGeneral instantiation */);
void launch_render_bwd(hipStream_t s, const ViewParams& v, const RenderConsts& c, const float* density12,
                       const float* feat, const float* ray_ori, const float* ray_dir, const uint32_t* ranges,
                       const uint32_t* sorted_ids, const float* rgba, const float* rgba_grad, const float* dist_grad,
                       float* grad16, uint32_t* tile_traversed, const uint32_t* tile_order,
                       const uint32_t* tile_walked /* forward's per-tile traversal depth: the backward stops there */,
                       int kernel_degree = 2);
// the same two launches for render.particle_kernel_degree != 2 (gut_render_general.hip; reached through launch_render / launch_render_bwd)
void launch_render_general(hipStream_t s, const ViewParams& v, const RenderConsts& c, const float* density12, const float* feat,
                           const float* ray_ori, const float* ray_dir, const uint32_t* ranges, const uint32_t* sorted_ids,
                           const uint32_t* d_num_intersections, float* rgba, float* dist, float* hits, uint32_t* tile_traversed,
                           const uint64_t* tile_keys, uint32_t* ordered_ids, uint32_t* tile_ordered, const uint32_t* tile_order,
                           int kernel_degree);
void launch_render_bwd_general(hipStream_t s, const ViewParams& v, const RenderConsts& c, const float* density12, const float* feat,
                               const float* ray_ori, const float* ray_dir, const uint32_t* ranges, const uint32_t* sorted_ids,
                               const float* rgba, const float* rgba_grad, const float* dist_grad, float* grad16,
                               uint32_t* tile_traversed, const uint32_t* tile_order, const uint32_t* tile_walked, int kernel_degree);
void launch_tile_order(hipStream_t s, uint32_t tiles, const uint32_t* traversed, uint32_t* order, const uint32_t* ranges = nullptr,
                       bool by_length = false, uint32_t* walk_sums = nullptr);
// sorted (k_buffer_size > 0) compositor variant, gut_render_sorted.hip
void launch_render_sorted(hipStream_t s, const ViewParams& v, const RenderConsts& c, int K, const float* density12, const float* feat,
                          const float* ray_ori, const float* ray_dir, const uint32_t* ranges, const uint32_t* sorted_ids,
                          const uint32_t* d_num_intersections, float* rgba, float* dist, float* hits, int kernel_degree = 2);
void launch_render_sorted_bwd(hipStream_t s, const ViewParams& v, const RenderConsts& c, int K, const float* density12,
                              const float* feat, const float* ray_ori, const float* ray_dir, const uint32_t* ranges,
                              const uint32_t* sorted_ids, const float* rgba, const float* dist, const float* rgba_grad,
                              const float* dist_grad, float* grad16, bool reference_undo, int kernel_degree = 2);
void launch_project_bwd_compact(hipStream_t s, uint32_t n, const float* density12, const uint32_t* tiles_count,
                                const float* feat, float* grad16 /* rows read are left zero */, float* raw_grad12, float* mrgb);
// fused per-Gaussian backward epilogue + SH-gradient + Adam (gut_train.hip), single view, reads the handle's gradient rows
void launch_sh_adam_from_scratch(hipStream_t s, uint32_t n, int sh_degree, const float* d_camera_position, float* grad16,
                                 const uint32_t* tiles_count, const float* feat, float* raw12, float* raw_m, float* raw_v,
                                 float* sh48, float* sh_m, float* sh_v, const float* lr12, const float* lr48, float beta1, float beta2,
                                 float eps, uint32_t step, const float* visibility, float* act12_out, bool rows_with_tiles_only,
                                 const uint8_t* wave_walked, uint32_t split_block, uint32_t extra_end,
                                 const LazyMoments& lazy, const uint8_t* rule_walked /* per-wave marks valid for EVERY wave, or null */,
                                 float* stat_accum = nullptr, int32_t* stat_denom = nullptr /* gs.py:106-115 statistics, or null */);
// Adam step of the rows that get no gradient this iteration (tiles_count == 0), see k_adam_rows_without_gradient
void launch_compact_gradient_rows(hipStream_t s, uint32_t n, const float* act12, const uint32_t* tiles_count, const float* feat,
                                  float* grad16, float* records, uint32_t capacity, uint32_t* count);
void launch_adam_rows_without_gradient(hipStream_t s, uint32_t n, const uint32_t* tiles_count, float* raw12, float* raw_m, float* raw_v,
                                       float* sh48, float* sh_m, float* sh_v, const float* lr12, const float* lr48, float beta1,
                                       float beta2, float eps, uint32_t step, float* act12_out,
                                       uint32_t block_begin, uint32_t block_end /* range of 256-row blocks */,
                                       const uint8_t* wave_walked, uint32_t split_block, uint32_t extra_end, bool second_launch,
                                       const LazyMoments& lazy);
void launch_count_side_stream_rows(hipStream_t s, uint32_t n, const uint32_t* tiles_count, const uint8_t* wave_walked,
                                   uint32_t split_block, uint32_t extra_end, Counters* out);
void launch_mark_waves_with_tiles(hipStream_t s, uint32_t n, const uint32_t* tiles_count, uint8_t* wave_flags);
void launch_mark_walked_waves(hipStream_t s, uint32_t n, uint32_t tiles, const uint32_t* ranges, const uint32_t* tile_walked,
                              const uint32_t* ids, uint8_t* wave_walked);
void launch_stats_reduce(hipStream_t s, uint32_t n, const uint32_t* tiles_count, uint32_t t, const uint32_t* trav_fwd,
                         const uint32_t* trav_bwd, Counters* out);

// scan / sort (rocPRIM device-wide primitives; temp storage owned by the caller)
size_t scan_temp_bytes(uint32_t n);
hipError_t run_scan(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint32_t n);
size_t sort_temp_bytes(uint32_t m, int end_bit);
hipError_t run_sort(hipStream_t s, void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                    const uint32_t* vals_in, uint32_t* vals_out, uint32_t m, int end_bit);
size_t sort_tiles_temp_bytes(uint32_t m, int end_bit);
hipError_t run_sort_tiles(hipStream_t s, void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                          const uint32_t* vals_in, uint32_t* vals_out, uint32_t m, int end_bit);

}  // namespace gut
