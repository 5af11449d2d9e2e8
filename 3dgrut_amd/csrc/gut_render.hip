// gut_render.hip — per-tile compositing kernels of the 3DGUT path for gfx950 (wave64):
//   K6 render          (reference: gutRenderer.cuh:83-115, gutKBufferRenderer.cuh:108-170,217-292 with K=0,
//                       slang/models/gaussianParticles.slang:96-254, rayPayload.cuh:76-129)
//   K7 render_backward (reference: gutKBufferRenderer.cuh:294-386, models/gaussianParticles.cuh:480-738,
//                       shRadiativeGaussianParticles.cuh:409-482)
//
// Layout: one 256-thread workgroup (4 wave64) per 16x16 tile, one lane per pixel; each wave owns an
// 8x8 block.  The tile's depth-sorted list is consumed in chunks of 256 entries staged in LDS by the
// whole workgroup (one entry per lane, coalesced id read + 48-byte parameter gather), converted once
// to the canonical-space form the inner loop needs (M = diag(1/s) * rotationT, so the per-pixel work
// is two 3x3 mat-vecs, a cross product and one v_exp_f32).  LDS reads in the inner loop are
// wave-uniform (broadcast, conflict-free).
//
// Backward: the 14 per-(pixel, entry) partial derivatives are summed over the 64 lanes of a wave with
// DPP row shifts / row broadcasts (no LDS traffic), then over the 4 waves with ds_add_f32 into a
// per-chunk LDS accumulator, and leave the CU as ONE global float atomic per (tile, entry, component)
// into a 64-byte per-Gaussian row — 8x fewer atomics than the reference's lane-0-per-32-lane-warp
// scheme, and each wave-instruction of the flush covers 4 rows x 16 contiguous floats.  Waves in
// which no lane hit the entry skip the reduction altogether (wave-uniform branch on __ballot).
//
// Three further pieces live in gut_render_common.h: the per-wave strip culling (double-wedge test of each staged entry's
// cut-off ellipsoid against the elevation and azimuth wedges of each wave's 8x8 block; every wave walks its own compacted entry list), the lazy per-tile depth order (the global sort only
// groups by tile; K6 orders 512 entries at a time as far as the tile is walked and hands the ordered ids to K7), and the
// staging form of an entry.  The backward launches its tiles deepest-first (k_tile_order).
//
// Colour/gradient buffers are compared with the oracle by tolerance, so this file is compiled with FMA
// contraction on and uses the hardware exp/rcp/rsq approximations.  Build flags that matter: -fno-slp-vectorize (the
// SLP vectoriser's v_pk_* packing is a net loss here) and the launch bounds of k_render (5 waves per SIMD).
#include "gut_internal.h"
#include "gut_render_common.h"

#include <type_traits>

namespace gut {

// This file is compiled twice: as itself — the default quadratic kernel, `kGeneral` = false everywhere — and through
// gut_render_general.hip (GUT_RENDER_GENERAL_TU), which instantiates the same kernels with kGeneral = true for the reference's other
// generalised Gaussian kernels (render.particle_kernel_degree = 0, 1, 3, 4, 5, 8) under the launcher names *_general.  Two translation
// units, not four instantiations in one: with the extra callers in the unit the inliner stops inlining helpers into the DEFAULT
// forward compositor (31 -> 102 spilled VGPRs, 64 -> 192 bytes of scratch per lane, measured on the kernel-resource remarks).
#ifdef GUT_RENDER_GENERAL_TU
constexpr bool kTuGeneral = true;
#define GUT_LAUNCH_RENDER launch_render_general
#define GUT_LAUNCH_RENDER_BWD launch_render_bwd_general
#else
constexpr bool kTuGeneral = false;
#define GUT_LAUNCH_RENDER launch_render
#define GUT_LAUNCH_RENDER_BWD launch_render_bwd
#endif

#ifdef GUT_CLOCK_STAMPS
// DIAGNOSTIC BUILD ONLY (tools/clock_probe.py builds tools/bin/libgut_hip_stamps.so with -DGUT_CLOCK_STAMPS; the product library
// has none of this): every workgroup of the backward compositor stamps the shader clock (s_memtime, cycles) and the constant
// 100 MHz counter (s_memrealtime) at its start and end into a buffer no other code reads — delta cycles / delta real time is the
// clock the chip actually held during the launch (MI355X_MICROARCH.md, "DVFS give-back" item 6).
__device__ unsigned long long g_clock_stamps[8192][4];
// ... and thread 0 of every workgroup of the FORWARD compositor adds up the cycles it spent in the phases of its chunk loop:
// [0] whole workgroup, [1] lazy selection, [2] staging (id + parameter gather, conversion, up to the barrier), [3] walking the
// chunk (incl. the wait for the slowest wave at the next loop head)
__device__ unsigned long long g_k6_phases[8192][4];
__device__ unsigned long long g_k7_phases[8192][4];   // backward: [0] staging, [1] walk, [2] epilogue + flush, [3] whole workgroup
#define GUT_K6_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define GUT_K6_ADD(slot, a, b) do { if (threadIdx.x == 0) k6_acc[slot] += (b) - (a); } while (0)
#else
#define GUT_K6_STAMP(var) do { } while (0)
#define GUT_K6_ADD(slot, a, b) do { } while (0)
#endif

// Staged entry of the two unsorted compositors: the thirteen floats every (pixel, entry) test needs sit in the first 13 dwords read
// per entry (three ds_read_b128 + one ds_read_b32); the scales — needed by a HIT only (hit distance) — follow as one aligned
// ds_read_b96, the colour and the id as before.  Against FwdEntry's rows [m_i.xyz | s_i] this frees three VGPRs in each of the two
// register sets of the software pipeline (round 4: k_render no longer spills inside its loop).
struct PackEntry {     // 80 bytes
    float4 q0;         // oc = M (sensor_pos - mean), density
    float4 q1;         // m00 m01 m02 m10
    float4 q2;         // m11 m12 m20 m21
    float4 q3;         // s.x s.y s.z m22
    float4 feat_id;    // max(rgb, 0), particle id (bit pattern)
};
__device__ __forceinline__ PackEntry pack_entry(const FwdEntry& e) {
    PackEntry p;
    p.q0 = e.mu_sigma;
    p.q1 = make_float4(e.m0.x, e.m0.y, e.m0.z, e.m1.x);
    p.q2 = make_float4(e.m1.y, e.m1.z, e.m2.x, e.m2.y);
    p.q3 = make_float4(e.m0.w, e.m1.w, e.m2.w, e.m2.z);
    p.feat_id = e.feat_id;
    return p;
}

// stage[j] for a list index held in a VGPR: 24-bit multiply (full rate) instead of the quarter-rate 32-bit one
__device__ __forceinline__ const PackEntry& stage_at(const PackEntry* stage, uint32_t j) {
    return *reinterpret_cast<const PackEntry*>(reinterpret_cast<const char*>(stage) + __umul24(j, (uint32_t)sizeof(PackEntry)));
}

template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, kRowMask, 0xF, true);
    return v + __builtin_bit_cast(float, moved);
}

// sum over the 64 lanes of the wave; the total ends up in lane 63
__device__ __forceinline__ float wave_sum_lane63(float v) {
#ifdef GUT_REDUCE_SHFL
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    return v;
#else
    v = dpp_add<0x111, 0xF>(v);  // row_shr:1
    v = dpp_add<0x112, 0xF>(v);  // row_shr:2
    v = dpp_add<0x114, 0xF>(v);  // row_shr:4
    v = dpp_add<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of every row holds the row total
    v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave total
    return v;
#endif
}

// ---------------------------------------------------------------------------------------------------
// K6 forward
// ---------------------------------------------------------------------------------------------------
// (5 waves per SIMD: at the 103 VGPRs the compiler takes otherwise the kernel runs 7 % slower, at 6 it spills)
// kLazy: sorted_ids / tile_keys are only grouped by tile; the kernel orders each chunk itself (lazy_select) and records the
// ids it consumed in ordered_ids for the backward.
// kGeneral: render.particle_kernel_degree != 2 — the response comes from kernel_response(kernel_degree, .) (gut_render_common.h).
template <bool kLazy, bool kGeneral = false>
__global__ __launch_bounds__(kBlock, 5) void k_render(ViewParams v, RenderConsts c, const float4* __restrict__ density12,
                                                     const float* __restrict__ feat, const float* __restrict__ ray_ori,
                                                     const float* __restrict__ ray_dir, const uint2* __restrict__ ranges,
                                                     const uint32_t* __restrict__ sorted_ids,
                                                     const uint32_t* __restrict__ d_num_intersections,
                                                     float4* __restrict__ rgba, float* __restrict__ dist,
                                                     float* __restrict__ hits, uint32_t* __restrict__ tile_traversed,
                                                     const uint2* __restrict__ tile_keys, uint32_t* __restrict__ ordered_ids,
                                                     uint32_t* __restrict__ tile_ordered, const uint32_t* __restrict__ tile_order,
                                                     int kernel_degree /* read by kGeneral only */) {
    __shared__ PackEntry stage[kBlock];
    __shared__ uint32_t s_deepest, s_first_invalid;
    __shared__ uint32_t s_mask[kBlock];  // per staged entry: which of the four waves (8x8 blocks) can hit it at all
    __shared__ uint16_t s_list[kBlock / 64][kBlock];  // per wave: the staged entries it has to evaluate (index), in list order
    __shared__ StripPlanes s_planes;
    __shared__ LazyOrder s_lazy;
#ifdef GUT_K6_EXTRA_LDS
    // DIAGNOSTIC BUILD ONLY: ballast that lowers the number of resident workgroups per CU (what a second staging area would cost)
    __shared__ float s_ballast6[GUT_K6_EXTRA_LDS / 4];
    if (threadIdx.x == 0 && v.width < 0) s_ballast6[0] = 1.0f;
    if (v.width < -1) dist[0] = s_ballast6[threadIdx.x];
#endif

    const uint32_t tile = tile_order ? tile_order[blockIdx.x] : blockIdx.x;  // (optional) longest lists first
    const uint32_t tid = threadIdx.x;
#ifdef GUT_CLOCK_STAMPS
    unsigned long long k6_acc[4] = {0, 0, 0, 0}, k6_prev = 0;
#endif
    GUT_K6_STAMP(k6_t_begin);
    const int px = (int)(tile % (uint32_t)v.grid_x) * kTile + tile_px(tid);   // wave = 8x8 block (gut_render_common.h)
    const int py = (int)(tile / (uint32_t)v.grid_x) * kTile + tile_py(tid);
    const bool inside = (px < v.width) && (py < v.height);
    const size_t pix = (size_t)py * (size_t)v.width + (size_t)px;
    // zero intersections: the reference returns before rendering and the outputs keep their initial values
    // (gutRenderer.cu:323-325); treating every ray as invalid writes exactly those.  The count is read on the device: the
    // host queues this launch before it knows it.
    const uint32_t num_intersections = *d_num_intersections;
    const RayState ray = make_ray(v, ray_ori, ray_dir, pix, inside && (num_intersections != 0));

    if (tid == 0) {
        s_deepest = 0;
        s_first_invalid = kBlock;
    }
    const bool centred = __syncthreads_and(ray.centred ? 1 : 0) != 0;  // block-uniform
    build_strip_planes(s_planes, ray, inside, centred, tid);
    const uint32_t wave_bit = 1u << (tid >> 6);

    const uint2 range = ranges[tile];
    const uint32_t total = range.y - range.x;
    // The ray's "alive" flag lives in the SIGN of its transmittance (alive <=> T > 0; a ray that ends keeps -T): one v_cmp per entry
    // serves the wave's "anyone left?" branch and the lane mask of the evaluation, where a separate boolean cost the loop five moves
    // and two more compares per entry (ISA of round 3).  T > 0 while alive: alpha <= max_alpha < 1 and T >= min_transmittance before
    // every multiplication.
    float T = ray.valid ? 1.0f : -1.0f, cr = 0.f, cg = 0.f, cb = 0.f, dsum = 0.f;
    uint32_t nhits = 0;
    bool have_lo = false;           // kLazy: the last list entry ordered so far (block-uniform)
    unsigned long long lo = 0;
    uint32_t batch_n = 0, batch_used = 0;

    uint32_t base = 0;  // list entries staged so far (kept after the loop: the ordered prefix handed to the backward)
    // The walk, instantiated twice on the block-uniform `centred` (every camera of the reference has centred rays): the centred form
    // carries neither the M e product nor the registers of e nor the per-entry branch and copies that merging the two forms cost.
    auto walk = [&](auto centred_tag) __attribute__((always_inline)) {
    constexpr bool kCentred = decltype(centred_tag)::value;
    for (; base < total; base += kBlock) {
        if (!__syncthreads_or(T > 0.0f ? 1 : 0)) break;  // whole tile terminated (gutKBufferRenderer.cuh:234-236)
        {
            const uint32_t k = range.x + base + tid;
            uint32_t id = kInvalid;
#ifdef GUT_CLOCK_STAMPS
            const unsigned long long k6_tA = __builtin_amdgcn_s_memtime();
            if (k6_prev) GUT_K6_ADD(3, k6_prev, k6_tA);    // walking the previous chunk + the wait for its slowest wave
            unsigned long long k6_tB = k6_tA;
#endif
            if (kLazy) {
                if (batch_used == batch_n) {  // block-uniform: order the next kLazyBatch entries of the tile
                    batch_n = min(kLazyBatch, total - base);
                    // (the staging area is free here: every wave has left the previous chunk at the barrier above)
                    static_assert(sizeof(stage) >= kLazyCache * sizeof(uint32_t), "depth cache aliases the staging area");
                    lazy_select(s_lazy, reinterpret_cast<uint32_t*>(stage), tile_keys + range.x, total, batch_n, have_lo, lo, tid);
#ifdef GUT_CLOCK_STAMPS
                    k6_tB = __builtin_amdgcn_s_memtime();
                    GUT_K6_ADD(1, k6_tA, k6_tB);
#endif
                    have_lo = true;
                    lo = s_lazy.sel[batch_n - 1];
                    batch_used = 0;
                }
                const uint32_t take = min((uint32_t)kBlock, batch_n - batch_used);
                if (tid < take) {
                    id = sorted_ids[range.x + (uint32_t)s_lazy.sel[batch_used + tid]];
                    ordered_ids[k] = id;
                }
                batch_used += take;
            } else if (k < range.y) {
                id = sorted_ids[k];
            }
            FwdEntry e;
            uint32_t strips = 0xFu;
            e.feat_id.w = __uint_as_float(id);
            if (id != kInvalid) {
                const float4 a = density12[3 * (size_t)id + 0];
                const float4 q = density12[3 * (size_t)id + 1];
                const float4 s = density12[3 * (size_t)id + 2];
                float r[3][3];
                quat_rows(q.x, q.y, q.z, q.w, r);
                const float i0 = 1.0f / s.x, i1 = 1.0f / s.y, i2 = 1.0f / s.z;
                e.m0 = make_float4(r[0][0] * i0, r[0][1] * i0, r[0][2] * i0, s.x);
                e.m1 = make_float4(r[1][0] * i1, r[1][1] * i1, r[1][2] * i1, s.y);
                e.m2 = make_float4(r[2][0] * i2, r[2][1] * i2, r[2][2] * i2, s.z);
                const float c0 = v.s2w.t[0] - a.x, c1 = v.s2w.t[1] - a.y, c2 = v.s2w.t[2] - a.z;
                e.mu_sigma = make_float4(e.m0.x * c0 + e.m0.y * c1 + e.m0.z * c2, e.m1.x * c0 + e.m1.y * c1 + e.m1.z * c2,
                                         e.m2.x * c0 + e.m2.y * c1 + e.m2.z * c2, a.w);
                e.feat_id.x = fmaxf(feat[3 * (size_t)id + 0], 0.0f);
                e.feat_id.y = fmaxf(feat[3 * (size_t)id + 1], 0.0f);
                e.feat_id.z = fmaxf(feat[3 * (size_t)id + 2], 0.0f);
                strips = strip_mask<kGeneral>(s_planes, v, c, a, r, s, kernel_degree);
            }
            stage[tid] = pack_entry(e);
            s_mask[tid] = strips;
            if (id == kInvalid && k < range.y) atomicMin(&s_first_invalid, tid);
#ifdef GUT_CLOCK_STAMPS
            __syncthreads();
            k6_prev = __builtin_amdgcn_s_memtime();
            GUT_K6_ADD(2, k6_tB, k6_prev);
#endif
        }
        __syncthreads();

        // entries of this chunk that are real particles (padding ids end the list for everyone, gutKBufferRenderer.cuh:256-259)
        const uint32_t cnt = min(min((uint32_t)kBlock, total - base), s_first_invalid);
        // software pipeline: the parameters of entry j+1 are fetched from LDS while entry j is evaluated
        auto entry = [&](const float4& cs, const float4& q1, const float4& q2, const float m22, const uint32_t j)
                         __attribute__((always_inline)) {
            if (T > 0.0f) {
                float o0 = cs.x, o1 = cs.y, o2 = cs.z;
                if (!kCentred) {
                    o0 += q1.x * ray.ex + q1.y * ray.ey + q1.z * ray.ez;
                    o1 += q1.w * ray.ex + q2.x * ray.ey + q2.y * ray.ez;
                    o2 += q2.z * ray.ex + q2.w * ray.ey + m22 * ray.ez;
                }
                const float u0 = q1.x * ray.dx + q1.y * ray.dy + q1.z * ray.dz;
                const float u1 = q1.w * ray.dx + q2.x * ray.dy + q2.y * ray.dz;
                const float u2 = q2.z * ray.dx + q2.w * ray.dy + m22 * ray.dz;
                const float x0 = u1 * o2 - u2 * o1, x1 = u2 * o0 - u0 * o2, x2 = u0 * o1 - u1 * o0;
                const float l2 = u0 * u0 + u1 * u1 + u2 * u2;
                const float il2 = fast_rcp(l2);
                const float d2 = (x0 * x0 + x1 * x1 + x2 * x2) * il2;  // |grd x gro|^2 with grd = u/|u|
                if (d2 < c.max_d2) {
                    const float resp = kGeneral ? kernel_response(kernel_degree, d2) : fast_exp(-0.5f * d2);
                    const float alpha = fminf(c.max_alpha, resp * cs.w);
                    if ((resp > c.min_response) && (alpha > c.alpha_threshold)) {
                        // hitT = | s * grd * (grd . -gro) |
                        const float proj = -(u0 * o0 + u1 * o1 + u2 * o2) * il2;  // (grd.-gro)/|u|
                        const float4 sc = stage_at(stage, j).q3;                   // the scales: read by hits only
                        const float h0 = sc.x * u0 * proj, h1 = sc.y * u1 * proj, h2 = sc.z * u2 * proj;
                        const float hit_t = fast_sqrt(h0 * h0 + h1 * h1 + h2 * h2);
                        if ((hit_t > ray.tmin) && (hit_t < ray.tmax)) {
                            const float w = alpha * T;
                            dsum += hit_t * w;
                            T *= (1.0f - alpha);
                            if (w > 0.0f) {
                                const float4 fid = stage_at(stage, j).feat_id;
                                cr += fid.x * w;
                                cg += fid.y * w;
                                cb += fid.z * w;
                                nhits++;
                            }
                            if (T < c.min_transmittance) {
                                T = -T;                                   // the ray ends here
                                atomicMax(&s_deepest, base + j + 1);      // list position at which it ended (once per ray)
                            }
                        }
                    }
                }
            }
        };
        // The wave first compacts the chunk to the entries whose cut-off ellipsoid can reach its 8x8 block (ballot prefix,
        // indices into a wave-private LDS list), then walks that list.  Software pipeline, unrolled by two (two alternating
        // register sets, no per-iteration moves): the parameters of the next listed entry are fetched from LDS while the
        // current one is evaluated.
        {
            uint16_t* list = s_list[tid >> 6];
            const uint32_t lane = tid & 63u;
            uint32_t nw = 0;
#pragma unroll
            for (uint32_t r = 0; r < kBlock / 64; ++r) {
                const uint32_t e = r * 64u + lane;
                const bool keep = (e < cnt) && ((s_mask[e] & wave_bit) != 0u);
                const unsigned long long bal = __ballot(keep);
                if (keep) list[nw + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)e;
                nw += (uint32_t)__popcll(bal);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t ja = list[0], jb = list[1];  // list positions i and i+1 (garbage beyond nw is never evaluated)
            float4 a0 = stage_at(stage, ja).q0, a1 = stage_at(stage, ja).q1, a2 = stage_at(stage, ja).q2;
            float a3 = stage_at(stage, ja).q3.w;
            float4 b0, b1, b2;
            float b3;
            uint32_t i = 0;
            while (i < nw) {
                if (__ballot(T > 0.0f) == 0ull) break;  // wave-uniform
                b0 = stage_at(stage, jb).q0; b1 = stage_at(stage, jb).q1; b2 = stage_at(stage, jb).q2; b3 = stage_at(stage, jb).q3.w;
                const uint32_t jc = list[min(i + 2, (uint32_t)kBlock - 1)];
                entry(a0, a1, a2, a3, ja);
                if (++i >= nw) break;
                if (__ballot(T > 0.0f) == 0ull) break;
                a0 = stage_at(stage, jc).q0; a1 = stage_at(stage, jc).q1; a2 = stage_at(stage, jc).q2; a3 = stage_at(stage, jc).q3.w;
                const uint32_t jd = list[min(i + 2, (uint32_t)kBlock - 1)];
                entry(b0, b1, b2, b3, jb);
                ++i;
                ja = jc;
                jb = jd;
            }
            // a ray still alive has walked the whole chunk (skipped entries included)
            if (__ballot(T > 0.0f) != 0ull && lane == 0) atomicMax(&s_deepest, base + cnt);
        }
        if (cnt < min((uint32_t)kBlock, total - base)) T = -fabsf(T);  // list ended at a padding entry
    }
    };
    if (centred) walk(std::true_type{}); else walk(std::false_type{});

    if (inside) {
        if (ray.valid) {
            rgba[pix] = make_float4(cr, cg, cb, 1.0f - fabsf(T));
            dist[pix] = dsum;
            hits[pix] = (float)nhits;
        } else {  // initial values of the reference's output tensors (splatRaster.cpp:196-198)
            rgba[pix] = make_float4(0.f, 0.f, 0.f, 0.f);
            dist[pix] = 1e06f;
            hits[pix] = 0.0f;
        }
    }
    // traversal statistics (E_f of the roofline model): deepest list position any pixel of the tile consumed (s_deepest)
    __syncthreads();
#ifdef GUT_CLOCK_STAMPS
    if (tid == 0 && blockIdx.x < 8192) {
        const unsigned long long k6_t_end = __builtin_amdgcn_s_memtime();
        if (k6_prev) k6_acc[3] += k6_t_end - k6_prev;
        g_k6_phases[blockIdx.x][0] = k6_t_end - k6_t_begin;
        g_k6_phases[blockIdx.x][1] = k6_acc[1]; g_k6_phases[blockIdx.x][2] = k6_acc[2]; g_k6_phases[blockIdx.x][3] = k6_acc[3];
    }
#endif
    if (tid == 0) {
        tile_traversed[tile] = s_deepest;
        if (kLazy) tile_ordered[tile] = min(base, total);  // the backward reads ordered_ids[range.x .. range.x + that) only
    }
}

// ---------------------------------------------------------------------------------------------------
// K7 backward
//
// Per (pixel, particle) the reference chains dL/d(alpha) through cross/normalise/scale/rotation down to
// position, scale and quaternion inside the pixel loop (models/gaussianParticles.cuh:516-737) and reduces 11+3
// floats per pair.  Here the chain is cut at the canonical-space matrix M = diag(1/s) * rotationT:
//     o = M (ray_o - mu),  u = M ray_d,  d2 = |u x o|^2 / |u|^2
//     dL/do = 2 g_d2 * o_perp,   dL/du = -2 g_d2 * t * o_perp,   o_perp = o - t u,  t = (u.o)/|u|^2
// so dL/dM = h (x) m is rank one with h = 2 g_d2 o_perp and m = (ray_o - mu) - t ray_d, and dL/dmu = -M^T h.
// The pixel loop only accumulates  A += h m^T (9), H += h (3), d(density) (1), d(rgb) (3)  = 16 floats; the
// conversion of (A, H) into d(position), d(scale), d(quaternion) (matmul_bw_quat, mathUtils.cuh:468-533) runs
// once per (tile, entry) in a per-chunk epilogue, one entry per lane.  The hit-distance gradient terms of the
// reference (non-zero only when a loss is put on pred_dist) break the rank-one form; they live in the kDistGrad
// instantiation, which accumulates the general g_o p^T + g_u d^T and three extra direct scale terms.
//
// Reduction over the 64 pixels of a wave: a DPP transpose-reduce (4 exchange steps inside each 16-lane row in
// which every lane keeps half of its values and adds its partner's copy of them, then two row broadcasts) —
// ~50 VALU ops for 16 values instead of 16 x 6; lanes 48..63 end up holding one total each and issue ONE
// ds_add_f32 into the chunk accumulator.  Waves in which no lane hit the entry skip all of it.
// ---------------------------------------------------------------------------------------------------
template <int kCtrl, int kRowMask = 0xF, int kBankMask = 0xF>
__device__ __forceinline__ float dpp_mov(float oldv, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, oldv), __builtin_bit_cast(int, v),
                                                                 kCtrl, kRowMask, kBankMask, false));
}

// Sums each of the 16 values of v[] over the 64 lanes of the wave ("transpose-reduce": every exchange step halves the
// number of live registers).  gfx950's half-wave / row swaps do the first two steps without any select:
//   v_permlane32_swap a, b : a = [a.lanes0-31 | b.lanes0-31],  b = [a.lanes32-63 | b.lanes32-63]  -> a + b folds the halves
//   v_permlane16_swap a, b : a = [a.row0, b.row0, a.row2, b.row2], b = [a.row1, b.row1, a.row3, b.row3] (rows of 16 lanes)
// after which row r holds, per lane column, the 4-row partial sums of v[i + 4r] in register i (i = 0..3); two
// select+DPP steps inside each row and two row rotations finish it.  35 VALU ops for 16 values.
// On return every lane holds the wave total of v[reduce_slot(lane)]; lanes with (lane & 12) == 0 are the 16 distinct ones.
// (inline asm rather than __builtin_amdgcn_permlane{32,16}_swap: with ROCm 7.2's hipcc, adding the two results of the
//  builtin folds to r[0] + r[0] at -O3.  The leading s_nop 1 covers the "VALU write -> v_permlane*_swap read" hazard
//  (2 wait states) for the whole group: every operand is written before the group starts.)
__device__ __forceinline__ void permlane32_swap_x8(float (&a)[16]) {
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %8\n\t"
        "v_permlane32_swap_b32 %1, %9\n\t"
        "v_permlane32_swap_b32 %2, %10\n\t"
        "v_permlane32_swap_b32 %3, %11\n\t"
        "v_permlane32_swap_b32 %4, %12\n\t"
        "v_permlane32_swap_b32 %5, %13\n\t"
        "v_permlane32_swap_b32 %6, %14\n\t"
        "v_permlane32_swap_b32 %7, %15"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),
          "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
}
__device__ __forceinline__ void permlane16_swap_x4(float (&a)[8]) {
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %4\n\t"
        "v_permlane16_swap_b32 %1, %5\n\t"
        "v_permlane16_swap_b32 %2, %6\n\t"
        "v_permlane16_swap_b32 %3, %7"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}

__device__ __forceinline__ float wave_transpose_reduce16(const float (&v)[16], uint32_t lane) {
    float t[16], x[8], y[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = v[i];
    permlane32_swap_x8(t);  // pairs (v[i], v[i+8])
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = t[i] + t[i + 8];
    permlane16_swap_x4(x);  // pairs (x[i], x[i+4])
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = x[i] + x[i + 4];
    const bool b0 = lane & 1, b1 = lane & 2;
    float z[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // partner = lane ^ 1 (quad_perm [1,0,3,2]): keep y[2i + b0]
        const float keep = b0 ? y[2 * i + 1] : y[2 * i];
        const float send = b0 ? y[2 * i] : y[2 * i + 1];
        z[i] = keep + dpp_mov<0xB1>(0.f, send);
    }
    float w;
    {  // partner = lane ^ 2 (quad_perm [2,3,0,1]): keep z[b1]
        const float keep = b1 ? z[1] : z[0];
        const float send = b1 ? z[0] : z[1];
        w = keep + dpp_mov<0x4E>(0.f, send);
    }
    w += dpp_mov<0x124>(0.f, w);  // row_ror:4
    w += dpp_mov<0x128>(0.f, w);  // row_ror:8 -> all four quads of the row hold the row's totals
    return w;
}

// value index whose wave total wave_transpose_reduce16 leaves in this lane
__device__ __forceinline__ uint32_t reduce_slot(uint32_t lane) { return (lane & 1u) + 2u * ((lane >> 1) & 1u) + 4u * (lane >> 4); }

constexpr int kGradRow = 16;  // floats per global gradient row: pos3, density, quat4, scale3, rgb3, pad2

// accumulator slots inside a chunk row
//   0..8  A[i][j] = sum h_i m_j     9..11 H_i = sum h_i     12 d(density)     13..15 d(rgb)     [16..18 direct d(scale)]
template <bool kDistGrad>
struct AccLayout {
    static constexpr int kW = kDistGrad ? 20 : 16;
};

template <bool kDistGrad, bool kGeneral = false>
__global__ __launch_bounds__(kBlock, kDistGrad ? 3 : 4) void k_render_backward(ViewParams v, RenderConsts c,
                                                           const float4* __restrict__ density12,
                                                           const float* __restrict__ feat,
                                                           const float* __restrict__ ray_ori,
                                                           const float* __restrict__ ray_dir,
                                                           const uint2* __restrict__ ranges,
                                                           const uint32_t* __restrict__ sorted_ids,
                                                           const float4* __restrict__ rgba,
                                                           const float4* __restrict__ rgba_grad,
                                                           const float* __restrict__ dist_grad, float* __restrict__ grad16,
                                                           uint32_t* __restrict__ tile_traversed,
                                                           const uint32_t* __restrict__ tile_order,
                                                           const uint32_t* __restrict__ tile_walked,
                                                           int kernel_degree /* read by kGeneral only */) {
    constexpr int W = AccLayout<kDistGrad>::kW;
    __shared__ PackEntry stage[kBlock];
    __shared__ float acc[kBlock * W];
    __shared__ uint32_t s_deepest, s_first_invalid;
    __shared__ uint32_t s_mask[kBlock];              // see k_render
    __shared__ uint16_t s_list[kBlock / 64][kBlock];  // see k_render
    __shared__ StripPlanes s_planes;
#ifdef GUT_K7_EXTRA_LDS
    // DIAGNOSTIC BUILD ONLY: ballast that lowers the number of resident workgroups per CU (occupancy experiment, tools/clock_probe.py)
    __shared__ float s_ballast[GUT_K7_EXTRA_LDS / 4];
    if (threadIdx.x == 0 && v.width < 0) s_ballast[0] = 1.0f;
    if (v.width < -1) grad16[0] = s_ballast[threadIdx.x];
#endif

    const uint32_t tile = tile_order ? tile_order[blockIdx.x] : blockIdx.x;  // deepest tiles are dispatched first
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
#ifdef GUT_CLOCK_STAMPS
    unsigned long long k7_acc[3] = {0, 0, 0};
    if (tid == 0 && blockIdx.x < 8192) {
        g_clock_stamps[blockIdx.x][0] = __builtin_amdgcn_s_memtime();
        g_clock_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const int px = (int)(tile % (uint32_t)v.grid_x) * kTile + tile_px(tid);   // wave = 8x8 block (gut_render_common.h)
    const int py = (int)(tile / (uint32_t)v.grid_x) * kTile + tile_py(tid);
    const bool inside = (px < v.width) && (py < v.height);
    const size_t pix = (size_t)py * (size_t)v.width + (size_t)px;
    const RayState ray = make_ray(v, ray_ori, ray_dir, pix, inside);

    if (tid == 0) {
        s_deepest = 0;
        s_first_invalid = kBlock;
    }
    const bool centred = __syncthreads_and(ray.centred ? 1 : 0) != 0;  // block-uniform
    build_strip_planes(s_planes, ray, inside, centred, tid);
    const uint32_t wave_bit = 1u << (tid >> 6);
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k * kBlock + tid] = 0.0f;

    // forward results and upstream gradients of this pixel (rayPayloadBackward.cuh:30-58)
    float T_final = 1.f, Tg = 0.f, fr = 0.f, fg = 0.f, fb = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f;
    if (ray.valid) {
        const float4 o = rgba[pix];
        const float4 g = rgba_grad[pix];
        T_final = 1.0f - o.w;
        Tg = -g.w;  // transmittanceGradient = -dL/d(opacity)
        fr = o.x; fg = o.y; fb = o.z;
        gr = g.x; gg = g.y; gb = g.z;
        if (kDistGrad) gd = dist_grad[pix];
    }

    // out-of-image lanes (ragged right/bottom tiles) never hit, but they run the gradient arithmetic with zero weights
    // once any lane of their wave hits: give them a unit direction so that arithmetic stays finite
    const float dir2 = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
    const float ori1 = fabsf(ray.ex) + fabsf(ray.ey) + fabsf(ray.ez);
    const bool usable = inside && (dir2 > 0.0f) && (dir2 < 3.0e38f) && (ori1 < 3.0e38f);  // false for NaN / inf / zero rays
    const float rdx = usable ? ray.dx : 0.0f, rdy = usable ? ray.dy : 0.0f, rdz = usable ? ray.dz : 1.0f;
    const float rex = usable ? ray.ex : 0.0f, rey = usable ? ray.ey : 0.0f, rez = usable ? ray.ez : 0.0f;

    uint2 range = ranges[tile];
    // The backward walks no further than the forward did: tile_walked[tile] is the deepest list position any pixel of the tile
    // consumed in k_render (its whole-tile termination point).  With the lazy order that is also what makes sorted_ids — the
    // forward's ordered-id list — valid: only the prefix the forward staged is ordered.  The rays terminate by the same rule in
    // both kernels, so the bound only bites where the two evaluations of a transmittance disagree in the last bit; what it buys
    // is a guarantee: a Gaussian outside the walked prefixes receives no gradient (the side-stream optimiser pass relies on it).
    range.y = range.x + min(range.y - range.x, tile_walked[tile]);
    const uint32_t total = range.y - range.x;
    const uint32_t my_slot = reduce_slot(lane);
    bool alive = ray.valid && usable;
    float T = 1.0f, rr = 0.f, rg = 0.f, rb = 0.f;  // running transmittance / radiance
    uint32_t consumed = 0;
    // the walk, instantiated on the block-uniform `centred` (see k_render)
    auto walk = [&](auto centred_tag) __attribute__((always_inline)) {
    constexpr bool kCentred = decltype(centred_tag)::value;
    for (uint32_t base = 0; base < total; base += kBlock) {
        if (!__syncthreads_or(alive ? 1 : 0)) break;
#ifdef GUT_CLOCK_STAMPS
        const unsigned long long k7_tA = __builtin_amdgcn_s_memtime();
#endif
        {
            const uint32_t k = range.x + base + tid;
            uint32_t id = kInvalid;
            if (k < range.y) id = sorted_ids[k];
            FwdEntry e;
            uint32_t strips = 0xFu;
            e.feat_id.w = __uint_as_float(id);
            if (id != kInvalid) {
                const float4 a = density12[3 * (size_t)id + 0];
                const float4 q = density12[3 * (size_t)id + 1];
                const float4 sc = density12[3 * (size_t)id + 2];
                float r[3][3];
                quat_rows(q.x, q.y, q.z, q.w, r);
                const float i0 = 1.0f / sc.x, i1 = 1.0f / sc.y, i2 = 1.0f / sc.z;
                e.m0 = make_float4(r[0][0] * i0, r[0][1] * i0, r[0][2] * i0, sc.x);
                e.m1 = make_float4(r[1][0] * i1, r[1][1] * i1, r[1][2] * i1, sc.y);
                e.m2 = make_float4(r[2][0] * i2, r[2][1] * i2, r[2][2] * i2, sc.z);
                const float c0 = v.s2w.t[0] - a.x, c1 = v.s2w.t[1] - a.y, c2 = v.s2w.t[2] - a.z;
                e.mu_sigma = make_float4(e.m0.x * c0 + e.m0.y * c1 + e.m0.z * c2, e.m1.x * c0 + e.m1.y * c1 + e.m1.z * c2,
                                         e.m2.x * c0 + e.m2.y * c1 + e.m2.z * c2, a.w);
                e.feat_id.x = fmaxf(feat[3 * (size_t)id + 0], 0.0f);
                e.feat_id.y = fmaxf(feat[3 * (size_t)id + 1], 0.0f);
                e.feat_id.z = fmaxf(feat[3 * (size_t)id + 2], 0.0f);
                strips = strip_mask<kGeneral>(s_planes, v, c, a, r, sc, kernel_degree);
            }
            stage[tid] = pack_entry(e);
            s_mask[tid] = strips;
            if (id == kInvalid && k < range.y) atomicMin(&s_first_invalid, tid);
        }
        __syncthreads();
#ifdef GUT_CLOCK_STAMPS
        const unsigned long long k7_tB = __builtin_amdgcn_s_memtime();
#endif

        const uint32_t cnt_all = min((uint32_t)kBlock, total - base);
        const uint32_t cnt = min(cnt_all, s_first_invalid);  // padding ids end the list for everyone
        // one list entry for this wave's 64 pixels
        // (p1, p2, p3: PackEntry.q1 / q2 / q3; q3 = scales + m22; only the hit-distance instantiation looks at the scales)
        auto entry = [&](const float4& ms, const float4& p1, const float4& p2, const float4& p3, const uint32_t j)
                         __attribute__((always_inline)) {
            float o0 = ms.x, o1 = ms.y, o2 = ms.z;
            if (!kCentred) {
                o0 += p1.x * rex + p1.y * rey + p1.z * rez;
                o1 += p1.w * rex + p2.x * rey + p2.y * rez;
                o2 += p2.z * rex + p2.w * rey + p3.w * rez;
            }
            const float u0 = p1.x * rdx + p1.y * rdy + p1.z * rdz;
            const float u1 = p1.w * rdx + p2.x * rdy + p2.y * rdz;
            const float u2 = p2.z * rdx + p2.w * rdy + p3.w * rdz;
            const float c0 = u1 * o2 - u2 * o1, c1 = u2 * o0 - u0 * o2, c2 = u0 * o1 - u1 * o0;
            const float l2 = u0 * u0 + u1 * u1 + u2 * u2;
            const float il2 = fast_rcp(l2);
            const float d2 = (c0 * c0 + c1 * c1 + c2 * c2) * il2;
            const float resp = kGeneral ? kernel_response(kernel_degree, d2) : fast_exp(-0.5f * d2);
            const float a0 = resp * ms.w;
            // the literal 0.99 of processHitBwd (gaussianParticles.cuh:528), NOT render.particle_kernel_max_alpha: the reference's forward
            // clamps with the configured value (slang particleDensityHit), its unsorted backward with this constant
            const float alpha = fminf(0.99f, a0);
            // NB: no tmin/tmax test in the backward
            const bool hit = alive && (d2 < c.max_d2) && (resp > c.min_response) && (alpha > c.alpha_threshold);
            if (__ballot(hit) == 0ull) return;  // wave-uniform: no lane of this wave hit entry j

            // From here every lane runs the same instruction stream; lanes that did not hit carry alpha = 0 and a zero
            // upstream factor, which makes all 16 partial derivatives exactly zero without per-value selects (all their
            // intermediates are finite: rays of out-of-image lanes were given a unit direction above).
            const float am = hit ? alpha : 0.0f;
            const float rm = hit ? resp : 0.0f;
            float g[16];
            float gs0 = 0.f, gs1 = 0.f, gs2 = 0.f;  // direct scale terms (kDistGrad only)
            const float4 fid = stage_at(stage, j).feat_id;
            const float w = am * T;
            const float Tn = (1.0f - am) * T;
            const float t = (u0 * o0 + u1 * o1 + u2 * o2) * il2;  // (u.o)/|u|^2
            const float q0 = o0 - t * u0, q1 = o1 - t * u1, q2 = o2 - t * u2;  // o_perp
            float ga_hit = 0.f;
            float k0 = 0.f, k1 = 0.f, k2 = 0.f, d0 = 0.f, d1 = 0.f, d2n = 0.f, il = 0.f;
            if (kDistGrad) {
                // hit-distance terms (residualHitT == 0: quirk 1 of SURVEY §8a); grd = u/|u|
                il = fast_rsq(l2);
                d0 = u0 * il; d1 = u1 * il; d2n = u2 * il;
                const float proj = -(d0 * o0 + d1 * o1 + d2n * o2);
                const float s0 = p3.x * d0 * proj, s1 = p3.y * d1 * proj, s2 = p3.z * d2n * proj;  // grds
                const float gsq = s0 * s0 + s1 * s1 + s2 * s2;
                const float gdist = fast_sqrt(gsq);
                ga_hit = gdist * T * gd;
                if (gsq > 0.0f) {
                    const float kk = (w / gdist) * gd;
                    k0 = s0 * kk; k1 = s1 * kk; k2 = s2 * kk;  // grdsRayHitGrd
                }
                gs0 = d0 * proj * k0; gs1 = d1 * proj * k1; gs2 = d2n * proj * k2;  // gsclRayHitGrd
            }
            const float res_T = am < 0.999999f ? T_final * fast_rcp(1.0f - am) : T;
            const float ga_dns = res_T * -Tg;
            g[13] = gr * w; g[14] = gg * w; g[15] = gb * w;
            rr += w * fid.x; rg += w * fid.y; rb += w * fid.z;
            float e0 = 0.f, e1 = 0.f, e2 = 0.f;  // residual radiance behind this particle
            if (!(Tn <= c.min_transmittance)) {
                const float iT = fast_rcp(Tn);
                e0 = fmaxf((fr - rr) * iT, 0.0f);
                e1 = fmaxf((fg - rg) * iT, 0.0f);
                e2 = fmaxf((fb - rb) * iT, 0.0f);
            }
            const float Gall = ga_hit + ga_dns + T * ((fid.x - e0) * gr + (fid.y - e1) * gg + (fid.z - e2) * gb);
            const float G = hit ? Gall : 0.0f;
            g[12] = rm * G;                            // d density
            const float g_d2x2 = kGeneral ? 2.0f * kernel_response_grad(kernel_degree, d2, rm, ms.w * G)
                                          : -(rm * ms.w) * G;     // 2 * dL/d(d2) = 2 * (-1/2 resp sigma G)
            float h0 = g_d2x2 * q0, h1 = g_d2x2 * q1, h2 = g_d2x2 * q2;  // dL/do
            if (!kDistGrad) {
                // m = (ray_o - mu) - t d = (e - t d) + (sensor_pos - mu); the second, per-entry constant part is
                // added in the epilogue as H (x) (sensor_pos - mu)
                const float n0 = (kCentred ? 0.0f : rex) - t * rdx, n1 = (kCentred ? 0.0f : rey) - t * rdy, n2 = (kCentred ? 0.0f : rez) - t * rdz;
                g[0] = h0 * n0; g[1] = h0 * n1; g[2] = h0 * n2;
                g[3] = h1 * n0; g[4] = h1 * n1; g[5] = h1 * n2;
                g[6] = h2 * n0; g[7] = h2 * n1; g[8] = h2 * n2;
            } else {
                float v0 = -t * h0, v1 = -t * h1, v2 = -t * h2;  // dL/du from the response term
                // reference's diagonal hit-distance terms: d/d(gro) and d/d(grd) (gaussianParticles.cuh:559-567)
                const float x0 = d0 * o0, x1 = d1 * o1, x2 = d2n * o2;
                h0 += -p3.x * d0 * d0 * k0;
                h1 += -p3.y * d1 * d1 * k1;
                h2 += -p3.z * d2n * d2n * k2;
                const float hd0 = -p3.x * (2.0f * x0 + x1 + x2) * k0;
                const float hd1 = -p3.y * (x0 + 2.0f * x1 + x2) * k1;
                const float hd2 = -p3.z * (x0 + x1 + 2.0f * x2) * k2;
                const float dot = hd0 * d0 + hd1 * d1 + hd2 * d2n;  // safe_normalize_bw
                v0 += il * (hd0 - d0 * dot);
                v1 += il * (hd1 - d1 * dot);
                v2 += il * (hd2 - d2n * dot);
                const float ex_ = kCentred ? 0.0f : rex, ey_ = kCentred ? 0.0f : rey, ez_ = kCentred ? 0.0f : rez;
                g[0] = h0 * ex_ + v0 * rdx; g[1] = h0 * ey_ + v0 * rdy; g[2] = h0 * ez_ + v0 * rdz;
                g[3] = h1 * ex_ + v1 * rdx; g[4] = h1 * ey_ + v1 * rdy; g[5] = h1 * ez_ + v1 * rdz;
                g[6] = h2 * ex_ + v2 * rdx; g[7] = h2 * ey_ + v2 * rdy; g[8] = h2 * ez_ + v2 * rdz;
            }
            g[9] = h0; g[10] = h1; g[11] = h2;
            T = Tn;  // unchanged for lanes that did not hit
            if (hit && (T < c.min_transmittance)) {
                alive = false;
                consumed = base + j + 1;  // list position at which this ray terminated
            }

            const float r = wave_transpose_reduce16(g, lane);
            if ((lane & 12u) == 0u) atomicAdd(&acc[j * W + my_slot], r);  // 16 lanes, 16 distinct addresses
            if (kDistGrad) {
                const float t0 = wave_sum_lane63(gs0), t1 = wave_sum_lane63(gs1), t2 = wave_sum_lane63(gs2);
                if (lane == 63) {
                    atomicAdd(&acc[j * W + 16], t0);
                    atomicAdd(&acc[j * W + 17], t1);
                    atomicAdd(&acc[j * W + 18], t2);
                }
            }
        };
        // as in k_render: the wave compacts the chunk to the entries that can reach its strip, then walks that list with the
        // two-register-set software pipeline
        {
            uint16_t* list = s_list[tid >> 6];
            uint32_t nw = 0;
#pragma unroll
            for (uint32_t r = 0; r < kBlock / 64; ++r) {
                const uint32_t e = r * 64u + lane;
                const bool keep = (e < cnt) && ((s_mask[e] & wave_bit) != 0u);
                const unsigned long long bal = __ballot(keep);
                if (keep) list[nw + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)e;
                nw += (uint32_t)__popcll(bal);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t ja = list[0], jb = list[1];
            // (without the hit-distance terms only q3.w = m22 is used: the compiler keeps one dword of the fourth read)
            float4 a0 = stage_at(stage, ja).q0, a1 = stage_at(stage, ja).q1, a2 = stage_at(stage, ja).q2, a3 = stage_at(stage, ja).q3;
            float4 b0, b1, b2, b3;
            uint32_t i = 0;
            while (i < nw) {
                if (__ballot(alive) == 0ull) break;
                b0 = stage_at(stage, jb).q0; b1 = stage_at(stage, jb).q1; b2 = stage_at(stage, jb).q2; b3 = stage_at(stage, jb).q3;
                const uint32_t jc = list[min(i + 2, (uint32_t)kBlock - 1)];
                entry(a0, a1, a2, a3, ja);
                if (++i >= nw) break;
                if (__ballot(alive) == 0ull) break;
                a0 = stage_at(stage, jc).q0; a1 = stage_at(stage, jc).q1; a2 = stage_at(stage, jc).q2; a3 = stage_at(stage, jc).q3;
                const uint32_t jd = list[min(i + 2, (uint32_t)kBlock - 1)];
                entry(b0, b1, b2, b3, jb);
                ++i;
                ja = jc;
                jb = jd;
            }
            if (alive) consumed = base + cnt;  // walked the whole chunk (skipped entries included) and is still alive
        }

        if (cnt < cnt_all) alive = false;  // list ended at a padding entry
        // ---- chunk epilogue: (A, H) -> d(position), d(scale), d(quaternion), one entry per lane ----
        __syncthreads();
#ifdef GUT_CLOCK_STAMPS
        const unsigned long long k7_tC = __builtin_amdgcn_s_memtime();
#endif
        if (tid < cnt) {
            float* a = &acc[tid * W];
            const uint32_t id = __float_as_uint(stage[tid].feat_id.w);
            bool any = false;
#pragma unroll
            for (int k = 0; k < W; ++k) any = any || (a[k] != 0.0f);
            if (any && id != kInvalid) {
                const float4 mu = density12[3 * (size_t)id + 0];
                const float4 q = density12[3 * (size_t)id + 1];
                const float4 sc = density12[3 * (size_t)id + 2];
                float r[3][3];
                quat_rows(q.x, q.y, q.z, q.w, r);
                const float is[3] = {1.0f / sc.x, 1.0f / sc.y, 1.0f / sc.z};
                const float H[3] = {a[9], a[10], a[11]};
                const float pc[3] = {v.s2w.t[0] - mu.x, v.s2w.t[1] - mu.y, v.s2w.t[2] - mu.z};
                const float A[3][3] = {{a[0] + H[0] * pc[0], a[1] + H[0] * pc[1], a[2] + H[0] * pc[2]},
                                       {a[3] + H[1] * pc[0], a[4] + H[1] * pc[1], a[5] + H[1] * pc[2]},
                                       {a[6] + H[2] * pc[0], a[7] + H[2] * pc[1], a[8] + H[2] * pc[2]}};
                const float d_dens = a[12], d_r = a[13], d_g = a[14], d_b = a[15];
                float out[16];
                // d mu = -M^T H
                out[0] = -(is[0] * r[0][0] * H[0] + is[1] * r[1][0] * H[1] + is[2] * r[2][0] * H[2]);
                out[1] = -(is[0] * r[0][1] * H[0] + is[1] * r[1][1] * H[1] + is[2] * r[2][1] * H[2]);
                out[2] = -(is[0] * r[0][2] * H[0] + is[1] * r[1][2] * H[1] + is[2] * r[2][2] * H[2]);
                out[3] = d_dens;
                // d s_i = -(1/s_i)^2 sum_j A_ij R_ij  (+ direct hit-distance term)
                float ds[3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    ds[i] = -is[i] * is[i] * (A[i][0] * r[i][0] + A[i][1] * r[i][1] + A[i][2] * r[i][2]);
                if (kDistGrad) { ds[0] += a[16]; ds[1] += a[17]; ds[2] += a[18]; }
                // d q through rotationT(q): dmat_ij = dL/dR_ij = (1/s_i) A_ij   (matmul_bw_quat, mathUtils.cuh:468-533)
                const float d00 = is[0] * A[0][0], d01 = is[0] * A[0][1], d02 = is[0] * A[0][2];
                const float d10 = is[1] * A[1][0], d11 = is[1] * A[1][1], d12 = is[1] * A[1][2];
                const float d20 = is[2] * A[2][0], d21 = is[2] * A[2][1], d22 = is[2] * A[2][2];
                const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
                out[4] = 2.0f * (qz * (d01 - d10) + qy * (d20 - d02) + qx * (d12 - d21));
                out[5] = 2.0f * (qy * (d01 + d10) + qz * (d02 + d20) + qr * (d12 - d21)) - 4.0f * qx * (d11 + d22);
                out[6] = 2.0f * (qx * (d01 + d10) + qr * (d20 - d02) + qz * (d12 + d21)) - 4.0f * qy * (d00 + d22);
                out[7] = 2.0f * (qr * (d01 - d10) + qx * (d02 + d20) + qy * (d12 + d21)) - 4.0f * qz * (d00 + d11);
                out[8] = ds[0]; out[9] = ds[1]; out[10] = ds[2];
                out[11] = d_r; out[12] = d_g; out[13] = d_b;
                out[14] = 0.0f; out[15] = 0.0f;
#pragma unroll
                for (int k = 0; k < 16; ++k) a[k] = out[k];
            }
        }
        __syncthreads();
        // flush: one global float atomic per (entry, component); a wave-instruction covers 4 entries x 16 consecutive
        // floats of their 64-byte gradient rows
#pragma unroll 4
        for (uint32_t it = 0; it < kBlock / 16; ++it) {
            const uint32_t e = it * 16 + (tid >> 4);
            const uint32_t k = tid & 15;
            const float val = acc[e * W + k];
            if (val != 0.0f && k < 14) {
                const uint32_t id = __float_as_uint(stage[e].feat_id.w);
                atomicAdd(&grad16[(size_t)id * kGradRow + k], val);
            }
            acc[e * W + k] = 0.0f;
            if (kDistGrad && k < 4) acc[e * W + 16 + k] = 0.0f;
        }
#ifdef GUT_CLOCK_STAMPS
        if (tid == 0) {   // [0] staging, [1] walking + wait for the slowest wave, [2] epilogue + flush
            const unsigned long long k7_tD = __builtin_amdgcn_s_memtime();
            k7_acc[0] += k7_tB - k7_tA; k7_acc[1] += k7_tC - k7_tB; k7_acc[2] += k7_tD - k7_tC;
        }
#endif
    }
    };
    if (centred) walk(std::true_type{}); else walk(std::false_type{});

    atomicMax(&s_deepest, consumed);
    __syncthreads();
    if (tid == 0) tile_traversed[tile] = s_deepest;
#ifdef GUT_CLOCK_STAMPS
    if (tid == 0 && blockIdx.x < 8192) {
        g_clock_stamps[blockIdx.x][2] = __builtin_amdgcn_s_memtime();
        g_clock_stamps[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
        g_k7_phases[blockIdx.x][0] = k7_acc[0]; g_k7_phases[blockIdx.x][1] = k7_acc[1]; g_k7_phases[blockIdx.x][2] = k7_acc[2];
        g_k7_phases[blockIdx.x][3] = g_clock_stamps[blockIdx.x][2] - g_clock_stamps[blockIdx.x][0];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
void GUT_LAUNCH_RENDER(hipStream_t s, const ViewParams& v, const RenderConsts& c, const float* density12, const float* feat,
                   const float* ray_ori, const float* ray_dir, const uint32_t* ranges, const uint32_t* sorted_ids,
                   const uint32_t* d_num_intersections, float* rgba, float* dist, float* hits, uint32_t* tile_traversed,
                   const uint64_t* tile_keys, uint32_t* ordered_ids, uint32_t* tile_ordered, const uint32_t* tile_order,
                   int kernel_degree) {
    const uint32_t tiles = (uint32_t)(v.grid_x * v.grid_y);
    if (tiles == 0) return;
#ifndef GUT_RENDER_GENERAL_TU
    if (kernel_degree != 2) {
        launch_render_general(s, v, c, density12, feat, ray_ori, ray_dir, ranges, sorted_ids, d_num_intersections, rgba, dist, hits,
                              tile_traversed, tile_keys, ordered_ids, tile_ordered, tile_order, kernel_degree);
        return;
    }
#endif
    auto kern = ordered_ids != nullptr ? k_render<true, kTuGeneral> : k_render<false, kTuGeneral>;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(kBlock), 0, s, v, c, reinterpret_cast<const float4*>(density12), feat,
                       ray_ori, ray_dir, reinterpret_cast<const uint2*>(ranges), sorted_ids, d_num_intersections,
                       reinterpret_cast<float4*>(rgba), dist, hits, tile_traversed, reinterpret_cast<const uint2*>(tile_keys),
                       ordered_ids, tile_ordered, tile_order, kernel_degree);
}

#ifndef GUT_RENDER_GENERAL_TU
// Launch order for the backward: tiles by decreasing forward traversal depth (the backward walks exactly as deep),
// longest-processing-time-first, so the few deep tiles do not become the tail of the grid.  One workgroup: 256
// linear buckets between 0 and the deepest tile, counting sort in LDS; the order inside a bucket is irrelevant.
// by_length: the key is the tile's list LENGTH — the forward's order, known before anything has been walked.
// walk_sums (may be null; backward order only): [0] = sum of the traversal depths, [1] = sum of the list lengths of this frame — the
// share of the lists the forward walked, which the host reads with the next frame's count and uses to decide whether the forward
// compositor of the frames after it is launched longest-lists-first (it pays when most of every list is walked).
__global__ __launch_bounds__(1024) void k_tile_order(uint32_t tiles, const uint32_t* __restrict__ traversed_in,
                                                     uint32_t* __restrict__ order, const uint2* __restrict__ ranges, bool by_length,
                                                     uint32_t* __restrict__ walk_sums) {
    __shared__ uint32_t s_max, s_hist[256], s_base[256], s_sum[2];
    const uint32_t tid = threadIdx.x;
    auto key = [&](uint32_t t) { return by_length ? ranges[t].y - ranges[t].x : traversed_in[t]; };
    if (walk_sums) {   // (uniform)
        if (tid < 2) s_sum[tid] = 0;
        __syncthreads();
        uint32_t a = 0, b = 0;
        for (uint32_t t = tid; t < tiles; t += 1024) { a += traversed_in[t]; b += ranges[t].y - ranges[t].x; }
        for (int o = 32; o > 0; o >>= 1) { a += (uint32_t)__shfl_xor((int)a, o); b += (uint32_t)__shfl_xor((int)b, o); }
        if ((tid & 63) == 0) { atomicAdd(&s_sum[0], a); atomicAdd(&s_sum[1], b); }
        __syncthreads();
        if (tid < 2) walk_sums[tid] = s_sum[tid];
    }
    if (tid == 0) s_max = 0;
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (uint32_t t = tid; t < tiles; t += 1024) mx = max(mx, key(t));
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if ((tid & 63) == 0) atomicMax(&s_max, mx);
    __syncthreads();
    const uint32_t width = s_max / 256 + 1;
    for (uint32_t t = tid; t < tiles; t += 1024) atomicAdd(&s_hist[255 - min(255u, key(t) / width)], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int b = 0; b < 256; ++b) {
            s_base[b] = run;
            run += s_hist[b];
        }
    }
    __syncthreads();
    for (uint32_t t = tid; t < tiles; t += 1024) order[atomicAdd(&s_base[255 - min(255u, key(t) / width)], 1u)] = t;
}

#ifdef GUT_CLOCK_STAMPS
}  // namespace gut
// diagnostic build only: copies the backward compositor's clock stamps [8192][4] u64 to the host
extern "C" int gut_debug_clock_stamps(void* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(gut::g_clock_stamps), sizeof(unsigned long long) * 8192 * 4) == hipSuccess ? 0 : 1;
}
extern "C" int gut_debug_k7_phases(void* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(gut::g_k7_phases), sizeof(unsigned long long) * 8192 * 4) == hipSuccess ? 0 : 1;
}
extern "C" int gut_debug_k6_phases(void* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(gut::g_k6_phases), sizeof(unsigned long long) * 8192 * 4) == hipSuccess ? 0 : 1;
}
namespace gut {
#endif

void launch_tile_order(hipStream_t s, uint32_t tiles, const uint32_t* traversed, uint32_t* order, const uint32_t* ranges, bool by_length,
                       uint32_t* walk_sums) {
    if (tiles == 0) return;
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, tiles, traversed, order, reinterpret_cast<const uint2*>(ranges), by_length,
                       ranges ? walk_sums : nullptr);
}
#endif  // !GUT_RENDER_GENERAL_TU

void GUT_LAUNCH_RENDER_BWD(hipStream_t s, const ViewParams& v, const RenderConsts& c, const float* density12,
                       const float* feat, const float* ray_ori, const float* ray_dir, const uint32_t* ranges,
                       const uint32_t* sorted_ids, const float* rgba, const float* rgba_grad, const float* dist_grad,
                       float* grad16, uint32_t* tile_traversed, const uint32_t* tile_order, const uint32_t* tile_walked,
                       int kernel_degree) {
    const uint32_t tiles = (uint32_t)(v.grid_x * v.grid_y);
    if (tiles == 0) return;
#ifndef GUT_RENDER_GENERAL_TU
    if (kernel_degree != 2) {
        launch_render_bwd_general(s, v, c, density12, feat, ray_ori, ray_dir, ranges, sorted_ids, rgba, rgba_grad, dist_grad, grad16,
                                  tile_traversed, tile_order, tile_walked, kernel_degree);
        return;
    }
#endif
    auto kern = dist_grad != nullptr ? k_render_backward<true, kTuGeneral> : k_render_backward<false, kTuGeneral>;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(kBlock), 0, s, v, c, reinterpret_cast<const float4*>(density12),
                       feat, ray_ori, ray_dir, reinterpret_cast<const uint2*>(ranges), sorted_ids,
                       reinterpret_cast<const float4*>(rgba), reinterpret_cast<const float4*>(rgba_grad), dist_grad, grad16,
                       tile_traversed, tile_order, tile_walked, kernel_degree);
}

}  // namespace gut
