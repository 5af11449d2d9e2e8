// gut_render.hip — per-tile compositing kernels of the 3DGUT path for gfx950 (wave64):
//   K6 render          (reference: gutRenderer.cuh:83-115, gutKBufferRenderer.cuh:108-170,217-292 with K=0,
//                       slang/models/gaussianParticles.slang:96-254, rayPayload.cuh:76-129)
//   K7 render_backward (reference: gutKBufferRenderer.cuh:294-386, models/gaussianParticles.cuh:480-738,
//                       shRadiativeGaussianParticles.cuh:409-482)
//
// Layout: one 256-thread workgroup (4 wave64) per 16x16 tile, one lane per pixel; each wave owns 4
// image rows.  The tile's depth-sorted list is consumed in chunks of 256 entries staged in LDS by the
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
// Colour/gradient buffers are compared with the oracle by tolerance, so this file is compiled with FMA
// contraction on and uses the hardware exp/rcp/rsq approximations.
#include "gut_internal.h"

namespace gut {

// ---- small helpers ------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

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

struct RayState {
    float ox, oy, oz, dx, dy, dz, tmin, tmax;
    bool valid;
};

// camera-space ray -> world, slab test against the +-1e6 scene box (rayPayload.cuh:76-108,
// utils/bounding_box.h:88-134)
__device__ __forceinline__ RayState make_ray(const ViewParams& v, const float* __restrict__ ray_ori,
                                             const float* __restrict__ ray_dir, size_t pix, bool inside) {
    RayState r;
    r.valid = false;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = 0.f;
    r.tmin = 0.f;
    r.tmax = 0.f;
    if (!inside) return r;
    const float a0 = ray_ori[3 * pix], a1 = ray_ori[3 * pix + 1], a2 = ray_ori[3 * pix + 2];
    const float b0 = ray_dir[3 * pix], b1 = ray_dir[3 * pix + 1], b2 = ray_dir[3 * pix + 2];
    const Affine& m = v.s2w;
    r.ox = m.r[0][0] * a0 + m.r[0][1] * a1 + m.r[0][2] * a2 + m.t[0];
    r.oy = m.r[1][0] * a0 + m.r[1][1] * a1 + m.r[1][2] * a2 + m.t[1];
    r.oz = m.r[2][0] * a0 + m.r[2][1] * a1 + m.r[2][2] * a2 + m.t[2];
    r.dx = m.r[0][0] * b0 + m.r[0][1] * b1 + m.r[0][2] * b2;
    r.dy = m.r[1][0] * b0 + m.r[1][1] * b1 + m.r[1][2] * b2;
    r.dz = m.r[2][0] * b0 + m.r[2][1] * b1 + m.r[2][2] * b2;
    const float lo = -1e06f, hi = 1e06f, fmx = 3.4028235e+38f;
    float tmin = (lo - r.ox) / r.dx, tmax = (hi - r.ox) / r.dx;
    if (tmin > tmax) { const float t = tmin; tmin = tmax; tmax = t; }
    float t0 = (lo - r.oy) / r.dy, t1 = (hi - r.oy) / r.dy;
    if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
    bool miss = (tmin > t1) || (t0 > tmax);
    if (t0 > tmin) tmin = t0;
    if (t1 < tmax) tmax = t1;
    t0 = (lo - r.oz) / r.dz;
    t1 = (hi - r.oz) / r.dz;
    if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
    miss = miss || (tmin > t1) || (t0 > tmax);
    if (t0 > tmin) tmin = t0;
    if (t1 < tmax) tmax = t1;
    if (miss) { tmin = fmx; tmax = fmx; }
    r.tmin = fmaxf(tmin, 0.0f);
    r.tmax = tmax;
    r.valid = r.tmax > r.tmin;
    return r;
}

__device__ __forceinline__ void quat_rows(float w, float x, float y, float z, float r[3][3]) {
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float rx = w * x, ry = w * y, rz = w * z;
    r[0][0] = 1.0f - 2.0f * (yy + zz); r[0][1] = 2.0f * (xy + rz); r[0][2] = 2.0f * (xz - ry);
    r[1][0] = 2.0f * (xy - rz); r[1][1] = 1.0f - 2.0f * (xx + zz); r[1][2] = 2.0f * (yz + rx);
    r[2][0] = 2.0f * (xz + ry); r[2][1] = 2.0f * (yz - rx); r[2][2] = 1.0f - 2.0f * (xx + yy);
}

// ---------------------------------------------------------------------------------------------------
// K6 forward
// ---------------------------------------------------------------------------------------------------
struct FwdEntry {      // 80 bytes, 16-byte aligned: five ds_read_b128 broadcasts per entry
    float4 mu_sigma;   // mean.xyz, density
    float4 m0;         // row 0 of M = diag(1/s) * rotationT, s.x
    float4 m1;         // row 1, s.y
    float4 m2;         // row 2, s.z
    float4 feat_id;    // max(rgb, 0), particle id (bit pattern)
};

__global__ __launch_bounds__(kBlock) void k_render(ViewParams v, RenderConsts c, const float4* __restrict__ density12,
                                                  const float* __restrict__ feat, const float* __restrict__ ray_ori,
                                                  const float* __restrict__ ray_dir, const uint2* __restrict__ ranges,
                                                  const uint32_t* __restrict__ sorted_ids, uint32_t num_intersections,
                                                  float4* __restrict__ rgba, float* __restrict__ dist,
                                                  float* __restrict__ hits, uint32_t* __restrict__ tile_traversed) {
    __shared__ FwdEntry stage[kBlock];
    __shared__ uint32_t s_deepest;

    const uint32_t tile = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const int px = (int)(tile % (uint32_t)v.grid_x) * kTile + (int)(tid & 15);
    const int py = (int)(tile / (uint32_t)v.grid_x) * kTile + (int)(tid >> 4);
    const bool inside = (px < v.width) && (py < v.height);
    const size_t pix = (size_t)py * (size_t)v.width + (size_t)px;
    // zero intersections: the reference returns before rendering and the outputs keep their initial values
    // (gutRenderer.cu:323-325); treating every ray as invalid writes exactly those
    const RayState ray = make_ray(v, ray_ori, ray_dir, pix, inside && (num_intersections != 0));

    if (tid == 0) s_deepest = 0;

    const uint2 range = ranges[tile];
    const uint32_t total = range.y - range.x;
    bool alive = ray.valid;
    float T = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, dsum = 0.f;
    uint32_t nhits = 0, consumed = 0;

    for (uint32_t base = 0; base < total; base += kBlock) {
        if (!__syncthreads_or(alive ? 1 : 0)) break;  // whole tile terminated (gutKBufferRenderer.cuh:234-236)
        {
            const uint32_t k = range.x + base + tid;
            uint32_t id = kInvalid;
            if (k < range.y) id = sorted_ids[k];
            FwdEntry e;
            e.feat_id.w = __uint_as_float(id);
            if (id != kInvalid) {
                const float4 a = density12[3 * (size_t)id + 0];
                const float4 q = density12[3 * (size_t)id + 1];
                const float4 s = density12[3 * (size_t)id + 2];
                float r[3][3];
                quat_rows(q.x, q.y, q.z, q.w, r);
                const float i0 = 1.0f / s.x, i1 = 1.0f / s.y, i2 = 1.0f / s.z;
                e.mu_sigma = a;
                e.m0 = make_float4(r[0][0] * i0, r[0][1] * i0, r[0][2] * i0, s.x);
                e.m1 = make_float4(r[1][0] * i1, r[1][1] * i1, r[1][2] * i1, s.y);
                e.m2 = make_float4(r[2][0] * i2, r[2][1] * i2, r[2][2] * i2, s.z);
                e.feat_id.x = fmaxf(feat[3 * (size_t)id + 0], 0.0f);
                e.feat_id.y = fmaxf(feat[3 * (size_t)id + 1], 0.0f);
                e.feat_id.z = fmaxf(feat[3 * (size_t)id + 2], 0.0f);
            }
            stage[tid] = e;
        }
        __syncthreads();

        const uint32_t cnt = min((uint32_t)kBlock, total - base);
        for (uint32_t j = 0; j < cnt; ++j) {
            if (__ballot(alive) == 0ull) break;  // wave-uniform
            const float4 fid = stage[j].feat_id;
            if (__float_as_uint(fid.w) == kInvalid) {  // padding entry: list ends here for everyone
                alive = false;
                break;
            }
            if (alive) {
                consumed = base + j + 1;
                const float4 ms = stage[j].mu_sigma;
                const float4 m0 = stage[j].m0, m1 = stage[j].m1, m2 = stage[j].m2;
                const float gx = ray.ox - ms.x, gy = ray.oy - ms.y, gz = ray.oz - ms.z;
                const float o0 = m0.x * gx + m0.y * gy + m0.z * gz;
                const float o1 = m1.x * gx + m1.y * gy + m1.z * gz;
                const float o2 = m2.x * gx + m2.y * gy + m2.z * gz;
                const float u0 = m0.x * ray.dx + m0.y * ray.dy + m0.z * ray.dz;
                const float u1 = m1.x * ray.dx + m1.y * ray.dy + m1.z * ray.dz;
                const float u2 = m2.x * ray.dx + m2.y * ray.dy + m2.z * ray.dz;
                const float c0 = u1 * o2 - u2 * o1, c1 = u2 * o0 - u0 * o2, c2 = u0 * o1 - u1 * o0;
                const float l2 = u0 * u0 + u1 * u1 + u2 * u2;
                const float il2 = fast_rcp(l2);
                const float d2 = (c0 * c0 + c1 * c1 + c2 * c2) * il2;  // |grd x gro|^2 with grd = u/|u|
                if (d2 < c.max_d2) {
                    const float resp = fast_exp(-0.5f * d2);
                    const float alpha = fminf(c.max_alpha, resp * ms.w);
                    if ((resp > c.min_response) && (alpha > c.alpha_threshold)) {
                        // hitT = | s * grd * (grd . -gro) |
                        const float proj = -(u0 * o0 + u1 * o1 + u2 * o2) * il2;  // (grd.-gro)/|u|
                        const float h0 = m0.w * u0 * proj, h1 = m1.w * u1 * proj, h2 = m2.w * u2 * proj;
                        const float hit_t = sqrtf(h0 * h0 + h1 * h1 + h2 * h2);
                        if ((hit_t > ray.tmin) && (hit_t < ray.tmax)) {
                            const float w = alpha * T;
                            dsum += hit_t * w;
                            T *= (1.0f - alpha);
                            if (w > 0.0f) {
                                cr += fid.x * w;
                                cg += fid.y * w;
                                cb += fid.z * w;
                                nhits++;
                            }
                            if (T < c.min_transmittance) alive = false;
                        }
                    }
                }
            }
        }
    }

    if (inside) {
        if (ray.valid) {
            rgba[pix] = make_float4(cr, cg, cb, 1.0f - T);
            dist[pix] = dsum;
            hits[pix] = (float)nhits;
        } else {  // initial values of the reference's output tensors (splatRaster.cpp:196-198)
            rgba[pix] = make_float4(0.f, 0.f, 0.f, 0.f);
            dist[pix] = 1e06f;
            hits[pix] = 0.0f;
        }
    }
    // traversal statistics (E_f of the roofline model): deepest list position any pixel of the tile consumed
    atomicMax(&s_deepest, consumed);
    __syncthreads();
    if (tid == 0) tile_traversed[tile] = s_deepest;
}

// ---------------------------------------------------------------------------------------------------
// K7 backward
// ---------------------------------------------------------------------------------------------------
struct BwdEntry {       // 112 bytes
    float4 mu_sigma;    // mean.xyz, density
    float4 quat;        // w,x,y,z
    float4 r0;          // rotationT row 0, 1/s.x
    float4 r1;          // row 1, 1/s.y
    float4 r2;          // row 2, 1/s.z
    float4 scale_id;    // s.xyz, particle id (bits)
    float4 feat;        // max(rgb,0), unused
};

constexpr int kGradRow = 16;  // floats per gradient row: pos3, density, quat4, scale3, rgb3, pad2

// d(out)/d(quat) for out = rotationT(q) * p, given g = dL/d(out)   (common/mathUtils.cuh:468-533)
__device__ __forceinline__ void matmul_bw_quat(float p0, float p1, float p2, float g0, float g1, float g2, float r,
                                               float x, float y, float z, float& dr, float& dx, float& dy, float& dz) {
    const float d00 = g0 * p0, d01 = g0 * p1, d02 = g0 * p2;
    const float d10 = g1 * p0, d11 = g1 * p1, d12 = g1 * p2;
    const float d20 = g2 * p0, d21 = g2 * p1, d22 = g2 * p2;
    dr += 2.0f * (z * (d01 - d10) + y * (d20 - d02) + x * (d12 - d21));
    dx += 2.0f * (y * (d01 + d10) + z * (d02 + d20) + r * (d12 - d21)) - 4.0f * x * (d11 + d22);
    dy += 2.0f * (x * (d01 + d10) + r * (d20 - d02) + z * (d12 + d21)) - 4.0f * y * (d00 + d22);
    dz += 2.0f * (r * (d01 - d10) + x * (d02 + d20) + y * (d12 + d21)) - 4.0f * z * (d00 + d11);
}

__global__ __launch_bounds__(kBlock) void k_render_backward(ViewParams v, RenderConsts c,
                                                           const float4* __restrict__ density12,
                                                           const float* __restrict__ feat,
                                                           const float* __restrict__ ray_ori,
                                                           const float* __restrict__ ray_dir,
                                                           const uint2* __restrict__ ranges,
                                                           const uint32_t* __restrict__ sorted_ids,
                                                           const float4* __restrict__ rgba,
                                                           const float4* __restrict__ rgba_grad,
                                                           const float* __restrict__ dist_grad, float* __restrict__ grad16,
                                                           uint32_t* __restrict__ tile_traversed) {
    __shared__ BwdEntry stage[kBlock];
    __shared__ float acc[kBlock * kGradRow];
    __shared__ uint32_t s_deepest;

    const uint32_t tile = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const int px = (int)(tile % (uint32_t)v.grid_x) * kTile + (int)(tid & 15);
    const int py = (int)(tile / (uint32_t)v.grid_x) * kTile + (int)(tid >> 4);
    const bool inside = (px < v.width) && (py < v.height);
    const size_t pix = (size_t)py * (size_t)v.width + (size_t)px;
    const RayState ray = make_ray(v, ray_ori, ray_dir, pix, inside);

    if (tid == 0) s_deepest = 0;
#pragma unroll
    for (int k = 0; k < kGradRow; ++k) acc[k * kBlock + tid] = 0.0f;

    // forward results and upstream gradients of this pixel (rayPayloadBackward.cuh:30-58)
    float T_final = 1.f, Tg = 0.f, fr = 0.f, fg = 0.f, fb = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gd = 0.f;
    if (ray.valid) {
        const float4 o = rgba[pix];
        const float4 g = rgba_grad[pix];
        T_final = 1.0f - o.w;
        Tg = -g.w;  // transmittanceGradient = -dL/d(opacity)
        fr = o.x; fg = o.y; fb = o.z;
        gr = g.x; gg = g.y; gb = g.z;
        gd = dist_grad[pix];
    }

    const uint2 range = ranges[tile];
    const uint32_t total = range.y - range.x;
    bool alive = ray.valid;
    float T = 1.0f, rr = 0.f, rg = 0.f, rb = 0.f;  // running transmittance / radiance
    uint32_t consumed = 0;

    for (uint32_t base = 0; base < total; base += kBlock) {
        if (!__syncthreads_or(alive ? 1 : 0)) break;
        {
            const uint32_t k = range.x + base + tid;
            uint32_t id = kInvalid;
            if (k < range.y) id = sorted_ids[k];
            BwdEntry e;
            e.scale_id.w = __uint_as_float(id);
            if (id != kInvalid) {
                const float4 a = density12[3 * (size_t)id + 0];
                const float4 q = density12[3 * (size_t)id + 1];
                const float4 s = density12[3 * (size_t)id + 2];
                float r[3][3];
                quat_rows(q.x, q.y, q.z, q.w, r);
                e.mu_sigma = a;
                e.quat = q;
                e.r0 = make_float4(r[0][0], r[0][1], r[0][2], 1.0f / s.x);
                e.r1 = make_float4(r[1][0], r[1][1], r[1][2], 1.0f / s.y);
                e.r2 = make_float4(r[2][0], r[2][1], r[2][2], 1.0f / s.z);
                e.scale_id.x = s.x; e.scale_id.y = s.y; e.scale_id.z = s.z;
                e.feat = make_float4(fmaxf(feat[3 * (size_t)id + 0], 0.0f), fmaxf(feat[3 * (size_t)id + 1], 0.0f),
                                     fmaxf(feat[3 * (size_t)id + 2], 0.0f), 0.0f);
            }
            stage[tid] = e;
        }
        __syncthreads();

        const uint32_t cnt = min((uint32_t)kBlock, total - base);
        for (uint32_t j = 0; j < cnt; ++j) {
            if (__ballot(alive) == 0ull) break;
            const float4 sid = stage[j].scale_id;
            if (__float_as_uint(sid.w) == kInvalid) {
                alive = false;
                break;
            }
            float g[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) g[k] = 0.0f;
            bool hit = false;
            if (alive) {
                consumed = base + j + 1;
                const float4 ms = stage[j].mu_sigma;
                const float4 r0 = stage[j].r0, r1 = stage[j].r1, r2 = stage[j].r2;
                const float p0 = ray.ox - ms.x, p1 = ray.oy - ms.y, p2 = ray.oz - ms.z;      // gposc
                const float pr0 = r0.x * p0 + r0.y * p1 + r0.z * p2;                           // gposcr
                const float pr1 = r1.x * p0 + r1.y * p1 + r1.z * p2;
                const float pr2 = r2.x * p0 + r2.y * p1 + r2.z * p2;
                const float dr0 = r0.x * ray.dx + r0.y * ray.dy + r0.z * ray.dz;               // rayDirR
                const float dr1 = r1.x * ray.dx + r1.y * ray.dy + r1.z * ray.dz;
                const float dr2 = r2.x * ray.dx + r2.y * ray.dy + r2.z * ray.dz;
                const float o0 = pr0 * r0.w, o1 = pr1 * r1.w, o2 = pr2 * r2.w;                 // gro
                const float u0 = dr0 * r0.w, u1 = dr1 * r1.w, u2 = dr2 * r2.w;                 // grdu
                const float l2 = u0 * u0 + u1 * u1 + u2 * u2;
                const float il = l2 > 0.0f ? fast_rsq(l2) : 1.0f;
                const float d0 = u0 * il, d1 = u1 * il, d2_ = u2 * il;                         // grd
                const float c0 = d1 * o2 - d2_ * o1, c1 = d2_ * o0 - d0 * o2, c2 = d0 * o1 - d1 * o0;
                const float dsq = c0 * c0 + c1 * c1 + c2 * c2;
                if (dsq < c.max_d2) {
                    const float resp = fast_exp(-0.5f * dsq);
                    const float alpha = fminf(c.max_alpha, resp * ms.w);
                    if ((resp > c.min_response) && (alpha > c.alpha_threshold)) {  // NB: no tmin/tmax test in the backward
                        hit = true;
                        const float4 q = stage[j].quat;
                        const float4 ft = stage[j].feat;
                        const float proj = -(d0 * o0 + d1 * o1 + d2_ * o2);
                        const float dd0 = d0 * proj, dd1 = d1 * proj, dd2 = d2_ * proj;       // grdd
                        const float s0 = sid.x * dd0, s1 = sid.y * dd1, s2 = sid.z * dd2;       // grds
                        const float gsq = s0 * s0 + s1 * s1 + s2 * s2;
                        const float gdist = sqrtf(gsq);
                        const float w = alpha * T;
                        const float Tn = (1.0f - alpha) * T;
                        // hit-distance terms (residualHitT == 0: quirk 1 of SURVEY §8a)
                        const float ga_hit = gdist * T * gd;
                        float k0 = 0.f, k1 = 0.f, k2 = 0.f;  // grdsRayHitGrd
                        if (gsq > 0.0f) {
                            const float kk = (w / gdist) * gd;
                            k0 = s0 * kk; k1 = s1 * kk; k2 = s2 * kk;
                        }
                        const float x0 = d0 * o0, x1 = d1 * o1, x2 = d2_ * o2;
                        const float hd0 = -sid.x * (2.0f * x0 + x1 + x2) * k0;                   // grdRayHitGrd
                        const float hd1 = -sid.y * (x0 + 2.0f * x1 + x2) * k1;
                        const float hd2 = -sid.z * (x0 + x1 + 2.0f * x2) * k2;
                        const float ho0 = -sid.x * d0 * d0 * k0;                                 // groRayHitGrd
                        const float ho1 = -sid.y * d1 * d1 * k1;
                        const float ho2 = -sid.z * d2_ * d2_ * k2;
                        const float res_T = alpha < 0.999999f ? T_final * fast_rcp(1.0f - alpha) : T;
                        const float ga_dns = res_T * -Tg;
                        // radiance: dL/dRGB of this particle, running radiance, residual radiance behind it
                        g[11] = gr * w; g[12] = gg * w; g[13] = gb * w;
                        rr += w * ft.x; rg += w * ft.y; rb += w * ft.z;
                        float q0 = 0.f, q1 = 0.f, q2 = 0.f;
                        if (!(Tn <= c.min_transmittance)) {
                            const float iT = fast_rcp(Tn);
                            q0 = fmaxf((fr - rr) * iT, 0.0f);
                            q1 = fmaxf((fg - rg) * iT, 0.0f);
                            q2 = fmaxf((fb - rb) * iT, 0.0f);
                        }
                        const float G = ga_hit + ga_dns + T * ((ft.x - q0) * gr + (ft.y - q1) * gg + (ft.z - q2) * gb);
                        g[3] = resp * G;                          // d density
                        const float g_resp = ms.w * G;
                        const float g_d2 = -0.5f * resp * g_resp;
                        const float e0 = 2.0f * c0 * g_d2, e1 = 2.0f * c1 * g_d2, e2 = 2.0f * c2 * g_d2;  // d cross
                        const float gd0 = e2 * o1 - e1 * o2 + hd0;  // d grd (incl. hit term)
                        const float gd1 = e0 * o2 - e2 * o0 + hd1;
                        const float gd2 = e1 * o0 - e0 * o1 + hd2;
                        const float go0 = e1 * d2_ - e2 * d1 + ho0;  // d gro (incl. hit term)
                        const float go1 = e2 * d0 - e0 * d2_ + ho1;
                        const float go2 = e0 * d1 - e1 * d0 + ho2;
                        const float gp0 = go0 * r0.w, gp1 = go1 * r1.w, gp2 = go2 * r2.w;        // d gposcr
                        // d position = -(rotationT^T * d gposcr)
                        g[0] = -(gp0 * r0.x + gp1 * r1.x + gp2 * r2.x);
                        g[1] = -(gp0 * r0.y + gp1 * r1.y + gp2 * r2.y);
                        g[2] = -(gp0 * r0.z + gp1 * r1.z + gp2 * r2.z);
                        // d grdu = (I - grd grd^T)/|grdu| * d grd      (safe_normalize_bw, mathUtils.cuh:420-430)
                        float gu0 = 0.f, gu1 = 0.f, gu2 = 0.f;
                        if (l2 > 0.0f) {
                            const float dot = gd0 * d0 + gd1 * d1 + gd2 * d2_;
                            gu0 = il * (gd0 - d0 * dot);
                            gu1 = il * (gd1 - d1 * dot);
                            gu2 = il * (gd2 - d2_ * dot);
                        }
                        // d scale
                        g[8] = dd0 * k0 - o0 * r0.w * go0 - u0 * r0.w * gu0;
                        g[9] = dd1 * k1 - o1 * r1.w * go1 - u1 * r1.w * gu1;
                        g[10] = dd2 * k2 - o2 * r2.w * go2 - u2 * r2.w * gu2;
                        // d quaternion through both mat-vecs
                        float qr = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
                        matmul_bw_quat(p0, p1, p2, gp0, gp1, gp2, q.x, q.y, q.z, q.w, qr, qx, qy, qz);
                        matmul_bw_quat(ray.dx, ray.dy, ray.dz, gu0 * r0.w, gu1 * r1.w, gu2 * r2.w, q.x, q.y, q.z, q.w, qr,
                                       qx, qy, qz);
                        g[4] = qr; g[5] = qx; g[6] = qy; g[7] = qz;
                        T = Tn;
                        if (T < c.min_transmittance) alive = false;
                    }
                }
            }
            if (__ballot(hit) != 0ull) {  // wave-uniform: skip the reduction when no lane of this wave hit entry j
#pragma unroll
                for (int k = 0; k < 14; ++k) {
                    const float s = wave_sum_lane63(g[k]);
                    if (lane == 63) atomicAdd(&acc[j * kGradRow + k], s);
                }
            }
        }

        // flush this chunk's accumulators: one float atomic per (entry, component); a wave-instruction covers
        // 4 entries x 16 consecutive floats of their 64-byte gradient rows
        __syncthreads();
#pragma unroll 4
        for (uint32_t it = 0; it < kBlock / 16; ++it) {
            const uint32_t e = it * 16 + (tid >> 4);
            const uint32_t k = tid & 15;
            const float val = acc[e * kGradRow + k];
            if (val != 0.0f) {
                const uint32_t id = __float_as_uint(stage[e].scale_id.w);
                atomicAdd(&grad16[(size_t)id * kGradRow + k], val);
                acc[e * kGradRow + k] = 0.0f;
            }
        }
    }

    atomicMax(&s_deepest, consumed);
    __syncthreads();
    if (tid == 0) tile_traversed[tile] = s_deepest;
}

// ---------------------------------------------------------------------------------------------------
void launch_render(hipStream_t s, const ViewParams& v, const RenderConsts& c, const float* density12, const float* feat,
                   const float* ray_ori, const float* ray_dir, const uint32_t* ranges, const uint32_t* sorted_ids,
                   uint32_t num_intersections, float* rgba, float* dist, float* hits, uint32_t* tile_traversed) {
    const uint32_t tiles = (uint32_t)(v.grid_x * v.grid_y);
    if (tiles == 0) return;
    hipLaunchKernelGGL(k_render, dim3(tiles), dim3(kBlock), 0, s, v, c, reinterpret_cast<const float4*>(density12), feat,
                       ray_ori, ray_dir, reinterpret_cast<const uint2*>(ranges), sorted_ids, num_intersections,
                       reinterpret_cast<float4*>(rgba), dist, hits, tile_traversed);
}

void launch_render_bwd(hipStream_t s, const ViewParams& v, const RenderConsts& c, const float* density12,
                       const float* feat, const float* ray_ori, const float* ray_dir, const uint32_t* ranges,
                       const uint32_t* sorted_ids, const float* rgba, const float* rgba_grad, const float* dist_grad,
                       float* grad16, uint32_t* tile_traversed) {
    const uint32_t tiles = (uint32_t)(v.grid_x * v.grid_y);
    if (tiles == 0) return;
    hipLaunchKernelGGL(k_render_backward, dim3(tiles), dim3(kBlock), 0, s, v, c, reinterpret_cast<const float4*>(density12),
                       feat, ray_ori, ray_dir, reinterpret_cast<const uint2*>(ranges), sorted_ids,
                       reinterpret_cast<const float4*>(rgba), reinterpret_cast<const float4*>(rgba_grad), dist_grad, grad16,
                       tile_traversed);
}

}  // namespace gut
