// gut_train.hip — per-Gaussian streaming kernels around the renderer ("next" rows N1/N2 of SURVEY §8f):
//   k_activate_pack : raw parameter rows -> the activated [N,12] particle_density rows the tracer consumes
//                     (what model.py:74-93 + tracer.py:176-178 do with normalize / exp / sigmoid / cat in torch)
//   k_adam_step     : fused Adam over an [N,C] tensor with per-column learning rates and an optional per-row
//                     visibility mask (reference: threedgrut/optimizers/optimizers.cu:47-117 SelectiveAdam, and
//                     torch.optim.Adam when bias correction is on and no mask is given)
// Both are pure HBM streams: 16-byte accesses, one row (or one float4 of a row) per lane.
#include <cstdlib>

#include "gut_internal.h"

namespace gut {

// raw row: pos3 | density logit | quat4 (unnormalised) | log-scale3 | unused
// act row: pos3 | sigmoid       | quat4 / |quat|       | exp3       | |quat|   (the norm rides in the pad column so
//          that the backward epilogue can chain through the normalisation without re-reading the raw row)
// (the three float4 of the row depend on one raw float4 each, plus the quaternion's norm for the last)
__device__ __forceinline__ float4 activate_position_density(const float4& a) {
    return make_float4(a.x, a.y, a.z, 1.0f / (1.0f + expf(-a.w)));
}
__device__ __forceinline__ float4 activate_quaternion(const float4& q, float* clamped_norm) {
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // torch.nn.functional.normalize eps
    *clamped_norm = fmaxf(nrm, 1e-12f);
    return make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
}
__device__ __forceinline__ float4 activate_scale(const float4& s, float clamped_norm) {
    return make_float4(expf(s.x), expf(s.y), expf(s.z), clamped_norm);
}
__device__ __forceinline__ void activate_row(const float4& a, const float4& q, const float4& s, float4* __restrict__ act_row) {
    float nrm;
    act_row[0] = activate_position_density(a);
    act_row[1] = activate_quaternion(q, &nrm);
    act_row[2] = activate_scale(s, nrm);
}

__global__ __launch_bounds__(kBlock) void k_activate_pack(uint32_t n, const float4* __restrict__ raw, float4* __restrict__ act) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    activate_row(raw[3 * (size_t)i + 0], raw[3 * (size_t)i + 1], raw[3 * (size_t)i + 2], act + 3 * (size_t)i);
}

// k_pack_activate_fields: the model's PRE-ACTIVATION tensors (positions [N,3], density logit [N,1], un-normalised quaternion [N,4],
// log-scale [N,3]; threedgrut/model/model.py:74-93 with base_gs.yaml's sigmoid / normalize / exp) -> the activated [N,12] rows the
// renderer reads, |quat| in the pad column for the backward's chain rule: the model's three activation kernels and the tracer's
// torch.cat as one coalesced pass (gut_trace_raw_model_fields)
__global__ __launch_bounds__(kBlock) void k_pack_activate_fields(uint32_t n, const float* __restrict__ pos, const float* __restrict__ dns,
                                                                const float4* __restrict__ rot, const float* __restrict__ scl,
                                                                float4* __restrict__ act) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    activate_row(make_float4(pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2], dns[i]), rot[i],
                 make_float4(scl[3 * (size_t)i], scl[3 * (size_t)i + 1], scl[3 * (size_t)i + 2], 0.0f), act + 3 * (size_t)i);
}

struct AdamParams {
    float lr[64];  // per column
    float beta1, beta2, eps;
    float bias1, bias2_sqrt;  // 1-beta1^t, sqrt(1-beta2^t); both 1 when bias correction is off
    uint32_t cols;            // multiple of 4
};

// one float4 (4 consecutive columns of one row) per lane
__global__ __launch_bounds__(kBlock) void k_adam_step(AdamParams ap, uint64_t n_vec4, float4* __restrict__ p,
                                                     const float4* __restrict__ g, float4* __restrict__ m,
                                                     float4* __restrict__ v, const float* __restrict__ visibility) {
    const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= n_vec4) return;
    const uint32_t vec_per_row = ap.cols / 4;
    const uint64_t row = idx / vec_per_row;
    const uint32_t c0 = (uint32_t)(idx - row * vec_per_row) * 4;
    if (visibility && !(visibility[row] != 0.0f)) return;  // SelectiveAdam: untouched row (no moment decay either)
    const float4 gg = g[idx];
    float4 mm = m[idx], vv = v[idx], pp = p[idx];
    const float b1 = ap.beta1, b2 = ap.beta2;
#define GUT_ADAM_LANE(X, K)                                                   \
    mm.X = b1 * mm.X + (1.0f - b1) * gg.X;                                    \
    vv.X = b2 * vv.X + (1.0f - b2) * gg.X * gg.X;                             \
    pp.X -= (ap.lr[c0 + K] / ap.bias1) * mm.X / (sqrtf(vv.X) / ap.bias2_sqrt + ap.eps);
    GUT_ADAM_LANE(x, 0)
    GUT_ADAM_LANE(y, 1)
    GUT_ADAM_LANE(z, 2)
    GUT_ADAM_LANE(w, 3)
#undef GUT_ADAM_LANE
    p[idx] = pp;
    m[idx] = mm;
    v[idx] = vv;
}

// k_selective_adam: the reference's SelectiveAdam update for ANY row width (its parameters are [N,3], [N,1], [N,4], [N,45] tensors;
// k_adam_step above wants whole float4 groups) — optimizers.cu:47-79: one element per lane, rows with visibility == 0 untouched
// (no moment decay either), no bias correction.  The mask is one BYTE per row (a torch.bool tensor).
__global__ __launch_bounds__(kBlock) void k_selective_adam(uint64_t n_elements, uint32_t cols, float* __restrict__ p,
                                                          const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          const uint8_t* __restrict__ visibility, float lr, float b1, float b2, float eps) {
    const uint64_t idx = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= n_elements) return;
    if (!visibility[idx / cols]) return;
    const float gg = g[idx];
    const float mm = b1 * m[idx] + (1.0f - b1) * gg;
    const float vv = b2 * v[idx] + (1.0f - b2) * gg * gg;
    p[idx] += -lr * mm / (sqrtf(vv) + eps);
    m[idx] = mm;
    v[idx] = vv;
}

// k_mcmc_relocation: new opacity / scale of Gaussians sampled `ratio` times by the MCMC relocation
// (reference: threedgrut/strategy/src/gaussian_mcmc.cu:33-73; binoms is the [n_max,n_max] Pascal table)
__global__ __launch_bounds__(kBlock) void k_mcmc_relocation(int n, const float* __restrict__ opacities,
                                                           const float* __restrict__ scales, const int* __restrict__ ratios,
                                                           const float* __restrict__ binoms, int n_max,
                                                           float* __restrict__ new_opacities, float* __restrict__ new_scales) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= n) return;
    const int n_idx = ratios[idx];
    const float op = opacities[idx];
    const float new_op = 1.0f - powf(1.0f - op, 1.0f / (float)n_idx);
    new_opacities[idx] = new_op;
    float denom = 0.0f;
    for (int i = 1; i <= n_idx; ++i) {
        float pw = new_op;  // new_op^(k+1)
        for (int k = 0; k <= i - 1; ++k) {
            const float sign = (k & 1) ? -1.0f : 1.0f;
            denom += binoms[(i - 1) * n_max + k] * (sign / sqrtf((float)(k + 1))) * pw;
            pw *= new_op;
        }
    }
    const float coeff = op / denom;
    for (int c = 0; c < 3; ++c) new_scales[idx * 3 + c] = coeff * scales[idx * 3 + c];
}

// k_mcmc_perturb: MCMCStrategy.perturb_gaussians (threedgrut/strategy/mcmc.py:147-164) as one pass over the raw rows:
//   positions += Sigma @ (unit_normal * op_sigmoid(1 - density) * noise_lr * position_lr),  Sigma = R S S^T R^T (model.py:95-105),
//   op_sigmoid(x) = 1 / (1 + exp(-100 (x - 0.995))).
// (the reference builds [N,3,3] covariances with four batched 3x3 matmuls for it: 250 ms at 6 M Gaussians on this chip.)
// unit == nullptr: three standard-normal draws per Gaussian from Philox4x32-10 keyed by (seed, step), counter = the row index
// (Box-Muller on its four words) — the same draws on every data-parallel rank, whatever the launch geometry.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(kBlock) void k_mcmc_perturb(uint32_t n, float4* __restrict__ raw, float4* __restrict__ act, float noise_scale,
                                                        uint32_t key0, uint32_t key1, uint32_t ctr_hi, const float* __restrict__ unit) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 a = raw[3 * (size_t)i + 0];
    const float4 q = raw[3 * (size_t)i + 1], ls = raw[3 * (size_t)i + 2];
    float u0, u1, u2;
    if (unit) {
        u0 = unit[3 * (size_t)i + 0]; u1 = unit[3 * (size_t)i + 1]; u2 = unit[3 * (size_t)i + 2];
    } else {
        uint32_t w[4];
        philox4x32_10(i, ctr_hi, 0u, 0u, key0, key1, w);
        // Box-Muller: (0,1] uniforms from the top 24 bits, two independent pairs
        const float f0 = ((float)(w[0] >> 8) + 1.0f) * 0x1p-24f, f1 = (float)(w[1] >> 8) * 0x1p-24f;
        const float f2 = ((float)(w[2] >> 8) + 1.0f) * 0x1p-24f, f3 = (float)(w[3] >> 8) * 0x1p-24f;
        const float r0 = sqrtf(-2.0f * logf(f0)), r1 = sqrtf(-2.0f * logf(f2));
        u0 = r0 * cosf(6.283185307179586f * f1);
        u1 = r0 * sinf(6.283185307179586f * f1);
        u2 = r1 * cosf(6.283185307179586f * f3);
    }
    const float dens = 1.0f / (1.0f + expf(-a.w));
    const float gate = 1.0f / (1.0f + expf(-100.0f * ((1.0f - dens) - 0.995f)));
    const float g = gate * noise_scale;
    const float n0 = u0 * g, n1 = u1 * g, n2 = u2 * g;
    // R from the normalised quaternion (wxyz; utils/misc.py:69-90), Sigma n = R S^2 R^T n
    const float nrm = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    const float r = q.x / nrm, x = q.y / nrm, y = q.z / nrm, z = q.w / nrm;
    const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                           {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                           {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
    const float s0 = expf(ls.x), s1 = expf(ls.y), s2 = expf(ls.z);
    const float t0 = (R[0][0] * n0 + R[1][0] * n1 + R[2][0] * n2) * s0 * s0;   // S^2 R^T n
    const float t1 = (R[0][1] * n0 + R[1][1] * n1 + R[2][1] * n2) * s1 * s1;
    const float t2 = (R[0][2] * n0 + R[1][2] * n1 + R[2][2] * n2) * s2 * s2;
    a.x += R[0][0] * t0 + R[0][1] * t1 + R[0][2] * t2;
    a.y += R[1][0] * t0 + R[1][1] * t1 + R[1][2] * t2;
    a.z += R[2][0] * t0 + R[2][1] * t1 + R[2][2] * t2;
    raw[3 * (size_t)i + 0] = a;
    if (act) {   // the activated row of the next forward: positions are not activated, the rest is unchanged
        float4 b = act[3 * (size_t)i + 0];
        b.x = a.x; b.y = a.y; b.z = a.z;
        act[3 * (size_t)i + 0] = b;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_sh_adam: fused (multi-view) SH-gradient rebuild + Adam for both parameter tensors, one wave per 64 Gaussians.
// The [N,48] SH gradient never exists in memory: each lane rebuilds its Gaussian's 48 gradient values from the
// compact per-view rows (3 floats per view), drops them into a wave-private LDS tile (row stride 49 dwords), and
// the wave then streams the contiguous 12 KiB blocks of p / m / v with fully coalesced 16-byte accesses, picking
// the matching gradients out of LDS.  Traffic per Gaussian: 12 B x views + 48 (grad12) + 6 x 240 (p,m,v in/out).
// ---------------------------------------------------------------------------------------------------
// Which 64-row waves the side-stream optimiser pass owns (everything else belongs to k_sh_adam<true>); both kernels call this.
//   first launch  (walked == nullptr): waves without a tile, in the 256-row blocks [block_begin, block_end)
//   second launch (walked != nullptr): the waves without a tile of the blocks >= split_block (those below were the first launch's)
//                                      + in the blocks < extra_end, waves WITH tiles in which the forward walked no Gaussian
//                                        (walked[w] == 0): the backward walks no further than the forward did, so they cannot
//                                        receive a gradient either — known only once the forward compositor has finished
struct EarlyOwnership {
    const uint8_t* walked;   // per wave, or nullptr
    uint32_t split_block;    // first block of the second launch's share of the waves without tiles
    uint32_t extra_end;      // blocks < extra_end: unwalked waves with tiles also go to the second launch
};
__device__ __forceinline__ bool side_stream_owns_wave(const EarlyOwnership& o, bool has_tiles, uint32_t wave_index, uint32_t blk,
                                                      bool second_launch) {
    if (!has_tiles) return second_launch ? blk >= o.split_block : blk < o.split_block;
    if (!o.walked || blk >= o.extra_end) return false;
    return second_launch && o.walked[wave_index] == 0;
}
// either launch (what k_sh_adam<true> must leave alone)
__device__ __forceinline__ bool side_stream_owns_wave(const EarlyOwnership& o, bool has_tiles, uint32_t wave_index, uint32_t blk) {
    if (!has_tiles) return true;
    return o.walked && blk < o.extra_end && o.walked[wave_index] == 0;
}

// (beta1^d, beta2^d) for the moments of wave `wave`, d = optimiser steps they have missed BEFORE the one being applied;
// (1, 1) when the moments are written every step
__device__ __forceinline__ float2 missed_decay(const LazyMoments& lz, uint32_t wave) {
    if (!lz.wave_step) return make_float2(1.0f, 1.0f);
    const uint32_t seen = lz.wave_step[wave];
    const uint32_t d = lz.t > seen ? lz.t - seen - 1u : 0u;
    const uint32_t k = d < lz.len ? d : lz.len - 1u;   // (the trainer brings every wave up to date long before the table ends ...
    if (d >= lz.len && lz.overrun) *lz.overrun = 1u;   //  ... and is told when a wave was not: its decay below is wrong)
    return make_float2(lz.pow1[k], lz.pow2[k]);
}

struct ShAdamParams {
    AdamParams a12, a48;
    LazyMoments lazy;
    const uint8_t* rule_walked;    // kScratch: per-wave "the forward walked a Gaussian of this wave" for EVERY wave, or null
    const float* cam;  // device [views,3]: sensor positions in world space
    uint32_t n, views;
    uint32_t view_stride;  // rows between consecutive views in mrgb (>= n)
    int32_t sh_degree;
    float grad_scale;
    int32_t rows_with_tiles_only;  // kScratch: the side-stream pass (k_adam_rows_without_gradient) already updated its waves
    EarlyOwnership own;            // ... which are these
    int32_t clear_consumed;        // !kScratch: gradient rows that were non-zero are written back as zeros (sparse exchange: the
                                   // dense accumulators are only ever touched where a record landed, never cleared wholesale)
    float* stat_accum;             // kScratch: the densification statistics of strategy/gs.py:106-115, or null
    int32_t* stat_denom;           //           (gut_set_position_gradient_statistics)
};

// strategy/gs.py:106-115 for one Gaussian with a non-zero position gradient of this view:
//   accum += | grad * |position - sensor| | / 2,   denom += 1
__device__ __forceinline__ void position_gradient_statistic(float gx, float gy, float gz, float px, float py, float pz,
                                                            const float* __restrict__ cam, float* accum, int32_t* denom) {
    const float dx = px - cam[0], dy = py - cam[1], dz = pz - cam[2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    const float sx = gx * d, sy = gy * d, sz = gz * d;
    *accum += sqrtf(sx * sx + sy * sy + sz * sz) * 0.5f;
    *denom += 1;
}

// the same as a kernel of its own, for the trainer paths whose optimiser kernel does not see the view's own gradient (data
// parallel exchange, unfused optimiser): grad / pos are rows of `*_stride` floats whose first three columns are used
__global__ __launch_bounds__(kBlock) void k_position_gradient_statistics(uint32_t n, const float* __restrict__ grad, uint32_t grad_stride,
                                                                        const float* __restrict__ pos, uint32_t pos_stride,
                                                                        const float* __restrict__ cam, float* __restrict__ accum,
                                                                        int32_t* __restrict__ denom) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float gx = grad[(size_t)i * grad_stride + 0], gy = grad[(size_t)i * grad_stride + 1], gz = grad[(size_t)i * grad_stride + 2];
    if (!(gx != 0.0f || gy != 0.0f || gz != 0.0f)) return;
    position_gradient_statistic(gx, gy, gz, pos[(size_t)i * pos_stride + 0], pos[(size_t)i * pos_stride + 1],
                                pos[(size_t)i * pos_stride + 2], cam, accum + i, denom + i);
}

__device__ __forceinline__ void sh_basis_fast(int deg, float x, float y, float z, float Y[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Y[i] = 0.0f;
    Y[0] = 0.28209479177387814f;
    if (deg > 0) {
        Y[1] = -0.4886025119029199f * y; Y[2] = 0.4886025119029199f * z; Y[3] = -0.4886025119029199f * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = 1.0925484305920792f * xy; Y[5] = -1.0925484305920792f * yz;
            Y[6] = 0.31539156525252005f * (2.0f * zz - xx - yy); Y[7] = -1.0925484305920792f * xz;
            Y[8] = 0.5462742152960396f * (xx - yy);
            if (deg > 2) {
                Y[9] = -0.5900435899266435f * y * (3.0f * xx - yy); Y[10] = 2.890611442640554f * xy * z;
                Y[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
                Y[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                Y[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy); Y[14] = 1.445305721320277f * z * (xx - yy);
                Y[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
            }
        }
    }
}

// sqrt(x), x >= 0, from the hardware approximation (v_sqrt_f32, 1 ulp) — moved into its normal range first, because second
// moments of tiny gradients are denormal and the instruction does not take those
__device__ __forceinline__ float sqrt_approx_pos(float x) {
    const bool tiny = x < 0x1p-96f;
    const float y = __builtin_amdgcn_sqrtf(tiny ? x * 0x1p+64f : x);
    return tiny ? y * 0x1p-32f : y;
}

// One Adam update of four consecutive columns:  p -= (lr / bias1) * m / (sqrt(v) / bias2_sqrt + eps)   (torch.optim.Adam).
// The two quotients are a hardware reciprocal (v_rcp_f32, 1 ulp; its argument is >= eps, never denormal) and host-side
// reciprocals of the bias corrections: ~12 VALU instructions per element instead of ~60 for two IEEE divisions and an IEEE
// square root.  That matters because the optimiser shares the chip with the VALU-bound compositing kernels
// (k_adam_rows_without_gradient); the update differs from the IEEE form by <= 3 ulp (tests: 1e-6 against torch.optim.Adam).
__device__ __forceinline__ void adam4(const AdamParams& ap, uint32_t c0, const float4& g, float4& p, float4& m, float4& v) {
#define GUT_ADAM_LANE(X, K)                               \
    m.X = ap.beta1 * m.X + (1.0f - ap.beta1) * g.X;        \
    v.X = ap.beta2 * v.X + (1.0f - ap.beta2) * g.X * g.X;  \
    p.X -= ap.lr[c0 + K] * m.X * __builtin_amdgcn_rcpf(sqrt_approx_pos(v.X) * ap.bias2_sqrt + ap.eps);
    GUT_ADAM_LANE(x, 0)
    GUT_ADAM_LANE(y, 1)
    GUT_ADAM_LANE(z, 2)
    GUT_ADAM_LANE(w, 3)
#undef GUT_ADAM_LANE
}

__device__ __forceinline__ void scale4(float4& a, float f) { a.x *= f; a.y *= f; a.z *= f; a.w *= f; }

__device__ __forceinline__ bool all_zero(const float4& m, const float4& v) {
    return m.x == 0.0f && m.y == 0.0f && m.z == 0.0f && m.w == 0.0f && v.x == 0.0f && v.y == 0.0f && v.z == 0.0f && v.w == 0.0f;
}

// adam4 with a zero gradient and the four learning rates handed over (bit-identical to adam4(ap, c0, 0, ...): the gradient
// terms are exact zeros there too)
__device__ __forceinline__ void adam4_zero_grad(const AdamParams& ap, const float4& lr, float4& p, float4& m, float4& v) {
#define GUT_ADAM_LANE(X)                                   \
    m.X = ap.beta1 * m.X + (1.0f - ap.beta1) * 0.0f;       \
    v.X = ap.beta2 * v.X + (1.0f - ap.beta2) * 0.0f * 0.0f; \
    p.X -= lr.X * m.X * __builtin_amdgcn_rcpf(sqrt_approx_pos(v.X) * ap.bias2_sqrt + ap.eps);
    GUT_ADAM_LANE(x)
    GUT_ADAM_LANE(y)
    GUT_ADAM_LANE(z)
    GUT_ADAM_LANE(w)
#undef GUT_ADAM_LANE
}

// The zero-gradient update of a wave whose moments are decayed lazily (read, never stored): with k1, k2 = beta^(steps missed)
//     m' = (beta1 k1) m,   v' = (beta2 k2) v,   p -= lr m' / (sqrt(v') / bias2_sqrt + eps)
// in 7 VALU instructions per value instead of 14 — this is the arithmetic of the side-stream pass, whose instructions issue on
// the SIMDs of the VALU-bound compositing kernels next door, so every one of them is taken from K6 / K7.  The constant factors
// are folded on the scalar side (c1 = beta1 k1; c2s = beta2 k2 2^32; b2s = 2^-16 / bias2_sqrt...) — the power of two lifts a
// denormal second moment into the hardware square root's range without the compare-and-select of sqrt_approx_pos, and being a
// power of two changes no bit of the result.  For a wave that is up to date (k = 1) the three values are those of adam4 with a
// zero gradient, bit for bit (beta m + (1 - beta) 0 = beta m); for k > 1 the decay is one rounding of beta k instead of two of
// the moment — inside the tolerance the lazy decay is specified with (deviation 10).  Every kernel that updates a
// lazy wave uses THIS function (the side-stream pass and k_sh_adam's lazy branch): the one- and two-pass forms stay bit-identical.
struct LazyAdam {
    float c1, c2s, b2s, eps;
};
__device__ __forceinline__ LazyAdam make_lazy_adam(const AdamParams& ap, const float2& dk) {
    LazyAdam la;
    la.c1 = ap.beta1 * dk.x;
    la.c2s = (ap.beta2 * dk.y) * 0x1p+32f;
    la.b2s = ap.bias2_sqrt * 0x1p-16f;
    la.eps = ap.eps;
    return la;
}
__device__ __forceinline__ void adam4_lazy(const LazyAdam& la, const float4& lr, float4& p, const float4& m, const float4& v) {
#define GUT_ADAM_LANE(X) p.X -= lr.X * (la.c1 * m.X) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(la.c2s * v.X) * la.b2s + la.eps);
    GUT_ADAM_LANE(x)
    GUT_ADAM_LANE(y)
    GUT_ADAM_LANE(z)
    GUT_ADAM_LANE(w)
#undef GUT_ADAM_LANE
}

// kScratch = true (one view, no exchange): the per-Gaussian epilogue of the backward (K8c) is folded in — grad12 then points
// at the renderer's 64-byte gradient rows [pos3, density, quat4, scale3, rgb3, pad2] w.r.t. the ACTIVATED parameters, which
// are chained to the raw parameters here (the activations are recomputed from the raw row the optimiser loads anyway, with
// the very function that produced the forward's inputs), and the masked dL/dRGB never leaves registers.
template <bool kScratch>
__global__ __launch_bounds__(kBlock) void k_sh_adam(ShAdamParams sp, float* __restrict__ mrgb,
                                                   float4* __restrict__ grad12, float4* __restrict__ p12,
                                                   float4* __restrict__ m12, float4* __restrict__ v12, float4* __restrict__ p48,
                                                   float4* __restrict__ m48, float4* __restrict__ v48,
                                                   const float* __restrict__ visibility, float4* __restrict__ act12,
                                                   const uint32_t* __restrict__ tiles_count, const float* __restrict__ feat) {
    constexpr int kRow = 49;
    __shared__ float tile[(kBlock / 64) * 64 * kRow];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wl = tile + wave * 64 * kRow;
    const uint32_t wave_first = blockIdx.x * kBlock + wave * 64u;
    const uint32_t rows_here = wave_first < sp.n ? min(64u, sp.n - wave_first) : 0u;
    // two-pass optimiser: waves without a single tile were updated whole by k_adam_rows_without_gradient
    if (kScratch && sp.rows_with_tiles_only &&
        side_stream_owns_wave(sp.own, __ballot(i < sp.n && tiles_count[i] != 0) != 0ull, wave_first >> 6, blockIdx.x))
        return;
    // data-parallel runs: waves no rank's forward walked were updated by gut_adam_unwalked_waves (wave-uniform test)
    if (!kScratch && sp.own.walked && wave_first < sp.n && sp.own.walked[wave_first >> 6] == 0) return;
    // lazy moment decay: a wave that cannot receive a gradient (by the rule the side-stream pass uses too, whoever processes the
    // wave) updates its parameters from moments it does not write back; every other wave brings its moments up to date first
    const uint32_t wave_index = wave_first >> 6;
    bool lazy_wave = false;
    float2 dk = make_float2(1.0f, 1.0f);
    if (sp.lazy.wave_step && wave_first < sp.n) {
        dk = missed_decay(sp.lazy, wave_index);
        if (kScratch) {
            const bool has_tiles = __ballot(i < sp.n && tiles_count[i] != 0) != 0ull;
            lazy_wave = !has_tiles || (sp.rule_walked && sp.rule_walked[wave_index] == 0);
        }
    }
    if (kScratch && lazy_wave) {
        // (wave-uniform) no row of this wave can receive a gradient — its gradient rows are exactly zero and are not even read —:
        // the arithmetic of the side-stream pass, value for value (adam4_lazy), so that the one-pass and two-pass forms agree bit
        // for bit; the stored moments stay those of wave_step[wave]
        const LazyAdam l12 = make_lazy_adam(sp.a12, dk), l48 = make_lazy_adam(sp.a48, dk);
        if (i < sp.n) {
            float4 a = p12[3 * (size_t)i + 0], b = p12[3 * (size_t)i + 1], c = p12[3 * (size_t)i + 2];
            adam4_lazy(l12, make_float4(sp.a12.lr[0], sp.a12.lr[1], sp.a12.lr[2], sp.a12.lr[3]), a, m12[3 * (size_t)i + 0], v12[3 * (size_t)i + 0]);
            adam4_lazy(l12, make_float4(sp.a12.lr[4], sp.a12.lr[5], sp.a12.lr[6], sp.a12.lr[7]), b, m12[3 * (size_t)i + 1], v12[3 * (size_t)i + 1]);
            adam4_lazy(l12, make_float4(sp.a12.lr[8], sp.a12.lr[9], sp.a12.lr[10], sp.a12.lr[11]), c, m12[3 * (size_t)i + 2], v12[3 * (size_t)i + 2]);
            p12[3 * (size_t)i + 0] = a; p12[3 * (size_t)i + 1] = b; p12[3 * (size_t)i + 2] = c;
            if (act12) activate_row(a, b, c, act12 + 3 * (size_t)i);
        }
        float4* bp = p48 + (size_t)wave_first * 12;
        const float4* bm = m48 + (size_t)wave_first * 12;
        const float4* bv = v48 + (size_t)wave_first * 12;
#pragma unroll 4
        for (int it = 0; it < 12; ++it) {
            const uint32_t q = (uint32_t)it * 64u + lane;
            if (q >= rows_here * 12u) continue;
            const uint32_t row = q / 12u, col = (q - row * 12u) * 4u;
            float4 pp = bp[q];
            adam4_lazy(l48, make_float4(sp.a48.lr[col], sp.a48.lr[col + 1], sp.a48.lr[col + 2], sp.a48.lr[col + 3]), pp, bm[q], bv[q]);
            bp[q] = pp;
        }
        return;
    }
    float G[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) G[k] = 0.0f;
    bool active = false;
    if (i < sp.n) {
        active = !(visibility && !(visibility[i] != 0.0f));
        if (active) {
            // --- raw [N,12] row ---
            float4 a = p12[3 * (size_t)i + 0];
            const float px = a.x, py = a.y, pz = a.z;  // pre-update position: the direction the forward used
            float4 b = p12[3 * (size_t)i + 1], c = p12[3 * (size_t)i + 2];
            float4 g0, g1, g2;
            float own_r = 0.f, own_g = 0.f, own_b = 0.f;  // kScratch: this view's masked dL/dRGB
            const float gs = sp.grad_scale;
            if (kScratch) {
                g0 = make_float4(0.f, 0.f, 0.f, 0.f); g1 = g0; g2 = g0;
                if (tiles_count[i] != 0) {
                    g0 = grad12[4 * (size_t)i + 0];
                    g1 = grad12[4 * (size_t)i + 1];
                    g2 = grad12[4 * (size_t)i + 2];
                    const float4 g3 = grad12[4 * (size_t)i + 3];
                    // the renderer's gradient row is consumed: leave it zero for the next backward
                    grad12[4 * (size_t)i + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
                    grad12[4 * (size_t)i + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
                    grad12[4 * (size_t)i + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
                    grad12[4 * (size_t)i + 3] = make_float4(0.f, 0.f, 0.f, 0.f);
                    own_r = feat[3 * (size_t)i + 0] > 0.0f ? g2.w : 0.0f;
                    own_g = feat[3 * (size_t)i + 1] > 0.0f ? g3.x : 0.0f;
                    own_b = feat[3 * (size_t)i + 2] > 0.0f ? g3.y : 0.0f;
                    if (sp.stat_accum && (g0.x != 0.0f || g0.y != 0.0f || g0.z != 0.0f))
                        position_gradient_statistic(g0.x, g0.y, g0.z, px, py, pz, sp.cam, sp.stat_accum + i, sp.stat_denom + i);
                    float4 act[3];
                    activate_row(a, b, c, act);
                    g0.w = g0.w * act[0].w * (1.0f - act[0].w);                      // sigmoid
                    const float dot = g1.x * act[1].x + g1.y * act[1].y + g1.z * act[1].z + g1.w * act[1].w;
                    const float inv = 1.0f / act[2].w;                                // 1 / |quat|
                    g1 = make_float4((g1.x - act[1].x * dot) * inv, (g1.y - act[1].y * dot) * inv, (g1.z - act[1].z * dot) * inv,
                                     (g1.w - act[1].w * dot) * inv);                  // normalise
                    g2.x *= act[2].x; g2.y *= act[2].y; g2.z *= act[2].z;             // exp
                }
            } else {
                g0 = grad12[3 * (size_t)i + 0]; g1 = grad12[3 * (size_t)i + 1]; g2 = grad12[3 * (size_t)i + 2];
                if (sp.clear_consumed && !(all_zero(g0, g1) && g2.x == 0.0f && g2.y == 0.0f && g2.z == 0.0f)) {
                    grad12[3 * (size_t)i + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
                    grad12[3 * (size_t)i + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
                    grad12[3 * (size_t)i + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            g0.x *= gs; g0.y *= gs; g0.z *= gs; g0.w *= gs; g1.x *= gs; g1.y *= gs; g1.z *= gs; g1.w *= gs;
            g2.x *= gs; g2.y *= gs; g2.z *= gs; g2.w = 0.0f;
            float4 ma = m12[3 * (size_t)i + 0], mb = m12[3 * (size_t)i + 1], mc = m12[3 * (size_t)i + 2];
            float4 va = v12[3 * (size_t)i + 0], vb = v12[3 * (size_t)i + 1], vc = v12[3 * (size_t)i + 2];
            scale4(ma, dk.x); scale4(mb, dk.x); scale4(mc, dk.x);
            scale4(va, dk.y); scale4(vb, dk.y); scale4(vc, dk.y);
            adam4(sp.a12, 0, g0, a, ma, va);
            adam4(sp.a12, 4, g1, b, mb, vb);
            adam4(sp.a12, 8, g2, c, mc, vc);
            p12[3 * (size_t)i + 0] = a; p12[3 * (size_t)i + 1] = b; p12[3 * (size_t)i + 2] = c;
            if (!lazy_wave) {
                m12[3 * (size_t)i + 0] = ma; m12[3 * (size_t)i + 1] = mb; m12[3 * (size_t)i + 2] = mc;
                v12[3 * (size_t)i + 0] = va; v12[3 * (size_t)i + 1] = vb; v12[3 * (size_t)i + 2] = vc;
            }
            // the next forward's activated row, while the updated raw row is still in registers (saves k_activate_pack's
            // separate pass over [N,12]); rows SelectiveAdam leaves untouched keep their previous activation
            if (act12) activate_row(a, b, c, act12 + 3 * (size_t)i);
            // --- rebuild the SH gradient of this Gaussian from the compact per-view rows ---
            for (uint32_t vw = 0; vw < sp.views; ++vw) {
                float r, g, bl;
                if (kScratch) {
                    r = own_r * sp.grad_scale; g = own_g * sp.grad_scale; bl = own_b * sp.grad_scale;
                } else {
                    float* mr = mrgb + ((size_t)vw * sp.view_stride + i) * 3;
                    r = mr[0]; g = mr[1]; bl = mr[2];
                    if (sp.clear_consumed && !(r == 0.0f && g == 0.0f && bl == 0.0f)) { mr[0] = 0.0f; mr[1] = 0.0f; mr[2] = 0.0f; }
                    r *= sp.grad_scale; g *= sp.grad_scale; bl *= sp.grad_scale;
                }
                if (r == 0.0f && g == 0.0f && bl == 0.0f) continue;
                const float dx = px - sp.cam[3 * vw + 0], dy = py - sp.cam[3 * vw + 1], dz = pz - sp.cam[3 * vw + 2];
                const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
                float Y[16];
                sh_basis_fast(sp.sh_degree, dx * inv, dy * inv, dz * inv, Y);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    G[3 * k + 0] += Y[k] * r;
                    G[3 * k + 1] += Y[k] * g;
                    G[3 * k + 2] += Y[k] * bl;
                }
            }
        }
    }
    if (__ballot(active) == 0ull) return;  // wave-uniform: none of the wave's 64 rows is updated by this launch
    // stage gradients + row-active flag (column 48) in the wave-private LDS tile
#pragma unroll
    for (int k = 0; k < 48; ++k) wl[lane * kRow + k] = G[k];
    wl[lane * kRow + 48] = active ? 1.0f : 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // coalesced sweep over the wave's contiguous [rows_here, 48] blocks of p, m, v
    float4* bp = p48 + (size_t)wave_first * 12;
    float4* bm = m48 + (size_t)wave_first * 12;
    float4* bv = v48 + (size_t)wave_first * 12;
#pragma unroll 4
    for (int it = 0; it < 12; ++it) {
        const uint32_t q = (uint32_t)it * 64u + lane;
        if (q >= rows_here * 12u) continue;
        const uint32_t row = q / 12u, col = (q - row * 12u) * 4u;
        if (wl[row * kRow + 48] == 0.0f) continue;  // SelectiveAdam: untouched row
        const float* gsrc = wl + row * kRow + col;
        const float4 g = make_float4(gsrc[0], gsrc[1], gsrc[2], gsrc[3]);
        float4 pp = bp[q], mm = bm[q], vv = bv[q];
        scale4(mm, dk.x); scale4(vv, dk.y);
        adam4(sp.a48, col, g, pp, mm, vv);
        bp[q] = pp;
        if (!lazy_wave) { bm[q] = mm; bv[q] = vv; }
    }
    // the wave's stored moments are those of this step now
    if (sp.lazy.wave_step && !lazy_wave && lane == 0 && rows_here) sp.lazy.wave_step[wave_index] = sp.lazy.t;
}

// k_adam_rows_without_gradient: the Adam step of Gaussians that cannot receive a gradient from the current view — the ones the
// projection gave no tile (tiles_count == 0: culled, off-screen, transparent) and, once the forward compositor has run, the
// ones it walked nothing of (EarlyOwnership above) — needs nothing from the backward pass.  It is the zero-gradient update
// (moments decay, parameters keep moving on their momentum, exactly as torch.optim.Adam does with a zero gradient), pure HBM
// streaming, so it runs on a side stream UNDER and beside the VALU-bound compositing kernels of the same iteration;
// k_sh_adam<true> then only walks the remaining waves.  Ownership is by WHOLE WAVES of 64 rows; rows without tiles inside
// other waves stay with k_sh_adam<true>, which sees their zero gradient.  Same adam4 arithmetic, same activation: every row
// this pass updates is bit-identical to the one-pass kernel's.  One wave per 64 Gaussians; the [N,48] sweep is the same
// coalesced 16-byte pattern as in k_sh_adam.
// Launch shape: PERSISTENT with a fixed, small footprint — gridDim.x = 1 workgroup per CU (one wave per SIMD, 83 VGPRs, 192 B
// of LDS, no scratch), every workgroup striding over the 256-row blocks.  Its workgroups take their wave slot per SIMD once and
// keep it until the pass is done; the compositing and loss kernels get everything else and are never queued behind a wall of
// streaming workgroups (a one-block-per-256-rows grid on a low-priority stream starved the image-sized loss kernels: measured
// 0.14 -> 0.8 ms).  The raw row's nine 16-byte loads are issued together, the [N,48] sweep issues three per (p, m, v) group and
// the next group's loads behind the previous group's stores.  Measured and not kept: the loads of 2 / 4 / 6 groups issued before
// the first use (84-116 VGPRs, which still leaves the compositors their waves): no faster, beside the compositors or alone —
// the pass is not latency-bound; two waves per SIMD at half the registers: same bytes, but scratch at the 64-VGPR bound;
// skipping the stores of groups whose moments are all zero (fixed points of the update): the start of a training run 4 %
// faster, its steady state 8 % slower (the test sits between the loads and their use).
template <bool kLazy>   // kLazy: lazy moment decay (the moments are read, never stored) — uniform over the launch
__global__ __launch_bounds__(kBlock, 4) void k_adam_rows_without_gradient(AdamParams a12, AdamParams a48, uint32_t n,
                                                                          const uint32_t* __restrict__ tiles_count,
                                                                          float4* __restrict__ p12, float4* __restrict__ m12,
                                                                          float4* __restrict__ v12, float4* __restrict__ p48,
                                                                          float4* __restrict__ m48, float4* __restrict__ v48,
                                                                          float4* __restrict__ act12, uint32_t block_begin,
                                                                          uint32_t block_end, EarlyOwnership own,
                                                                          uint32_t second_launch, LazyMoments lazy) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_s_setprio(1);  // its few instructions issue ahead of the VALU-saturated compositor next door (+1 % step rate)
    // the [N,48] sweep picks its four learning rates by a per-lane column: keep them in LDS (a register-indexed kernel
    // argument array would live in scratch, and this kernel must not use any: it runs beside other kernels on another queue)
    __shared__ float4 s_lr48[12];
    if (threadIdx.x < 12) s_lr48[threadIdx.x] = make_float4(a48.lr[4 * threadIdx.x], a48.lr[4 * threadIdx.x + 1], a48.lr[4 * threadIdx.x + 2],
                                                           a48.lr[4 * threadIdx.x + 3]);
    __syncthreads();
    for (uint32_t blk = block_begin + blockIdx.x; blk < block_end; blk += gridDim.x) {
        const uint32_t i = blk * kBlock + threadIdx.x;
        const uint32_t wave_first = blk * kBlock + wave * 64u;
        const uint32_t rows_here = wave_first < n ? min(64u, n - wave_first) : 0u;
        // whole waves only (side_stream_owns_wave): first of all the waves in which NO row has a tile (k_sh_adam<true> applies
        // the same test and takes every other wave whole).  Inside the view frustum about one row in ten has no tile of its own
        // (sub-pixel or transparent), sprinkled between rows that have: row-granular, this pass visited 55 k such waves for
        // 13 % of its bytes with mostly-masked loads, and both passes touched those waves' cache lines.
        const bool mine = i < n;
        if (wave_first >= n) continue;   // tail waves of the last 256-row block: no rows, and no entry in the per-wave flags
        // (tiles_count == nullptr: ownership by the wave flags alone — data-parallel runs, where the flags are the union over the
        //  ranks' views of the walked waves)
        const bool has_tiles = tiles_count ? (__ballot(i < n && tiles_count[i] != 0) != 0ull) : true;
        if (!side_stream_owns_wave(own, has_tiles, wave_first >> 6, blk, second_launch != 0u)) continue;
        // lazy moment decay: the moments are brought up to date in registers and NOT written back (the wave cannot receive a
        // gradient: its stored moments stay those of wave_step[wave], whoever reads them next decays them by the steps missed)
        const float2 dk = missed_decay(lazy, wave_first >> 6);
        // (kLazy == false: left a run-time value — with `true` known at compile time the schedule needs 88 VGPRs instead of 69, and
        //  this kernel shares its SIMDs with the compositors' waves)
        const bool store_mv = kLazy ? false : (lazy.wave_step == nullptr);
        if (kLazy) {   // lazy moment decay: read p, m, v; write p and the activation row
            const LazyAdam l12 = make_lazy_adam(a12, dk), l48 = make_lazy_adam(a48, dk);
            if (mine) {
                float4 a = p12[3 * (size_t)i + 0], b = p12[3 * (size_t)i + 1], c = p12[3 * (size_t)i + 2];
                adam4_lazy(l12, make_float4(a12.lr[0], a12.lr[1], a12.lr[2], a12.lr[3]), a, m12[3 * (size_t)i + 0], v12[3 * (size_t)i + 0]);
                p12[3 * (size_t)i + 0] = a;
                adam4_lazy(l12, make_float4(a12.lr[4], a12.lr[5], a12.lr[6], a12.lr[7]), b, m12[3 * (size_t)i + 1], v12[3 * (size_t)i + 1]);
                p12[3 * (size_t)i + 1] = b;
                adam4_lazy(l12, make_float4(a12.lr[8], a12.lr[9], a12.lr[10], a12.lr[11]), c, m12[3 * (size_t)i + 2], v12[3 * (size_t)i + 2]);
                p12[3 * (size_t)i + 2] = c;
                if (act12) activate_row(a, b, c, act12 + 3 * (size_t)i);
            }
            float4* bp = p48 + (size_t)wave_first * 12;
            const float4* bm = m48 + (size_t)wave_first * 12;
            const float4* bv = v48 + (size_t)wave_first * 12;
#pragma unroll 4
            for (int it = 0; it < 12; ++it) {
                const uint32_t q = (uint32_t)it * 64u + lane;
                if (q >= rows_here * 12u) continue;
                const uint32_t row = q / 12u, col = (q - row * 12u) * 4u;
                float4 pp = bp[q];
                adam4_lazy(l48, s_lr48[col >> 2], pp, bm[q], bv[q]);
                bp[q] = pp;
            }
            continue;
        }
        if (mine) {
            // one float4 of (p, m, v) at a time: at most 12 of the row's 36 values are live besides the updated parameters
            // the activation needs (the kernel must stay within 64 VGPRs WITHOUT scratch, see below)
            float4 a = p12[3 * (size_t)i + 0], m = m12[3 * (size_t)i + 0], v = v12[3 * (size_t)i + 0];
            scale4(m, dk.x); scale4(v, dk.y);
            adam4(a12, 0, zero, a, m, v);
            p12[3 * (size_t)i + 0] = a;
            if (store_mv) { m12[3 * (size_t)i + 0] = m; v12[3 * (size_t)i + 0] = v; }
            float4 b = p12[3 * (size_t)i + 1];
            m = m12[3 * (size_t)i + 1]; v = v12[3 * (size_t)i + 1];
            scale4(m, dk.x); scale4(v, dk.y);
            adam4(a12, 4, zero, b, m, v);
            p12[3 * (size_t)i + 1] = b;
            if (store_mv) { m12[3 * (size_t)i + 1] = m; v12[3 * (size_t)i + 1] = v; }
            float4 c = p12[3 * (size_t)i + 2];
            m = m12[3 * (size_t)i + 2]; v = v12[3 * (size_t)i + 2];
            scale4(m, dk.x); scale4(v, dk.y);
            adam4(a12, 8, zero, c, m, v);
            p12[3 * (size_t)i + 2] = c;
            if (store_mv) { m12[3 * (size_t)i + 2] = m; v12[3 * (size_t)i + 2] = v; }
            if (act12) activate_row(a, b, c, act12 + 3 * (size_t)i);
        }
        float4* bp = p48 + (size_t)wave_first * 12;
        float4* bm = m48 + (size_t)wave_first * 12;
        float4* bv = v48 + (size_t)wave_first * 12;
#pragma unroll 4
        for (int it = 0; it < 12; ++it) {
            const uint32_t q = (uint32_t)it * 64u + lane;
            if (q >= rows_here * 12u) continue;
            const uint32_t row = q / 12u, col = (q - row * 12u) * 4u;
            float4 pp = bp[q], mm = bm[q], vv = bv[q];
            scale4(mm, dk.x); scale4(vv, dk.y);
            adam4_zero_grad(a48, s_lr48[col >> 2], pp, mm, vv);
            bp[q] = pp;
            if (store_mv) { bm[q] = mm; bv[q] = vv; }
        }
    }
}

// The same pass (lazy moments) within 32 registers per lane, for the launch that runs beside the FORWARD compositor: K6 holds five
// 96-register waves per SIMD (480 of 512), so a co-resident wave of this size costs it none of them, where the 72-register form
// above takes the fifth.  One (p, m, v) group in flight per lane, wave-uniform bases with 32-bit lane offsets: latency-bound,
// about a third of the streaming rate — still more rows than the wide form could be given under K6.  Same arithmetic, bit for
// bit (adam4_lazy, the pieces of activate_row).  (amdgpu_num_vgpr counts each half of the unified file: 16 = 32 registers.)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_vgpr(16))) void k_adam_rows_without_gradient_narrow(
    AdamParams a12, AdamParams a48, uint32_t n, const uint32_t* __restrict__ tiles_count, float4* __restrict__ p12,
    float4* __restrict__ m12, float4* __restrict__ v12, float4* __restrict__ p48, float4* __restrict__ m48,
    float4* __restrict__ v48, float4* __restrict__ act12, uint32_t block_begin, uint32_t block_end, EarlyOwnership own,
    uint32_t second_launch, LazyMoments lazy) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __builtin_amdgcn_s_setprio(1);
    __shared__ float4 s_lr48[12];
    if (threadIdx.x < 12) s_lr48[threadIdx.x] = make_float4(a48.lr[4 * threadIdx.x], a48.lr[4 * threadIdx.x + 1], a48.lr[4 * threadIdx.x + 2],
                                                           a48.lr[4 * threadIdx.x + 3]);
    __syncthreads();
    for (uint32_t blk = block_begin + blockIdx.x; blk < block_end; blk += gridDim.x) {
        const uint32_t wave_first = blk * kBlock + wave * 64u;  // wave-uniform
        if (wave_first >= n) continue;
        const uint32_t rows_here = min(64u, n - wave_first);
        const bool has_tiles = tiles_count ? (__ballot(lane < rows_here && tiles_count[wave_first + lane] != 0) != 0ull) : true;
        if (!side_stream_owns_wave(own, has_tiles, wave_first >> 6, blk, second_launch != 0u)) continue;
        const float2 dk = missed_decay(lazy, wave_first >> 6);
        const LazyAdam l12 = make_lazy_adam(a12, dk), l48 = make_lazy_adam(a48, dk);
        if (lane < rows_here) {
            float4* rp = p12 + 3 * (size_t)wave_first;
            const float4* rm = m12 + 3 * (size_t)wave_first;
            const float4* rv = v12 + 3 * (size_t)wave_first;
            float4* ra = act12 + 3 * (size_t)wave_first;
            uint32_t o = 3u * lane;
            asm volatile("" : "+v"(o));  // (keeps the six per-lane 64-bit addresses from being hoisted out of the loop and spilled)
            float4 x = rp[o];
            adam4_lazy(l12, make_float4(a12.lr[0], a12.lr[1], a12.lr[2], a12.lr[3]), x, rm[o], rv[o]);
            rp[o] = x;
            if (act12) ra[o] = activate_position_density(x);
            __builtin_amdgcn_sched_barrier(0);
            x = rp[o + 1];
            adam4_lazy(l12, make_float4(a12.lr[4], a12.lr[5], a12.lr[6], a12.lr[7]), x, rm[o + 1], rv[o + 1]);
            rp[o + 1] = x;
            float nrm = 0.0f;
            if (act12) ra[o + 1] = activate_quaternion(x, &nrm);
            __builtin_amdgcn_sched_barrier(0);
            x = rp[o + 2];
            adam4_lazy(l12, make_float4(a12.lr[8], a12.lr[9], a12.lr[10], a12.lr[11]), x, rm[o + 2], rv[o + 2]);
            rp[o + 2] = x;
            if (act12) ra[o + 2] = activate_scale(x, nrm);
            __builtin_amdgcn_sched_barrier(0);
        }
        float4* bp = p48 + (size_t)wave_first * 12;
        const float4* bm = m48 + (size_t)wave_first * 12;
        const float4* bv = v48 + (size_t)wave_first * 12;
        uint32_t q0 = lane;
        asm volatile("" : "+v"(q0));
#pragma unroll 1
        for (uint32_t q = q0; q < rows_here * 12u; q += 64u) {
            const uint32_t col4 = q % 12u;
            float4 pp = bp[q];
            adam4_lazy(l48, s_lr48[col4], pp, bm[q], bv[q]);
            bp[q] = pp;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Sparse gradient exchange of the data-parallel trainer.  A view gives a gradient only to the Gaussians its rays hit, a
// small share of the scene, so what crosses xGMI per view is a list of 64-byte RECORDS
//     [ d pos3 | d density logit | d quat4 (un-normalised) | d log-scale3 | row id (uint bits) | masked dL/dRGB 3 | 0 ]
// (the [N,12] gradient already chained to the RAW parameters, and the generator of the SH gradient — see K8c), one per
// Gaussian whose renderer gradient row is non-zero, instead of dense [N,12] + [N,3] tensors.
//   k_compact_gradient_rows    : renderer rows (gut_trace_bwd_ex(..., GUT_BWD_SKIP_EPILOGUE)) -> records, consumed rows zeroed
//   k_scatter_gradient_records : one view's records -> += dense raw-gradient accumulator, = that view's [N,3] dL/dRGB slab
// Every id occurs at most once per view and the views are scattered one launch after the other in rank order, so the sums
// are formed in the same order on every rank: the replicas stay bit-identical.  The dense targets are zero everywhere except
// where records landed, and k_sh_adam<false>(clear_consumed) zeroes exactly those again.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool any_bits(const float4& a) {
    return ((__float_as_uint(a.x) | __float_as_uint(a.y) | __float_as_uint(a.z) | __float_as_uint(a.w)) & 0x7fffffffu) != 0u;
}

__global__ __launch_bounds__(kBlock) void k_compact_gradient_rows(uint32_t n, const float4* __restrict__ act12,
                                                                 const uint32_t* __restrict__ tiles_count,
                                                                 const float* __restrict__ feat, float4* __restrict__ grad16,
                                                                 float4* __restrict__ records, uint32_t capacity,
                                                                 uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    float4 g0, g1, g2, g3;
    bool have = false;
    if (i < n && tiles_count[i] != 0) {
        g0 = grad16[4 * (size_t)i + 0];
        g1 = grad16[4 * (size_t)i + 1];
        g2 = grad16[4 * (size_t)i + 2];
        g3 = grad16[4 * (size_t)i + 3];
        have = any_bits(g0) || any_bits(g1) || any_bits(g2) || any_bits(g3);
    }
    const unsigned long long bal = __ballot(have);
    if (bal == 0ull) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(bal));
    base = __shfl(base, 0);
    const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (!have || slot >= capacity) return;  // (capacity is the number of Gaussians: never exceeded)
    const float m0 = feat[3 * (size_t)i + 0] > 0.0f ? g2.w : 0.0f;
    const float m1 = feat[3 * (size_t)i + 1] > 0.0f ? g3.x : 0.0f;
    const float m2 = feat[3 * (size_t)i + 2] > 0.0f ? g3.y : 0.0f;
    const float4 a = act12[3 * (size_t)i], qn = act12[3 * (size_t)i + 1], sc = act12[3 * (size_t)i + 2];
    g0.w = g0.w * a.w * (1.0f - a.w);                                               // sigmoid
    const float dot = g1.x * qn.x + g1.y * qn.y + g1.z * qn.z + g1.w * qn.w;
    const float inv = 1.0f / sc.w;                                                  // 1 / |quat| (pad column of the activated row)
    g1 = make_float4((g1.x - qn.x * dot) * inv, (g1.y - qn.y * dot) * inv, (g1.z - qn.z * dot) * inv, (g1.w - qn.w * dot) * inv);
    records[4 * (size_t)slot + 0] = g0;
    records[4 * (size_t)slot + 1] = g1;
    records[4 * (size_t)slot + 2] = make_float4(g2.x * sc.x, g2.y * sc.y, g2.z * sc.z, __uint_as_float(i));   // exp
    records[4 * (size_t)slot + 3] = make_float4(m0, m1, m2, 0.0f);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    grad16[4 * (size_t)i + 0] = z; grad16[4 * (size_t)i + 1] = z; grad16[4 * (size_t)i + 2] = z; grad16[4 * (size_t)i + 3] = z;
}

// d_count != nullptr: the number of valid records is read on the device (at most `count` of them are taken): the data-parallel
// step queues the scatter before the host knows the ranks' record counts (dp.RecordExchange)
__global__ __launch_bounds__(kBlock) void k_scatter_gradient_records(const float4* __restrict__ records, uint32_t count, uint32_t n,
                                                                    float4* __restrict__ grad12, float* __restrict__ mrgb_view,
                                                                    const uint32_t* __restrict__ d_count) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (d_count) count = min(count, *d_count);
    if (r >= count) return;
    const float4 g0 = records[4 * (size_t)r + 0], g1 = records[4 * (size_t)r + 1], g2 = records[4 * (size_t)r + 2];
    const float4 m = records[4 * (size_t)r + 3];
    const uint32_t id = __float_as_uint(g2.w);
    if (id >= n) return;  // not a row of this model: a corrupt record must not become a wild store
    float4 t = grad12[3 * (size_t)id + 0];
    grad12[3 * (size_t)id + 0] = make_float4(t.x + g0.x, t.y + g0.y, t.z + g0.z, t.w + g0.w);
    t = grad12[3 * (size_t)id + 1];
    grad12[3 * (size_t)id + 1] = make_float4(t.x + g1.x, t.y + g1.y, t.z + g1.z, t.w + g1.w);
    t = grad12[3 * (size_t)id + 2];
    grad12[3 * (size_t)id + 2] = make_float4(t.x + g2.x, t.y + g2.y, t.z + g2.z, 0.0f);
    mrgb_view[3 * (size_t)id + 0] = m.x;
    mrgb_view[3 * (size_t)id + 1] = m.y;
    mrgb_view[3 * (size_t)id + 2] = m.z;
}

void launch_compact_gradient_rows(hipStream_t s, uint32_t n, const float* act12, const uint32_t* tiles_count, const float* feat,
                                  float* grad16, float* records, uint32_t capacity, uint32_t* count) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_compact_gradient_rows, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n,
                       reinterpret_cast<const float4*>(act12), tiles_count, feat, reinterpret_cast<float4*>(grad16),
                       reinterpret_cast<float4*>(records), capacity, count);
}

}  // namespace gut

// AdamParams for the adam4 kernels: lr[] already holds lr / (1 - beta1^t) and bias2_sqrt holds 1 / sqrt(1 - beta2^t)
// (both 1-free when bias correction is off), so the device code multiplies only.  k_adam_step keeps the plain form.
static void fill_adam(gut::AdamParams& ap, const float* lr, uint32_t cols, float beta1, float beta2, float eps, uint32_t step) {
    ap.beta1 = beta1; ap.beta2 = beta2; ap.eps = eps; ap.cols = cols;
    float bias1 = 1.0f, bias2_sqrt = 1.0f;
    if (step) {
        bias1 = (float)(1.0 - pow((double)beta1, (double)step));
        bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    }
    for (uint32_t i = 0; i < 64; ++i) ap.lr[i] = i < cols ? lr[i] / bias1 : 0.0f;
    ap.bias1 = 1.0f;
    ap.bias2_sqrt = 1.0f / bias2_sqrt;
}

namespace gut {
void launch_sh_adam_from_scratch(hipStream_t s, uint32_t n, int sh_degree, const float* d_camera_position, float* grad16,
                                 const uint32_t* tiles_count, const float* feat, float* raw12, float* raw_m, float* raw_v,
                                 float* sh48, float* sh_m, float* sh_v, const float* lr12, const float* lr48, float beta1, float beta2,
                                 float eps, uint32_t step, const float* visibility, float* act12_out, bool rows_with_tiles_only,
                                 const uint8_t* wave_walked, uint32_t split_block, uint32_t extra_end, const LazyMoments& lazy,
                                 const uint8_t* rule_walked, float* stat_accum, int32_t* stat_denom) {
    if (n == 0) return;
    ShAdamParams sp;
    sp.stat_accum = stat_accum; sp.stat_denom = stat_denom;
    sp.lazy = lazy;
    sp.rule_walked = rule_walked;
    sp.rows_with_tiles_only = rows_with_tiles_only ? 1 : 0;
    sp.own.walked = wave_walked; sp.own.split_block = split_block; sp.own.extra_end = extra_end;
    sp.clear_consumed = 0;
    fill_adam(sp.a12, lr12, 12, beta1, beta2, eps, step);
    fill_adam(sp.a48, lr48, 48, beta1, beta2, eps, step);
    sp.cam = d_camera_position;
    sp.n = n; sp.views = 1; sp.sh_degree = sh_degree; sp.grad_scale = 1.0f; sp.view_stride = n;
    hipLaunchKernelGGL(k_sh_adam<true>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, sp, (float*)nullptr,
                       reinterpret_cast<float4*>(grad16), reinterpret_cast<float4*>(raw12), reinterpret_cast<float4*>(raw_m),
                       reinterpret_cast<float4*>(raw_v), reinterpret_cast<float4*>(sh48), reinterpret_cast<float4*>(sh_m),
                       reinterpret_cast<float4*>(sh_v), visibility, reinterpret_cast<float4*>(act12_out), tiles_count, feat);
}

void launch_adam_rows_without_gradient(hipStream_t s, uint32_t n, const uint32_t* tiles_count, float* raw12, float* raw_m, float* raw_v,
                                       float* sh48, float* sh_m, float* sh_v, const float* lr12, const float* lr48, float beta1,
                                       float beta2, float eps, uint32_t step, float* act12_out, uint32_t block_begin,
                                       uint32_t block_end, const uint8_t* wave_walked, uint32_t split_block, uint32_t extra_end,
                                       bool second_launch, const LazyMoments& lazy) {
    if (n == 0 || block_end <= block_begin) return;
    EarlyOwnership own;
    own.walked = wave_walked; own.split_block = split_block; own.extra_end = extra_end;
    AdamParams a12, a48;
    fill_adam(a12, lr12, 12, beta1, beta2, eps, step);
    fill_adam(a48, lr48, 48, beta1, beta2, eps, step);
    static int num_cus = 0;  // same for every device of a node
    if (num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cus = prop.multiProcessorCount;
        if (num_cus <= 0) num_cus = 256;
    }
    const uint32_t nblocks = block_end - block_begin;
    static int wgs_per_cu = 0, wgs_per_cu2 = 0;
    if (wgs_per_cu == 0) {
        const char* e = getenv("GUT_EARLY_WGS_PER_CU");  // tuning experiments only
        wgs_per_cu = e ? atoi(e) : 1;
        if (wgs_per_cu < 1 || wgs_per_cu > 8) wgs_per_cu = 1;
        // second launch (under the backward compositor): two workgroups per CU since the lazy moment decay — the kernel then reads
        // 720 B and writes 288 B per row, and with one wave per SIMD it is latency-bound (4.5 TB/s alone); measured on a box whose
        // stream was on the step's critical path, interleaved runs: 2.70 / 2.73 -> 2.65 / 2.65 ms per step (second launch
        // 1.31 -> 1.15 ms, K7 0.90 -> 0.95 ms beside it); two per CU in the FIRST launch too (under K6, which is the more
        // sensitive compositor): no further gain
        const char* e2 = getenv("GUT_EARLY_WGS_PER_CU2");
        wgs_per_cu2 = e2 ? atoi(e2) : (e ? wgs_per_cu : 2);
        if (wgs_per_cu2 < 1 || wgs_per_cu2 > 8) wgs_per_cu2 = wgs_per_cu;
    }
    const uint32_t cap = (uint32_t)(second_launch ? wgs_per_cu2 : wgs_per_cu) * (uint32_t)num_cus;
    const uint32_t grid = nblocks < cap ? nblocks : cap;
    // the first launch (beside K6) in its 32-register form wherever that form exists (lazy moments): measured on the bench frame,
    // interleaved runs on one box, K6 0.476 -> 0.445 ms beside it (it keeps its fifth wave) and, with 60 % instead of 25 % of the
    // row blocks given to the first launch, the second launch 1.05 -> 0.87 ms and the step's tail 0.27 -> 0.23 ms
    static int narrow_first = -1;
    if (narrow_first < 0) {
        const char* e = getenv("GUT_EARLY_NARROW");  // tuning experiments only
        narrow_first = e ? atoi(e) : 1;
    }
    auto kern = lazy.wave_step ? ((narrow_first && !second_launch) ? k_adam_rows_without_gradient_narrow : k_adam_rows_without_gradient<true>)
                               : k_adam_rows_without_gradient<false>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, s, a12, a48, n, tiles_count,
                       reinterpret_cast<float4*>(raw12), reinterpret_cast<float4*>(raw_m), reinterpret_cast<float4*>(raw_v),
                       reinterpret_cast<float4*>(sh48), reinterpret_cast<float4*>(sh_m), reinterpret_cast<float4*>(sh_v),
                       reinterpret_cast<float4*>(act12_out), block_begin, block_end, own, second_launch ? 1u : 0u, lazy);
}

// k_mark_walked_waves: wave_walked[id / 64] = 1 for every Gaussian id among the list entries the forward compositor walked
// (the first tile_walked[tile] entries of each tile's ordered list).  One workgroup per tile; racing stores all write 1.
__global__ __launch_bounds__(kBlock) void k_mark_walked_waves(uint32_t n, const uint2* __restrict__ ranges,
                                                             const uint32_t* __restrict__ tile_walked,
                                                             const uint32_t* __restrict__ ids, uint8_t* __restrict__ wave_walked) {
    const uint2 r = ranges[blockIdx.x];
    const uint32_t depth = min(r.y - r.x, tile_walked[blockIdx.x]);
    for (uint32_t j = threadIdx.x; j < depth; j += kBlock) {
        const uint32_t id = ids[r.x + j];
        if (id < n) wave_walked[id >> 6] = 1;
    }
}

// statistics (gut_get_stats): rows in the waves the side-stream pass owned in the last step
__global__ __launch_bounds__(kBlock) void k_count_side_stream_rows(uint32_t n, const uint32_t* __restrict__ tiles_count,
                                                                  EarlyOwnership own, Counters* __restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t wave_first = i & ~63u;
    const bool has_tiles = __ballot(i < n && tiles_count[i] != 0) != 0ull;
    if (wave_first < n && (threadIdx.x & 63) == 0 && side_stream_owns_wave(own, has_tiles, wave_first >> 6, blockIdx.x)) {
        atomicAdd(&out->side_stream_rows, (unsigned long long)min(64u, n - wave_first));
        if (side_stream_owns_wave(own, has_tiles, wave_first >> 6, blockIdx.x, false))
            atomicAdd(&out->side_stream_rows_first, (unsigned long long)min(64u, n - wave_first));
    }
}

void launch_count_side_stream_rows(hipStream_t s, uint32_t n, const uint32_t* tiles_count, const uint8_t* wave_walked,
                                   uint32_t split_block, uint32_t extra_end, Counters* out) {
    if (n == 0) return;
    EarlyOwnership own;
    own.walked = wave_walked; own.split_block = split_block; own.extra_end = extra_end;
    hipLaunchKernelGGL(k_count_side_stream_rows, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, tiles_count, own, out);
}

__global__ __launch_bounds__(kBlock) void k_mark_waves_with_tiles(uint32_t n, const uint32_t* __restrict__ tiles_count,
                                                                 uint8_t* __restrict__ wave_flags) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool has = __ballot(i < n && tiles_count[i] != 0) != 0ull;
    if ((threadIdx.x & 63) == 0 && i < n && has) wave_flags[i >> 6] = 1;
}

void launch_mark_waves_with_tiles(hipStream_t s, uint32_t n, const uint32_t* tiles_count, uint8_t* wave_flags) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_mark_waves_with_tiles, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, tiles_count, wave_flags);
}

void launch_mark_walked_waves(hipStream_t s, uint32_t n, uint32_t tiles, const uint32_t* ranges, const uint32_t* tile_walked,
                              const uint32_t* ids, uint8_t* wave_walked) {
    if (n == 0 || tiles == 0) return;
    hipLaunchKernelGGL(k_mark_walked_waves, dim3(tiles), dim3(kBlock), 0, s, n, reinterpret_cast<const uint2*>(ranges), tile_walked,
                       ids, wave_walked);
}
}  // namespace gut

namespace gut {
// k_sync_moments: brings the stored moments of every wave up to step t (m *= beta1^(t - wave_step), same for v) and marks them so:
// what any reader would compute on the fly.  Before anything that moves rows between waves or looks at the moments from outside.
__global__ __launch_bounds__(kBlock) void k_sync_moments(uint32_t n, float4* __restrict__ m12, float4* __restrict__ v12,
                                                        float4* __restrict__ m48, float4* __restrict__ v48, LazyMoments lz) {
    const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t wave_first = wave * 64u;
    if (wave_first >= n) return;
    const uint32_t seen = lz.wave_step[wave];
    if (seen >= lz.t) return;
    const uint32_t d = lz.t - seen;
    const uint32_t k = d < lz.len ? d : lz.len - 1u;
    if (d >= lz.len && lz.overrun && lane == 0) *lz.overrun = 1u;
    const float f1 = lz.pow1[k], f2 = lz.pow2[k];
    const uint32_t rows_here = min(64u, n - wave_first);
    for (uint32_t q = lane; q < rows_here * 3u; q += 64u) {
        float4 a = m12[(size_t)wave_first * 3 + q], b = v12[(size_t)wave_first * 3 + q];
        scale4(a, f1); scale4(b, f2);
        m12[(size_t)wave_first * 3 + q] = a; v12[(size_t)wave_first * 3 + q] = b;
    }
    for (uint32_t q = lane; q < rows_here * 12u; q += 64u) {
        float4 a = m48[(size_t)wave_first * 12 + q], b = v48[(size_t)wave_first * 12 + q];
        scale4(a, f1); scale4(b, f2);
        m48[(size_t)wave_first * 12 + q] = a; v48[(size_t)wave_first * 12 + q] = b;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) lz.wave_step[wave] = lz.t;
}
}  // namespace gut

namespace gut {
__global__ __launch_bounds__(256) void k_resize_sph_rows(uint32_t n, uint32_t in_width, uint32_t out_width, const float* __restrict__ in,
                                                         float* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)n * out_width) return;
    const uint32_t row = (uint32_t)(idx / out_width), col = (uint32_t)(idx - (size_t)row * out_width);
    out[idx] = col < in_width ? in[(size_t)row * in_width + col] : 0.0f;
}

void launch_resize_sph_rows(hipStream_t s, uint32_t n, uint32_t in_width, uint32_t out_width, const float* in, float* out) {
    const size_t total = (size_t)n * out_width;
    if (total == 0) return;
    hipLaunchKernelGGL(k_resize_sph_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n, in_width, out_width, in, out);
}

void launch_pack_activate_fields(hipStream_t s, uint32_t n, const float* pos, const float* dns, const float* rot, const float* scl,
                                 float* act12) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack_activate_fields, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, pos, dns,
                       reinterpret_cast<const float4*>(rot), scl, reinterpret_cast<float4*>(act12));
}
}  // namespace gut

// GutLazyMoments of the C ABI + the step being applied -> the kernels' argument (NULL / no wave_step: moments written every step)
gut::LazyMoments gut_make_lazy(const GutLazyMoments* lazy, uint32_t step) {
    gut::LazyMoments lz;
    if (lazy && lazy->d_wave_step && lazy->d_pow_beta1 && lazy->d_pow_beta2 && lazy->table_len >= 2 && step >= 1) {
        lz.wave_step = lazy->d_wave_step; lz.pow1 = lazy->d_pow_beta1; lz.pow2 = lazy->d_pow_beta2; lz.len = lazy->table_len; lz.t = step;
        lz.overrun = lazy->d_overrun;
    }
    return lz;
}

extern "C" {

int gut_sync_moments(void* stream, uint32_t num_particles, float* d_raw_m, float* d_raw_v, float* d_sh_m, float* d_sh_v,
                     const GutLazyMoments* lazy, uint32_t step) {
    if (num_particles == 0) return 0;
    if (!d_raw_m || !d_raw_v || !d_sh_m || !d_sh_v || !lazy) return 1;
    gut::LazyMoments lz = gut_make_lazy(lazy, step ? step : 1u);
    if (!lz.wave_step) return 3;
    lz.t = step;   // (step 0: nothing has been applied yet, nothing to bring up to date)
    const uint32_t waves = (num_particles + 63u) / 64u;
    hipLaunchKernelGGL(gut::k_sync_moments, dim3((waves + 3u) / 4u), dim3(gut::kBlock), 0, static_cast<hipStream_t>(stream), num_particles,
                       reinterpret_cast<float4*>(d_raw_m), reinterpret_cast<float4*>(d_raw_v), reinterpret_cast<float4*>(d_sh_m),
                       reinterpret_cast<float4*>(d_sh_v), lz);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_activate_pack(void* stream, uint32_t num_particles, const float* d_raw12, float* d_act12) {
    if (num_particles == 0) return 0;
    if (!d_raw12 || !d_act12) return 1;
    hipLaunchKernelGGL(gut::k_activate_pack, dim3((num_particles + gut::kBlock - 1) / gut::kBlock), dim3(gut::kBlock), 0,
                       static_cast<hipStream_t>(stream), num_particles, reinterpret_cast<const float4*>(d_raw12),
                       reinterpret_cast<float4*>(d_act12));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_adam_step(void* stream, uint64_t rows, uint32_t cols, float* d_param, const float* d_grad, float* d_exp_avg,
                  float* d_exp_avg_sq, const float* lr_per_col /* host, cols floats */, float beta1, float beta2, float eps,
                  uint32_t step /* 1-based; 0 = no bias correction (reference SelectiveAdam) */,
                  const float* d_visibility /* [rows] or NULL */) {
    if (rows == 0) return 0;
    if (!d_param || !d_grad || !d_exp_avg || !d_exp_avg_sq || !lr_per_col) return 1;
    if (cols == 0 || cols > 64 || (cols & 3)) return 3;  // rows are processed as float4 groups
    gut::AdamParams ap;
    for (uint32_t i = 0; i < 64; ++i) ap.lr[i] = i < cols ? lr_per_col[i] : 0.0f;
    ap.beta1 = beta1; ap.beta2 = beta2; ap.eps = eps; ap.cols = cols;
    if (step) {
        ap.bias1 = (float)(1.0 - pow((double)beta1, (double)step));
        ap.bias2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    } else {
        ap.bias1 = 1.0f; ap.bias2_sqrt = 1.0f;
    }
    const uint64_t n_vec4 = rows * (cols / 4);
    const uint64_t blocks = (n_vec4 + gut::kBlock - 1) / gut::kBlock;
    if (blocks > 0x7fffffffull) return 4;
    hipLaunchKernelGGL(gut::k_adam_step, dim3((uint32_t)blocks), dim3(gut::kBlock), 0, static_cast<hipStream_t>(stream), ap, n_vec4,
                       reinterpret_cast<float4*>(d_param), reinterpret_cast<const float4*>(d_grad),
                       reinterpret_cast<float4*>(d_exp_avg), reinterpret_cast<float4*>(d_exp_avg_sq), d_visibility);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_selective_adam(void* stream, uint64_t rows, uint32_t cols, float* d_param, const float* d_grad, float* d_exp_avg,
                       float* d_exp_avg_sq, const uint8_t* d_visibility, float lr, float beta1, float beta2, float eps) {
    if (rows == 0 || cols == 0) return 0;
    if (!d_param || !d_grad || !d_exp_avg || !d_exp_avg_sq || !d_visibility) return 1;
    const uint64_t n = rows * cols;
    const uint64_t blocks = (n + gut::kBlock - 1) / gut::kBlock;
    if (blocks > 0x7fffffffull) return 4;
    hipLaunchKernelGGL(gut::k_selective_adam, dim3((uint32_t)blocks), dim3(gut::kBlock), 0, static_cast<hipStream_t>(stream), n, cols,
                       d_param, d_grad, d_exp_avg, d_exp_avg_sq, d_visibility, lr, beta1, beta2, eps);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_position_gradient_statistics(void* stream, uint32_t n, const float* d_position_grad, uint32_t grad_stride, const float* d_positions,
                                     uint32_t position_stride, const float* d_sensor_position, float* d_norm_accum,
                                     int32_t* d_norm_denom) {
    if (n == 0) return 0;
    if (!d_position_grad || !d_positions || !d_sensor_position || !d_norm_accum || !d_norm_denom) return 1;
    if (grad_stride < 3 || position_stride < 3) return 3;
    hipLaunchKernelGGL(gut::k_position_gradient_statistics, dim3((n + gut::kBlock - 1) / gut::kBlock), dim3(gut::kBlock), 0,
                       static_cast<hipStream_t>(stream), n, d_position_grad, grad_stride, d_positions, position_stride,
                       d_sensor_position, d_norm_accum, d_norm_denom);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_mcmc_perturb(void* stream, uint32_t n, float* d_raw12, float* d_act12, float noise_scale, uint64_t seed, uint64_t step,
                     const float* d_unit_normals) {
    if (n == 0) return 0;
    if (!d_raw12) return 1;
    if (!(noise_scale == noise_scale)) return 3;
    hipLaunchKernelGGL(gut::k_mcmc_perturb, dim3((n + gut::kBlock - 1) / gut::kBlock), dim3(gut::kBlock), 0, static_cast<hipStream_t>(stream),
                       n, reinterpret_cast<float4*>(d_raw12), reinterpret_cast<float4*>(d_act12), noise_scale, (uint32_t)seed,
                       (uint32_t)(seed >> 32) ^ (uint32_t)step, (uint32_t)(step >> 32), d_unit_normals);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_mcmc_relocation(void* stream, int32_t n, const float* d_opacities, const float* d_scales, const int32_t* d_ratios,
                        const float* d_binoms, int32_t n_max, float* d_new_opacities, float* d_new_scales) {
    if (n <= 0) return 0;
    if (!d_opacities || !d_scales || !d_ratios || !d_binoms || !d_new_opacities || !d_new_scales || n_max <= 0) return 1;
    hipLaunchKernelGGL(gut::k_mcmc_relocation, dim3((n + gut::kBlock - 1) / gut::kBlock), dim3(gut::kBlock), 0,
                       static_cast<hipStream_t>(stream), n, d_opacities, d_scales, d_ratios, d_binoms, n_max, d_new_opacities,
                       d_new_scales);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}


int gut_sh_adam_step_ex(void* stream, uint32_t num_particles, int32_t sh_degree, uint32_t num_views, const float* d_camera_positions,
                        float* d_mrgb, float* d_raw_grad12, float grad_scale, float* d_raw12, float* d_raw_m,
                        float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v, const float* lr12, const float* lr48,
                        float beta1, float beta2, float eps, uint32_t step, const float* d_visibility, float* d_act12_out,
                        uint32_t mrgb_view_stride, uint32_t flags, const uint8_t* d_wave_flags, const GutLazyMoments* lazy) {
    if (num_particles == 0) return 0;
    if (!d_camera_positions || !d_mrgb || !d_raw_grad12 || !d_raw12 || !d_raw_m || !d_raw_v || !d_sh48 || !d_sh_m || !d_sh_v ||
        !lr12 || !lr48)
        return 1;
    if (num_views == 0 || num_views > 1024 || sh_degree < 0 || sh_degree > 3) return 3;
    if (flags & ~(uint32_t)GUT_ADAM_CLEAR_CONSUMED_GRADS) return 3;
    gut::ShAdamParams sp;
    fill_adam(sp.a12, lr12, 12, beta1, beta2, eps, step);
    fill_adam(sp.a48, lr48, 48, beta1, beta2, eps, step);
    sp.cam = d_camera_positions;
    sp.n = num_particles; sp.views = num_views; sp.sh_degree = sh_degree; sp.grad_scale = grad_scale;
    sp.view_stride = mrgb_view_stride ? mrgb_view_stride : num_particles;
    sp.rows_with_tiles_only = 0;
    sp.own.walked = d_wave_flags; sp.own.split_block = 0; sp.own.extra_end = 0;
    sp.clear_consumed = (flags & GUT_ADAM_CLEAR_CONSUMED_GRADS) ? 1 : 0;
    if (sp.view_stride < num_particles) return 3;
    if (lazy && d_visibility) return 3;   // a visibility mask leaves rows untouched: their moments do not decay at all
    sp.lazy = gut_make_lazy(lazy, step);
    sp.rule_walked = nullptr;
    sp.stat_accum = nullptr; sp.stat_denom = nullptr;
    hipLaunchKernelGGL(gut::k_sh_adam<false>, dim3((num_particles + gut::kBlock - 1) / gut::kBlock), dim3(gut::kBlock), 0,
                       static_cast<hipStream_t>(stream), sp, d_mrgb, reinterpret_cast<float4*>(d_raw_grad12),
                       reinterpret_cast<float4*>(d_raw12), reinterpret_cast<float4*>(d_raw_m), reinterpret_cast<float4*>(d_raw_v),
                       reinterpret_cast<float4*>(d_sh48), reinterpret_cast<float4*>(d_sh_m), reinterpret_cast<float4*>(d_sh_v),
                       d_visibility, reinterpret_cast<float4*>(d_act12_out), (const uint32_t*)nullptr, (const float*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_sh_adam_step(void* stream, uint32_t num_particles, int32_t sh_degree, uint32_t num_views, const float* d_camera_positions,
                     const float* d_mrgb, const float* d_raw_grad12, float grad_scale, float* d_raw12, float* d_raw_m,
                     float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v, const float* lr12, const float* lr48,
                     float beta1, float beta2, float eps, uint32_t step, const float* d_visibility, float* d_act12_out,
                     uint32_t mrgb_view_stride) {
    return gut_sh_adam_step_ex(stream, num_particles, sh_degree, num_views, d_camera_positions, const_cast<float*>(d_mrgb),
                               const_cast<float*>(d_raw_grad12), grad_scale, d_raw12, d_raw_m, d_raw_v, d_sh48, d_sh_m, d_sh_v, lr12,
                               lr48, beta1, beta2, eps, step, d_visibility, d_act12_out, mrgb_view_stride, 0u, nullptr, nullptr);
}

int gut_adam_unwalked_waves_ex(void* stream, uint32_t num_particles, const uint8_t* d_wave_flags, float* d_raw12, float* d_raw_m,
                               float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v, const float* lr12, const float* lr48,
                               float beta1, float beta2, float eps, uint32_t step, float* d_act12_out, const GutLazyMoments* lazy) {
    if (num_particles == 0) return 0;
    if (!d_wave_flags || !d_raw12 || !d_raw_m || !d_raw_v || !d_sh48 || !d_sh_m || !d_sh_v || !lr12 || !lr48) return 1;
    const uint32_t nblocks = (num_particles + gut::kBlock - 1) / gut::kBlock;
    gut::launch_adam_rows_without_gradient(static_cast<hipStream_t>(stream), num_particles, nullptr, d_raw12, d_raw_m, d_raw_v, d_sh48,
                                           d_sh_m, d_sh_v, lr12, lr48, beta1, beta2, eps, step, d_act12_out, 0, nblocks, d_wave_flags,
                                           0, nblocks, true, gut_make_lazy(lazy, step));
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_adam_unwalked_waves(void* stream, uint32_t num_particles, const uint8_t* d_wave_flags, float* d_raw12, float* d_raw_m,
                            float* d_raw_v, float* d_sh48, float* d_sh_m, float* d_sh_v, const float* lr12, const float* lr48,
                            float beta1, float beta2, float eps, uint32_t step, float* d_act12_out) {
    return gut_adam_unwalked_waves_ex(stream, num_particles, d_wave_flags, d_raw12, d_raw_m, d_raw_v, d_sh48, d_sh_m, d_sh_v, lr12, lr48,
                                      beta1, beta2, eps, step, d_act12_out, nullptr);
}

int gut_scatter_gradient_records(void* stream, const float* d_records, uint32_t count, uint32_t num_particles, float* d_raw_grad12,
                                 float* d_mrgb_view) {
    if (count == 0) return 0;
    if (!d_records || !d_raw_grad12 || !d_mrgb_view) return 1;
    hipLaunchKernelGGL(gut::k_scatter_gradient_records, dim3((count + gut::kBlock - 1) / gut::kBlock), dim3(gut::kBlock), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(d_records), count, num_particles,
                       reinterpret_cast<float4*>(d_raw_grad12), d_mrgb_view, (const uint32_t*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_scatter_gradient_records_dev(void* stream, const float* d_records, const uint32_t* d_count, uint32_t max_count,
                                     uint32_t num_particles, float* d_raw_grad12, float* d_mrgb_view) {
    if (max_count == 0) return 0;
    if (!d_records || !d_count || !d_raw_grad12 || !d_mrgb_view) return 1;
    hipLaunchKernelGGL(gut::k_scatter_gradient_records, dim3((max_count + gut::kBlock - 1) / gut::kBlock), dim3(gut::kBlock), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(d_records), max_count, num_particles,
                       reinterpret_cast<float4*>(d_raw_grad12), d_mrgb_view, d_count);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
