// gut_render_common.h — device helpers shared by the compositing kernels (gut_render.hip, gut_render_sorted.hip)
#pragma once
#include "gut_internal.h"

namespace gut {

// ---- small helpers ------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, ~1 ulp
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// ---- generalised Gaussian kernels (render.particle_kernel_degree != 2) ----------------------------------------------------
// particleResponse<n> / particleResponseGrd<n> of gaussianParticles.cuh:211-306: resp = exp(s_n d2^(n/2)), s_n = -4.5 / 3^n
// (n = 0: the linear hat max(1 + s sqrt(d2), 0)).  The degree is a kernel ARGUMENT here (block-uniform scalar branch) and only
// the `kGeneral` instantiations of the compositors look at it: the default quadratic path is compiled exactly as before.
__device__ __forceinline__ float kernel_response(int degree, float d2) {
    switch (degree) {
    case 8: { const float q = d2 * d2; return fast_exp(-0.000685871056241f * (q * q)); }
    case 5: return fast_exp(-0.0185185185185f * (d2 * d2 * fast_sqrt(d2)));
    case 4: return fast_exp(-0.0555555555556f * (d2 * d2));
    case 3: return fast_exp(-0.166666666667f * (d2 * fast_sqrt(d2)));
    case 1: return fast_exp(-1.5f * fast_sqrt(d2));
    case 0: return fmaxf(1.0f + -0.329630334487f * fast_sqrt(d2), 0.0f);
    default: return fast_exp(-0.5f * d2);
    }
}
// dL/d(d2) from dL/d(resp), as the reference writes it — including its degree-1 form, which multiplies by sqrt(d2) where the
// derivative of exp(s sqrt(d2)) divides by it (gaussianParticles.cuh:248-252): restated, not corrected.  That is the CUDA backward of
// the unsorted compositor (processHitBwd); the sorted variant is differentiated by slang autodiff from the forward expression
// (slang/models/gaussianParticles.slang:119-164) and gets the true derivative: as_written = false.
__device__ __forceinline__ float kernel_response_grad(int degree, float d2, float resp, float g_resp, bool as_written = true) {
    switch (degree) {
    case 8: return (float)(-0.000685871056241 * 4.0) * (d2 * d2) * d2 * resp * g_resp;
    case 5: return (float)(-0.0185185185185 * 2.5) * d2 * fast_sqrt(d2) * resp * g_resp;
    case 4: return (float)(-0.0555555555556 * 2.0) * d2 * resp * g_resp;
    case 3: return (float)(-0.166666666667 * 1.5) * fast_sqrt(d2) * resp * g_resp;
    case 1: return (-1.5f * 0.5f) * (as_written ? fast_sqrt(d2) : fast_rsq(d2)) * resp * g_resp;
    case 0: return resp > 0.0f ? (0.5f * -0.329630334487f * fast_rsq(d2)) * g_resp : 0.0f;
    default: return -0.5f * resp * g_resp;
    }
}
struct RayState {
    float ox, oy, oz, dx, dy, dz, tmin, tmax;
    float ex, ey, ez;  // ray origin minus the sensor position (all zero for pinhole / fisheye rays)
    bool valid;
    bool centred;      // the camera-space ray origin is exactly (0,0,0)
};

// camera-space ray -> world, slab test against the +-1e6 scene box (rayPayload.cuh:76-108,
// utils/bounding_box.h:88-134)
__device__ __forceinline__ RayState make_ray(const ViewParams& v, const float* __restrict__ ray_ori,
                                             const float* __restrict__ ray_dir, size_t pix, bool inside) {
    RayState r;
    r.valid = false;
    r.centred = true;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = 0.f;
    r.ex = r.ey = r.ez = 0.f;
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
    r.centred = (a0 == 0.0f) && (a1 == 0.0f) && (a2 == 0.0f);
    r.ex = m.r[0][0] * a0 + m.r[0][1] * a1 + m.r[0][2] * a2;
    r.ey = m.r[1][0] * a0 + m.r[1][1] * a1 + m.r[1][2] * a2;
    r.ez = m.r[2][0] * a0 + m.r[2][1] * a1 + m.r[2][2] * a2;
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

// Canonical-space ray origin:  o = M (ray_o - mu) = M (sensor_pos - mu) + M (ray_o - sensor_pos) = oc + M e.
// oc is a per-entry constant computed once at staging; e is a per-pixel constant that is exactly zero for every
// camera the reference has (rays start at the sensor position), in which case the whole tile skips the M e term
// (block-uniform flag) and the per-pair work is one 3x3 mat-vec instead of two.
struct FwdEntry {      // 80 bytes, 16-byte aligned: five ds_read_b128 broadcasts per entry
    float4 mu_sigma;   // oc = M (sensor_pos - mean), density
    float4 m0;         // row 0 of M = diag(1/s) * rotationT, s.x
    float4 m1;         // row 1, s.y
    float4 m2;         // row 2, s.z
    float4 feat_id;    // max(rgb, 0), particle id (bit pattern)
};


// ---- pixel layout of a tile and per-wave culling -----------------------------------------------------
// A wave owns an 8x8-pixel block of the 16x16 tile (wave w: block column w & 1, block row w >> 1; lane l: pixel (l & 7, l >> 3)
// of the block).  Tile lists are built per 16x16 tile, so a particle whose footprint covers only part of the tile is listed for
// all four waves although some of them cannot hit it.  A compact 8x8 block meets fewer footprints than a 16x4 strip does
// (oracle_count_wave_pairs on the bench frame: 2.55 M (wave, entry) pairs with a hit lane against 2.75 M, -7.3 %).
// Each wave encloses the directions of its rays in two double wedges — elevation and azimuth about the tile's central
// direction — each between two planes through the sensor position, and the lane that stages an entry tests the entry's
// cut-off ellipsoid {x : |M (x - mu)|^2 <= D} against the four wedges of each wave once; D is the largest d2 at which a hit is
// still possible (response and alpha thresholds).  The test is exact for a wedge and conservative for the rays inside it: an
// entry is skipped by a wave only if the ellipsoid lies entirely outside the wave's elevation wedge or entirely outside its
// azimuth wedge, i.e. if no line through the sensor position with a direction inside both wedges comes within D of it.
//   wedge:      directions d with tau0 <= (a.d)/(c.d) <= tau1   (c: the tile's central direction; a: n0, its image-down direction
//               orthogonal to c, or m0 = c x n0, its image-right direction; tau0/tau1: min/max over the wave's rays, widened by
//               a relative 1e-5)
//   planes:     a(tau) = a - tau c      (a(tau).d has the sign of tau_d - tau for c.d > 0)
//   ellipsoid entirely outside  <=>  both plane distances delta_k = a_k.(mu - sensor) exceed the support half-width
//               h_k = sqrt(D) |diag(s) R a_k| with the same sign  (lines extend both ways, hence the double wedge).
// Rays that do not start at the sensor position, ragged tiles whose reference pixels fall outside the image and tiles whose
// rays spread too widely around their central direction switch the test off for the whole tile (all masks 0xF).
__device__ __forceinline__ int tile_px(uint32_t tid) { return (int)(((tid >> 6) & 1u) * 8u + (tid & 7u)); }
__device__ __forceinline__ int tile_py(uint32_t tid) { return (int)((tid >> 7) * 8u + ((tid >> 3) & 7u)); }
__device__ __forceinline__ uint32_t tid_of_tile_pixel(int x, int y) {
    return (uint32_t)(((y >> 3) * 2 + (x >> 3)) * 64 + (y & 7) * 8 + (x & 7));
}

struct StripPlanes {
    float n[4][4][3];  // per wave: elevation planes (tau_lo, tau_hi), azimuth planes (sigma_lo, sigma_hi)
    uint32_t usable;
    float ref[3][3];  // scratch: rays of the top-middle, bottom-middle and centre pixel
    uint32_t ref_ok;
};

// Called by all 256 threads of the tile (contains barriers).
__device__ __forceinline__ void build_strip_planes(StripPlanes& sp, const RayState& ray, bool inside, bool centred, uint32_t tid) {
    if (tid == 0) sp.ref_ok = 1u;
    __syncthreads();
    const int slot = tid == tid_of_tile_pixel(8, 0) ? 0 : (tid == tid_of_tile_pixel(8, 15) ? 1 : (tid == tid_of_tile_pixel(8, 8) ? 2 : -1));
    if (slot >= 0) {
        sp.ref[slot][0] = ray.dx; sp.ref[slot][1] = ray.dy; sp.ref[slot][2] = ray.dz;
        if (!inside) sp.ref_ok = 0u;
    }
    __syncthreads();
    bool ok = centred && (sp.ref_ok != 0u);
    float c0 = 0.f, c1 = 0.f, c2 = 1.f, n0 = 0.f, n1 = 1.f, n2 = 0.f;
    {
        const float* a = sp.ref[0]; const float* b = sp.ref[1]; const float* m = sp.ref[2];
        const float la = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], lb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
        const float lm = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
        ok = ok && (la > 0.f) && (lb > 0.f) && (lm > 0.f) && (la < 1e30f) && (lb < 1e30f) && (lm < 1e30f);
        if (ok) {
            const float ia = 1.0f / sqrtf(la), ib = 1.0f / sqrtf(lb), im = 1.0f / sqrtf(lm);
            c0 = m[0] * im; c1 = m[1] * im; c2 = m[2] * im;
            float d0 = b[0] * ib - a[0] * ia, d1 = b[1] * ib - a[1] * ia, d2 = b[2] * ib - a[2] * ia;
            const float dc = d0 * c0 + d1 * c1 + d2 * c2;
            d0 -= dc * c0; d1 -= dc * c1; d2 -= dc * c2;
            const float ld = d0 * d0 + d1 * d1 + d2 * d2;
            ok = ld > 1e-12f;
            if (ok) {
                const float id = 1.0f / sqrtf(ld);
                n0 = d0 * id; n1 = d1 * id; n2 = d2 * id;
            }
        }
    }
    // image-right direction: m0 = c x n0 (unit, orthogonal to both; its sign is irrelevant: the wedge is [min, max] of sigma)
    const float m0 = c1 * n2 - c2 * n1, m1 = c2 * n0 - c0 * n2, m2 = c0 * n1 - c1 * n0;
    // this lane's elevation / azimuth tangents inside the wedge parametrisation
    float tau_lo = 3.0e38f, tau_hi = -3.0e38f, sig_lo = 3.0e38f, sig_hi = -3.0e38f;
    bool lane_ok = true;
    if (inside) {
        const float cd = c0 * ray.dx + c1 * ray.dy + c2 * ray.dz;
        const float nd = n0 * ray.dx + n1 * ray.dy + n2 * ray.dz;
        const float md = m0 * ray.dx + m1 * ray.dy + m2 * ray.dz;
        const float l2 = ray.dx * ray.dx + ray.dy * ray.dy + ray.dz * ray.dz;
        lane_ok = (l2 > 0.f) && (l2 < 1e30f) && (cd * cd > 0.01f * l2) && (cd > 0.f);  // within ~84 degrees of the central direction
        const float icd = 1.0f / cd;
        const float tau = nd * icd, sig = md * icd;
        if (lane_ok) { tau_lo = tau; tau_hi = tau; sig_lo = sig; sig_hi = sig; }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        tau_lo = fminf(tau_lo, __shfl_xor(tau_lo, m));
        tau_hi = fmaxf(tau_hi, __shfl_xor(tau_hi, m));
        sig_lo = fminf(sig_lo, __shfl_xor(sig_lo, m));
        sig_hi = fmaxf(sig_hi, __shfl_xor(sig_hi, m));
    }
    const bool all_ok = __syncthreads_and((ok && lane_ok) ? 1 : 0) != 0;
    if ((tid & 63u) == 0u) {
        const uint32_t w = tid >> 6;
        const float mag = 1.0f + fmaxf(fabsf(tau_lo), fabsf(tau_hi));
        const float t0 = tau_lo - 1e-5f * mag, t1 = tau_hi + 1e-5f * mag;
        sp.n[w][0][0] = n0 - t0 * c0; sp.n[w][0][1] = n1 - t0 * c1; sp.n[w][0][2] = n2 - t0 * c2;
        sp.n[w][1][0] = n0 - t1 * c0; sp.n[w][1][1] = n1 - t1 * c1; sp.n[w][1][2] = n2 - t1 * c2;
        const float mags = 1.0f + fmaxf(fabsf(sig_lo), fabsf(sig_hi));
        const float s0 = sig_lo - 1e-5f * mags, s1 = sig_hi + 1e-5f * mags;
        sp.n[w][2][0] = m0 - s0 * c0; sp.n[w][2][1] = m1 - s0 * c1; sp.n[w][2][2] = m2 - s0 * c2;
        sp.n[w][3][0] = m0 - s1 * c0; sp.n[w][3][1] = m1 - s1 * c1; sp.n[w][3][2] = m2 - s1 * c2;
    }
    if (tid == 0) sp.usable = all_ok ? 1u : 0u;
    __syncthreads();
}

// bit w set <=> wave w has to evaluate the entry.  a = (mean, density), r = rows of rotationT, s = scale
template <bool kGeneral = false>
__device__ __forceinline__ uint32_t strip_mask(const StripPlanes& sp, const ViewParams& v, const RenderConsts& c, const float4& a,
                                               const float (&r)[3][3], const float4& s, int kernel_degree = 2) {
    if (sp.usable == 0u) return 0xFu;
    const float ratio = a.w / c.alpha_threshold;  // alpha = resp * density > threshold  <=>  d2 < 2 ln(density / threshold)
    if (!(ratio > 1.0f)) return 0u;               // cannot be hit at all (K1 culls these already)
    const float D = (!kGeneral ? fminf(c.max_d2, 2.0f * __logf(ratio))
                               : kernel_cutoff_d2(kernel_degree, fmaxf(c.min_response, 1.0f / ratio))) * 1.001f + 1e-3f;
    const float m0 = a.x - v.s2w.t[0], m1 = a.y - v.s2w.t[1], m2 = a.z - v.s2w.t[2];
    uint32_t mask = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        bool outside = false;
#pragma unroll
        for (int axis = 0; axis < 2; ++axis) {  // elevation wedge, azimuth wedge
            bool above = true, below = true;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float n0 = sp.n[w][2 * axis + k][0], n1 = sp.n[w][2 * axis + k][1], n2 = sp.n[w][2 * axis + k][2];
                const float delta = n0 * m0 + n1 * m1 + n2 * m2;
                const float t0 = s.x * (r[0][0] * n0 + r[0][1] * n1 + r[0][2] * n2);
                const float t1 = s.y * (r[1][0] * n0 + r[1][1] * n1 + r[1][2] * n2);
                const float t2 = s.z * (r[2][0] * n0 + r[2][1] * n1 + r[2][2] * n2);
                const bool clear = delta * delta > D * (t0 * t0 + t1 * t1 + t2 * t2);  // |delta| > h
                above = above && clear && (delta > 0.0f);
                below = below && clear && (delta < 0.0f);
            }
            outside = outside || above || below;
        }
        if (!outside) mask |= 1u << w;
    }
    return mask;
}

// ---- lazy per-tile depth order ------------------------------------------------------------------------
// Whole-tile termination leaves most of a tile's list untouched (bench frame: 1.28 M of 9.36 M entries are ever staged),
// so the global sort only groups the (tile | depth, id) pairs by TILE (the two high radix passes; stable, so every tile's
// entries stay in particle-id order) and the forward compositor orders each tile on demand: before it stages a chunk it
// selects the next <= 512 entries in (depth bits, list position) order — list position == particle-id order, i.e. exactly
// the order the full stable sort on (tile | depth) produces — writes their ids to the ordered-id list (the backward and
// the tests read that) and stages them.  Selection = up to 4 x 8-bit radix select on the depth bits among the entries behind
// the last one taken, one gather pass, one 512-element bitonic sort in LDS (lazy_select below).
constexpr uint32_t kLazyBatch = 512;  // entries ordered per selection: two 256-entry chunks (power of two for the bitonic sort)

struct LazyOrder {
    uint32_t hist[256];
    unsigned long long sel[kLazyBatch];  // (depth bits << 32) | list position: one 64-bit compare orders two entries
    uint32_t wave_cnt[2][4][4];  // [less | equal][unrolled position][wave]
    uint32_t bin, need, bin_count;
    uint32_t cnt_lt, cnt_eq;  // slots handed out by the unordered gather
};

// keys: the tile's slice of the tile-grouped (tile << 32 | depth bits) keys; total = its length; want = min(kLazyBatch,
// entries not yet taken); (have_lo, lo) = the last entry taken so far, in sel[]'s form.  Called by all 256 threads (contains
// barriers).  On return sel[t], t < want, hold the next `want` entries in final order.
// The tile's depths (the first kLazyCache of them) are copied ONCE per selection into LDS the compositing loop is not using at
// that moment (kcache = its staging area: the previous chunk has been walked, the next is not staged yet) and the four digit
// passes and the gather read them from there: one round of L2 / HBM latency per selection instead of five.  Only lists longer
// than the cache stream their remainder from memory in every pass, four 256-entry rows at a time with the four loads issued
// back to back.  Histogram increments are aggregated per wave on the digit of the wave's first counting lane (the high digits
// of a tile's depths are nearly all equal: one LDS atomic instead of 64 serialised ones).
// (Not inlined: inlined, its register needs made the compiler spill the compositing loop's state, 2.4x slower; keeping 8 or 12
// rows of depths in REGISTERS across the passes instead of in LDS does the same to the caller even when not inlined — the
// values the loop keeps live across the call no longer fit beside the callee's: 9 scratch accesses inside the per-entry loop.)
constexpr uint32_t kLazyCache = 4096;  // depth words: 16 KB = sizeof(PackEntry) * kBlock

__device__ __noinline__ void lazy_select(LazyOrder& S, uint32_t* __restrict__ kcache, const uint2* __restrict__ keys, uint32_t total,
                                         uint32_t want, bool have_lo, unsigned long long lo, uint32_t tid) {
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    auto behind_lo = [&](uint32_t d, uint32_t p) { return !have_lo || (((unsigned long long)d << 32) | p) > lo; };
    // (the ballot of a condition as it stands in the condition register: __ballot() materialises an int per lane first)
    auto ballot = [](bool b) -> unsigned long long { return __builtin_amdgcn_ballot_w64(b); };
    auto lanes_below = [](unsigned long long b) -> uint32_t {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    };
    const uint32_t cached = min(total, kLazyCache);
    for (uint32_t base = 0; base < cached; base += 8 * kBlock) {
        uint32_t d[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) d[u] = (base + u * kBlock + tid) < cached ? keys[base + u * kBlock + tid].x : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u)
            if ((base + u * kBlock + tid) < cached) kcache[base + u * kBlock + tid] = d[u];
    }
    // four consecutive 256-entry rows of depths starting at the block-uniform `base` (a multiple of 4 * kBlock, so a group is
    // cached as a whole or not at all)
    auto fetch4 = [&](uint32_t base, uint32_t (&d)[4]) {
        if (base < kLazyCache) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) d[u] = (base + u * kBlock + tid) < total ? kcache[base + u * kBlock + tid] : 0u;
        } else {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) d[u] = (base + u * kBlock + tid) < total ? keys[base + u * kBlock + tid].x : 0u;
        }
    };
    // 1. depth of the want-th smallest remaining entry, digit by digit
    // The passes stop as soon as the bin that holds the want-th entry has at most 64 members (normally after the second: 16
    // depth bits leave about a dozen): a single wave then ranks that bin's members exactly (step 2b) instead of two more
    // passes over the whole list.
    uint32_t prefix = 0, need = want, cmask = 0xFFFFFFFFu;  // cmask: the depth bits the passes have decided
    bool short_bin = false;
    for (int shift = 24; shift >= 0; shift -= 8) {
        S.hist[tid] = 0u;
        __syncthreads();  // (first pass: the cache is complete behind this barrier too)
        const uint32_t hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t base = 0; base < total; base += 4 * kBlock) {
            uint32_t d[4];
            fetch4(base, d);
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t p = base + u * kBlock + tid;
                const bool ok = p < total && behind_lo(d[u], p) && ((d[u] & hi_mask) == (prefix & hi_mask));
                const unsigned long long b_ok = ballot(ok);
                if (b_ok == 0ull) continue;  // wave-uniform
                const uint32_t digit = (d[u] >> shift) & 255u;
                const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane(__ffsll((long long)b_ok) - 1);
                const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)digit, (int)first);
                const bool same = ok && digit == lead;
                const unsigned long long b_same = ballot(same);
                if (lane == first) atomicAdd(&S.hist[lead], (uint32_t)__popcll(b_same));
                if (ok && !same) atomicAdd(&S.hist[digit], 1u);
            }
        }
        __syncthreads();
        if (wave == 0) {  // first bin at which the running count reaches `need`
            const uint32_t h0 = S.hist[4 * lane], h1 = S.hist[4 * lane + 1], h2 = S.hist[4 * lane + 2], h3 = S.hist[4 * lane + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            uint32_t incl = mine;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                if ((int)lane >= o) incl += up;
            }
            const unsigned long long reach = __ballot(incl >= need);
            const uint32_t first = (uint32_t)__ffsll((long long)reach) - 1u;  // always found: the candidates number >= need
            if (lane == first) {
                uint32_t c = incl - mine, b = 4 * lane;
                if (c + h0 < need) { c += h0; ++b; if (c + h1 < need) { c += h1; ++b; if (c + h2 < need) { c += h2; ++b; } } }
                S.bin = b;
                S.need = need - c;
                S.bin_count = S.hist[b];
                S.cnt_lt = 0u;
                S.cnt_eq = 0u;
            }
        }
        __syncthreads();
        prefix |= S.bin << shift;
        need = S.need;
        if (shift > 0 && S.bin_count <= 64u) {  // block-uniform
            short_bin = true;
            cmask = 0xFFFFFFFFu << shift;
            break;
        }
    }
    const uint32_t dstar = prefix;        // the decided depth bits of the want-th entry: `need` of the entries that share them are taken
    const uint32_t n_less = want - need;  // entries below them: all taken
    // the bin's members: straight into their slots in list order when the bin is one exact depth (only `need` of them fit), else
    // into a side list (the histogram's storage, free now) that step 2b ranks
    const uint32_t bin_count = S.bin_count;
    unsigned long long* const eq_dst = short_bin ? reinterpret_cast<unsigned long long*>(S.hist) : S.sel + n_less;
    const uint32_t eq_cap = short_bin ? bin_count : need;
    // 2. gather.  Short bin: in any order — the sort orders the slots and step 2b ranks the bin's members —, a wave takes the
    // slots of a row's entries with one returning LDS atomic.  Else ordered (list order = row, then wave, then lane): of the
    // entries at the exact depth the first `need` of the list are the ones to take.
    uint32_t got_lt = 0, got_eq = 0;
    if (short_bin) {
        for (uint32_t base = 0; base < total; base += 4 * kBlock) {
            uint32_t d[4];
            fetch4(base, d);
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const uint32_t p = base + u * kBlock + tid;
                const bool cand = (p < total) && behind_lo(d[u], p);
                const bool lt = cand && (d[u] & cmask) < dstar, eq = cand && (d[u] & cmask) == dstar;
                const unsigned long long key = ((unsigned long long)d[u] << 32) | p;
                const unsigned long long b_lt = ballot(lt), b_eq = ballot(eq);
                if (b_lt != 0ull) {  // wave-uniform
                    uint32_t off = 0;
                    if (lane == 0) off = atomicAdd(&S.cnt_lt, (uint32_t)__popcll(b_lt));
                    off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
                    if (lt) S.sel[off + lanes_below(b_lt)] = key;
                }
                if (b_eq != 0ull) {
                    uint32_t off = 0;
                    if (lane == 0) off = atomicAdd(&S.cnt_eq, (uint32_t)__popcll(b_eq));
                    off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
                    if (eq) eq_dst[off + lanes_below(b_eq)] = key;
                }
            }
        }
        __syncthreads();
    } else
    for (uint32_t base = 0; base < total; base += 4 * kBlock) {
        if (got_lt == n_less && got_eq >= eq_cap) break;  // block-uniform
        uint32_t d[4];
        unsigned long long b_lt[4], b_eq[4];
        bool lt[4], eq[4];
        fetch4(base, d);
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint32_t p = base + u * kBlock + tid;
            const bool cand = (p < total) && behind_lo(d[u], p);
            lt[u] = cand && (d[u] & cmask) < dstar;
            eq[u] = cand && (d[u] & cmask) == dstar;
            b_lt[u] = ballot(lt[u]);
            b_eq[u] = ballot(eq[u]);
            if (lane == 0) {
                S.wave_cnt[0][u][wave] = (uint32_t)__popcll(b_lt[u]);
                S.wave_cnt[1][u][wave] = (uint32_t)__popcll(b_eq[u]);
            }
        }
        __syncthreads();
        uint32_t run_lt = got_lt, run_eq = got_eq;
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            uint32_t my_lt = run_lt, my_eq = run_eq;
#pragma unroll
            for (uint32_t w = 0; w < kBlock / 64; ++w) {
                if (w < wave) { my_lt += S.wave_cnt[0][u][w]; my_eq += S.wave_cnt[1][u][w]; }
                run_lt += S.wave_cnt[0][u][w];
                run_eq += S.wave_cnt[1][u][w];
            }
            my_lt += lanes_below(b_lt[u]);
            my_eq += lanes_below(b_eq[u]);
            const uint32_t p = base + u * kBlock + tid;
            const unsigned long long key = ((unsigned long long)d[u] << 32) | p;
            if (lt[u]) S.sel[my_lt] = key;
            if (eq[u] && my_eq < eq_cap) eq_dst[my_eq] = key;
        }
        got_lt = run_lt;
        got_eq = run_eq;
        __syncthreads();
    }
    // 2b. the `need` smallest of a short bin's members, each to the slot of its rank among them (keys are distinct)
    if (short_bin && wave == 0) {
        const unsigned long long key = lane < bin_count ? eq_dst[lane] : ~0ull;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < bin_count; ++j) rank += eq_dst[j] < key ? 1u : 0u;  // (one broadcast LDS read per member)
        if (lane < bin_count && rank < need) S.sel[n_less + rank] = key;
    }
    for (uint32_t t = tid; t < kLazyBatch; t += kBlock)
        if (t >= want) S.sel[t] = ~0ull;
    __syncthreads();
    // 3. bitonic sort of the kLazyBatch slots by (depth, position); every thread owns one compare-exchange per stage
    for (uint32_t k = 2; k <= kLazyBatch; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < kLazyBatch / 2; t += kBlock) {
                const uint32_t e = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), o = e | j;
                const unsigned long long a = S.sel[e], b = S.sel[o];
                if ((a > b) == ((e & k) == 0u)) {
                    S.sel[e] = b;
                    S.sel[o] = a;
                }
            }
            __syncthreads();
        }
}

// stage one list entry (lane-private id) into its LDS slot in the canonical-space form
__device__ __forceinline__ FwdEntry make_entry(const ViewParams& v, const float4* __restrict__ density12,
                                               const float* __restrict__ feat, uint32_t id) {
    FwdEntry e;
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
    }
    return e;
}

}  // namespace gut
