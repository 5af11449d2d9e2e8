// gut_render_sorted.hip — the sorted variant of the compositor (render.splat.k_buffer_size = K, 1 <= K <= 16;
// the reference's paper config `paper/3dgut/base_sorted.yaml` uses 16).
//   forward : gutKBufferRenderer.cuh:28-76 (HitParticleKBuffer), :217-292 (evalKBuffer), :108-170 (processHitParticle)
//   backward: same traversal with Backward = true; the reference differentiates each composited hit with slang
//             autodiff of the back-to-front recurrence ("undo" form, gaussianParticles.slang:394-451,
//             shRadiativeParticles.slang:179-207) and adds every per-pixel contribution with its own atomics.
//
// Each ray keeps the K closest pending hits in registers, sorted by hit distance; when the buffer is full the
// closest one is composited before the new hit is inserted, and the rest is drained in order at the end.  The
// gradient of a composited hit is the exact derivative of the forward (incl. min(0.99,.) and the hit distance),
// written here in the front-to-back residual form the unsorted backward uses (mathematically identical to the
// reference's undo form).  Because a hit is composited up to K entries after it was found — possibly in a later
// list chunk — contributions cannot be wave-reduced per entry; like the reference, every (pixel, hit) issues its
// own float atomics into the 64-byte gradient rows.  This is the secondary variant (README numbers of the
// reference are all "unsorted"); it is built for parity, not tuned.
//
// Deviation: the reference's sorted BACKWARD reads the UNclamped precomputed colour while its forward composites
// max(colour, 0) (gutKBufferRenderer.cuh:130 vs :161); that breaks its own undo recurrence whenever a colour
// channel is negative.  The library follows the reference's formula by default (GUT_OPT_SORTED_REFERENCE_BACKWARD = 1);
// gut_set_option(GUT_OPT_SORTED_REFERENCE_BACKWARD, 0) selects the exact derivative of the forward (clamped colour in both passes).
#include "gut_internal.h"
#include "gut_render_common.h"

namespace gut {

constexpr int kKMax = 16;

struct KBuffer {
    float t[kKMax];
    float a[kKMax];
    uint32_t id[kKMax];
    int num;
};

__device__ __forceinline__ void kb_init(KBuffer& kb) {
#pragma unroll
    for (int i = 0; i < kKMax; ++i) {
        kb.t[i] = -1.0f;
        kb.a[i] = 0.0f;
        kb.id[i] = kInvalid;
    }
    kb.num = 0;
}

// slot `first` = kKMax - K is the closest stored hit when the buffer is full
__device__ __forceinline__ void kb_front(const KBuffer& kb, int first, float& t, float& a, uint32_t& id) {
    t = kb.t[0]; a = kb.a[0]; id = kb.id[0];
#pragma unroll
    for (int s = 1; s < kKMax; ++s)
        if (s == first) { t = kb.t[s]; a = kb.a[s]; id = kb.id[s]; }
}

__device__ __forceinline__ void kb_invalidate(KBuffer& kb, int first) {
#pragma unroll
    for (int s = 0; s < kKMax; ++s)
        if (s == first) kb.t[s] = -1.0f;
}

// HitParticleKBuffer::insert: bubble from the back while farther than the stored entries (slots >= first only)
__device__ __forceinline__ void kb_insert(KBuffer& kb, int first, float t, float a, uint32_t id) {
#pragma unroll
    for (int s = kKMax - 1; s >= 0; --s)
        if (s >= first && t > kb.t[s]) {
            const float tt = kb.t[s], ta = kb.a[s];
            const uint32_t ti = kb.id[s];
            kb.t[s] = t; kb.a[s] = a; kb.id[s] = id;
            t = tt; a = ta; id = ti;
        }
}

// response of a staged entry for this lane's ray; returns accept (response / alpha / range tests of the forward)
__device__ __forceinline__ bool eval_entry(const RenderConsts& c, int kernel_degree, const RayState& ray, bool centred, const FwdEntry& e,
                                           float& alpha, float& hit_t) {
    float o0 = e.mu_sigma.x, o1 = e.mu_sigma.y, o2 = e.mu_sigma.z;
    if (!centred) {
        o0 += e.m0.x * ray.ex + e.m0.y * ray.ey + e.m0.z * ray.ez;
        o1 += e.m1.x * ray.ex + e.m1.y * ray.ey + e.m1.z * ray.ez;
        o2 += e.m2.x * ray.ex + e.m2.y * ray.ey + e.m2.z * ray.ez;
    }
    const float u0 = e.m0.x * ray.dx + e.m0.y * ray.dy + e.m0.z * ray.dz;
    const float u1 = e.m1.x * ray.dx + e.m1.y * ray.dy + e.m1.z * ray.dz;
    const float u2 = e.m2.x * ray.dx + e.m2.y * ray.dy + e.m2.z * ray.dz;
    const float x0 = u1 * o2 - u2 * o1, x1 = u2 * o0 - u0 * o2, x2 = u0 * o1 - u1 * o0;
    const float l2 = u0 * u0 + u1 * u1 + u2 * u2;
    const float il2 = fast_rcp(l2);
    const float d2 = (x0 * x0 + x1 * x1 + x2 * x2) * il2;
    if (!(d2 < c.max_d2)) return false;
    const float resp = kernel_response(kernel_degree, d2);
    alpha = fminf(c.max_alpha, resp * e.mu_sigma.w);
    if (!((resp > c.min_response) && (alpha > c.alpha_threshold))) return false;
    const float proj = -(u0 * o0 + u1 * o1 + u2 * o2) * il2;
    const float h0 = e.m0.w * u0 * proj, h1 = e.m1.w * u1 * proj, h2 = e.m2.w * u2 * proj;
    hit_t = fast_sqrt(h0 * h0 + h1 * h1 + h2 * h2);
    return (hit_t > ray.tmin) && (hit_t < ray.tmax);
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_render_sorted(ViewParams v, RenderConsts c, int K, const float4* __restrict__ density12,
                                                         const float* __restrict__ feat, const float* __restrict__ ray_ori,
                                                         const float* __restrict__ ray_dir, const uint2* __restrict__ ranges,
                                                         const uint32_t* __restrict__ sorted_ids, const uint32_t* __restrict__ d_num_intersections,
                                                         float4* __restrict__ rgba, float* __restrict__ dist,
                                                         float* __restrict__ hits, int kernel_degree) {
    __shared__ FwdEntry stage[kBlock];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const int px = (int)(tile % (uint32_t)v.grid_x) * kTile + (int)(tid & 15);
    const int py = (int)(tile / (uint32_t)v.grid_x) * kTile + (int)(tid >> 4);
    const bool inside = (px < v.width) && (py < v.height);
    const size_t pix = (size_t)py * (size_t)v.width + (size_t)px;
    const uint32_t num_intersections = *d_num_intersections;  // read on the device: the host queues this launch before it knows it
    const RayState ray = make_ray(v, ray_ori, ray_dir, pix, inside && (num_intersections != 0));
    const bool centred = __syncthreads_and(ray.centred ? 1 : 0) != 0;
    const uint2 range = ranges[tile];
    const uint32_t total = range.y - range.x;
    const int first = kKMax - K;

    bool alive = ray.valid;
    float T = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, dsum = 0.f;
    uint32_t nhits = 0;
    KBuffer kb;
    kb_init(kb);

    auto composite = [&](uint32_t id, float a, float t) {
        const float w = a * T;
        dsum += t * w;
        T *= (1.0f - a);
        if (w > 0.0f) {
            cr += fmaxf(feat[3 * (size_t)id + 0], 0.0f) * w;
            cg += fmaxf(feat[3 * (size_t)id + 1], 0.0f) * w;
            cb += fmaxf(feat[3 * (size_t)id + 2], 0.0f) * w;
            nhits++;
        }
        if (T < c.min_transmittance) alive = false;
    };

    for (uint32_t base = 0; base < total; base += kBlock) {
        if (!__syncthreads_or(alive ? 1 : 0)) break;
        {
            const uint32_t k = range.x + base + tid;
            stage[tid] = make_entry(v, density12, feat, k < range.y ? sorted_ids[k] : kInvalid);
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)kBlock, total - base);
        for (uint32_t j = 0; j < cnt; ++j) {
            if (__ballot(alive) == 0ull) break;
            const FwdEntry e = stage[j];
            const uint32_t id = __float_as_uint(e.feat_id.w);
            if (id == kInvalid) {  // padding entry ends the list for everyone
                alive = alive && false;
                base = total;
                break;
            }
            float a = 0.f, t = 0.f;
            if (alive && eval_entry(c, kernel_degree, ray, centred, e, a, t)) {
                if (kb.num == K) {
                    float ft, fa; uint32_t fi;
                    kb_front(kb, first, ft, fa, fi);
                    composite(fi, fa, ft);
                    kb_invalidate(kb, first);
                } else {
                    kb.num++;
                }
                kb_insert(kb, first, t, a, id);
            }
        }
    }
    // drain the pending hits, closest first
#pragma unroll
    for (int s = 0; s < kKMax; ++s)
        if (alive && s >= kKMax - kb.num) composite(kb.id[s], kb.a[s], kb.t[s]);

    if (inside) {
        if (ray.valid) {
            rgba[pix] = make_float4(cr, cg, cb, 1.0f - T);
            dist[pix] = dsum;
            hits[pix] = (float)nhits;
        } else {
            rgba[pix] = make_float4(0.f, 0.f, 0.f, 0.f);
            dist[pix] = 1e06f;
            hits[pix] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
struct PixelGrad {
    float T_final, rgbF[3], distF;     // forward results
    float g_rgb[3], g_opacity, g_dist;  // upstream gradients
    float T, rgb_run[3], dist_run;      // running front-to-back state
    float undo_rgb[3];                  // reference_undo: the reference's "integrated features" state (starts at the final colour)
    bool reference_undo;                // colour term of d(alpha) exactly as the reference computes it, see backward_hit
};

// exact derivatives of one composited hit w.r.t. the particle's parameters; 14 atomics into its gradient row
__device__ void backward_hit(const ViewParams& v, const RenderConsts& c, int kernel_degree, const RayState& ray, const float4* __restrict__ density12,
                             const float* __restrict__ feat, float* __restrict__ grad16, uint32_t id, float alpha, float hit_t,
                             PixelGrad& pg) {
    const float4 mu = density12[3 * (size_t)id + 0];
    const float4 q = density12[3 * (size_t)id + 1];
    const float4 sc = density12[3 * (size_t)id + 2];
    float r[3][3];
    quat_rows(q.x, q.y, q.z, q.w, r);
    const float is[3] = {1.0f / sc.x, 1.0f / sc.y, 1.0f / sc.z};
    const float s[3] = {sc.x, sc.y, sc.z};
    const float p[3] = {ray.ox - mu.x, ray.oy - mu.y, ray.oz - mu.z};
    const float d[3] = {ray.dx, ray.dy, ray.dz};
    float o[3], u[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o[i] = is[i] * (r[i][0] * p[0] + r[i][1] * p[1] + r[i][2] * p[2]);
        u[i] = is[i] * (r[i][0] * d[0] + r[i][1] * d[1] + r[i][2] * d[2]);
    }
    const float l2 = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    const float il2 = 1.0f / l2, il = fast_rsq(l2);
    const float uo = u[0] * o[0] + u[1] * o[1] + u[2] * o[2];
    const float t = uo * il2;
    const float op[3] = {o[0] - t * u[0], o[1] - t * u[1], o[2] - t * u[2]};  // o_perp
    const float d2 = op[0] * op[0] + op[1] * op[1] + op[2] * op[2];          // == |u x o|^2 / |u|^2
    const float resp = kernel_response(kernel_degree, d2);
    const float a0 = resp * mu.w;

    const float f[3] = {fmaxf(feat[3 * (size_t)id + 0], 0.0f), fmaxf(feat[3 * (size_t)id + 1], 0.0f),
                        fmaxf(feat[3 * (size_t)id + 2], 0.0f)};
    const float T = pg.T, w = alpha * T, Tn = T * (1.0f - alpha);
    pg.dist_run += w * hit_t;
    float b_rgb[3] = {0.f, 0.f, 0.f}, b_dist = 0.f;  // composite of everything behind this hit
#pragma unroll
    for (int k = 0; k < 3; ++k) pg.rgb_run[k] += w * f[k];
    if (Tn > 1e-20f) {
        const float iT = 1.0f / Tn;
#pragma unroll
        for (int k = 0; k < 3; ++k) b_rgb[k] = (pg.rgbF[k] - pg.rgb_run[k]) * iT;
        b_dist = (pg.distF - pg.dist_run) * iT;
    }
    float g_alpha = pg.g_opacity * pg.T_final / (1.0f - alpha) + pg.g_dist * T * (hit_t - b_dist);
    if (!pg.reference_undo) {
#pragma unroll
        for (int k = 0; k < 3; ++k) g_alpha += T * pg.g_rgb[k] * (f[k] - b_rgb[k]);
    } else {
        // GUT_OPT_SORTED_REFERENCE_BACKWARD: the reference's sorted backward un-does the back-to-front recurrence
        // C_k = lerp(C_{k+1}, colour, alpha) from the forward's final colour with the UNCLAMPED precomputed colour
        // (gutKBufferRenderer.cuh:127-131 hands particleFeatures[idx] to featuresIntegrateBwd, while the forward composited
        // max(colour, 0), :159-161; recurrence: shRadiativeParticles.slang:179-207 around integrateRadiance<true>, :83-99):
        //     C_{k+1} = (C_k - colour * alpha) / (1 - alpha),   d(alpha) += (colour - C_{k+1}) . dL/dC_k,  dL/dC_k = T_k dL/dC
        // With a negative colour channel C_{k+1} is no longer what lies behind the hit, and stays off for every later hit
        // of the ray.  The feature gradient w * dL/dC and its (colour > 0) mask (K8) are the same in both forms.
        const float wgt = 1.0f / (1.0f - alpha);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float fu = feat[3 * (size_t)id + k];
            pg.undo_rgb[k] = (pg.undo_rgb[k] - fu * alpha) * wgt;
            g_alpha += T * pg.g_rgb[k] * (fu - pg.undo_rgb[k]);
        }
    }
    const float g_hit = w * pg.g_dist;
    pg.T = Tn;

    float out[14];
    // colour: K8 applies the (precomputed RGB > 0) mask and the SH basis
    out[11] = w * pg.g_rgb[0]; out[12] = w * pg.g_rgb[1]; out[13] = w * pg.g_rgb[2];
    // alpha = min(max_alpha, sigma * resp): zero derivative where clamped
    const bool clamped = a0 >= c.max_alpha;
    out[3] = clamped ? 0.0f : resp * g_alpha;
    const float g_d2x2 = clamped ? 0.0f : 2.0f * kernel_response_grad(kernel_degree, d2, resp, mu.w * g_alpha, false);  // 2 dL/d(d2)
    float go[3] = {g_d2x2 * op[0], g_d2x2 * op[1], g_d2x2 * op[2]};
    float gu[3] = {-t * go[0], -t * go[1], -t * go[2]};
    float gs_direct[3] = {0.f, 0.f, 0.f};
    if (g_hit != 0.0f && hit_t > 0.0f) {
        // hit_t = | s * grd * pr |, grd = u/|u|, pr = -(grd . o)
        const float grd[3] = {u[0] * il, u[1] * il, u[2] * il};
        const float pr = -(grd[0] * o[0] + grd[1] * o[1] + grd[2] * o[2]);
        float gv[3], K = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float vk = s[k] * grd[k] * pr;
            gv[k] = g_hit * vk / hit_t;
            K += gv[k] * s[k] * grd[k];
        }
        float ggrd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gs_direct[k] = gv[k] * grd[k] * pr;
            ggrd[k] = pr * s[k] * gv[k] - K * o[k];
            go[k] += -K * grd[k];
        }
        const float dot = ggrd[0] * grd[0] + ggrd[1] * grd[1] + ggrd[2] * grd[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) gu[k] += il * (ggrd[k] - grd[k] * dot);
    }
    // dL/dM_ij = go_i p_j + gu_i d_j, M = diag(1/s) rotationT
    float A[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = go[i] * p[j] + gu[i] * d[j];
    out[0] = -(is[0] * r[0][0] * go[0] + is[1] * r[1][0] * go[1] + is[2] * r[2][0] * go[2]);
    out[1] = -(is[0] * r[0][1] * go[0] + is[1] * r[1][1] * go[1] + is[2] * r[2][1] * go[2]);
    out[2] = -(is[0] * r[0][2] * go[0] + is[1] * r[1][2] * go[1] + is[2] * r[2][2] * go[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        out[8 + i] = -is[i] * is[i] * (A[i][0] * r[i][0] + A[i][1] * r[i][1] + A[i][2] * r[i][2]) + gs_direct[i];
    const float d00 = is[0] * A[0][0], d01 = is[0] * A[0][1], d02 = is[0] * A[0][2];
    const float d10 = is[1] * A[1][0], d11 = is[1] * A[1][1], d12 = is[1] * A[1][2];
    const float d20 = is[2] * A[2][0], d21 = is[2] * A[2][1], d22 = is[2] * A[2][2];
    const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
    out[4] = 2.0f * (qz * (d01 - d10) + qy * (d20 - d02) + qx * (d12 - d21));
    out[5] = 2.0f * (qy * (d01 + d10) + qz * (d02 + d20) + qr * (d12 - d21)) - 4.0f * qx * (d11 + d22);
    out[6] = 2.0f * (qx * (d01 + d10) + qr * (d20 - d02) + qz * (d12 + d21)) - 4.0f * qy * (d00 + d22);
    out[7] = 2.0f * (qr * (d01 - d10) + qx * (d02 + d20) + qy * (d12 + d21)) - 4.0f * qz * (d00 + d11);
#pragma unroll
    for (int k = 0; k < 14; ++k)
        if (out[k] != 0.0f) atomicAdd(&grad16[(size_t)id * 16 + k], out[k]);
}

__global__ __launch_bounds__(kBlock) void k_render_sorted_backward(ViewParams v, RenderConsts c, int K,
                                                                  const float4* __restrict__ density12,
                                                                  const float* __restrict__ feat,
                                                                  const float* __restrict__ ray_ori,
                                                                  const float* __restrict__ ray_dir,
                                                                  const uint2* __restrict__ ranges,
                                                                  const uint32_t* __restrict__ sorted_ids,
                                                                  const float4* __restrict__ rgba, const float* __restrict__ dist,
                                                                  const float4* __restrict__ rgba_grad,
                                                                  const float* __restrict__ dist_grad, float* __restrict__ grad16,
                                                                  int reference_undo, int kernel_degree) {
    __shared__ FwdEntry stage[kBlock];
    const uint32_t tile = blockIdx.x, tid = threadIdx.x;
    const int px = (int)(tile % (uint32_t)v.grid_x) * kTile + (int)(tid & 15);
    const int py = (int)(tile / (uint32_t)v.grid_x) * kTile + (int)(tid >> 4);
    const bool inside = (px < v.width) && (py < v.height);
    const size_t pix = (size_t)py * (size_t)v.width + (size_t)px;
    const RayState ray = make_ray(v, ray_ori, ray_dir, pix, inside);
    const bool centred = __syncthreads_and(ray.centred ? 1 : 0) != 0;
    const uint2 range = ranges[tile];
    const uint32_t total = range.y - range.x;
    const int first = kKMax - K;

    PixelGrad pg;
    pg.T_final = 1.f; pg.distF = 0.f; pg.g_opacity = 0.f; pg.g_dist = 0.f; pg.T = 1.0f; pg.dist_run = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { pg.rgbF[k] = 0.f; pg.g_rgb[k] = 0.f; pg.rgb_run[k] = 0.f; pg.undo_rgb[k] = 0.f; }
    pg.reference_undo = reference_undo != 0;
    if (ray.valid) {
        const float4 o = rgba[pix], g = rgba_grad[pix];
        pg.T_final = 1.0f - o.w;
        pg.rgbF[0] = o.x; pg.rgbF[1] = o.y; pg.rgbF[2] = o.z;
        pg.undo_rgb[0] = o.x; pg.undo_rgb[1] = o.y; pg.undo_rgb[2] = o.z;
        pg.distF = dist[pix];
        pg.g_rgb[0] = g.x; pg.g_rgb[1] = g.y; pg.g_rgb[2] = g.z;
        pg.g_opacity = g.w;
        pg.g_dist = dist_grad ? dist_grad[pix] : 0.0f;
    }
    bool alive = ray.valid;
    KBuffer kb;
    kb_init(kb);

    for (uint32_t base = 0; base < total; base += kBlock) {
        if (!__syncthreads_or(alive ? 1 : 0)) break;
        {
            const uint32_t k = range.x + base + tid;
            stage[tid] = make_entry(v, density12, feat, k < range.y ? sorted_ids[k] : kInvalid);
        }
        __syncthreads();
        const uint32_t cnt = min((uint32_t)kBlock, total - base);
        for (uint32_t j = 0; j < cnt; ++j) {
            if (__ballot(alive) == 0ull) break;
            const FwdEntry e = stage[j];
            const uint32_t id = __float_as_uint(e.feat_id.w);
            if (id == kInvalid) {
                alive = alive && false;
                base = total;
                break;
            }
            float a = 0.f, t = 0.f;
            if (alive && eval_entry(c, kernel_degree, ray, centred, e, a, t)) {
                if (kb.num == K) {
                    float ft, fa; uint32_t fi;
                    kb_front(kb, first, ft, fa, fi);
                    backward_hit(v, c, kernel_degree, ray, density12, feat, grad16, fi, fa, ft, pg);
                    if (pg.T < c.min_transmittance) alive = false;
                    kb_invalidate(kb, first);
                } else {
                    kb.num++;
                }
                kb_insert(kb, first, t, a, id);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < kKMax; ++s)
        if (alive && s >= kKMax - kb.num) {
            backward_hit(v, c, kernel_degree, ray, density12, feat, grad16, kb.id[s], kb.a[s], kb.t[s], pg);
            if (pg.T < c.min_transmittance) alive = false;
        }
}

void launch_render_sorted(hipStream_t s, const ViewParams& v, const RenderConsts& c, int K, const float* density12, const float* feat,
                          const float* ray_ori, const float* ray_dir, const uint32_t* ranges, const uint32_t* sorted_ids,
                          const uint32_t* d_num_intersections, float* rgba, float* dist, float* hits, int kernel_degree) {
    const uint32_t tiles = (uint32_t)(v.grid_x * v.grid_y);
    if (tiles == 0) return;
    hipLaunchKernelGGL(k_render_sorted, dim3(tiles), dim3(kBlock), 0, s, v, c, K, reinterpret_cast<const float4*>(density12), feat,
                       ray_ori, ray_dir, reinterpret_cast<const uint2*>(ranges), sorted_ids, d_num_intersections,
                       reinterpret_cast<float4*>(rgba), dist, hits, kernel_degree);
}

void launch_render_sorted_bwd(hipStream_t s, const ViewParams& v, const RenderConsts& c, int K, const float* density12,
                              const float* feat, const float* ray_ori, const float* ray_dir, const uint32_t* ranges,
                              const uint32_t* sorted_ids, const float* rgba, const float* dist, const float* rgba_grad,
                              const float* dist_grad, float* grad16, bool reference_undo, int kernel_degree) {
    const uint32_t tiles = (uint32_t)(v.grid_x * v.grid_y);
    if (tiles == 0) return;
    hipLaunchKernelGGL(k_render_sorted_backward, dim3(tiles), dim3(kBlock), 0, s, v, c, K, reinterpret_cast<const float4*>(density12),
                       feat, ray_ori, ray_dir, reinterpret_cast<const uint2*>(ranges), sorted_ids,
                       reinterpret_cast<const float4*>(rgba), dist, reinterpret_cast<const float4*>(rgba_grad), dist_grad, grad16,
                       reference_undo ? 1 : 0, kernel_degree);
}

}  // namespace gut
