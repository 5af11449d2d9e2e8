// gut_render_common.h — device helpers shared by the compositing kernels (gut_render.hip, gut_render_sorted.hip)
#pragma once
#include "gut_internal.h"

namespace gut {

// ---- small helpers ------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, ~1 ulp
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

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
