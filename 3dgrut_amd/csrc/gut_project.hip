// gut_project.hip — per-Gaussian kernels of the 3DGUT path for gfx950:
//   K1 project_on_tiles   (reference: gutProjector.cuh:81-322  GUTProjector::eval)
//   K3 expand_tiles       (reference: gutProjector.cuh:324-388 GUTProjector::expand)
//   K5 tile_ranges        (reference: src/gutRenderer.cu:46-76)
//   K8 project_backward   (reference: gutProjector.cuh:390-430 + gaussianParticles.cuh:120-187)
//
// THIS FILE IS COMPILED WITH -ffp-contract=off.  Everything that feeds an integer decision (tile
// bounding boxes, per-tile culling, the 32 depth bits of the sort key) is evaluated in fp32 in a
// fixed, documented operation order with IEEE-exact +,-,*,/,sqrt and with det_logf / det_atan2f_pos
// instead of the device math library, so that tile/key buffers are reproducible bit for bit
// (numerics contract in DESIGN.md §4).  These kernels are HBM-streaming (one thread per Gaussian,
// 48 B in / 56 B out, +192 B of SH for survivors), so the missing FMA contraction costs nothing.
#include "gut_internal.h"

namespace gut {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// natural log for finite x > 0: fdlibm e_logf algorithm, integer + {+,-,*,/} only
__device__ float det_logf(float x) {
    uint32_t ix = f2u(x);
    int k = 0;
    if (ix >= 0x7f800000u) return x;
    if (ix < 0x00800000u) {
        if (ix == 0) return -__builtin_inff();
        x = x * 33554432.0f;
        ix = f2u(x);
        k -= 25;
    }
    k += (int)(ix >> 23) - 127;
    ix &= 0x007fffffu;
    const uint32_t i = (ix + (0x95f64u << 3)) & 0x800000u;
    x = u2f(ix | (i ^ 0x3f800000u));
    k += (int)(i >> 23);
    const float f = x - 1.0f;
    const float s = f / (2.0f + f);
    const float dk = (float)k;
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (0.40000972152f + w * 0.24279078841f);
    const float t2 = z * (0.66666662693f + w * 0.28498786688f);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    return dk * 6.9313812256e-01f - ((hfsq - (s * (hfsq + R) + dk * 9.0580006145e-06f)) - f);
}

// atan2(y, x), y > 0: Cephes atanf reduction + polynomial
__device__ float det_atan2f_pos(float y, float x) {
    const float ax = fabsf(x);
    const float lo = y < ax ? y : ax;
    const float hi = y < ax ? ax : y;
    float t = lo / hi;
    float base = 0.0f;
    if (t > 0.4142135679721832f) {
        base = 0.7853981852531433f;
        t = (t - 1.0f) / (t + 1.0f);
    }
    const float z = t * t;
    float a = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t;
    a = base + a;
    if (y > ax) a = 1.5707963705062866f - a;
    if (x < 0.0f) a = 3.1415927410125732f - a;
    return a;
}

__device__ __forceinline__ bool within_resolution(float rx, float ry, float tol, float px, float py) {
    const float mx = rx * tol, my = ry * tol;
    return (px > -mx) && (py > -my) && (px < rx + mx) && (py < ry + my);
}

// OpenCV pinhole with rational radial / tangential / thin-prism distortion (cameraProjections.cuh:57-103)
template <bool kDistorted>
__device__ __forceinline__ bool project_pinhole(const ViewParams& v, float px, float py, float pz, float tol, float& ox,
                                                float& oy) {
    if (pz <= 0.0f) {
        ox = 0.0f;
        oy = 0.0f;
        return false;
    }
    // xy / z as multiplication by one IEEE reciprocal (numerics contract, mirrored by the oracle)
    const float rz = 1.0f / pz;
    const float un = px * rz, vn = py * rz;
    if (!kDistorted) {  // all distortion coefficients are zero: icd == 1, delta == 0 exactly
        ox = un * v.focal_length[0] + v.principal_point[0];
        oy = vn * v.focal_length[1] + v.principal_point[1];
        return within_resolution((float)v.width, (float)v.height, tol, ox, oy);
    }
    const float u2 = un * un, v2 = vn * vn;
    const float r2 = u2 + v2;
    const float a1 = 2.0f * un * vn;
    const float a2 = r2 + 2.0f * u2;
    const float a3 = r2 + 2.0f * v2;
    const float num = 1.0f + r2 * (v.radial[0] + r2 * (v.radial[1] + r2 * v.radial[2]));
    const float den = 1.0f + r2 * (v.radial[3] + r2 * (v.radial[4] + r2 * v.radial[5]));
    const float icd = num / den;
    const float dx = v.tangential[0] * a1 + v.tangential[1] * a2 + r2 * (v.thin_prism[0] + r2 * v.thin_prism[1]);
    const float dy = v.tangential[0] * a3 + v.tangential[1] * a1 + r2 * (v.thin_prism[2] + r2 * v.thin_prism[3]);
    const float xd = icd * un + dx, yd = icd * vn + dy;
    const bool radial_ok = (icd > 0.8f) && (icd < 1.2f);
    if (radial_ok) {
        ox = xd * v.focal_length[0] + v.principal_point[0];
        oy = yd * v.focal_length[1] + v.principal_point[1];
    } else {
        const float fw = (float)v.width, fh = (float)v.height;
        const float clip = sqrtf(fw * fw + fh * fh);
        const float k = clip / sqrtf(r2);
        ox = k * un + v.principal_point[0];
        oy = k * vn + v.principal_point[1];
    }
    return radial_ok && within_resolution((float)v.width, (float)v.height, tol, ox, oy);
}

// OpenCV fisheye, theta clamped to max_angle (cameraProjections.cuh:105-128)
__device__ __forceinline__ bool project_fisheye(const ViewParams& v, float px, float py, float pz, float tol, float& ox, float& oy) {
    const float eps = 1.1920929e-07f;
    float rho = sqrtf(px * px + py * py);
    rho = rho > eps ? rho : eps;
    const float theta_full = det_atan2f_pos(rho, pz);
    const float theta = theta_full < v.max_angle ? theta_full : v.max_angle;
    const float t2 = theta * theta;
    float poly = v.radial[3];
    poly = t2 * poly + v.radial[2];
    poly = t2 * poly + v.radial[1];
    poly = t2 * poly + v.radial[0];
    const float delta = (theta * (poly * t2 + 1.0f)) / rho;
    ox = v.focal_length[0] * px * delta + v.principal_point[0];
    oy = v.focal_length[1] * py * delta + v.principal_point[1];
    return (theta < v.max_angle) && within_resolution((float)v.width, (float)v.height, tol, ox, oy);
}

// acos on [0,1] / sin on [0, pi/2] for the rolling-shutter slerp: fdlibm e_acosf / k_sinf / k_cosf in plain fp32
__device__ float det_acosf01(float x) {
    const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666586697e-01f, pS1 = -4.2743422091e-02f, pS2 = -8.6563630030e-03f, qS1 = -7.0662963390e-01f;
    if (x >= 1.0f) return 0.0f;
    if (x < 0.5f) {
        const float z = x * x;
        const float r = (z * (pS0 + z * (pS1 + z * pS2))) / (1.0f + z * qS1);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    const float z = (1.0f - x) * 0.5f;
    const float sq = sqrtf(z);
    const float df = u2f(f2u(sq) & 0xfffff000u);
    const float c = (z - df * df) / (sq + df);
    const float r = (z * (pS0 + z * (pS1 + z * pS2))) / (1.0f + z * qS1);
    const float w = r * sq + c;
    return 2.0f * (df + w);
}
__device__ float det_sinf_0_pio2(float x) {
    if (x <= 0.78539818525f) {
        const float z = x * x;
        const float vv = z * x;
        const float r = 8.3333337680e-03f + z * (-1.9841270114e-04f + z * (2.7557314297e-06f + z * (-2.5050759689e-08f + z * 1.5896910177e-10f)));
        return x + vv * (-1.6666667163e-01f + z * r);
    }
    const float y = (1.5707962513e+00f - x) + 7.5497894159e-08f;
    const float z = y * y;
    const float r = z * (4.1666667908e-02f + z * (-1.3888889225e-03f + z * (2.4801587642e-05f + z * (-2.7557314297e-07f + z * (2.0875723372e-09f + z * -1.1359647598e-11f)))));
    return 1.0f - (0.5f * z - z * r);
}

// kVariant: 0 = pinhole without distortion, 1 = pinhole with distortion, 2 = fisheye (compile-time specialisation of
// the reference's run-time camera-model switch, cameraProjections.cuh:130-144)
template <int kVariant>
__device__ __forceinline__ int project_camera(const ViewParams& v, float cx, float cy, float cz, float tol, float& ox, float& oy) {
    if (kVariant == 0) return project_pinhole<false>(v, cx, cy, cz, tol, ox, oy) ? 1 : 0;
    if (kVariant == 1) return project_pinhole<true>(v, cx, cy, cz, tol, ox, oy) ? 1 : 0;
    return project_fisheye(v, cx, cy, cz, tol, ox, oy) ? 1 : 0;
}

__device__ __forceinline__ float relative_shutter_time(const ViewParams& v, float px, float py) {
    switch (v.shutter) {
    case GUT_SHUTTER_ROLLING_TOP_TO_BOTTOM: return floorf(py) / ((float)v.height - 1.0f);
    case GUT_SHUTTER_ROLLING_LEFT_TO_RIGHT: return floorf(px) / ((float)v.width - 1.0f);
    case GUT_SHUTTER_ROLLING_BOTTOM_TO_TOP: return ((float)v.height - ceilf(py)) / ((float)v.height - 1.0f);
    case GUT_SHUTTER_ROLLING_RIGHT_TO_LEFT: return ((float)v.width - ceilf(px)) / ((float)v.width - 1.0f);
    default: return 0.5f;
    }
}

// world point -> sensor space at relative shutter time a: slerp of the two pose quaternions + mix of the translations
__device__ void shutter_pose_transform(const ViewParams& v, float a, float wx, float wy, float wz, float& cx, float& cy,
                                       float& cz) {
    const float* ps = v.pose_start;
    const float* pe = v.pose_end;
    float q0[4] = {ps[6], ps[3], ps[4], ps[5]}, q1[4] = {pe[6], pe[3], pe[4], pe[5]}, q[4];
    float cosT = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
    if (cosT < 0.0f) {
        for (int i = 0; i < 4; ++i) q1[i] = -q1[i];
        cosT = -cosT;
    }
    if (cosT > 1.0f - 1.1920929e-07f) {
        for (int i = 0; i < 4; ++i) q[i] = q0[i] * (1.0f - a) + q1[i] * a;
    } else {
        const float ang = det_acosf01(cosT);
        const float s0 = det_sinf_0_pio2((1.0f - a) * ang), s1 = det_sinf_0_pio2(a * ang), sd = det_sinf_0_pio2(ang);
        for (int i = 0; i < 4; ++i) q[i] = (s0 * q0[i] + s1 * q1[i]) / sd;
    }
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
    // GLM mat3_cast, columns c0,c1,c2
    const float r00 = 1.0f - 2.0f * (qyy + qzz), r10 = 2.0f * (qxy + qwz), r20 = 2.0f * (qxz - qwy);
    const float r01 = 2.0f * (qxy - qwz), r11 = 1.0f - 2.0f * (qxx + qzz), r21 = 2.0f * (qyz + qwx);
    const float r02 = 2.0f * (qxz + qwy), r12 = 2.0f * (qyz - qwx), r22 = 1.0f - 2.0f * (qxx + qyy);
    const float t0 = ps[0] * (1.0f - a) + pe[0] * a, t1 = ps[1] * (1.0f - a) + pe[1] * a, t2 = ps[2] * (1.0f - a) + pe[2] * a;
    cx = r00 * wx + r01 * wy + r02 * wz + t0;
    cy = r10 * wx + r11 * wy + r12 * wz + t1;
    cz = r20 * wx + r21 * wy + r22 * wz + t2;
}

// projectPointWithShutter<N> (cameraProjections.cuh:146-185), N = render.splat.n_rolling_shutter_iterations (5 by default)
template <int kVariant, bool kRolling>
__device__ __forceinline__ int project_world(const ViewParams& v, float wx, float wy, float wz, float tol, float& ox,
                                             float& oy) {
    const Affine& a = v.w2s_start;
    float cx = a.r[0][0] * wx + a.r[0][1] * wy + a.r[0][2] * wz + a.t[0];
    float cy = a.r[1][0] * wx + a.r[1][1] * wy + a.r[1][2] * wz + a.t[1];
    float cz = a.r[2][0] * wx + a.r[2][1] * wy + a.r[2][2] * wz + a.t[2];
    int valid = project_camera<kVariant>(v, cx, cy, cz, tol, ox, oy);
    if (!kRolling) return valid;
    if (!valid) {
        const Affine& e = v.w2s_end;
        cx = e.r[0][0] * wx + e.r[0][1] * wy + e.r[0][2] * wz + e.t[0];
        cy = e.r[1][0] * wx + e.r[1][1] * wy + e.r[1][2] * wz + e.t[1];
        cz = e.r[2][0] * wx + e.r[2][1] * wy + e.r[2][2] * wz + e.t[2];
        valid = project_camera<kVariant>(v, cx, cy, cz, tol, ox, oy);
        if (!valid) return 0;
    }
    for (int it = 0; it < v.shutter_iterations; ++it) {
        const float al = relative_shutter_time(v, ox, oy);
        shutter_pose_transform(v, al, wx, wy, wz, cx, cy, cz);
        valid = project_camera<kVariant>(v, cx, cy, cz, tol, ox, oy);
    }
    return valid;
}

// rows of rotationT from a wxyz quaternion (slang/common/transforms.slang:22-39)
__device__ __forceinline__ void quat_rows(float w, float x, float y, float z, float r[3][3]) {
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float rx = w * x, ry = w * y, rz = w * z;
    r[0][0] = 1.0f - 2.0f * (yy + zz); r[0][1] = 2.0f * (xy + rz); r[0][2] = 2.0f * (xz - ry);
    r[1][0] = 2.0f * (xy - rz); r[1][1] = 1.0f - 2.0f * (xx + zz); r[1][2] = 2.0f * (yz + rx);
    r[2][0] = 2.0f * (xz + ry); r[2][1] = 2.0f * (yz - rx); r[2][2] = 1.0f - 2.0f * (xx + yy);
}

// real SH basis up to degree 3 (gaussianParticles.cuh:57-96)
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float Y[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) Y[i] = 0.0f;
    Y[0] = 0.28209479177387814f;
    if (deg > 0) {
        Y[1] = -0.4886025119029199f * y;
        Y[2] = 0.4886025119029199f * z;
        Y[3] = -0.4886025119029199f * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = 1.0925484305920792f * xy;
            Y[5] = -1.0925484305920792f * yz;
            Y[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
            Y[7] = -1.0925484305920792f * xz;
            Y[8] = 0.5462742152960396f * (xx - yy);
            if (deg > 2) {
                Y[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
                Y[10] = 2.890611442640554f * xy * z;
                Y[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
                Y[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                Y[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
                Y[14] = 1.445305721320277f * z * (xx - yy);
                Y[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
            }
        }
    }
}

__device__ __forceinline__ int clamp_tile(float v, int grid) {
    if (!(v > 0.0f)) return 0;
    if (v >= (float)grid) return grid;
    return (int)v;
}

struct TileBox {
    int x0, y0, x1, y1;
};

__device__ __forceinline__ TileBox tile_bbox(int gx, int gy, float px, float py, float ex, float ey) {
    TileBox b;
    b.x0 = clamp_tile(floorf((px - 0.5f - ex) / 16.0f), gx);
    b.y0 = clamp_tile(floorf((py - 0.5f - ey) / 16.0f), gy);
    b.x1 = clamp_tile(ceilf((px - 0.5f + ex) / 16.0f), gx);
    b.y1 = clamp_tile(ceilf((py - 0.5f + ey) / 16.0f), gy);
    return b;
}

__device__ __forceinline__ float saturate(float v) { return v > 0.0f ? (v < 1.0f ? v : 1.0f) : 0.0f; }

// smallest conic power 0.5 x^T C x over the tile rectangle (gutProjector.cuh:49-78)
__device__ float tile_min_power(float tx, float ty, float c0, float c1, float c2, float mx, float my) {
    const float ts = 16.0f;
    const float tminx = ts * tx, tminy = ts * ty;
    const float tmaxx = ts + tminx, tmaxy = ts + tminy;
    const float offx = tminx - mx, offy = tminy - my;
    const float lax = offx > 0.0f ? 1.0f : 0.0f, lay = offy > 0.0f ? 1.0f : 0.0f;
    const float nrx = lax + (mx > tmaxx ? 1.0f : 0.0f);
    const float nry = lay + (my > tmaxy ? 1.0f : 0.0f);
    if ((nrx + nry) > 0.0f) {
        const float px = lax > 0.0f ? tminx : tmaxx;
        const float py = lay > 0.0f ? tminy : tmaxy;
        const float dxx = copysignf(ts, offx), dxy = copysignf(ts, offy);
        const float dfx = mx - px, dfy = my - py;
        const float rcx = 1.0f / (ts * ts * c0);
        const float rcy = 1.0f / (ts * ts * c2);
        const float tx_ = nry * saturate((dxx * c0 * dfx + dxx * c1 * dfy) * rcx);
        const float ty_ = nrx * saturate((dxy * c1 * dfx + dxy * c2 * dfy) * rcy);
        const float qx = mx - (px + tx_ * dxx);
        const float qy = my - (py + ty_ * dxy);
        return 0.5f * (c0 * qx * qx + c2 * qy * qy) + c1 * qx * qy;
    }
    return 0.0f;
}

// ---------------------------------------------------------------------------------------------------
// K1
//
// MI355X shape: one lane per Gaussian, 4 waves per workgroup.  Two things keep this kernel on the HBM
// roofline instead of the L2 one:
//   * tile counting: footprints of up to kSerialTiles tiles are walked by their own lane; larger ones are
//     handed to the whole wave one at a time (64 tiles tested per step, counted with ballot+popcount), so a
//     screen-filling splat costs area/64 steps instead of stalling 63 idle lanes for `area` steps;
//   * SH coefficients: the 64 Gaussians of a wave own one contiguous 12 KiB block of the [N,48] tensor, which
//     the wave streams into LDS with fully coalesced 16-byte loads (row stride 49 dwords -> the later
//     row-per-lane reads are bank-conflict-free) instead of 12 strided float4 loads per lane that touch 64
//     cache lines each.
// Neither changes any arithmetic: the per-tile test and the SH dot products are evaluated exactly as before.
// ---------------------------------------------------------------------------------------------------
constexpr int kSerialTiles = 12;
constexpr int kShRow = 49;  // padded LDS row stride (dwords)

__device__ __forceinline__ float bcast(float v, int src_lane) { return __shfl(v, src_lane); }
__device__ __forceinline__ int bcast(int v, int src_lane) { return __shfl(v, src_lane); }

// (launch bound of 5 waves per SIMD: the compiler's own choice is 124 VGPRs = 4 waves; at 92 VGPRs nothing spills and the
//  kernel is 12 % faster, at 6 waves it spills and is 18 % slower)
// kSplitSH: the SH coefficients arrive as the model's two tensors (features_albedo [N,3] = the degree-0 triple, features_specular
// [N,45]; threedgrut/model/model.py:68-75) instead of their [N,48] concatenation: `sph48` is then the [N,45] tensor.
template <int kVariant, bool kRolling, bool kSplitSH>
__global__ __launch_bounds__(kBlock, 5) void k_project_on_tiles(ViewParams v, RenderConsts c, uint32_t n, int sh_degree,
                                                            const float4* __restrict__ density12,
                                                            const float* __restrict__ sph48,
                                                            const float* __restrict__ sph_albedo,
                                                            uint32_t* __restrict__ tiles_count, float2* __restrict__ proj_pos,
                                                            float4* __restrict__ conic_opacity, float2* __restrict__ extent,
                                                            float* __restrict__ depth, float* __restrict__ feat,
                                                            float* __restrict__ visibility, uint32_t* __restrict__ wave_sums,
                                                            FrameClears clr) {
    __shared__ float sh_lds[(kBlock / 64) * 32 * kShRow];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    const int wave = (int)(threadIdx.x >> 6);
    // the frame's small clears (three fill launches until round 4): per-tile ranges and backward traversal depths, one byte per wave
    for (uint32_t t = i; t < clr.tiles; t += gridDim.x * kBlock) {
        clr.ranges[t] = make_uint2(0u, 0u);
        clr.trav_bwd[t] = 0u;
    }
    if (lane == 0 && clr.wave_walked) clr.wave_walked[blockIdx.x * (kBlock / 64) + (uint32_t)wave] = 0;
    // ... and the sensor position the view-dependent colours below are evaluated from, for the fused optimiser's SH gradient
    // (gut_optimize_after_bwd with a null camera position): the same three floats in both passes, no host-to-device copy in the step
    if (i == 0 && clr.cam_pos) {
        clr.cam_pos[0] = v.s2w.t[0]; clr.cam_pos[1] = v.s2w.t[1]; clr.cam_pos[2] = v.s2w.t[2];
    }
    uint32_t cnt = 0;
    bool conic_ok = false, ok = false;
    float cx = 0.f, cy = 0.f, con0 = 0.f, con1 = 0.f, con2 = 0.f, con3 = 0.f, ex = 0.f, ey = 0.f, zkey = 0.f;
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, max_power = 0.f, zcam = 0.f;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t wave_first = blockIdx.x * kBlock + (uint32_t)wave * 64u;  // first Gaussian of this wave
    const uint32_t rows_here = wave_first < n ? min(64u, n - wave_first) : 0u;
    if (i < n) {
        a = density12[3 * (size_t)i + 0];                // pos.xyz, density
        const float4 b = density12[3 * (size_t)i + 1];   // quat wxyz
        const float4 d = density12[3 * (size_t)i + 2];   // scale.xyz, pad
        const float opacity_in = a.w;
        const Affine& m = v.w2s_mid;
        zcam = a.x * m.r[2][0] + a.y * m.r[2][1] + a.z * m.r[2][2] + m.t[2];
        ok = !(opacity_in < c.alpha_threshold) && !(zcam < c.min_sensor_z);
        if (ok) {
            // unscented transform: 7 sigma points through the full camera model (gutProjector.cuh:118-215)
            float rows[3][3];
            quat_rows(b.x, b.y, b.z, b.w, rows);
            const float scl[3] = {d.x, d.y, d.z};
            float sx[7], sy[7];
            int nvalid = project_world<kVariant, kRolling>(v, a.x, a.y, a.z, c.ut_margin, sx[0], sy[0]);
            cx = sx[0] * c.ut_w0_mean;
            cy = sy[0] * c.ut_w0_mean;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float kk = c.ut_delta * scl[k];
                const float dx = kk * rows[k][0], dy = kk * rows[k][1], dz = kk * rows[k][2];
                nvalid += project_world<kVariant, kRolling>(v, a.x + dx, a.y + dy, a.z + dz, c.ut_margin, sx[k + 1], sy[k + 1]);
                cx += c.ut_wi * sx[k + 1];
                cy += c.ut_wi * sy[k + 1];
                nvalid += project_world<kVariant, kRolling>(v, a.x - dx, a.y - dy, a.z - dz, c.ut_margin, sx[k + 4], sy[k + 4]);
                cx += c.ut_wi * sx[k + 4];
                cy += c.ut_wi * sy[k + 4];
            }
            ok = nvalid != 0;
            if (ok) {
                float cov0, cov1, cov2;
                {
                    const float e0 = sx[0] - cx, e1 = sy[0] - cy;
                    cov0 = c.ut_w0_cov * (e0 * e0);
                    cov1 = c.ut_w0_cov * (e0 * e1);
                    cov2 = c.ut_w0_cov * (e1 * e1);
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const float e0 = sx[k + 1] - cx, e1 = sy[k + 1] - cy;
                    cov0 += c.ut_wi * (e0 * e0);
                    cov1 += c.ut_wi * (e0 * e1);
                    cov2 += c.ut_wi * (e1 * e1);
                }
                // dilation, conic, mip-splatting opacity compensation, tight extent (gutProjector.cuh:81-116)
                const float dcx = cov0 + c.cov_dilation, dcy = cov1, dcz = cov2 + c.cov_dilation;
                const float ddet = dcx * dcz - dcy * dcy;
                ok = !(ddet == 0.0f);
                if (ok) {
                    const float inv_det = 1.0f / ddet;
                    con0 = dcz * inv_det;
                    con1 = -dcy * inv_det;
                    con2 = dcx * inv_det;
                    const float cdet = cov0 * cov2 - cov1 * cov1;
                    const float ratio = cdet * inv_det;
                    const float conv = sqrtf(ratio > 0.000025f ? ratio : 0.000025f);
                    con3 = opacity_in * conv;
                    ok = !(con3 < c.alpha_threshold);
                }
                if (ok) {
                    max_power = det_logf(con3 / c.alpha_threshold);
                    float extent_factor = 3.33f;
                    if (c.tight_opacity_bounding) {
                        const float e = sqrtf(2.0f * max_power);
                        extent_factor = e < 3.33f ? e : 3.33f;
                    }
                    const float mid = 0.5f * (dcx + dcz);
                    const float disc = mid * mid - ddet;
                    const float lam = mid + sqrtf(disc > 0.01f ? disc : 0.01f);
                    const float radius = extent_factor * sqrtf(lam);
                    if (c.rect_bounding) {
                        const float rx = extent_factor * sqrtf(dcx), ry = extent_factor * sqrtf(dcz);
                        ex = rx < radius ? rx : radius;
                        ey = ry < radius ? ry : radius;
                    } else {
                        ex = radius;
                        ey = radius;
                    }
                    ok = radius > 0.0f;
                    conic_ok = ok;
                }
            }
        }
    }

    // ---- tile counting ----
    TileBox bb = {0, 0, 0, 0};
    int area = 0;
    if (ok) {
        bb = tile_bbox(v.grid_x, v.grid_y, cx, cy, ex, ey);
        area = (bb.x1 - bb.x0) * (bb.y1 - bb.y0);
    }
    if (!c.tile_culling) {
        cnt = (uint32_t)area;
    } else {
        if (area > 0 && area <= kSerialTiles) {
            for (int y = bb.y0; y < bb.y1; ++y)
                for (int x = bb.x0; x < bb.x1; ++x)
                    if (tile_min_power((float)x, (float)y, con0, con1, con2, cx, cy) < max_power) cnt++;
        }
        unsigned long long todo = __ballot(area > kSerialTiles);
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int jx0 = bcast(bb.x0, j), jy0 = bcast(bb.y0, j), jw = bcast(bb.x1 - bb.x0, j), jarea = bcast(area, j);
            const float j0 = bcast(con0, j), j1 = bcast(con1, j), j2 = bcast(con2, j), jcx = bcast(cx, j), jcy = bcast(cy, j),
                        jmp = bcast(max_power, j);
            uint32_t total = 0;
            for (int base = 0; base < jarea; base += 64) {
                const int t = base + lane;
                bool pass = false;
                if (t < jarea) {
                    const int ty = t / jw;
                    const int tx = t - ty * jw;
                    pass = tile_min_power((float)(jx0 + tx), (float)(jy0 + ty), j0, j1, j2, jcx, jcy) < jmp;
                }
                total += (uint32_t)__popcll(__ballot(pass));
            }
            if (lane == j) cnt = total;
        }
    }

    // ---- view-dependent colour ----
    // The 64 Gaussians of a wave own one contiguous 12 KiB block of the [N,48] SH tensor.  It is read with fully
    // coalesced 16-byte loads and transposed through a wave-private LDS region (32 rows x 49 dwords, two passes;
    // no workgroup barrier: LDS operations of one wave execute in order) so that each lane ends up with its own row.
    // q / 12 never exceeds 63, so the mask shift is in range.
    const bool vis = cnt != 0;
    const unsigned long long vis_mask = __ballot(vis);
    if (vis_mask != 0ull) {
        float* wl = sh_lds + wave * 32 * kShRow;
        const float4* src = reinterpret_cast<const float4*>(sph48 + (size_t)wave_first * 48);
        float Y[16];
        float inv_dist = 0.f, dist = 0.f;
        if (vis) {
            const float rx = a.x - v.s2w.t[0], ry = a.y - v.s2w.t[1], rz = a.z - v.s2w.t[2];
            dist = sqrtf(rx * rx + ry * ry + rz * rz);
            inv_dist = 1.0f / dist;
            sh_basis(sh_degree, rx * inv_dist, ry * inv_dist, rz * inv_dist, Y);
        }
        const int ncoef = (sh_degree + 1) * (sh_degree + 1);
        float4 shq[12];
        float alb0 = 0.f, alb1 = 0.f, alb2 = 0.f;
        if (kSplitSH) {
            // the wave's block of the [N,45] tensor: 64 x 180 B = 720 float4, contiguous and 16-byte aligned (wave_first is a
            // multiple of 64); a half (32 rows) is exactly 360 of them.  A float4 may straddle two rows: fetched if either is visible.
            const float* spec = sph48 + (size_t)wave_first * 45;
            const uint32_t lim = rows_here * 45u;  // floats of the block that exist (the tensor may end inside the last float4)
            if (vis) {
                alb0 = sph_albedo[3 * (size_t)i + 0];
                alb1 = sph_albedo[3 * (size_t)i + 1];
                alb2 = sph_albedo[3 * (size_t)i + 2];
            }
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const uint32_t ql = (uint32_t)(it % 6) * 64u + (uint32_t)lane;
                const uint32_t f = 4u * ((uint32_t)(it / 6) * 360u + ql);
                const uint32_t r0 = min(f / 45u, 63u), r1 = min((f + 3u) / 45u, 63u);
                const bool wanted = (ql < 360u) && (f < lim) && (sh_degree > 0) && (((vis_mask >> r0) | (vis_mask >> r1)) & 1ull);
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (wanted) {
                    if (f + 3u < lim) {
                        val = *reinterpret_cast<const float4*>(spec + f);
                    } else {
                        val.x = spec[f];
                        if (f + 1u < lim) val.y = spec[f + 1u];
                        if (f + 2u < lim) val.z = spec[f + 2u];
                    }
                }
                shq[it] = val;
            }
        } else {
#pragma unroll
            for (int it = 0; it < 12; ++it) {
                const uint32_t q = (uint32_t)it * 64u + (uint32_t)lane;  // float4 index inside the 12 KiB block
                // rows are 192 B = three whole 64-byte lines: rows of culled Gaussians are not fetched at all
                const bool wanted = (q < rows_here * 12u) && ((vis_mask >> (q / 12u)) & 1ull);
                shq[it] = wanted ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (kSplitSH) {
                // row r of the half -> LDS row r: [albedo(3) | specular(45)], the same 48 columns the concatenated tensor has
#pragma unroll
                for (int it6 = 0; it6 < 6; ++it6) {
                    const uint32_t ql = (uint32_t)it6 * 64u + (uint32_t)lane;
                    if (ql < 360u) {
                        const uint32_t f = 4u * ql;
                        const uint32_t row = f / 45u, col = f - row * 45u;
                        const float4 val = shq[half * 6 + it6];
                        const float vals[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                        for (uint32_t j = 0; j < 4u; ++j) {
                            const uint32_t cj = col + j;
                            const uint32_t rj = cj >= 45u ? row + 1u : row;
                            if (rj < 32u) wl[rj * kShRow + 3u + (cj >= 45u ? cj - 45u : cj)] = vals[j];
                        }
                    }
                }
                if ((lane >> 5) == half) {
                    float* own = wl + (lane & 31) * kShRow;
                    own[0] = alb0; own[1] = alb1; own[2] = alb2;
                }
            } else {
#pragma unroll
                for (int it6 = 0; it6 < 6; ++it6) {
                    const uint32_t q = (uint32_t)it6 * 64u + (uint32_t)lane;  // float4 index inside this half (32 rows)
                    const uint32_t row = q / 12u, col = (q - row * 12u) * 4u;
                    float* dst = wl + row * kShRow + col;
                    const float4 val = shq[half * 6 + it6];
                    dst[0] = val.x; dst[1] = val.y; dst[2] = val.z; dst[3] = val.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (vis && (lane >> 5) == half) {
                const float* sh = wl + (lane & 31) * kShRow;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < ncoef) {
                        f0 += Y[k] * sh[3 * k + 0];
                        f1 += Y[k] * sh[3 * k + 1];
                        f2 += Y[k] * sh[3 * k + 2];
                    }
                f0 += 0.5f; f1 += 0.5f; f2 += 0.5f;
                zkey = c.global_z_order ? zcam : dist;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }

    if (i < n) {
        if (!vis) cx = cy = con0 = con1 = con2 = con3 = ex = ey = 0.0f;
        tiles_count[i] = cnt;
        proj_pos[i] = make_float2(cx, cy);
        conic_opacity[i] = make_float4(con0, con1, con2, con3);
        extent[i] = make_float2(ex, ey);
        depth[i] = zkey;
        feat[3 * (size_t)i + 0] = f0;
        feat[3 * (size_t)i + 1] = f1;
        feat[3 * (size_t)i + 2] = f2;
        // Deviation from the reference (documented): 1.0f/0.0f instead of int bit
        // patterns in a float tensor, and validProjection && validConic instead of validConic alone.
        visibility[i] = conic_ok ? 1.0f : 0.0f;
    }
    // NB: no global "visible" counter here — 94 k same-address atomics cost ~1 ms at N = 6 M (they serialise at the
    // memory side).  V is counted on demand from tiles_count (k_stats_reduce, gut_get_stats).
    // K2, first level: the tile count of this wave's 64 Gaussians (every wave of the grid writes one, zero beyond n) — a block's four sums
    // are one uint4.  k_scan_wave_sums turns them into per-block offsets, k_expand_tiles finishes the scan inside each wave: the
    // reference's (and rounds 1-3's) device-wide inclusive scan of the [N] counts (gutRenderer.cu:300-311: cub) is gone from the frame.
    {
        uint32_t wsum = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wsum += (uint32_t)__shfl_xor((int)wsum, o);
        if (lane == 0) wave_sums[blockIdx.x * (kBlock / 64) + (uint32_t)wave] = wsum;
    }
}

// K2, second level: exclusive scan of the per-256-row-block sums (one uint4 of wave sums per block) by ONE workgroup — 23 k values at
// 6 M Gaussians, three iterations of 8192 through LDS — and the frame's intersection count.  block_prefix[b] = list entries of all Gaussians before
// block b; *total = M.
// host_out (pinned host memory, may be null): [0] = M, [2], [3] = walk_sums of the last frame that had a backward — what the host reads
// once it has queued the rest of the frame; written from here instead of by two stream-ordered 4- and 8-byte copies (~12 us each).
__global__ __launch_bounds__(1024) void k_scan_wave_sums(const uint4* __restrict__ wave_sums4, uint32_t nblocks,
                                                         uint32_t* __restrict__ block_prefix, uint32_t* __restrict__ total,
                                                         uint32_t* __restrict__ host_out, const uint32_t* __restrict__ walk_sums) {
    constexpr uint32_t kPer = 8;                 // block sums per thread and iteration
    constexpr uint32_t kChunk = 1024u * kPer;    // 8192 blocks = 2.1 M Gaussians per iteration
    __shared__ uint32_t s_val[kChunk + kChunk / 32];   // (+1 word per 32: a thread's eight consecutive words spread over the banks)
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    auto slot = [](uint32_t e) { return e + (e >> 5); };
    for (uint32_t base = 0; base < nblocks; base += kChunk) {
        // coalesced 16-byte loads (consecutive lanes, consecutive blocks): the first version gave every lane 24 consecutive uint4 —
        // 64 different cache lines per load instruction, 25 k line requests through one CU's L1, and took 28 us
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t e = k * 1024u + tid, b = base + e;
            uint32_t v = 0;
            if (b < nblocks) {
                const uint4 w = wave_sums4[b];
                v = w.x + w.y + w.z + w.w;
            }
            s_val[slot(e)] = v;
        }
        __syncthreads();
        uint32_t local[kPer];
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {   // this thread's eight consecutive blocks
            local[k] = sum;
            sum += s_val[slot(tid * kPer + k)];
        }
        uint32_t incl = sum;   // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if ((int)lane >= o) incl += up;
        }
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t wave_excl = 0, iter_total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16; ++w) {
            const uint32_t t = s_wave[w];
            if (w < wave) wave_excl += t;
            iter_total += t;
        }
        const uint32_t excl = s_carry + wave_excl + (incl - sum);
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) s_val[slot(tid * kPer + k)] = excl + local[k];
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {   // coalesced stores
            const uint32_t e = k * 1024u + tid, b = base + e;
            if (b < nblocks) block_prefix[b] = s_val[slot(e)];
        }
        if (tid == 0) s_carry += iter_total;
        __syncthreads();
    }
    if (tid == 0) {
        *total = s_carry;
        if (host_out) {
            // system-scope stores into coherent host memory: they bypass the caches, and the end of the kernel orders them before the
            // event the host waits on.  (NOT __threadfence_system(): on gfx950 that writes back the whole L2 — dirty from K1 — and made
            // this kernel 28 us long.)
            __hip_atomic_store(&host_out[0], s_carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (walk_sums) {
                __hip_atomic_store(&host_out[2], walk_sums[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&host_out[3], walk_sums[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K3: emit (tile|depth, id) pairs at tilesOffset[i-1]...  Same serial/cooperative split as K1; the cooperative
// path writes in the same row-major tile order (ballot prefix), so the unsorted buffers are identical.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_expand_tiles(ViewParams v, RenderConsts c, uint32_t n,
                                                        const uint32_t* __restrict__ tiles_count,
                                                        const uint4* __restrict__ wave_sums4,
                                                        const uint32_t* __restrict__ block_prefix,
                                                        const uint32_t* __restrict__ total,
                                                        const float2* __restrict__ proj_pos,
                                                        const float4* __restrict__ conic_opacity,
                                                        const float2* __restrict__ extent, const float* __restrict__ depth,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ ids, uint32_t capacity) {
    // capacity: entries the key / id buffers hold.  The forward sizes them from the previous frames' intersection counts
    // without waiting for this frame's (gut_api.cpp); should this frame need more, the writes beyond the buffers are dropped
    // here and the host, which reads the count once everything is queued, grows the buffers and redoes the binning.
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63);
    {
        // Padding behind the last real entry (a separate launch until round 4): the sort runs over `capacity` >= M slots (sized on the
        // host before M is known), the tail [M, capacity) carries the reference's padding pair (gutProjector.cuh:372-376), which sorts
        // behind every tile.  Every workgroup fills one slice of the tail.
        const uint32_t m = min(*total, capacity);
        const uint32_t per = (capacity - m + gridDim.x - 1) / gridDim.x;
        const uint32_t beg = m + blockIdx.x * per, end = min(beg + per, capacity);
        for (uint32_t k = beg + threadIdx.x; k < end; k += kBlock) {
            keys[k] = ((uint64_t)kInvalid << 32) | f2u(3.4028235e+38f);
            ids[k] = kInvalid;
        }
    }
    bool active = false;
    float2 e = make_float2(0.f, 0.f), p = make_float2(0.f, 0.f);
    float4 con = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t dkey = 0, off = 0, max_off = 0;
    float max_power = 0.f;
    TileBox bb = {0, 0, 0, 0};
    int area = 0;
    uint32_t my_cnt = 0;
    if (i < n) {
        e = extent[i];
        active = !(e.x <= 1e-06f);
        my_cnt = tiles_count[i];
    }
    // K2, third level: this Gaussian's first list slot = entries of the blocks before this one (k_scan_wave_sums) + of the waves before
    // this one in the block (K1's wave sums) + of the lanes before this one (scan over the wave)
    uint32_t incl = my_cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= o) incl += up;
    }
    const uint4 ws = wave_sums4[blockIdx.x];
    const uint32_t wv = threadIdx.x >> 6;
    const uint32_t first = block_prefix[blockIdx.x] + (wv > 0u ? ws.x : 0u) + (wv > 1u ? ws.y : 0u) + (wv > 2u ? ws.z : 0u) + (incl - my_cnt);
    if (active) {
        dkey = f2u(depth[i]);
        max_off = min(first + my_cnt, capacity);
        off = min(first, max_off);
        p = proj_pos[i];
        bb = tile_bbox(v.grid_x, v.grid_y, p.x, p.y, e.x, e.y);
        area = (bb.x1 - bb.x0) * (bb.y1 - bb.y0);
        con = conic_opacity[i];
        if (c.tile_culling) max_power = det_logf(con.w / c.alpha_threshold);
    }
    if (!c.tile_culling) {
        if (active)
            for (int y = bb.y0; y < bb.y1; ++y)
                for (int x = bb.x0; x < bb.x1; ++x) {
                    if (off < max_off) {
                        keys[off] = ((uint64_t)(uint32_t)(y * v.grid_x + x) << 32) | dkey;
                        ids[off] = i;
                    }
                    off++;
                }
        return;
    }
    const bool serial = active && area <= kSerialTiles;
    if (serial) {
        for (int y = bb.y0; (y < bb.y1) && (off < max_off); ++y)
            for (int x = bb.x0; (x < bb.x1) && (off < max_off); ++x)
                if (tile_min_power((float)x, (float)y, con.x, con.y, con.z, p.x, p.y) < max_power) {
                    keys[off] = ((uint64_t)(uint32_t)(y * v.grid_x + x) << 32) | dkey;
                    ids[off] = i;
                    off++;
                }
    }
    unsigned long long todo = __ballot(active && area > kSerialTiles);
    while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int jx0 = bcast(bb.x0, j), jy0 = bcast(bb.y0, j), jw = bcast(bb.x1 - bb.x0, j), jarea = bcast(area, j);
        const float j0 = bcast(con.x, j), j1 = bcast(con.y, j), j2 = bcast(con.z, j), jcx = bcast(p.x, j), jcy = bcast(p.y, j),
                    jmp = bcast(max_power, j);
        const uint32_t jdkey = (uint32_t)bcast((int)dkey, j), jmax = (uint32_t)bcast((int)max_off, j);
        const uint32_t jid = blockIdx.x * kBlock + (threadIdx.x & ~63u) + (uint32_t)j;
        uint32_t joff = (uint32_t)bcast((int)off, j);
        for (int base = 0; base < jarea; base += 64) {
            const int t = base + lane;
            bool pass = false;
            int tx = 0, ty = 0;
            if (t < jarea) {
                ty = t / jw;
                tx = t - ty * jw;
                pass = tile_min_power((float)(jx0 + tx), (float)(jy0 + ty), j0, j1, j2, jcx, jcy) < jmp;
            }
            const unsigned long long mask = __ballot(pass);
            const uint32_t my = joff + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (pass && my < jmax) {
                keys[my] = ((uint64_t)(uint32_t)((jy0 + ty) * v.grid_x + (jx0 + tx)) << 32) | jdkey;
                ids[my] = jid;
            }
            joff += (uint32_t)__popcll(mask);
        }
        if (lane == j) off = joff < jmax ? joff : jmax;
    }
    if (active)
        for (; off < max_off; ++off) {  // pad (cannot happen while K1 and K3 evaluate the same test; kept for parity)
            keys[off] = ((uint64_t)kInvalid << 32) | f2u(3.4028235e+38f);
            ids[off] = kInvalid;
        }
}

// Padding behind the last real entry: the sort runs over `sort_n` >= M entries (sized on the host before M is known), the tail
// [M, sort_n) carries the reference's padding pair (gutProjector.cuh:372-376), which sorts behind every tile.
__global__ __launch_bounds__(kBlock) void k_pad_keys(const uint32_t* __restrict__ count, uint32_t sort_n, uint64_t* __restrict__ keys,
                                                    uint32_t* __restrict__ ids) {
    const uint32_t m = *count;
    for (uint32_t k = m + blockIdx.x * kBlock + threadIdx.x; k < sort_n; k += gridDim.x * kBlock) {
        keys[k] = ((uint64_t)kInvalid << 32) | f2u(3.4028235e+38f);
        ids[k] = kInvalid;
    }
}

__global__ __launch_bounds__(kBlock) void k_mask_unordered(uint32_t tiles, const uint2* __restrict__ ranges,
                                                          const uint32_t* __restrict__ tile_ordered, uint32_t* __restrict__ ordered_ids) {
    const uint32_t t = blockIdx.x;
    if (t >= tiles) return;
    const uint2 r = ranges[t];
    for (uint32_t k = r.x + min(r.y - r.x, tile_ordered[t]) + threadIdx.x; k < r.y; k += kBlock) ordered_ids[k] = kInvalid;
}

void launch_mask_unordered(hipStream_t s, uint32_t tiles, const uint32_t* ranges, const uint32_t* tile_ordered, uint32_t* ordered_ids) {
    if (tiles == 0) return;
    hipLaunchKernelGGL(k_mask_unordered, dim3(tiles), dim3(kBlock), 0, s, tiles, reinterpret_cast<const uint2*>(ranges), tile_ordered,
                       ordered_ids);
}

void launch_pad_keys(hipStream_t s, const uint32_t* count, uint32_t sort_n, uint64_t* keys, uint32_t* ids) {
    if (sort_n == 0) return;
    hipLaunchKernelGGL(k_pad_keys, dim3(64), dim3(kBlock), 0, s, count, sort_n, keys, ids);
}

// ---------------------------------------------------------------------------------------------------
// K5: ranges[tile] = (first, last+1) over the sorted keys; ranges pre-zeroed by the caller
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_tile_ranges(uint32_t m, const uint64_t* __restrict__ sorted_keys,
                                                       uint2* __restrict__ ranges) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= m) return;
    const uint32_t tile = (uint32_t)(sorted_keys[k] >> 32);
    const bool valid = tile != kInvalid;
    if (k == 0) {
        if (valid) ranges[tile].x = 0;
    } else {
        const uint32_t prev = (uint32_t)(sorted_keys[k - 1] >> 32);
        if (prev != tile) {
            if (prev != kInvalid) ranges[prev].y = k;
            if (valid) ranges[tile].x = k;
        }
    }
    if (valid && (k == m - 1)) ranges[tile].y = m;
}

// ---------------------------------------------------------------------------------------------------
// K8: per-Gaussian epilogue of the backward.  Reads the 64-byte gradient row K7 accumulated
// (pos3, density, quat4, scale3, dRGB3, pad2), writes the [N,12] density gradient and the [N,48] SH
// gradient in full (zeros for Gaussians that touched no tile), so the caller never zero-fills them.
// ---------------------------------------------------------------------------------------------------
// kRawGrads: additionally chain d(position, density, quaternion, scale) through the model's activations
// (sigmoid, normalise, exp — threedgrut/model/model.py:74-93) so that the rows can be fed to the optimiser as
// gradients of the RAW parameters; |quat| is read from the pad column written by k_activate_pack.
// kSplitSH (with `fields` only): the SH gradient goes out as the model's two tensors, fields.alb [N,3] and fields.spec [N,45]
// (model.py:68-75) — the wave's 64 x 45 block is transposed through LDS and written with coalesced 16-byte stores.
template <bool kRawGrads, bool kSplitSH>
__global__ __launch_bounds__(kBlock) void k_project_backward(ViewParams v, uint32_t n, int sh_degree,
                                                            const float4* __restrict__ density12,
                                                            const uint32_t* __restrict__ tiles_count,
                                                            const float* __restrict__ feat, float4* __restrict__ grad16,
                                                            float4* __restrict__ density_grad12,
                                                            float4* __restrict__ sph_grad48, GradFields fields) {
    __shared__ __attribute__((aligned(16))) float spec_lds[kSplitSH ? (kBlock / 64) * 64 * 45 : 4];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    if (!kSplitSH && !live) return;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
    float out[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) out[k] = 0.0f;
    if (live && tiles_count[i] != 0) {
        g0 = grad16[4 * (size_t)i + 0];
        g1 = grad16[4 * (size_t)i + 1];
        g2 = grad16[4 * (size_t)i + 2];
        const float4 g3 = grad16[4 * (size_t)i + 3];
        // the row is consumed: leave it zero for the next backward (no 64 N-byte clear per step)
        grad16[4 * (size_t)i + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        grad16[4 * (size_t)i + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        grad16[4 * (size_t)i + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        grad16[4 * (size_t)i + 3] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float dr = g2.w, dg = g3.x, db = g3.y;
        g2.w = 0.0f;
        const float4 a = density12[3 * (size_t)i];
        if (kRawGrads) {
            const float4 qn = density12[3 * (size_t)i + 1];
            const float4 sc = density12[3 * (size_t)i + 2];
            g0.w = g0.w * a.w * (1.0f - a.w);                                   // sigmoid'
            const float dot = g1.x * qn.x + g1.y * qn.y + g1.z * qn.z + g1.w * qn.w;
            const float inv = 1.0f / sc.w;                                      // 1/|quat_raw|
            g1 = make_float4((g1.x - qn.x * dot) * inv, (g1.y - qn.y * dot) * inv, (g1.z - qn.z * dot) * inv,
                             (g1.w - qn.w * dot) * inv);                        // normalise'
            g2.x *= sc.x; g2.y *= sc.y; g2.z *= sc.z;                           // exp'
        }
        const float rx = a.x - v.s2w.t[0], ry = a.y - v.s2w.t[1], rz = a.z - v.s2w.t[2];
        const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
        const float inv_dist = 1.0f / dist;
        float Y[16];
        sh_basis(sh_degree, rx * inv_dist, ry * inv_dist, rz * inv_dist, Y);
        const int ncoef = (sh_degree + 1) * (sh_degree + 1);
        const float m0 = feat[3 * (size_t)i + 0] > 0.0f ? dr : 0.0f;
        const float m1 = feat[3 * (size_t)i + 1] > 0.0f ? dg : 0.0f;
        const float m2 = feat[3 * (size_t)i + 2] > 0.0f ? db : 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < ncoef) {
                out[3 * k + 0] = Y[k] * m0;
                out[3 * k + 1] = Y[k] * m1;
                out[3 * k + 2] = Y[k] * m2;
            }
    }
    if (live) {
        if (fields.pos) {
            // the four gradient tensors the reference's _Autograd.backward hands back (tracer.py:268-286), written directly
            // instead of one [N,12] tensor that torch then splits and copies four times
            fields.pos[3 * (size_t)i + 0] = g0.x; fields.pos[3 * (size_t)i + 1] = g0.y; fields.pos[3 * (size_t)i + 2] = g0.z;
            fields.dns[i] = g0.w;
            reinterpret_cast<float4*>(fields.rot)[i] = g1;
            fields.scl[3 * (size_t)i + 0] = g2.x; fields.scl[3 * (size_t)i + 1] = g2.y; fields.scl[3 * (size_t)i + 2] = g2.z;
        } else {
            density_grad12[3 * (size_t)i + 0] = g0;
            density_grad12[3 * (size_t)i + 1] = g1;
            density_grad12[3 * (size_t)i + 2] = g2;
        }
    }
    if (kSplitSH) {
        const int lane = (int)(threadIdx.x & 63);
        const uint32_t wave_first = i - (uint32_t)lane;
        const uint32_t rows_here = wave_first < n ? min(64u, n - wave_first) : 0u;
        float* wl = spec_lds + (threadIdx.x >> 6) * (64 * 45);
        if (live) {
            fields.alb[3 * (size_t)i + 0] = out[0]; fields.alb[3 * (size_t)i + 1] = out[1]; fields.alb[3 * (size_t)i + 2] = out[2];
        }
#pragma unroll
        for (int k = 0; k < 45; ++k) wl[lane * 45 + k] = out[3 + k];   // (stride 45 dwords: conflict-free)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float* dst = fields.spec + (size_t)wave_first * 45;             // 16-byte aligned: wave_first is a multiple of 64
        const uint32_t lim = rows_here * 45u;
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const uint32_t f = 4u * ((uint32_t)it * 64u + (uint32_t)lane);
            if (f < 2880u && f < lim) {
                const float4 val = *reinterpret_cast<const float4*>(wl + f);
                if (f + 3u < lim) {
                    *reinterpret_cast<float4*>(dst + f) = val;
                } else {                                                 // the tensor ends inside this float4
                    dst[f] = val.x;
                    if (f + 1u < lim) dst[f + 1u] = val.y;
                    if (f + 2u < lim) dst[f + 2u] = val.z;
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 12; ++k)
            sph_grad48[12 * (size_t)i + k] = make_float4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
    }
}

// k_pack_fields: (positions [N,3], density [N,1], rotation [N,4], scale [N,3]) -> the [N,12] rows the kernels read
// ([pos | density | quat wxyz | scale | 0], tracer.py:176-178) — the reference's torch.cat, as one coalesced pass
__global__ __launch_bounds__(kBlock) void k_pack_fields(uint32_t n, const float* __restrict__ pos, const float* __restrict__ dns,
                                                       const float4* __restrict__ rot, const float* __restrict__ scl,
                                                       float4* __restrict__ density12) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    density12[3 * (size_t)i + 0] = make_float4(pos[3 * (size_t)i], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2], dns[i]);
    density12[3 * (size_t)i + 1] = rot[i];
    density12[3 * (size_t)i + 2] = make_float4(scl[3 * (size_t)i], scl[3 * (size_t)i + 1], scl[3 * (size_t)i + 2], 0.0f);
}

// K8c: compact per-Gaussian epilogue for the fused optimiser / compact data-parallel exchange.  Writes the [N,12]
// gradient chained to the RAW parameters and, instead of the [N,48] SH gradient, only its generator:
// mrgb = dL/dRGB masked by (precomputed RGB > 0).  The SH gradient of a view is the rank-one product
// Y_k(dir_view) * mrgb, which k_sh_adam rebuilds on the fly (per view) — 12 bytes per Gaussian and view cross the
// wire instead of 192.
__global__ __launch_bounds__(kBlock) void k_project_backward_compact(uint32_t n, const float4* __restrict__ density12,
                                                                    const uint32_t* __restrict__ tiles_count,
                                                                    const float* __restrict__ feat,
                                                                    float4* __restrict__ grad16,
                                                                    float4* __restrict__ raw_grad12, float* __restrict__ mrgb) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if (tiles_count[i] != 0) {
        g0 = grad16[4 * (size_t)i + 0];
        g1 = grad16[4 * (size_t)i + 1];
        g2 = grad16[4 * (size_t)i + 2];
        const float4 g3 = grad16[4 * (size_t)i + 3];
        // the row is consumed: leave it zero for the next backward (no 64 N-byte clear per step)
        grad16[4 * (size_t)i + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
        grad16[4 * (size_t)i + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        grad16[4 * (size_t)i + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        grad16[4 * (size_t)i + 3] = make_float4(0.f, 0.f, 0.f, 0.f);
        m0 = feat[3 * (size_t)i + 0] > 0.0f ? g2.w : 0.0f;
        m1 = feat[3 * (size_t)i + 1] > 0.0f ? g3.x : 0.0f;
        m2 = feat[3 * (size_t)i + 2] > 0.0f ? g3.y : 0.0f;
        g2.w = 0.0f;
        const float4 a = density12[3 * (size_t)i];
        const float4 qn = density12[3 * (size_t)i + 1];
        const float4 sc = density12[3 * (size_t)i + 2];
        g0.w = g0.w * a.w * (1.0f - a.w);
        const float dot = g1.x * qn.x + g1.y * qn.y + g1.z * qn.z + g1.w * qn.w;
        const float inv = 1.0f / sc.w;
        g1 = make_float4((g1.x - qn.x * dot) * inv, (g1.y - qn.y * dot) * inv, (g1.z - qn.z * dot) * inv, (g1.w - qn.w * dot) * inv);
        g2.x *= sc.x; g2.y *= sc.y; g2.z *= sc.z;
    }
    raw_grad12[3 * (size_t)i + 0] = g0;
    raw_grad12[3 * (size_t)i + 1] = g1;
    raw_grad12[3 * (size_t)i + 2] = g2;
    mrgb[3 * (size_t)i + 0] = m0;
    mrgb[3 * (size_t)i + 1] = m1;
    mrgb[3 * (size_t)i + 2] = m2;
}

void launch_project_bwd_compact(hipStream_t s, uint32_t n, const float* density12, const uint32_t* tiles_count, const float* feat,
                                float* grad16, float* raw_grad12, float* mrgb) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_project_backward_compact, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n,
                       reinterpret_cast<const float4*>(density12), tiles_count, feat, reinterpret_cast<float4*>(grad16),
                       reinterpret_cast<float4*>(raw_grad12), mrgb);
}

// ---------------------------------------------------------------------------------------------------
// statistics on demand (gut_get_stats): V = #{tiles_count > 0}, E_f / E_b = sums of the per-tile traversal depths
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_stats_reduce(uint32_t n, const uint32_t* __restrict__ tiles_count, uint32_t t,
                                                        const uint32_t* __restrict__ trav_fwd,
                                                        const uint32_t* __restrict__ trav_bwd, Counters* __restrict__ out) {
    unsigned long long v = 0, ef = 0, eb = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) v += tiles_count[i] != 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < t; i += gridDim.x * kBlock) {
        ef += trav_fwd[i];
        eb += trav_bwd[i];
    }
    for (int m = 32; m >= 1; m >>= 1) {
        v += __shfl_xor(v, m);
        ef += __shfl_xor(ef, m);
        eb += __shfl_xor(eb, m);
    }
    if ((threadIdx.x & 63) == 0) {
        if (v) atomicAdd(&out->visible, v);
        if (ef) atomicAdd(&out->traversed_fwd, ef);
        if (eb) atomicAdd(&out->traversed_bwd, eb);
    }
}

void launch_stats_reduce(hipStream_t s, uint32_t n, const uint32_t* tiles_count, uint32_t t, const uint32_t* trav_fwd,
                         const uint32_t* trav_bwd, Counters* out) {
    hipLaunchKernelGGL(k_stats_reduce, dim3(256), dim3(kBlock), 0, s, n, tiles_count, t, trav_fwd, trav_bwd, out);
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

void launch_project(hipStream_t s, const ViewParams& v, const RenderConsts& c, uint32_t n, int sh_degree,
                    const float* density12, const float* sph48, uint32_t* tiles_count, float* proj_pos,
                    float* conic_opacity, float* extent, float* depth, float* feat, float* visibility,
                    uint32_t* wave_sums, const float* sph_albedo, const FrameClears& clears) {
    if (n == 0) return;
    bool distorted = false;
    for (float k : v.radial) distorted |= (k != 0.0f);
    for (float k : v.tangential) distorted |= (k != 0.0f);
    for (float k : v.thin_prism) distorted |= (k != 0.0f);
    const int variant = v.model == GUT_CAMERA_OPENCV_FISHEYE ? 2 : (distorted ? 1 : 0);
    const bool rolling = v.shutter != GUT_SHUTTER_GLOBAL;
    auto kern = rolling ? (variant == 0 ? k_project_on_tiles<0, true, false> : (variant == 1 ? k_project_on_tiles<1, true, false> : k_project_on_tiles<2, true, false>))
                        : (variant == 0 ? k_project_on_tiles<0, false, false> : (variant == 1 ? k_project_on_tiles<1, false, false> : k_project_on_tiles<2, false, false>));
    if (sph_albedo)   // sph48 is the [N,45] specular tensor
        kern = rolling ? (variant == 0 ? k_project_on_tiles<0, true, true> : (variant == 1 ? k_project_on_tiles<1, true, true> : k_project_on_tiles<2, true, true>))
                       : (variant == 0 ? k_project_on_tiles<0, false, true> : (variant == 1 ? k_project_on_tiles<1, false, true> : k_project_on_tiles<2, false, true>));
    hipLaunchKernelGGL(kern, dim3(blocks_for(n)), dim3(kBlock), 0, s, v, c, n, sh_degree,
                       reinterpret_cast<const float4*>(density12), sph48, sph_albedo, tiles_count, reinterpret_cast<float2*>(proj_pos),
                       reinterpret_cast<float4*>(conic_opacity), reinterpret_cast<float2*>(extent), depth, feat, visibility,
                       wave_sums, clears);
}

void launch_scan_wave_sums(hipStream_t s, uint32_t n, const uint32_t* wave_sums, uint32_t* block_prefix, uint32_t* total,
                           uint32_t* host_out, const uint32_t* walk_sums) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_scan_wave_sums, dim3(1), dim3(1024), 0, s, reinterpret_cast<const uint4*>(wave_sums), blocks_for(n), block_prefix,
                       total, host_out, walk_sums);
}

void launch_expand(hipStream_t s, const ViewParams& v, const RenderConsts& c, uint32_t n, const uint32_t* tiles_count,
                   const uint32_t* wave_sums, const uint32_t* block_prefix, const uint32_t* total,
                   const float* proj_pos, const float* conic_opacity, const float* extent, const float* depth,
                   uint64_t* keys, uint32_t* ids, uint32_t capacity) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_expand_tiles, dim3(blocks_for(n)), dim3(kBlock), 0, s, v, c, n, tiles_count,
                       reinterpret_cast<const uint4*>(wave_sums), block_prefix, total,
                       reinterpret_cast<const float2*>(proj_pos), reinterpret_cast<const float4*>(conic_opacity),
                       reinterpret_cast<const float2*>(extent), depth, keys, ids, capacity);
}

void launch_tile_ranges(hipStream_t s, uint32_t m, const uint64_t* sorted_keys, uint32_t* ranges) {
    if (m == 0) return;
    hipLaunchKernelGGL(k_tile_ranges, dim3(blocks_for(m)), dim3(kBlock), 0, s, m, sorted_keys,
                       reinterpret_cast<uint2*>(ranges));
}

void launch_project_bwd(hipStream_t s, const ViewParams& v, uint32_t n, int sh_degree, const float* density12,
                        const uint32_t* tiles_count, const float* feat, float* grad16, float* density_grad12,
                        float* sph_grad48, bool raw_grads, const GradFields& fields) {
    if (n == 0) return;
    auto kern = raw_grads ? k_project_backward<true, false> : k_project_backward<false, false>;
    if (fields.spec) kern = raw_grads ? k_project_backward<true, true> : k_project_backward<false, true>;   // gut_trace_bwd_model_fields
    hipLaunchKernelGGL(kern, dim3(blocks_for(n)), dim3(kBlock), 0, s, v, n, sh_degree,
                       reinterpret_cast<const float4*>(density12), tiles_count, feat,
                       reinterpret_cast<float4*>(grad16), reinterpret_cast<float4*>(density_grad12),
                       reinterpret_cast<float4*>(sph_grad48), fields);
}

void launch_pack_fields(hipStream_t s, uint32_t n, const float* pos, const float* dns, const float* rot, const float* scl,
                        float* density12) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack_fields, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, pos, dns, reinterpret_cast<const float4*>(rot), scl,
                       reinterpret_cast<float4*>(density12));
}

}  // namespace gut
