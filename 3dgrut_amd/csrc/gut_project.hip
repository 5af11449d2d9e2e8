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
__device__ bool project_pinhole(const ViewParams& v, float px, float py, float pz, float tol, float& ox, float& oy) {
    if (pz <= 0.0f) {
        ox = 0.0f;
        oy = 0.0f;
        return false;
    }
    const float un = px / pz, vn = py / pz;
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
__device__ bool project_fisheye(const ViewParams& v, float px, float py, float pz, float tol, float& ox, float& oy) {
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

__device__ __forceinline__ int project_world(const ViewParams& v, float wx, float wy, float wz, float tol, float& ox,
                                             float& oy) {
    const Affine& a = v.w2s_start;
    const float cx = a.r[0][0] * wx + a.r[0][1] * wy + a.r[0][2] * wz + a.t[0];
    const float cy = a.r[1][0] * wx + a.r[1][1] * wy + a.r[1][2] * wz + a.t[1];
    const float cz = a.r[2][0] * wx + a.r[2][1] * wy + a.r[2][2] * wz + a.t[2];
    if (v.model == GUT_CAMERA_OPENCV_PINHOLE) return project_pinhole(v, cx, cy, cz, tol, ox, oy) ? 1 : 0;
    return project_fisheye(v, cx, cy, cz, tol, ox, oy) ? 1 : 0;
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
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_project_on_tiles(ViewParams v, RenderConsts c, uint32_t n, int sh_degree,
                                                            const float4* __restrict__ density12,
                                                            const float* __restrict__ sph48,
                                                            uint32_t* __restrict__ tiles_count, float2* __restrict__ proj_pos,
                                                            float4* __restrict__ conic_opacity, float2* __restrict__ extent,
                                                            float* __restrict__ depth, float* __restrict__ feat,
                                                            float* __restrict__ visibility, Counters* __restrict__ counters) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t cnt = 0;
    bool conic_ok = false;
    float cx = 0.f, cy = 0.f, con0 = 0.f, con1 = 0.f, con2 = 0.f, con3 = 0.f, ex = 0.f, ey = 0.f, zkey = 0.f;
    float f0 = 0.f, f1 = 0.f, f2 = 0.f;
    if (i < n) {
        const float4 a = density12[3 * (size_t)i + 0];  // pos.xyz, density
        const float4 b = density12[3 * (size_t)i + 1];  // quat wxyz
        const float4 d = density12[3 * (size_t)i + 2];  // scale.xyz, pad
        const float opacity_in = a.w;
        const Affine& m = v.w2s_mid;
        const float zcam = a.x * m.r[2][0] + a.y * m.r[2][1] + a.z * m.r[2][2] + m.t[2];
        bool ok = !(opacity_in < c.alpha_threshold) && !(zcam < c.min_sensor_z);
        float max_power = 0.f;
        if (ok) {
            // unscented transform: 7 sigma points through the full camera model (gutProjector.cuh:118-215)
            float rows[3][3];
            quat_rows(b.x, b.y, b.z, b.w, rows);
            const float scl[3] = {d.x, d.y, d.z};
            float sx[7], sy[7];
            int nvalid = project_world(v, a.x, a.y, a.z, c.ut_margin, sx[0], sy[0]);
            cx = sx[0] * c.ut_w0_mean;
            cy = sy[0] * c.ut_w0_mean;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float kk = c.ut_delta * scl[k];
                const float dx = kk * rows[k][0], dy = kk * rows[k][1], dz = kk * rows[k][2];
                nvalid += project_world(v, a.x + dx, a.y + dy, a.z + dz, c.ut_margin, sx[k + 1], sy[k + 1]);
                cx += c.ut_wi * sx[k + 1];
                cy += c.ut_wi * sy[k + 1];
                nvalid += project_world(v, a.x - dx, a.y - dy, a.z - dz, c.ut_margin, sx[k + 4], sy[k + 4]);
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
                    con0 = dcz / ddet;
                    con1 = -dcy / ddet;
                    con2 = dcx / ddet;
                    const float cdet = cov0 * cov2 - cov1 * cov1;
                    const float ratio = cdet / ddet;
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
        if (ok) {
            const TileBox bb = tile_bbox(v.grid_x, v.grid_y, cx, cy, ex, ey);
            if (c.tile_culling) {
                for (int y = bb.y0; y < bb.y1; ++y)
                    for (int x = bb.x0; x < bb.x1; ++x)
                        if (tile_min_power((float)x, (float)y, con0, con1, con2, cx, cy) < max_power) cnt++;
            } else {
                cnt = (uint32_t)((bb.x1 - bb.x0) * (bb.y1 - bb.y0));
            }
        }
        if (cnt != 0) {
            // view-dependent colour evaluated once per Gaussian along (mean - sensor position), unclamped, +0.5
            const float rx = a.x - v.s2w.t[0], ry = a.y - v.s2w.t[1], rz = a.z - v.s2w.t[2];
            const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
            float Y[16];
            sh_basis(sh_degree, rx / dist, ry / dist, rz / dist, Y);
            const int ncoef = (sh_degree + 1) * (sh_degree + 1);
            const float4* sh4 = reinterpret_cast<const float4*>(sph48 + (size_t)i * 48);
            float sh[48];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const float4 q = sh4[k];
                sh[4 * k + 0] = q.x; sh[4 * k + 1] = q.y; sh[4 * k + 2] = q.z; sh[4 * k + 3] = q.w;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < ncoef) {
                    f0 += Y[k] * sh[3 * k + 0];
                    f1 += Y[k] * sh[3 * k + 1];
                    f2 += Y[k] * sh[3 * k + 2];
                }
            f0 += 0.5f; f1 += 0.5f; f2 += 0.5f;
            zkey = c.global_z_order ? zcam : dist;
        } else {
            cx = cy = con0 = con1 = con2 = con3 = ex = ey = 0.0f;
        }
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
    const unsigned long long vis_mask = __ballot(cnt != 0);
    if ((threadIdx.x & 63) == 0 && vis_mask) atomicAdd(&counters->visible, (unsigned long long)__popcll(vis_mask));
}

// ---------------------------------------------------------------------------------------------------
// K3
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_expand_tiles(ViewParams v, RenderConsts c, uint32_t n,
                                                        const uint32_t* __restrict__ offset,
                                                        const float2* __restrict__ proj_pos,
                                                        const float4* __restrict__ conic_opacity,
                                                        const float2* __restrict__ extent, const float* __restrict__ depth,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float2 e = extent[i];
    if (e.x <= 1e-06f) return;
    const uint32_t dkey = f2u(depth[i]);
    uint32_t off = (i == 0) ? 0u : offset[i - 1];
    const uint32_t max_off = offset[i];
    const float2 p = proj_pos[i];
    const TileBox bb = tile_bbox(v.grid_x, v.grid_y, p.x, p.y, e.x, e.y);
    if (c.tile_culling) {
        const float4 con = conic_opacity[i];
        const float max_power = det_logf(con.w / c.alpha_threshold);
        for (int y = bb.y0; (y < bb.y1) && (off < max_off); ++y)
            for (int x = bb.x0; (x < bb.x1) && (off < max_off); ++x)
                if (tile_min_power((float)x, (float)y, con.x, con.y, con.z, p.x, p.y) < max_power) {
                    keys[off] = ((uint64_t)(uint32_t)(y * v.grid_x + x) << 32) | dkey;
                    ids[off] = i;
                    off++;
                }
        for (; off < max_off; ++off) {  // pad (cannot happen while K1 and K3 evaluate the same test; kept for parity)
            keys[off] = ((uint64_t)kInvalid << 32) | f2u(3.4028235e+38f);
            ids[off] = kInvalid;
        }
    } else {
        for (int y = bb.y0; y < bb.y1; ++y)
            for (int x = bb.x0; x < bb.x1; ++x) {
                keys[off] = ((uint64_t)(uint32_t)(y * v.grid_x + x) << 32) | dkey;
                ids[off] = i;
                off++;
            }
    }
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
__global__ __launch_bounds__(kBlock) void k_project_backward(ViewParams v, uint32_t n, int sh_degree,
                                                            const float4* __restrict__ density12,
                                                            const uint32_t* __restrict__ tiles_count,
                                                            const float* __restrict__ feat, const float4* __restrict__ grad16,
                                                            float4* __restrict__ density_grad12,
                                                            float4* __restrict__ sph_grad48) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
    float out[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) out[k] = 0.0f;
    if (tiles_count[i] != 0) {
        g0 = grad16[4 * (size_t)i + 0];
        g1 = grad16[4 * (size_t)i + 1];
        g2 = grad16[4 * (size_t)i + 2];
        const float4 g3 = grad16[4 * (size_t)i + 3];
        const float dr = g2.w, dg = g3.x, db = g3.y;
        g2.w = 0.0f;
        const float4 a = density12[3 * (size_t)i];
        const float rx = a.x - v.s2w.t[0], ry = a.y - v.s2w.t[1], rz = a.z - v.s2w.t[2];
        const float dist = sqrtf(rx * rx + ry * ry + rz * rz);
        float Y[16];
        sh_basis(sh_degree, rx / dist, ry / dist, rz / dist, Y);
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
    density_grad12[3 * (size_t)i + 0] = g0;
    density_grad12[3 * (size_t)i + 1] = g1;
    density_grad12[3 * (size_t)i + 2] = g2;
#pragma unroll
    for (int k = 0; k < 12; ++k)
        sph_grad48[12 * (size_t)i + k] = make_float4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

void launch_project(hipStream_t s, const ViewParams& v, const RenderConsts& c, uint32_t n, int sh_degree,
                    const float* density12, const float* sph48, uint32_t* tiles_count, float* proj_pos,
                    float* conic_opacity, float* extent, float* depth, float* feat, float* visibility,
                    Counters* counters) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_project_on_tiles, dim3(blocks_for(n)), dim3(kBlock), 0, s, v, c, n, sh_degree,
                       reinterpret_cast<const float4*>(density12), sph48, tiles_count, reinterpret_cast<float2*>(proj_pos),
                       reinterpret_cast<float4*>(conic_opacity), reinterpret_cast<float2*>(extent), depth, feat, visibility,
                       counters);
}

void launch_expand(hipStream_t s, const ViewParams& v, const RenderConsts& c, uint32_t n, const uint32_t* offset,
                   const float* proj_pos, const float* conic_opacity, const float* extent, const float* depth,
                   uint64_t* keys, uint32_t* ids) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_expand_tiles, dim3(blocks_for(n)), dim3(kBlock), 0, s, v, c, n, offset,
                       reinterpret_cast<const float2*>(proj_pos), reinterpret_cast<const float4*>(conic_opacity),
                       reinterpret_cast<const float2*>(extent), depth, keys, ids);
}

void launch_tile_ranges(hipStream_t s, uint32_t m, const uint64_t* sorted_keys, uint32_t* ranges) {
    if (m == 0) return;
    hipLaunchKernelGGL(k_tile_ranges, dim3(blocks_for(m)), dim3(kBlock), 0, s, m, sorted_keys,
                       reinterpret_cast<uint2*>(ranges));
}

void launch_project_bwd(hipStream_t s, const ViewParams& v, uint32_t n, int sh_degree, const float* density12,
                        const uint32_t* tiles_count, const float* feat, const float* grad16, float* density_grad12,
                        float* sph_grad48) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_project_backward, dim3(blocks_for(n)), dim3(kBlock), 0, s, v, n, sh_degree,
                       reinterpret_cast<const float4*>(density12), tiles_count, feat,
                       reinterpret_cast<const float4*>(grad16), reinterpret_cast<float4*>(density_grad12),
                       reinterpret_cast<float4*>(sph_grad48));
}

}  // namespace gut
