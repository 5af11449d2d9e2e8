/*
 * gut_oracle.c — CPU restatement of the reference's 3DGUT renderer (forward + backward).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (3dgrut_amd/, bench.py's timed GPU
 * region) may link, load or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg use it, and only as the checker.
 *
 * PARITY STATUS: "parity unpinned" for the device kernels — the reference holds no tests, golden
 * images or known-answer vectors for this path, and its CUDA/Slang sources
 * cannot be compiled here (no nvcc, no slangc, tiny-cuda-nn un-vendored).  This file restates the
 * reference sources read as text; every function cites the reference file:line it follows.  The
 * host-side pose math is pinned by tests/golden/pose_golden.npz (generated from the reference's
 * own Python, see tests/golden/gen_pose_golden.py).
 *
 * Numerics contract shared with the HIP kernels (so that integer tile/key buffers are bit-exact):
 *   - all arithmetic fp32, evaluated left-to-right as written, NO fused multiply-add
 *     (compile with -ffp-contract=off);
 *   - sqrtf and '/' are IEEE correctly rounded (true on x86-64 and in hipcc's default mode);
 *   - logf / atan2f are NOT taken from libm (glibc and the ROCm device library differ in the last
 *     ulp): both sides evaluate det_logf / det_atan2f_pos below (fdlibm- and Cephes-style
 *     published algorithms, +,-,*,/ and integer ops only);
 *   - float -> int conversions are clamped in the float domain first (the reference relies on
 *     CUDA's saturating cvt, gutProjector.cuh:35-40).
 * The reference itself is built with -use_fast_math (setup_3dgut.py:83), so not even two CUDA
 * GPUs of different generations agree with each other bit-for-bit; the contract above is what
 * "bit-exact" is defined against.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GUT_TILE 16
#define GUT_BLOCK 256
#define INVALID_IDX 0xFFFFFFFFu

/* ---- render constants: configs/render/3dgut.yaml, configs/render/3dgrt.yaml, threedgut.cuh:32-88 ---- */
typedef struct {
    float alpha_threshold;        /* particle_kernel_min_alpha = 1/255          (3dgrt.yaml:8)  */
    float max_alpha;              /* particle_kernel_max_alpha = 0.99           (3dgrt.yaml:9)  */
    float min_kernel_density;     /* particle_kernel_min_response = 0.0113      (3dgut.yaml:9)  */
    float min_transmittance;      /* min_transmittance = 1e-4                   (3dgut.yaml:10) */
    float min_sensor_z;           /* ParticleMinSensorZ = 0.2                   (threedgut.cuh:49) */
    float cov_dilation;           /* CovarianceDilation = 0.3                   (threedgut.cuh:50) */
    float ut_alpha, ut_beta, ut_kappa; /* 1, 2, 0                               (3dgut.yaml:19-21) */
    float ut_margin;              /* in_image_margin_factor = 0.1               (3dgut.yaml:22) */
    int32_t rect_bounding, tight_opacity_bounding, tile_culling, global_z_order; /* all 1 */
    /* kernel variants the reference compiles from its render config (setup_3dgut.py:47-56) */
    int32_t kernel_degree;              /* particle_kernel_degree = 2: GAUSSIAN_PARTICLE_KERNEL_DEGREE (threedgut.cuh:35); 0,1,3,4,5,8 */
    int32_t rolling_shutter_iterations; /* splat.n_rolling_shutter_iterations = 5 (threedgut.cuh:63) */
    int32_t enable_hitcounts;           /* enable_hitcounts = true: GAUSSIAN_ENABLE_HIT_COUNT (rayPayload.cuh:44-46,69-73,126-128) */
} OracleParams;

typedef struct {
    int32_t model;   /* 0 = OpenCV pinhole, 1 = OpenCV fisheye        (sensors/cameraModels.h:42-47) */
    int32_t shutter; /* 4 = global; 0..3 rolling                      (sensors/cameraModels.h:34-40) */
    float principal_point[2];
    float focal_length[2];
    float radial[6];      /* pinhole k1..k6; fisheye k1..k4 in radial[0..3] */
    float tangential[2];
    float thin_prism[4];
    float max_angle;
    float pose_start[7];  /* world->sensor: t(3), q(x,y,z,w)  (tracer.py:138-151, splatRaster.cpp:92-100) */
    float pose_end[7];
} OracleCamera;

void oracle_default_params(OracleParams* p) {
    p->alpha_threshold = 1.0f / 255.0f;
    p->max_alpha = 0.99f;
    p->min_kernel_density = 0.0113f;
    p->min_transmittance = 0.0001f;
    p->min_sensor_z = 0.2f;
    p->cov_dilation = 0.3f;
    p->ut_alpha = 1.0f; p->ut_beta = 2.0f; p->ut_kappa = 0.0f;
    p->ut_margin = 0.1f;
    p->rect_bounding = 1; p->tight_opacity_bounding = 1; p->tile_culling = 1; p->global_z_order = 1;
    p->kernel_degree = 2; p->rolling_shutter_iterations = 5; p->enable_hitcounts = 1;
}

/* ------------------------------------------------------------------------------------------------
 * deterministic elementary functions (see header)
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* natural log for finite x > 0 (fdlibm e_logf.c algorithm, general branch) */
static float det_logf(float x) {
    uint32_t ix = f2u(x);
    int k = 0;
    if (ix >= 0x7f800000u) return x;                 /* inf / nan passthrough */
    if (ix < 0x00800000u) {                          /* subnormal (or 0) */
        if (ix == 0) return -INFINITY;
        x = x * 33554432.0f; ix = f2u(x); k -= 25;
    }
    k += (int)(ix >> 23) - 127;
    ix &= 0x007fffffu;
    uint32_t i = (ix + (0x95f64u << 3)) & 0x800000u;
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

/* atan2(y, x) for y > 0 (Cephes atanf range reduction + polynomial); result in (0, pi) */
static float det_atan2f_pos(float y, float x) {
    const float ax = fabsf(x);
    const float lo = y < ax ? y : ax;
    const float hi = y < ax ? ax : y;
    float t = lo / hi;                               /* in [0,1] */
    float base = 0.0f;
    if (t > 0.4142135679721832f) { base = 0.7853981852531433f; t = (t - 1.0f) / (t + 1.0f); }
    const float z = t * t;
    float a = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t;
    a = base + a;
    if (y > ax) a = 1.5707963705062866f - a;
    if (x < 0.0f) a = 3.1415927410125732f - a;
    return a;
}

/* acos(x) for x in [0,1] and sin(x) for x in [0, pi/2]: fdlibm e_acosf.c / k_sinf.c / k_cosf.c algorithms in plain
 * fp32 (+,-,*,/,sqrt, bit masks).  Used only by the rolling-shutter slerp inside the projection, where the device
 * and the host must agree bit for bit. */
static float det_acosf01(float x) {
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
static float det_ksinf(float x) { /* |x| <= pi/4 */
    const float S1 = -1.6666667163e-01f, S2 = 8.3333337680e-03f, S3 = -1.9841270114e-04f, S4 = 2.7557314297e-06f,
                S5 = -2.5050759689e-08f, S6 = 1.5896910177e-10f;
    const float z = x * x;
    const float v = z * x;
    const float r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
static float det_kcosf(float x) { /* |x| <= pi/4 */
    const float C1 = 4.1666667908e-02f, C2 = -1.3888889225e-03f, C3 = 2.4801587642e-05f, C4 = -2.7557314297e-07f,
                C5 = 2.0875723372e-09f, C6 = -1.1359647598e-11f;
    const float z = x * x;
    const float r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return 1.0f - (0.5f * z - z * r);
}
static float det_sinf_0_pio2(float x) { /* x in [0, pi/2] */
    if (x <= 0.78539818525f) return det_ksinf(x);
    const float y = (1.5707962513e+00f - x) + 7.5497894159e-08f;
    return det_kcosf(y);
}

/* ------------------------------------------------------------------------------------------------
 * host-side pose math (float)
 * tiny-cuda-nn is not vendored: to_mat3 / quat(mat3) / slerp / mix are restated with
 * their GLM-equivalent textbook definitions, column-major, quaternion ctor (w,x,y,z).
 * Call sites: sensors/sensors.h:44-73, gutRenderer.cu:266-284.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float c[3][3]; } M3; /* c[col][row] */

static M3 quat_to_mat3(float w, float x, float y, float z) {
    M3 m;
    const float qxx = x * x, qyy = y * y, qzz = z * z;
    const float qxz = x * z, qxy = x * y, qyz = y * z;
    const float qwx = w * x, qwy = w * y, qwz = w * z;
    m.c[0][0] = 1.0f - 2.0f * (qyy + qzz); m.c[0][1] = 2.0f * (qxy + qwz); m.c[0][2] = 2.0f * (qxz - qwy);
    m.c[1][0] = 2.0f * (qxy - qwz); m.c[1][1] = 1.0f - 2.0f * (qxx + qzz); m.c[1][2] = 2.0f * (qyz + qwx);
    m.c[2][0] = 2.0f * (qxz + qwy); m.c[2][1] = 2.0f * (qyz - qwx); m.c[2][2] = 1.0f - 2.0f * (qxx + qyy);
    return m;
}

/* interpolatedSensorPose(start, end, 0.5), sensors/sensors.h:55-68 (slerp + mix) */
static void interpolate_pose(const float* a, const float* b, float t, float* out) {
    float qa[4] = {a[6], a[3], a[4], a[5]}; /* w,x,y,z */
    float qb[4] = {b[6], b[3], b[4], b[5]};
    float cosT = qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3];
    if (cosT < 0.0f) { for (int i = 0; i < 4; ++i) qb[i] = -qb[i]; cosT = -cosT; }
    float q[4];
    if (cosT > 1.0f - 1.1920929e-07f) {
        for (int i = 0; i < 4; ++i) q[i] = qa[i] * (1.0f - t) + qb[i] * t;
    } else {
        const float ang = acosf(cosT);
        const float s0 = sinf((1.0f - t) * ang), s1 = sinf(t * ang), sd = sinf(ang);
        for (int i = 0; i < 4; ++i) q[i] = (s0 * qa[i] + s1 * qb[i]) / sd;
    }
    for (int i = 0; i < 3; ++i) out[i] = a[i] * (1.0f - t) + b[i] * t;
    out[3] = q[1]; out[4] = q[2]; out[5] = q[3]; out[6] = q[0];
}

typedef struct {
    M3 Re; float te[3];      /* end pose world->sensor (rolling shutter fallback, cameraProjections.cuh:162-170) */
    M3 Rs; float ts[3];      /* start pose world->sensor (projection of sigma points, cameraProjections.cuh:154-157) */
    M3 Rm; float tm[3];      /* mid pose world->sensor (depth key, gutProjector.cuh:137,317) */
    M3 Rinv; float cam[3];   /* sensor->world (ray transform rayPayload.cuh:93-94) and sensor world position */
} PoseSet;

static PoseSet make_pose_set(const OracleCamera* cam) {
    PoseSet p;
    const float* s = cam->pose_start;
    p.Rs = quat_to_mat3(s[6], s[3], s[4], s[5]);
    p.ts[0] = s[0]; p.ts[1] = s[1]; p.ts[2] = s[2];
    const float* e = cam->pose_end;
    p.Re = quat_to_mat3(e[6], e[3], e[4], e[5]);
    p.te[0] = e[0]; p.te[1] = e[1]; p.te[2] = e[2];
    float mid[7];
    interpolate_pose(cam->pose_start, cam->pose_end, 0.5f, mid);
    p.Rm = quat_to_mat3(mid[6], mid[3], mid[4], mid[5]);
    p.tm[0] = mid[0]; p.tm[1] = mid[1]; p.tm[2] = mid[2];
    /* sensorPoseInverse (sensors.h:44-53): R^T and -R^T t.  The reference round-trips R^T through
     * a quaternion (tcnn::quat{mat3}) and back; that only adds rounding noise and is skipped. */
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) p.Rinv.c[c][r] = p.Rm.c[r][c];
    for (int r = 0; r < 3; ++r)
        p.cam[r] = -1.0f * (p.Rinv.c[0][r] * p.tm[0] + p.Rinv.c[1][r] * p.tm[1] + p.Rinv.c[2][r] * p.tm[2]);
    return p;
}

/* exported for the pose tests: world->sensor (t,q xyzw) -> 3x4 matrices, row-major [3][4] */
void oracle_pose_matrices(const OracleCamera* cam, float* w2s_start, float* w2s_mid, float* s2w) {
    PoseSet p = make_pose_set(cam);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            w2s_start[r * 4 + c] = p.Rs.c[c][r];
            w2s_mid[r * 4 + c] = p.Rm.c[c][r];
            s2w[r * 4 + c] = p.Rinv.c[c][r];
        }
        w2s_start[r * 4 + 3] = p.ts[r];
        w2s_mid[r * 4 + 3] = p.tm[r];
        s2w[r * 4 + 3] = p.cam[r];
    }
}

/* ------------------------------------------------------------------------------------------------
 * camera projections — sensors/cameraProjections.cuh:52-128
 * ---------------------------------------------------------------------------------------------- */
static int within_resolution(float rx, float ry, float tol, float px, float py) {
    const float mx = rx * tol, my = ry * tol;
    return (px > -mx) && (py > -my) && (px < rx + mx) && (py < ry + my);
}

static int pinhole_is_distorted(const OracleCamera* c) {
    for (int i = 0; i < 6; ++i) if (c->radial[i] != 0.0f) return 1;
    for (int i = 0; i < 2; ++i) if (c->tangential[i] != 0.0f) return 1;
    for (int i = 0; i < 4; ++i) if (c->thin_prism[i] != 0.0f) return 1;
    return 0;
}

static int project_pinhole(const OracleCamera* c, int W, int H, const float p[3], float tol, float out[2]) {
    if (p[2] <= 0.0f) { out[0] = 0.0f; out[1] = 0.0f; return 0; }
    /* position.xy()/position.z evaluated as multiplication by one IEEE reciprocal (the reference's
     * -use_fast_math build lowers its divisions to reciprocal-multiplies too) */
    const float rz = 1.0f / p[2];
    const float u = p[0] * rz, v = p[1] * rz;
    if (!pinhole_is_distorted(c)) {
        /* all distortion coefficients zero: icD == 1 and delta == 0 exactly, so for finite inputs this is
         * bit-identical to the general path below; it only differs where r2 overflows (0*inf = NaN there) */
        out[0] = u * c->focal_length[0] + c->principal_point[0];
        out[1] = v * c->focal_length[1] + c->principal_point[1];
        return within_resolution((float)W, (float)H, tol, out[0], out[1]);
    }
    const float u2 = u * u, v2 = v * v;
    const float r2 = u2 + v2;
    const float a1 = 2.0f * u * v;
    const float a2 = r2 + 2.0f * u2;
    const float a3 = r2 + 2.0f * v2;
    const float num = 1.0f + r2 * (c->radial[0] + r2 * (c->radial[1] + r2 * c->radial[2]));
    const float den = 1.0f + r2 * (c->radial[3] + r2 * (c->radial[4] + r2 * c->radial[5]));
    const float icD = num / den;
    const float dx = c->tangential[0] * a1 + c->tangential[1] * a2 + r2 * (c->thin_prism[0] + r2 * c->thin_prism[1]);
    const float dy = c->tangential[0] * a3 + c->tangential[1] * a1 + r2 * (c->thin_prism[2] + r2 * c->thin_prism[3]);
    const float und_x = icD * u + dx, und_y = icD * v + dy;
    const int valid_radial = (icD > 0.8f) && (icD < 1.2f);
    if (valid_radial) {
        out[0] = und_x * c->focal_length[0] + c->principal_point[0];
        out[1] = und_y * c->focal_length[1] + c->principal_point[1];
    } else {
        /* hypotf(W,H) restated as sqrtf(W*W+H*H) (deterministic; same value for image-sized ints) */
        const float fw = (float)W, fh = (float)H;
        const float clip = sqrtf(fw * fw + fh * fh);
        const float k = clip / sqrtf(r2);
        out[0] = k * u + c->principal_point[0];
        out[1] = k * v + c->principal_point[1];
    }
    return valid_radial && within_resolution((float)W, (float)H, tol, out[0], out[1]);
}

static int project_fisheye(const OracleCamera* c, int W, int H, const float p[3], float tol, float out[2]) {
    const float eps = 1.1920929e-07f;
    float rho = sqrtf(p[0] * p[0] + p[1] * p[1]);
    rho = rho > eps ? rho : eps;
    const float theta_full = det_atan2f_pos(rho, p[2]);
    const float theta = theta_full < c->max_angle ? theta_full : c->max_angle;
    const float t2 = theta * theta;
    /* evalPolyHorner<4>(radialCoeffs, theta2): y = c3; y = x*y + c_i */
    float poly = c->radial[3];
    poly = t2 * poly + c->radial[2];
    poly = t2 * poly + c->radial[1];
    poly = t2 * poly + c->radial[0];
    const float delta = (theta * (poly * t2 + 1.0f)) / rho;
    out[0] = c->focal_length[0] * p[0] * delta + c->principal_point[0];
    out[1] = c->focal_length[1] * p[1] * delta + c->principal_point[1];
    return (theta < c->max_angle) && within_resolution((float)W, (float)H, tol, out[0], out[1]);
}

static int project_camera_point(const OracleCamera* c, int W, int H, const float p[3], float tol, float out[2]) {
    if (c->model == 0) return project_pinhole(c, W, H, p, tol, out);
    if (c->model == 1) return project_fisheye(c, W, H, p, tol, out);
    out[0] = 0.0f; out[1] = 0.0f;
    return 0;
}

/* relativeShutterTime, cameraProjections.cuh:35-50 */
static float relative_shutter_time(int shutter, int W, int H, const float pos[2]) {
    switch (shutter) {
    case 0: return floorf(pos[1]) / ((float)H - 1.0f);
    case 1: return floorf(pos[0]) / ((float)W - 1.0f);
    case 2: return ((float)H - ceilf(pos[1])) / ((float)H - 1.0f);
    case 3: return ((float)W - ceilf(pos[0])) / ((float)W - 1.0f);
    default: return 0.5f;
    }
}

/* slerp(q0,q1,a) (GLM definition, w,x,y,z) + mix(t0,t1,a) applied to a world point, deterministic trig */
static void shutter_pose_transform(const float* ps, const float* pe, float a, const float w[3], float out[3]) {
    float q0[4] = {ps[6], ps[3], ps[4], ps[5]}, q1[4] = {pe[6], pe[3], pe[4], pe[5]}, q[4];
    float cosT = q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2] + q0[3] * q1[3];
    if (cosT < 0.0f) { for (int i = 0; i < 4; ++i) q1[i] = -q1[i]; cosT = -cosT; }
    if (cosT > 1.0f - 1.1920929e-07f) {
        for (int i = 0; i < 4; ++i) q[i] = q0[i] * (1.0f - a) + q1[i] * a;
    } else {
        const float ang = det_acosf01(cosT);
        const float s0 = det_sinf_0_pio2((1.0f - a) * ang), s1 = det_sinf_0_pio2(a * ang), sd = det_sinf_0_pio2(ang);
        for (int i = 0; i < 4; ++i) q[i] = (s0 * q0[i] + s1 * q1[i]) / sd;
    }
    const M3 R = quat_to_mat3(q[0], q[1], q[2], q[3]);
    for (int r = 0; r < 3; ++r) {
        const float t = ps[r] * (1.0f - a) + pe[r] * a;
        out[r] = R.c[0][r] * w[0] + R.c[1][r] * w[1] + R.c[2][r] * w[2] + t;
    }
}

/* projectPointWithShutter<N>, cameraProjections.cuh:146-185 (N = GAUSSIAN_N_ROLLING_SHUTTER_ITERATIONS, 5 in render/3dgut.yaml) */
static int project_world_point(const OracleCamera* c, const PoseSet* ps, int W, int H, const float w[3], float tol, int iterations,
                               float out[2]) {
    float p[3];
    for (int r = 0; r < 3; ++r)
        p[r] = ps->Rs.c[0][r] * w[0] + ps->Rs.c[1][r] * w[1] + ps->Rs.c[2][r] * w[2] + ps->ts[r];
    int valid = project_camera_point(c, W, H, p, tol, out);
    if (c->shutter == 4) return valid;
    if (!valid) {
        for (int r = 0; r < 3; ++r)
            p[r] = ps->Re.c[0][r] * w[0] + ps->Re.c[1][r] * w[1] + ps->Re.c[2][r] * w[2] + ps->te[r];
        valid = project_camera_point(c, W, H, p, tol, out);
        if (!valid) return 0;
    }
    for (int it = 0; it < iterations; ++it) {
        const float a = relative_shutter_time(c->shutter, W, H, out);
        shutter_pose_transform(c->pose_start, c->pose_end, a, w, p);
        valid = project_camera_point(c, W, H, p, tol, out);
    }
    return valid;
}

/* quaternion (w,x,y,z) -> rows of rotationT — slang/common/transforms.slang:22-39 */
static void quat_to_rows(const float q[4], float r[3][3]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float rx = w * x, ry = w * y, rz = w * z;
    r[0][0] = 1.0f - 2.0f * (yy + zz); r[0][1] = 2.0f * (xy + rz); r[0][2] = 2.0f * (xz - ry);
    r[1][0] = 2.0f * (xy - rz); r[1][1] = 1.0f - 2.0f * (xx + zz); r[1][2] = 2.0f * (yz + rx);
    r[2][0] = 2.0f * (xz + ry); r[2][1] = 2.0f * (yz - rx); r[2][2] = 1.0f - 2.0f * (xx + yy);
}

/* SH basis Y_k(dir), k<16 — models/gaussianParticles.cuh:57-96 (same constants/signs as 3DGS) */
static void sh_basis(int deg, const float d[3], float Y[16]) {
    for (int i = 0; i < 16; ++i) Y[i] = 0.0f;
    Y[0] = 0.28209479177387814f;
    if (deg > 0) {
        const float x = d[0], y = d[1], z = d[2];
        Y[1] = -0.4886025119029199f * y; Y[2] = 0.4886025119029199f * z; Y[3] = -0.4886025119029199f * x;
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

/* ------------------------------------------------------------------------------------------------
 * tile helpers — gutProjector.cuh:32-78
 * ---------------------------------------------------------------------------------------------- */
static int clamp_tile(float v, int grid) {
    /* min(grid, max(0, (int)v)) with the clamp done in float (saturating-cvt semantics) */
    if (!(v > 0.0f)) return 0;                /* also NaN -> 0 */
    if (v >= (float)grid) return grid;
    return (int)v;
}

static void tile_bbox(int gx, int gy, const float pos[2], const float ext[2], int bmin[2], int bmax[2]) {
    bmin[0] = clamp_tile(floorf((pos[0] - 0.5f - ext[0]) / 16.0f), gx);
    bmin[1] = clamp_tile(floorf((pos[1] - 0.5f - ext[1]) / 16.0f), gy);
    bmax[0] = clamp_tile(ceilf((pos[0] - 0.5f + ext[0]) / 16.0f), gx);
    bmax[1] = clamp_tile(ceilf((pos[1] - 0.5f + ext[1]) / 16.0f), gy);
}

static float saturatef(float v) { return v > 0.0f ? (v < 1.0f ? v : 1.0f) : 0.0f; }

/* tileMinParticlePowerResponse, gutProjector.cuh:49-78 */
static float tile_min_power(float tx, float ty, const float conic[4], const float mean[2]) {
    const float ts = 16.0f;
    const float tminx = ts * tx, tminy = ts * ty;
    const float tmaxx = ts + tminx, tmaxy = ts + tminy;
    const float offx = tminx - mean[0], offy = tminy - mean[1];
    const float lax = offx > 0.0f ? 1.0f : 0.0f, lay = offy > 0.0f ? 1.0f : 0.0f;
    const float nrx = lax + (mean[0] > tmaxx ? 1.0f : 0.0f);
    const float nry = lay + (mean[1] > tmaxy ? 1.0f : 0.0f);
    if ((nrx + nry) > 0.0f) {
        const float px = lax > 0.0f ? tminx : tmaxx;
        const float py = lay > 0.0f ? tminy : tmaxy;
        const float dxx = copysignf(ts, offx), dxy = copysignf(ts, offy);
        const float dfx = mean[0] - px, dfy = mean[1] - py;
        const float rcx = 1.0f / (ts * ts * conic[0]);
        const float rcy = 1.0f / (ts * ts * conic[2]);
        const float tx_ = nry * saturatef((dxx * conic[0] * dfx + dxx * conic[1] * dfy) * rcx);
        const float ty_ = nrx * saturatef((dxy * conic[1] * dfx + dxy * conic[2] * dfy) * rcy);
        const float mx = mean[0] - (px + tx_ * dxx);
        const float my = mean[1] - (py + ty_ * dxy);
        return 0.5f * (conic[0] * mx * mx + conic[2] * my * my) + conic[1] * mx * my;
    }
    return 0.0f;
}

/* ------------------------------------------------------------------------------------------------
 * K1: projectOnTiles — gutProjector.cuh:81-322
 * Outputs per Gaussian: tiles_count, proj_pos(2), conic_opacity(4), extent(2), depth, feat(3), visibility.
 * Deviation: visibility := validProjection && validConic (the reference reads
 * an uninitialised covariance for culled Gaussians).
 * ---------------------------------------------------------------------------------------------- */
void oracle_project(const OracleParams* prm, const OracleCamera* cam, int W, int H, uint32_t N, int sh_degree,
                    const float* density12, const float* sph48,
                    uint32_t* tiles_count, float* proj_pos, float* conic_opacity, float* extent,
                    float* depth, float* feat, int32_t* visibility) {
    const PoseSet ps = make_pose_set(cam);
    const int gx = (W + GUT_TILE - 1) / GUT_TILE, gy = (H + GUT_TILE - 1) / GUT_TILE;
    const float D = 3.0f;
    const float lambda = prm->ut_alpha * prm->ut_alpha * (D + prm->ut_kappa) - D;
    /* GAUSSIAN_UT_DELTA: computed in double by the build script and rounded once (setup_3dgut.py:41-45, threedgut.cuh:70) */
    const float delta_f = (float)sqrt((double)prm->ut_alpha * (double)prm->ut_alpha * (3.0 + (double)prm->ut_kappa));
    const float w0_mean = lambda / (D + lambda);
    const float wi = 1.0f / (2.0f * (D + lambda));
    const float w0_cov = lambda / (D + lambda) + (1.0f - prm->ut_alpha * prm->ut_alpha + prm->ut_beta);

#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t ii = 0; ii < (int64_t)N; ++ii) {
        const uint32_t i = (uint32_t)ii;
        const float* g = density12 + (size_t)i * 12;
        const float pos[3] = {g[0], g[1], g[2]};
        const float opacity_in = g[3];
        const float quat[4] = {g[4], g[5], g[6], g[7]};
        const float scl[3] = {g[8], g[9], g[10]};

        tiles_count[i] = 0;
        proj_pos[2 * i] = 0.0f; proj_pos[2 * i + 1] = 0.0f;
        for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = 0.0f;
        extent[2 * i] = 0.0f; extent[2 * i + 1] = 0.0f;
        depth[i] = 0.0f;
        feat[3 * i] = 0.0f; feat[3 * i + 1] = 0.0f; feat[3 * i + 2] = 0.0f;
        visibility[i] = 0;

        /* unscentedParticleProjection, gutProjector.cuh:118-215 */
        if (opacity_in < prm->alpha_threshold) continue;
        const float zcam = pos[0] * ps.Rm.c[0][2] + pos[1] * ps.Rm.c[1][2] + pos[2] * ps.Rm.c[2][2] + ps.tm[2];
        if (zcam < prm->min_sensor_z) continue;

        float rows[3][3];
        quat_to_rows(quat, rows);
        float sig[7][2];
        int nvalid = 0;
        nvalid += project_world_point(cam, &ps, W, H, pos, prm->ut_margin, prm->rolling_shutter_iterations, sig[0]);
        float cx = sig[0][0] * w0_mean, cy = sig[0][1] * w0_mean;
        for (int a = 0; a < 3; ++a) {
            const float k = delta_f * scl[a];
            const float d[3] = {k * rows[a][0], k * rows[a][1], k * rows[a][2]};
            const float pp[3] = {pos[0] + d[0], pos[1] + d[1], pos[2] + d[2]};
            const float pm[3] = {pos[0] - d[0], pos[1] - d[1], pos[2] - d[2]};
            nvalid += project_world_point(cam, &ps, W, H, pp, prm->ut_margin, prm->rolling_shutter_iterations, sig[a + 1]);
            cx += wi * sig[a + 1][0]; cy += wi * sig[a + 1][1];
            nvalid += project_world_point(cam, &ps, W, H, pm, prm->ut_margin, prm->rolling_shutter_iterations, sig[a + 4]);
            cx += wi * sig[a + 4][0]; cy += wi * sig[a + 4][1];
        }
        if (nvalid == 0) continue;
        float cov[3];
        {
            const float ex = sig[0][0] - cx, ey = sig[0][1] - cy;
            cov[0] = w0_cov * (ex * ex); cov[1] = w0_cov * (ex * ey); cov[2] = w0_cov * (ey * ey);
        }
        for (int k = 0; k < 6; ++k) {
            const float ex = sig[k + 1][0] - cx, ey = sig[k + 1][1] - cy;
            cov[0] += wi * (ex * ex); cov[1] += wi * (ex * ey); cov[2] += wi * (ey * ey);
        }

        /* computeProjectedExtentConicOpacity, gutProjector.cuh:81-116 */
        const float dcx = cov[0] + prm->cov_dilation, dcy = cov[1], dcz = cov[2] + prm->cov_dilation;
        const float ddet = dcx * dcz - dcy * dcy;
        if (ddet == 0.0f) continue;
        float con[4];
        const float inv_det = 1.0f / ddet; /* vec3 / scalar as one reciprocal + multiplies */
        con[0] = dcz * inv_det; con[1] = -dcy * inv_det; con[2] = dcx * inv_det;
        const float cdet = cov[0] * cov[2] - cov[1] * cov[1];
        const float ratio = cdet * inv_det;
        const float conv = sqrtf(ratio > 0.000025f ? ratio : 0.000025f); /* fmaxf(0.000025, ratio); NaN -> 0.000025 */
        con[3] = opacity_in * conv;
        if (con[3] < prm->alpha_threshold) continue;
        const float max_power = det_logf(con[3] / prm->alpha_threshold);
        float extent_factor = 3.33f;
        if (prm->tight_opacity_bounding) {
            const float e = sqrtf(2.0f * max_power);
            extent_factor = e < 3.33f ? e : 3.33f;
        }
        const float mid = 0.5f * (dcx + dcz);
        const float disc = mid * mid - ddet;
        const float lam = mid + sqrtf(disc > 0.01f ? disc : 0.01f); /* fmaxf(0.01, disc) */
        const float radius = extent_factor * sqrtf(lam);
        float ext[2];
        if (prm->rect_bounding) {
            const float ex = extent_factor * sqrtf(dcx), ey = extent_factor * sqrtf(dcz);
            ext[0] = ex < radius ? ex : radius; ext[1] = ey < radius ? ey : radius;
        } else { ext[0] = radius; ext[1] = radius; }
        if (!(radius > 0.0f)) continue;

        visibility[i] = 1;

        /* tile count with per-tile culling, gutProjector.cuh:279-293 */
        const float mean2[2] = {cx, cy};
        int bmin[2], bmax[2];
        tile_bbox(gx, gy, mean2, ext, bmin, bmax);
        uint32_t cnt = 0;
        if (prm->tile_culling) {
            for (int y = bmin[1]; y < bmax[1]; ++y)
                for (int x = bmin[0]; x < bmax[0]; ++x)
                    if (tile_min_power((float)x, (float)y, con, mean2) < max_power) cnt++;
        } else {
            cnt = (uint32_t)((bmax[0] - bmin[0]) * (bmax[1] - bmin[1]));
        }
        tiles_count[i] = cnt;
        if (cnt == 0) continue;

        /* precomputed view-dependent RGB (unclamped), gutProjector.cuh:304-310 */
        const float sr[3] = {pos[0] - ps.cam[0], pos[1] - ps.cam[1], pos[2] - ps.cam[2]};
        const float dist = sqrtf(sr[0] * sr[0] + sr[1] * sr[1] + sr[2] * sr[2]);
        const float inv_dist = 1.0f / dist;
        const float dir[3] = {sr[0] * inv_dist, sr[1] * inv_dist, sr[2] * inv_dist};
        float Y[16];
        sh_basis(sh_degree, dir, Y);
        const int ncoef = (sh_degree + 1) * (sh_degree + 1);
        const float* sh = sph48 + (size_t)i * 48;
        for (int ch = 0; ch < 3; ++ch) {
            float acc = 0.0f;
            for (int k = 0; k < ncoef; ++k) acc += Y[k] * sh[3 * k + ch];
            feat[3 * i + ch] = acc + 0.5f;
        }
        proj_pos[2 * i] = cx; proj_pos[2 * i + 1] = cy;
        for (int k = 0; k < 4; ++k) conic_opacity[4 * i + k] = con[k];
        extent[2 * i] = ext[0]; extent[2 * i + 1] = ext[1];
        depth[i] = prm->global_z_order ? zcam : dist;
    }
}

/* K2: inclusive scan — gutRenderer.cu:302-310 */
uint32_t oracle_scan(uint32_t N, const uint32_t* count, uint32_t* offset) {
    uint32_t acc = 0;
    for (uint32_t i = 0; i < N; ++i) { acc += count[i]; offset[i] = acc; }
    return acc;
}

/* K3: expandTileProjections — gutProjector.cuh:324-388 */
void oracle_expand(const OracleParams* prm, int W, int H, uint32_t N,
                   const uint32_t* offset, const float* proj_pos, const float* conic_opacity,
                   const float* extent, const float* depth, uint64_t* keys, uint32_t* ids) {
    const int gx = (W + GUT_TILE - 1) / GUT_TILE, gy = (H + GUT_TILE - 1) / GUT_TILE;
    for (uint32_t i = 0; i < N; ++i) {
        const float ext[2] = {extent[2 * i], extent[2 * i + 1]};
        if (ext[0] <= 1e-06f) continue;
        const uint32_t dkey = f2u(depth[i]);
        uint32_t off = (i == 0) ? 0 : offset[i - 1];
        const uint32_t max_off = offset[i];
        const float mean2[2] = {proj_pos[2 * i], proj_pos[2 * i + 1]};
        int bmin[2], bmax[2];
        tile_bbox(gx, gy, mean2, ext, bmin, bmax);
        if (prm->tile_culling) {
            const float* con = conic_opacity + 4 * (size_t)i;
            const float max_power = det_logf(con[3] / prm->alpha_threshold);
            for (int y = bmin[1]; (y < bmax[1]) && (off < max_off); ++y)
                for (int x = bmin[0]; (x < bmax[0]) && (off < max_off); ++x)
                    if (tile_min_power((float)x, (float)y, con, mean2) < max_power) {
                        keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dkey;
                        ids[off] = i;
                        off++;
                    }
            for (; off < max_off; ++off) {
                keys[off] = ((uint64_t)INVALID_IDX << 32) | f2u(3.4028235e+38f);
                ids[off] = INVALID_IDX;
            }
        } else {
            for (int y = bmin[1]; y < bmax[1]; ++y)
                for (int x = bmin[0]; x < bmax[0]; ++x) {
                    keys[off] = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dkey;
                    ids[off] = i;
                    off++;
                }
        }
    }
}

/* higherMsb — gutRenderer.cu:79-94 (== bit_width for n >= 1) */
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

uint32_t oracle_higher_msb(uint32_t n) {
    uint32_t msb = 16, step = 16;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* K4: stable LSD radix sort of (key,value) on bits [0,end_bit) — cub::DeviceRadixSort::SortPairs,
 * gutRenderer.cu:356-365 */
void oracle_sort_pairs(uint32_t M, int end_bit, const uint64_t* keys_in, const uint32_t* vals_in,
                       uint64_t* keys_out, uint32_t* vals_out) {
    if (M == 0) return;
    uint64_t* ka = (uint64_t*)malloc(sizeof(uint64_t) * M);
    uint64_t* kb = (uint64_t*)malloc(sizeof(uint64_t) * M);
    uint32_t* va = (uint32_t*)malloc(sizeof(uint32_t) * M);
    uint32_t* vb = (uint32_t*)malloc(sizeof(uint32_t) * M);
    memcpy(ka, keys_in, sizeof(uint64_t) * M);
    memcpy(va, vals_in, sizeof(uint32_t) * M);
    for (int bit = 0; bit < end_bit; bit += 8) {
        const int nb = (end_bit - bit) < 8 ? (end_bit - bit) : 8;
        const uint64_t mask = ((uint64_t)1 << nb) - 1;
        size_t hist[257];
        memset(hist, 0, sizeof(hist));
        for (uint32_t i = 0; i < M; ++i) hist[((ka[i] >> bit) & mask) + 1]++;
        for (int d = 0; d < 256; ++d) hist[d + 1] += hist[d];
        for (uint32_t i = 0; i < M; ++i) {
            const size_t p = hist[(ka[i] >> bit) & mask]++;
            kb[p] = ka[i]; vb[p] = va[i];
        }
        uint64_t* tk = ka; ka = kb; kb = tk;
        uint32_t* tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, sizeof(uint64_t) * M);
    memcpy(vals_out, va, sizeof(uint32_t) * M);
    free(ka); free(kb); free(va); free(vb);
}

/* K5: computeSortedTileRangeIndices — gutRenderer.cu:46-76; ranges[2*T] pre-zeroed here */
void oracle_tile_ranges(uint32_t M, uint32_t T, const uint64_t* sorted_keys, uint32_t* ranges) {
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T);
    for (uint32_t k = 0; k < M; ++k) {
        const uint32_t tile = (uint32_t)(sorted_keys[k] >> 32);
        const int valid = tile != INVALID_IDX;
        if (k == 0) {
            if (valid) ranges[2 * tile] = 0;
        } else {
            const uint32_t prev = (uint32_t)(sorted_keys[k - 1] >> 32);
            if (prev != tile) {
                if (prev != INVALID_IDX) ranges[2 * prev + 1] = k;
                if (valid) ranges[2 * tile] = k;
            }
        }
        if (valid && (k == M - 1)) ranges[2 * tile + 1] = M;
    }
}

/* ------------------------------------------------------------------------------------------------
 * ray setup — common/rayPayload.cuh:76-108, utils/bounding_box.h:88-134
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float o[3], d[3], tmin, tmax; int alive; } Ray;

static void aabb_intersect(const float o[3], const float d[3], float* tmin_out, float* tmax_out) {
    const float lo = -1e06f, hi = 1e06f, fmax_ = 3.4028235e+38f;
    float tmin = (lo - o[0]) / d[0], tmax = (hi - o[0]) / d[0];
    if (tmin > tmax) { float t = tmin; tmin = tmax; tmax = t; }
    float tymin = (lo - o[1]) / d[1], tymax = (hi - o[1]) / d[1];
    if (tymin > tymax) { float t = tymin; tymin = tymax; tymax = t; }
    if (tmin > tymax || tymin > tmax) { *tmin_out = fmax_; *tmax_out = fmax_; return; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo - o[2]) / d[2], tzmax = (hi - o[2]) / d[2];
    if (tzmin > tzmax) { float t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tmin > tzmax || tzmin > tmax) { *tmin_out = fmax_; *tmax_out = fmax_; return; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    *tmin_out = tmin; *tmax_out = tmax;
}

static Ray make_ray(const PoseSet* ps, const float* ro, const float* rd) {
    Ray r;
    for (int k = 0; k < 3; ++k) {
        r.o[k] = ps->Rinv.c[0][k] * ro[0] + ps->Rinv.c[1][k] * ro[1] + ps->Rinv.c[2][k] * ro[2] + ps->cam[k];
        r.d[k] = ps->Rinv.c[0][k] * rd[0] + ps->Rinv.c[1][k] * rd[1] + ps->Rinv.c[2][k] * rd[2];
    }
    aabb_intersect(r.o, r.d, &r.tmin, &r.tmax);
    r.tmin = r.tmin > 0.0f ? r.tmin : 0.0f; /* fmaxf(tmin, 0) */
    r.alive = r.tmax > r.tmin;
    return r;
}

/* per-(ray, particle) response — slang/models/gaussianParticles.slang:96-222 */
typedef struct { float gro[3], grdu[3], grd[3], gposc[3], gposcr[3], rdr[3], d2, resp, alpha; } Hit;

/* Second, DIFFERENT BUT EQUALLY VALID fp32 evaluation of the same per-pair formulas (tests only: oracle_set_variant(1)) — the
 * form the HIP compositors use (gut_render.hip): the rows of diag(1/s) * rotationT rounded once, fused multiply-adds, the
 * response as |u x o|^2 / |u|^2 with one reciprocal (u = the UNnormalised canonical direction), exp2 of the pre-scaled
 * argument.  It follows no reference line that variant 0 does not follow; it exists so that the tolerance of the GPU parity
 * tests can be MEASURED as |variant 0 - variant 1| instead of estimated (tests/test_cpu_oracle.py, tests/common.py). */
static int g_variant = 0;   /* 0: the reference's operation order; 1: fused form; 2: fused form with 1-ulp transcendentals */
void oracle_set_variant(int v) { g_variant = v; }
int oracle_get_variant(void) { return g_variant; }

/* variant 2: the hardware's v_rcp_f32 / v_rsq_f32 / v_exp_f32 are specified to 1 ulp, not correctly rounded: move a correctly
 * rounded result by -1, 0 or +1 ulp, chosen by a hash of its bits (deterministic, about a third each) */
static float ulp_jitter(float x) {
    if (g_variant != 2 || !(x > 0.0f) || !(x < 3.0e38f)) return x;
    uint32_t b; memcpy(&b, &x, 4);
    uint32_t hsh = b * 2654435761u; hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    const uint32_t r = hsh % 3u;
    b += (r == 1u) ? 1u : (r == 2u ? 0xFFFFFFFFu : 0u);
    memcpy(&x, &b, 4);
    return x;
}

/* particleResponse<n> — kernels/cuda/models/gaussianParticles.cuh:256-306 (generalised Gaussian of degree n, s_n = -4.5 / 3^n;
 * n = 0 the linear hat; anything else the quadratic default), constants as written there */
static float kernel_response(int degree, float d2) {
    switch (degree) {
    case 8: { const float q = d2 * d2; return expf(-0.000685871056241f * q * q); }
    case 5: return expf(-0.0185185185185f * d2 * d2 * sqrtf(d2));
    case 4: return expf(-0.0555555555556f * d2 * d2);
    case 3: return expf(-0.166666666667f * d2 * sqrtf(d2));
    case 1: return expf(-1.5f * sqrtf(d2));
    case 0: return fmaxf(1.0f + -0.329630334487f * sqrtf(d2), 0.0f);
    default: return expf(-0.5f * d2);
    }
}
/* particleResponseGrd<n> — gaussianParticles.cuh:211-254: dL/d(d2) from dL/d(resp).  The `constexpr float s = <double literal> *
 * (0.5f * n)` of the reference is a double product rounded once.  Degree 1 is restated AS WRITTEN there: it multiplies by sqrt(d2)
 * where the derivative of exp(s sqrt(d2)) divides by it (:248-252) — the reference's behaviour, not a correct derivative. */
static float kernel_response_grad(int degree, float d2, float resp, float g_resp) {
    switch (degree) {
    case 8: return (float)(-0.000685871056241 * 4.0) * (d2 * d2) * d2 * resp * g_resp;
    case 5: return (float)(-0.0185185185185 * 2.5) * d2 * sqrtf(d2) * resp * g_resp;
    case 4: return (float)(-0.0555555555556 * 2.0) * d2 * resp * g_resp;
    case 3: return (float)(-0.166666666667 * 1.5) * sqrtf(d2) * resp * g_resp;
    case 1: return (-1.5f * 0.5f) * sqrtf(d2) * resp * g_resp;
    case 0: return resp > 0.0f ? (0.5f * -0.329630334487f * (1.0f / sqrtf(d2))) * g_resp : 0.0f;
    default: return -0.5f * resp * g_resp;
    }
}
/* second fp32 form of the response (variants 1, 2): exp2 of the pre-scaled argument, 1-ulp square roots */
static float kernel_response_fused(int degree, float d2) {
    switch (degree) {
    case 8: { const float q = d2 * d2; return ulp_jitter(exp2f(1.4426950408889634f * (-0.000685871056241f * (q * q)))); }
    case 5: return ulp_jitter(exp2f(1.4426950408889634f * (-0.0185185185185f * (d2 * d2 * ulp_jitter(sqrtf(d2))))));
    case 4: return ulp_jitter(exp2f(1.4426950408889634f * (-0.0555555555556f * (d2 * d2))));
    case 3: return ulp_jitter(exp2f(1.4426950408889634f * (-0.166666666667f * (d2 * ulp_jitter(sqrtf(d2))))));
    case 1: return ulp_jitter(exp2f(1.4426950408889634f * (-1.5f * ulp_jitter(sqrtf(d2)))));
    case 0: return fmaxf(fmaf(-0.329630334487f, ulp_jitter(sqrtf(d2)), 1.0f), 0.0f);
    default: return ulp_jitter(exp2f(-0.72134752f * d2));
    }
}

static void eval_hit_fused(const OracleParams* prm, const float* g, const float rows[3][3], const Ray* ray, Hit* h) {
    const float* mu = g; const float* s = g + 8; const float sigma = g[3];
    float is[3], m[3][3];
    for (int k = 0; k < 3; ++k) {
        is[k] = ulp_jitter(1.0f / s[k]);
        for (int j = 0; j < 3; ++j) m[k][j] = is[k] * rows[k][j];
        h->gposc[k] = ray->o[k] - mu[k];
    }
    for (int k = 0; k < 3; ++k) {
        h->gposcr[k] = fmaf(rows[k][0], h->gposc[0], fmaf(rows[k][1], h->gposc[1], rows[k][2] * h->gposc[2]));
        h->rdr[k] = fmaf(rows[k][0], ray->d[0], fmaf(rows[k][1], ray->d[1], rows[k][2] * ray->d[2]));
        h->gro[k] = fmaf(m[k][0], h->gposc[0], fmaf(m[k][1], h->gposc[1], m[k][2] * h->gposc[2]));
        h->grdu[k] = fmaf(m[k][0], ray->d[0], fmaf(m[k][1], ray->d[1], m[k][2] * ray->d[2]));
    }
    const float l2 = fmaf(h->grdu[0], h->grdu[0], fmaf(h->grdu[1], h->grdu[1], h->grdu[2] * h->grdu[2]));
    const float il = l2 > 0.0f ? ulp_jitter(1.0f / sqrtf(l2)) : 1.0f;
    for (int k = 0; k < 3; ++k) h->grd[k] = h->grdu[k] * il;
    /* |u x o|^2 / |u|^2 */
    const float c0 = fmaf(h->grdu[1], h->gro[2], -(h->grdu[2] * h->gro[1]));
    const float c1 = fmaf(h->grdu[2], h->gro[0], -(h->grdu[0] * h->gro[2]));
    const float c2 = fmaf(h->grdu[0], h->gro[1], -(h->grdu[1] * h->gro[0]));
    const float n2 = fmaf(c0, c0, fmaf(c1, c1, c2 * c2));
    h->d2 = l2 > 0.0f ? n2 * ulp_jitter(1.0f / l2) : 0.0f;   /* zero direction: grd = 0, as in variant 0 */
    h->resp = kernel_response_fused(prm->kernel_degree, h->d2);
    const float a = h->resp * sigma;
    h->alpha = a < prm->max_alpha ? a : prm->max_alpha;
}

static void eval_hit(const OracleParams* prm, const float* g, const float rows[3][3], const Ray* ray, Hit* h) {
    if (g_variant != 0) { eval_hit_fused(prm, g, rows, ray, h); return; }
    const float* mu = g; const float* s = g + 8; const float sigma = g[3];
    for (int k = 0; k < 3; ++k) h->gposc[k] = ray->o[k] - mu[k];
    for (int k = 0; k < 3; ++k) {
        h->gposcr[k] = rows[k][0] * h->gposc[0] + rows[k][1] * h->gposc[1] + rows[k][2] * h->gposc[2];
        h->rdr[k] = rows[k][0] * ray->d[0] + rows[k][1] * ray->d[1] + rows[k][2] * ray->d[2];
        h->gro[k] = (1.0f / s[k]) * h->gposcr[k];
        h->grdu[k] = (1.0f / s[k]) * h->rdr[k];
    }
    const float l2 = h->grdu[0] * h->grdu[0] + h->grdu[1] * h->grdu[1] + h->grdu[2] * h->grdu[2];
    const float il = l2 > 0.0f ? 1.0f / sqrtf(l2) : 1.0f;
    for (int k = 0; k < 3; ++k) h->grd[k] = h->grdu[k] * il;
    const float c0 = h->grd[1] * h->gro[2] - h->grd[2] * h->gro[1];
    const float c1 = h->grd[2] * h->gro[0] - h->grd[0] * h->gro[2];
    const float c2 = h->grd[0] * h->gro[1] - h->grd[1] * h->gro[0];
    h->d2 = c0 * c0 + c1 * c1 + c2 * c2;
    h->resp = kernel_response(prm->kernel_degree, h->d2);
    const float a = h->resp * sigma;
    h->alpha = a < prm->max_alpha ? a : prm->max_alpha;
}

/* fp32 noise of a pair's response, in units of eps = 2^-24, for ANY evaluation order of the formula (tests only: decision margins,
 * flip budgets, row conditioning).  resp = exp(-d2 / 2), d2 = |grd x gro|^2:
 *   - c = grd x gro is a cancellation: an absolute error delta(grd) |gro| + delta(gro) in each component of a vector of length sqrt(d2);
 *   - gro = diag(1/s) R (o - mu): R (o - mu) carries eps |o - mu| per component, divided by s_k: delta(gro) <= eps |o - mu| / s_min
 *     (for an isotropic Gaussian that IS eps |gro|; for a flat disc seen along its plane it is |1/s|_max / |1/s|_eff times more);
 *   - grd = normalize(diag(1/s) R d): delta(grd) <= eps (1 / s_min) / |diag(1/s) R d|  (1 for an isotropic Gaussian);
 *   - delta(d2) = 2 sqrt(d2) delta(c), delta(resp) / resp = delta(d2) / 2, plus exp's argument and result rounding (d2 / 2 + 2).
 * The two anisotropy factors were missing until round 4 (found with the surface-like stand-in, whose flat 8 : 1 discs made two CPU
 * evaluations differ by up to 9 x the old estimate, gpurun_out/r4/outliers_surface.json); the factor 1/2 keeps the estimate what it
 * was for isotropic Gaussians (sqrt(d2) |gro|), on which every constant of tests/common.py was measured. */
static float hit_noise_d2(const Hit* h, const float* s) {
    const float gn = sqrtf(h->gro[0] * h->gro[0] + h->gro[1] * h->gro[1] + h->gro[2] * h->gro[2]);
    float is_max = 1.0f / s[0];
    if (1.0f / s[1] > is_max) is_max = 1.0f / s[1];
    if (1.0f / s[2] > is_max) is_max = 1.0f / s[2];
    const float lu = sqrtf(h->grdu[0] * h->grdu[0] + h->grdu[1] * h->grdu[1] + h->grdu[2] * h->grdu[2]);
    const float lp = sqrtf(h->gposc[0] * h->gposc[0] + h->gposc[1] * h->gposc[1] + h->gposc[2] * h->gposc[2]);
    float spread = 0.5f * (gn * (lu > 0.0f ? is_max / lu : 1.0f) + lp * is_max);
    if (!(spread >= gn)) spread = gn;
    return sqrtf(h->d2) * spread;   /* delta(d2) / (2 eps) */
}
/* ... for the generalised kernels resp = exp(s d2^(n/2)): delta(resp) / resp = |s| (n/2) d2^(n/2-1) delta(d2) plus the rounding of the
 * exponent's argument (a few operations instead of one) and of the result; the linear hat (n = 0) has delta(resp) =
 * |s| delta(d2) / (2 sqrt(d2)), taken relative to the response like the rest (the margins compare against the thresholds). */
static float hit_noise(const OracleParams* prm, const Hit* h, const float* s) {
    const float base = hit_noise_d2(h, s);
    const float d2 = h->d2, r = sqrtf(d2);
    float slope, arg;   /* |d ln(resp) / d(d2)| and |ln(resp)| */
    switch (prm->kernel_degree) {
    case 8: arg = 0.000685871056241f * d2 * d2 * d2 * d2; slope = 4.0f * 0.000685871056241f * d2 * d2 * d2; break;
    case 5: arg = 0.0185185185185f * d2 * d2 * r; slope = 2.5f * 0.0185185185185f * d2 * r; break;
    case 4: arg = 0.0555555555556f * d2 * d2; slope = 2.0f * 0.0555555555556f * d2; break;
    case 3: arg = 0.166666666667f * d2 * r; slope = 1.5f * 0.166666666667f * r; break;
    case 1: arg = 1.5f * r; slope = r > 0.0f ? 0.75f / r : 0.0f; break;
    case 0: {
        const float resp = h->resp > 1e-3f ? h->resp : 1e-3f;
        return (r > 0.0f ? 0.329630334487f * base / r : 0.0f) / resp + 3.0f;
    }
    default: return base + 0.5f * d2 + 2.0f;   /* the quadratic kernel: exactly the estimate every constant of tests/common.py was measured on */
    }
    return 2.0f * slope * base + 4.0f * arg + 4.0f;
}

/* K6: render — gutRenderer.cuh:83-115, gutKBufferRenderer.cuh:108-170,217-292 (K=0),
 * rayPayload.cuh:110-129.  Also returns per-tile traversal counts (entries fetched before the
 * whole tile terminated) for the roofline statistics E_f. */
/* tests only: per-pixel FLIP BUDGET of the next oracle_render_margins call.  g_pixel_budget_out[2 pix] = how far a different but equally
 * valid fp32 evaluation may move the pixel's colour / opacity / (relative) hit distance because hit / no-hit decisions within
 * g_pixel_budget_bound noise widths of their thresholds flip: sum over those entries of 2 alpha T (a flipped hit adds or removes
 * alpha T (colour - what lies behind it), colours of magnitude <= 2), plus 2 T where the ray's termination is that close to its
 * threshold; [2 pix + 1] = by how many hits the count may differ (one per such entry; 64 more if the termination may flip). */
static float* g_pixel_budget_out = NULL;
static float g_pixel_budget_bound = 0.0f;
void oracle_set_pixel_budget_out(float* p, float bound) { g_pixel_budget_out = p; g_pixel_budget_bound = bound; }

/* tests only: when set, the next render / render_bwd also writes each tile's traversal depth (list entries fetched before every ray
 * of the tile had ended) there; the totals they return are the sums */
static uint32_t* g_tile_traversed_out = NULL;
void oracle_set_tile_traversed_out(uint32_t* p) { g_tile_traversed_out = p; }

static void render_impl(const OracleParams* prm, const OracleCamera* cam, int W, int H,
                   const float* density12, const float* feat,
                   const float* ray_ori, const float* ray_dir,
                   const uint32_t* ranges, const uint32_t* sorted_ids,
                   float* rgba, float* dist, float* hits, uint64_t* traversed_out, float* margins) {
    const PoseSet ps = make_pose_set(cam);
    const int gx = (W + GUT_TILE - 1) / GUT_TILE, gy = (H + GUT_TILE - 1) / GUT_TILE;
    uint64_t traversed_total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : traversed_total)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
        uint32_t deepest = 0;
        for (int py = ty * GUT_TILE; py < (ty + 1) * GUT_TILE && py < H; ++py)
            for (int px = tx * GUT_TILE; px < (tx + 1) * GUT_TILE && px < W; ++px) {
                const size_t pix = (size_t)py * W + px;
                Ray ray = make_ray(&ps, ray_ori + 3 * pix, ray_dir + 3 * pix);
                if (!ray.alive) continue; /* invalid rays keep the caller's initial outputs */
                float T = 1.0f, rgb[3] = {0, 0, 0}, dsum = 0.0f;
                uint32_t nh = 0;
                uint32_t k = beg;
                float m_thr = 3.4028235e+38f, m_trm = 3.4028235e+38f, t_noise = 1.0f; /* decision margins, see oracle_render_margins */
                float pb = 0.0f, pb_hits = 0.0f;   /* pixel flip budget (g_pixel_budget_out) */
                int pb_term = 0;
                for (; k < end && ray.alive; ++k) {
                    const uint32_t id = sorted_ids[k];
                    if (id == INVALID_IDX) break;
                    const float* g = density12 + (size_t)id * 12;
                    float rows[3][3];
                    quat_to_rows(g + 4, rows);
                    Hit h;
                    eval_hit(prm, g, rows, &ray, &h);
                    float nu = 0.0f;
                    if (margins) {
                        /* fp32 noise of the response in eps units: hit_noise() */
                        nu = hit_noise(prm, &h, g + 8);
                        const float eps = 5.9604645e-08f;
                        const float mr = fabsf(h.resp - prm->min_kernel_density) / (prm->min_kernel_density * eps * nu);
                        if (mr < m_thr) m_thr = mr;
                        float me = mr;
                        if (h.resp > prm->min_kernel_density) {
                            const float ma = fabsf(h.resp * g[3] - prm->alpha_threshold) / (prm->alpha_threshold * eps * nu);
                            if (ma < m_thr) m_thr = ma;
                            if (ma < me) me = ma;
                        }
                        if (g_pixel_budget_out && me < g_pixel_budget_bound) { pb += 2.0f * h.alpha * T; pb_hits += 1.0f; }
                    }
                    if ((h.resp > prm->min_kernel_density) && (h.alpha > prm->alpha_threshold)) {
                        const float* s = g + 8;
                        const float proj = h.grd[0] * -h.gro[0] + h.grd[1] * -h.gro[1] + h.grd[2] * -h.gro[2];
                        const float v0 = s[0] * h.grd[0] * proj, v1 = s[1] * h.grd[1] * proj, v2 = s[2] * h.grd[2] * proj;
                        const float hit_t = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
                        if ((hit_t > ray.tmin) && (hit_t < ray.tmax)) {
                            const float w = h.alpha * T;
                            dsum += hit_t * w;
                            T *= (1.0f - h.alpha);
                            if (w > 0.0f) {
                                for (int c = 0; c < 3; ++c) {
                                    const float f = feat[3 * (size_t)id + c];
                                    rgb[c] += (f > 0.0f ? f : 0.0f) * w;
                                }
                                nh++;
                            }
                            if (margins) {
                                /* relative noise of the running transmittance: every factor (1 - alpha) inherits alpha's */
                                t_noise += h.alpha * nu / (1.0f - h.alpha) + 1.0f;
                                const float mt = fabsf(T - prm->min_transmittance) / (prm->min_transmittance * 5.9604645e-08f * t_noise);
                                if (mt < m_trm) m_trm = mt;
                                if (g_pixel_budget_out && mt < g_pixel_budget_bound && !pb_term) { pb_term = 1; pb += 2.0f * T; pb_hits += 64.0f; }
                            }
                            if (T < prm->min_transmittance) ray.alive = 0;
                        }
                    }
                }
                if (k - beg > deepest) deepest = k - beg;
                rgba[4 * pix] = rgb[0]; rgba[4 * pix + 1] = rgb[1]; rgba[4 * pix + 2] = rgb[2];
                rgba[4 * pix + 3] = 1.0f - T;
                dist[pix] = dsum;
                hits[pix] = prm->enable_hitcounts ? (float)nh : 0.0f;   /* rayPayload.cuh:126-128: written only when compiled in; the tensor is created zeroed (splatRaster.cpp:198) */
                if (margins) { margins[2 * pix] = m_thr; margins[2 * pix + 1] = m_trm; }
                if (margins && g_pixel_budget_out) { g_pixel_budget_out[2 * pix] = pb; g_pixel_budget_out[2 * pix + 1] = pb_hits; }
            }
        traversed_total += deepest;
        if (g_tile_traversed_out) g_tile_traversed_out[tile] = deepest;
    }
    if (traversed_out) *traversed_out = traversed_total;
}

/* tests / investigations only: the entries one ray walks, as this evaluation sees them.  out[k] = {id, d2, resp, alpha, nu, accepted,
 * T after the entry, |gro|} for the k-th listed entry of the ray's tile, until the ray ends or `max_entries` are written; returns
 * the number written. */
int oracle_debug_ray(const OracleParams* prm, const OracleCamera* cam, int W, int H, const float* density12,
                     const float* ray_ori, const float* ray_dir, const uint32_t* ranges, const uint32_t* sorted_ids,
                     int px, int py, double* out, int max_entries) {
    const PoseSet ps = make_pose_set(cam);
    const int gx = (W + GUT_TILE - 1) / GUT_TILE;
    const int tile = (py / GUT_TILE) * gx + px / GUT_TILE;
    const size_t pix = (size_t)py * W + px;
    Ray ray = make_ray(&ps, ray_ori + 3 * pix, ray_dir + 3 * pix);
    int n = 0;
    float T = 1.0f;
    for (uint32_t k = ranges[2 * tile]; k < ranges[2 * tile + 1] && ray.alive && n < max_entries; ++k) {
        const uint32_t id = sorted_ids[k];
        if (id == INVALID_IDX) break;
        const float* g = density12 + (size_t)id * 12;
        float rows[3][3];
        quat_to_rows(g + 4, rows);
        Hit h;
        eval_hit(prm, g, rows, &ray, &h);
        const float gn = sqrtf(h.gro[0] * h.gro[0] + h.gro[1] * h.gro[1] + h.gro[2] * h.gro[2]);
        const float nu = hit_noise(prm, &h, g + 8);
        const int acc = (h.resp > prm->min_kernel_density) && (h.alpha > prm->alpha_threshold);
        if (acc) { T *= (1.0f - h.alpha); if (T < prm->min_transmittance) ray.alive = 0; }
        double* o = out + 8 * (size_t)n++;
        o[0] = id; o[1] = h.d2; o[2] = h.resp; o[3] = h.alpha; o[4] = nu; o[5] = acc; o[6] = T; o[7] = gn;
    }
    return n;
}

void oracle_render(const OracleParams* prm, const OracleCamera* cam, int W, int H,
                   const float* density12, const float* feat,
                   const float* ray_ori, const float* ray_dir,
                   const uint32_t* ranges, const uint32_t* sorted_ids,
                   float* rgba, float* dist, float* hits, uint64_t* traversed_out) {
    render_impl(prm, cam, W, H, density12, feat, ray_ori, ray_dir, ranges, sorted_ids, rgba, dist, hits, traversed_out, NULL);
}

/* The same render, additionally exporting per pixel how close any accept/reject decision along the ray came to flipping
 * (tests only: attributes colour differences between two fp32 evaluations of the same formula to threshold flips).
 * margins[2*pix+0]: min over walked entries of |resp - min_response| / min_response and (when resp passed)
 *                   |resp*sigma - min_alpha| / min_alpha, each divided by eps * nu, nu = the entry's fp32 noise estimate
 *                   in eps units (see render_impl) — i.e. "how many noise widths away from a hit/no-hit flip";
 * margins[2*pix+1]: the same for the running transmittance against min_transmittance (early-termination flips).
 * The hit-distance window (tmin, tmax) is not tracked: tmin = 0 and tmax ~ 1e6 for rays inside the +-1e6 scene box. */
void oracle_render_margins(const OracleParams* prm, const OracleCamera* cam, int W, int H,
                           const float* density12, const float* feat, const float* ray_ori, const float* ray_dir,
                           const uint32_t* ranges, const uint32_t* sorted_ids,
                           float* rgba, float* dist, float* hits, float* margins) {
    for (size_t i = 0; i < (size_t)W * H * 2; ++i) margins[i] = 3.4028235e+38f;
    render_impl(prm, cam, W, H, density12, feat, ray_ori, ray_dir, ranges, sorted_ids, rgba, dist, hits, NULL, margins);
}

/* K6, sorted variant (k_buffer_size = K > 0): render with a per-ray K-entry hit buffer kept sorted by hit distance —
 * gutKBufferRenderer.cuh:28-76 (HitParticleKBuffer::insert), :217-292 (evalKBuffer), :108-170 (processHitParticle).
 * Optionally records, per pixel, the particle ids in the order they were composited (for the autograd check of
 * the backward: oracle/per_ray_torch.py: composite_ordered). order_ids is [P, max_order], -1 padded. */
void oracle_render_kbuffer(const OracleParams* prm, const OracleCamera* cam, int W, int H, int K,
                           const float* density12, const float* feat, const float* ray_ori, const float* ray_dir,
                           const uint32_t* ranges, const uint32_t* sorted_ids, float* rgba, float* dist, float* hits,
                           int32_t* order_ids, int32_t* order_count, int max_order) {
    const PoseSet ps = make_pose_set(cam);
    const int gx = (W + GUT_TILE - 1) / GUT_TILE, gy = (H + GUT_TILE - 1) / GUT_TILE;
    if (K < 1) K = 1;
    if (K > 64) K = 64;
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
        for (int py = ty * GUT_TILE; py < (ty + 1) * GUT_TILE && py < H; ++py)
            for (int px = tx * GUT_TILE; px < (tx + 1) * GUT_TILE && px < W; ++px) {
                const size_t pix = (size_t)py * W + px;
                if (order_count) order_count[pix] = 0;
                Ray ray = make_ray(&ps, ray_ori + 3 * pix, ray_dir + 3 * pix);
                if (!ray.alive) continue;
                float T = 1.0f, rgb[3] = {0, 0, 0}, dsum = 0.0f;
                uint32_t nh = 0;
                float kb_t[64], kb_a[64]; uint32_t kb_i[64]; int num = 0;
                for (int i = 0; i < K; ++i) { kb_t[i] = -1.0f; kb_a[i] = 0.0f; kb_i[i] = INVALID_IDX; }
#define ORACLE_PROCESS(IDX, ALPHA, HITT)                                                    \
    do {                                                                                     \
        const float w_ = (ALPHA) * T;                                                        \
        dsum += (HITT) * w_;                                                                 \
        T *= (1.0f - (ALPHA));                                                               \
        if (w_ > 0.0f) {                                                                     \
            for (int c_ = 0; c_ < 3; ++c_) {                                                 \
                const float f_ = feat[3 * (size_t)(IDX) + c_];                               \
                rgb[c_] += (f_ > 0.0f ? f_ : 0.0f) * w_;                                     \
            }                                                                                \
            nh++;                                                                            \
        }                                                                                    \
        if (order_ids && order_count[pix] < max_order) order_ids[pix * (size_t)max_order + order_count[pix]] = (int32_t)(IDX); \
        if (order_count) order_count[pix]++;                                                 \
        if (T < prm->min_transmittance) ray.alive = 0;                                       \
    } while (0)
                for (uint32_t k = beg; k < end && ray.alive; ++k) {
                    const uint32_t id = sorted_ids[k];
                    if (id == INVALID_IDX) break;
                    const float* g = density12 + (size_t)id * 12;
                    float rows[3][3];
                    quat_to_rows(g + 4, rows);
                    Hit h;
                    eval_hit(prm, g, rows, &ray, &h);
                    if (!((h.resp > prm->min_kernel_density) && (h.alpha > prm->alpha_threshold))) continue;
                    const float* s = g + 8;
                    const float proj = h.grd[0] * -h.gro[0] + h.grd[1] * -h.gro[1] + h.grd[2] * -h.gro[2];
                    const float v0 = s[0] * h.grd[0] * proj, v1 = s[1] * h.grd[1] * proj, v2 = s[2] * h.grd[2] * proj;
                    float hit_t = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
                    if (!((hit_t > ray.tmin) && (hit_t < ray.tmax))) continue;
                    if (num == K) {  /* full: composite the closest stored hit, then free its slot */
                        ORACLE_PROCESS(kb_i[0], kb_a[0], kb_t[0]);
                        kb_t[0] = -1.0f;
                    } else {
                        num++;
                    }
                    /* insert (bubble towards the back while farther than the stored entries) */
                    float ct = hit_t, ca = h.alpha; uint32_t ci = id;
                    for (int i = K - 1; i >= 0; --i)
                        if (ct > kb_t[i]) {
                            const float tt = kb_t[i], ta = kb_a[i]; const uint32_t ti = kb_i[i];
                            kb_t[i] = ct; kb_a[i] = ca; kb_i[i] = ci;
                            ct = tt; ca = ta; ci = ti;
                        }
                }
                for (int i = 0; ray.alive && i < num; ++i) {
                    const int sl = K - num + i;
                    ORACLE_PROCESS(kb_i[sl], kb_a[sl], kb_t[sl]);
                }
#undef ORACLE_PROCESS
                rgba[4 * pix] = rgb[0]; rgba[4 * pix + 1] = rgb[1]; rgba[4 * pix + 2] = rgb[2];
                rgba[4 * pix + 3] = 1.0f - T;
                dist[pix] = dsum;
                hits[pix] = prm->enable_hitcounts ? (float)nh : 0.0f;   /* rayPayload.cuh:126-128: written only when compiled in; the tensor is created zeroed (splatRaster.cpp:198) */
            }
    }
}

/* matmul_bw_quat — common/mathUtils.cuh:468-533 */
static void matmul_bw_quat(const float p[3], const float g[3], const float q[4], float out[4]) {
    float dm[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dm[i][j] = g[i] * p[j];
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    float dr = 0, dx = 0, dy = 0, dz = 0;
    dy += -4 * y * dm[0][0]; dz += -4 * z * dm[0][0];
    dr += 2 * z * dm[0][1]; dx += 2 * y * dm[0][1]; dy += 2 * x * dm[0][1]; dz += 2 * r * dm[0][1];
    dr += -2 * y * dm[0][2]; dx += 2 * z * dm[0][2]; dy += -2 * r * dm[0][2]; dz += 2 * x * dm[0][2];
    dr += -2 * z * dm[1][0]; dx += 2 * y * dm[1][0]; dy += 2 * x * dm[1][0]; dz += -2 * r * dm[1][0];
    dx += -4 * x * dm[1][1]; dz += -4 * z * dm[1][1];
    dr += 2 * x * dm[1][2]; dx += 2 * r * dm[1][2]; dy += 2 * z * dm[1][2]; dz += 2 * y * dm[1][2];
    dr += 2 * y * dm[2][0]; dx += 2 * z * dm[2][0]; dy += 2 * r * dm[2][0]; dz += 2 * x * dm[2][0];
    dr += -2 * x * dm[2][1]; dx += -2 * r * dm[2][1]; dy += 2 * z * dm[2][1]; dz += 2 * y * dm[2][1];
    dx += -4 * x * dm[2][2]; dy += -4 * y * dm[2][2];
    out[0] = dr; out[1] = dx; out[2] = dy; out[3] = dz;
}

/* K7: renderBackward — gutKBufferRenderer.cuh:294-386, models/gaussianParticles.cuh:480-738,
 * rayPayloadBackward.cuh:30-58.  Per-pair math in fp32 exactly as the reference; the cross-pixel
 * sums (float atomics in the reference, order undefined) are taken in double here.
 * Quirk 1: integratedDepth == 0 so residualHitT == 0.
 * density_grad [N,12] (pos3, density, quat4 wxyz, scale3, pad) and feat_grad [N,3], both doubles.
 *
 * flip_budget (tests only, may be NULL): [N,10] doubles; columns 0..4 (positions, density, rotation, scale, colour) receive, per Gaussian,
 * how far a DIFFERENT BUT EQUALLY VALID fp32 evaluation of the same formulas may move that row of the gradient because a
 * hit / no-hit decision flips: the reference's per-hit gradient is discontinuous at the min_response and min_alpha
 * thresholds (a hit with alpha ~ 1/255 on a faint Gaussian has d alpha / d sigma = response ~ 1: the density row jumps).
 * A decision counts as flip-prone when its margin (the same noise model as oracle_render_margins) is below flip_bound:
 *   - the prone entry's own row may gain or lose its whole contribution from that pixel  -> += |contribution| (evaluated
 *     as if accepted, whether this evaluation accepted it or not);
 *   - every other entry the ray walks sees T, the final T and the final colour change by the flipped hit's alpha (<= 0.0113,
 *     or 1/255 / (1 - 1/255)): its contribution moves by that share -> += 4 x sum(alpha_f / (1 - alpha_f)) x |contribution|.
 * Rows that get no budget have no flip-prone decision on any ray that touches them.
 * Columns 5..9 of the [N,10] array: the fp32 conditioning of each row, sum over its hits of eps x nu x |contribution| with
 * nu the response's noise estimate of oracle_render_margins (the response is exp(-|grd x gro|^2 / 2) with |gro| = the
 * camera distance in units of the Gaussian's size: a relative error of eps x nu, up to 1e-3 for a small, distant Gaussian,
 * which every term of the hit's gradient inherits; the geometric rows additionally 8 eps |gro|, see below) — what ANY fp32
 * evaluation order may differ by on that row. */
static void render_bwd_impl(const OracleParams* prm, const OracleCamera* cam, int W, int H,
                       const float* density12, const float* feat,
                       const float* ray_ori, const float* ray_dir,
                       const uint32_t* ranges, const uint32_t* sorted_ids,
                       const float* rgba, const float* rgba_grad, const float* dist, const float* dist_grad,
                       double* density_grad, double* feat_grad, uint64_t* traversed_out, double* flip_budget, float flip_bound) {
    /* The unsorted backward recomputes alpha as fminf(0.99f, gres * density) with the LITERAL 0.99 (processHitBwd,
     * gaussianParticles.cuh:528), whatever render.particle_kernel_max_alpha is — the forward (any K) goes through the slang
     * particleDensityHit, which clamps with GAUSSIAN_PARTICLE_MAX_ALPHA (gaussianParticles.slang:211, threedgut.slang:20).  The two agree
     * under render/3dgut.yaml (0.99); with another max alpha the reference's backward walks a transmittance chain its forward did not,
     * and so does this restatement. */
    OracleParams bwd_prm = *prm;
    bwd_prm.max_alpha = 0.99f;
    prm = &bwd_prm;
    const PoseSet ps = make_pose_set(cam);
    const int gx = (W + GUT_TILE - 1) / GUT_TILE, gy = (H + GUT_TILE - 1) / GUT_TILE;
    uint64_t traversed_total = 0;
    (void)dist;
    /* tiles in parallel: the per-Gaussian double accumulators are shared, hence the atomic adds (their order, and with it the
     * last bits of the double sums, depends on the thread schedule; every consumer compares with a tolerance) */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : traversed_total)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
        uint32_t deepest = 0;
        for (int py = ty * GUT_TILE; py < (ty + 1) * GUT_TILE && py < H; ++py)
            for (int px = tx * GUT_TILE; px < (tx + 1) * GUT_TILE && px < W; ++px) {
                const size_t pix = (size_t)py * W + px;
                Ray ray = make_ray(&ps, ray_ori + 3 * pix, ray_dir + 3 * pix);
                if (!ray.alive) continue;
                const float T_final = 1.0f - rgba[4 * pix + 3];
                const float T_grad = -1.0f * rgba_grad[4 * pix + 3];
                const float rgb_final[3] = {rgba[4 * pix], rgba[4 * pix + 1], rgba[4 * pix + 2]};
                const float rgb_g[3] = {rgba_grad[4 * pix], rgba_grad[4 * pix + 1], rgba_grad[4 * pix + 2]};
                const float depth_g = dist_grad[pix];
                /* flip budget, first pass over the ray: the summed alpha share of its flip-prone decisions */
                float taint = 0.0f;
                /* (round 4) ... and what the first pass learns about the ray as a whole:
                 *  - chain noise: every later quantity of the ray inherits the fp32 noise of the alphas in front of it.  tn = the
                 *    relative noise of the running transmittance in eps units (as t_noise below); nrgb_total[c] = the absolute
                 *    noise, in eps units, of the final colour, sum_j w_j f_jc (nu_j + tn_j); tn_final that of the final T.
                 *  - termination flip: if the running transmittance comes within flip_bound noise widths of min_transmittance
                 *    the ray may end there or walk on in another evaluation, and then EVERY entry in front of that point sees
                 *    different finals (the reference's residual form feeds T_final and rgb_final back into every hit's
                 *    gradient): d_T / d_rgb = |state at the prone point - state where the extended walk ends|. */
                double tn_final = 1.0, nrgb_total[3] = {0, 0, 0}, term_dT = 0.0, term_drgb[3] = {0, 0, 0};
                if (flip_budget) {
                    float Tp = 1.0f, rgbp[3] = {0, 0, 0}, T_mark = 0.0f, rgb_mark[3] = {0, 0, 0};
                    double tn = 1.0;
                    int marked = 0, dead = 0;
                    for (uint32_t kk = beg; kk < end; ++kk) {
                        const uint32_t id = sorted_ids[kk];
                        if (id == INVALID_IDX) break;
                        const float* g = density12 + (size_t)id * 12;
                        float rows[3][3];
                        quat_to_rows(g + 4, rows);
                        Hit h;
                        eval_hit(prm, g, rows, &ray, &h);
                        const float nu = hit_noise(prm, &h, g + 8), eps = 5.9604645e-08f;
                        int prone = fabsf(h.resp - prm->min_kernel_density) / (prm->min_kernel_density * eps * nu) < flip_bound;
                        if (h.resp > prm->min_kernel_density)
                            prone |= fabsf(h.resp * g[3] - prm->alpha_threshold) / (prm->alpha_threshold * eps * nu) < flip_bound;
                        if (prone && !dead) taint += h.alpha / (1.0f - h.alpha);
                        if ((h.resp > prm->min_kernel_density) && (h.alpha > prm->alpha_threshold)) {
                            const float wp = h.alpha * Tp;
                            for (int c = 0; c < 3; ++c) {
                                const float fv = feat[3 * (size_t)id + c];
                                const float fc = fv > 0.0f ? fv : 0.0f;
                                rgbp[c] += wp * fc;
                                if (!dead) nrgb_total[c] += (double)wp * fc * ((double)nu + tn);
                            }
                            Tp *= (1.0f - h.alpha);
                            tn += (double)h.alpha * nu / (1.0 - (double)h.alpha) + 1.0;
                            if (!dead) tn_final = tn;
                            if (!marked && fabs((double)Tp - prm->min_transmittance) / (prm->min_transmittance * 5.9604645e-08 * tn) < flip_bound) {
                                marked = 1; T_mark = Tp;
                                for (int c = 0; c < 3; ++c) rgb_mark[c] = rgbp[c];
                            }
                            if (Tp < prm->min_transmittance) {
                                if (!marked) break;
                                dead = 1;                                   /* walk on for the alternative ending */
                                if (Tp < 0.25f * prm->min_transmittance) break;
                            }
                        }
                    }
                    if (marked) {
                        term_dT = fabs((double)T_mark - (double)Tp);
                        for (int c = 0; c < 3; ++c) term_drgb[c] = fabs((double)rgb_mark[c] - (double)rgbp[c]);
                    }
                }
                double nrgb_run[3] = {0, 0, 0}, tn_run = 1.0;   /* the chain noise up to and including the current entry */
                float T = 1.0f, rgb_run[3] = {0, 0, 0};
                /* flip budget: once the running transmittance has come within flip_bound noise widths of min_transmittance the
                 * ray may end one entry earlier or later in another evaluation: from there on every entry's contribution is
                 * budgeted whole, and a ray that ended there is walked on ("ghost") for the budget alone until T < Tmin / 4 */
                float t_noise = 1.0f;
                int term_prone = 0, ghost = 0;
                uint32_t k = beg, k_end = 0;
                for (; k < end && (ray.alive || ghost); ++k) {
                    const uint32_t id = sorted_ids[k];
                    if (id == INVALID_IDX) break;
                    const float* g = density12 + (size_t)id * 12;
                    const float* q = g + 4; const float* s = g + 8; const float sigma = g[3];
                    float rows[3][3];
                    quat_to_rows(q, rows);
                    Hit h;
                    eval_hit(prm, g, rows, &ray, &h);
                    const int accept = (h.resp > prm->min_kernel_density) && (h.alpha > prm->alpha_threshold);
                    int prone = 0;
                    float nu = 0.0f, gn = 0.0f;
                    if (flip_budget) {
                        gn = sqrtf(h.gro[0] * h.gro[0] + h.gro[1] * h.gro[1] + h.gro[2] * h.gro[2]);
                        const float eps = 5.9604645e-08f;
                        nu = hit_noise(prm, &h, g + 8);
                        prone = fabsf(h.resp - prm->min_kernel_density) / (prm->min_kernel_density * eps * nu) < flip_bound;
                        if (h.resp > prm->min_kernel_density)
                            prone |= fabsf(h.resp * g[3] - prm->alpha_threshold) / (prm->alpha_threshold * eps * nu) < flip_bound;
                    }
                    if (!accept && !prone) continue;
                    /* NB: no tmin/tmax test in the backward (gaussianParticles.cuh:530) */
                    const float proj = h.grd[0] * -h.gro[0] + h.grd[1] * -h.gro[1] + h.grd[2] * -h.gro[2];
                    const float grdd[3] = {h.grd[0] * proj, h.grd[1] * proj, h.grd[2] * proj};
                    const float grds[3] = {s[0] * grdd[0], s[1] * grdd[1], s[2] * grdd[2]};
                    const float gsq = grds[0] * grds[0] + grds[1] * grds[1] + grds[2] * grds[2];
                    const float gdist = sqrtf(gsq);
                    const float w = h.alpha * T;
                    const float Tn = (1.0f - h.alpha) * T;
                    const float res_hit = 0.0f; /* quirk 1 */
                    const float ga_hit = (gdist - res_hit) * T * depth_g;
                    float g_grds[3] = {0, 0, 0};
                    if (gsq > 0.0f) for (int c = 0; c < 3; ++c) g_grds[c] = ((2.0f * grds[c] * w) / (2.0f * gdist)) * depth_g;
                    const float gs_hit[3] = {grdd[0] * g_grds[0], grdd[1] * g_grds[1], grdd[2] * g_grds[2]};
                    const float gx0 = h.grd[0] * h.gro[0], gy0 = h.grd[1] * h.gro[1], gz0 = h.grd[2] * h.gro[2];
                    const float ggrd_hit[3] = {-s[0] * (2 * gx0 + gy0 + gz0) * g_grds[0],
                                               -s[1] * (gx0 + 2 * gy0 + gz0) * g_grds[1],
                                               -s[2] * (gx0 + gy0 + 2 * gz0) * g_grds[2]};
                    const float ggro_hit[3] = {-s[0] * h.grd[0] * h.grd[0] * g_grds[0],
                                               -s[1] * h.grd[1] * h.grd[1] * g_grds[1],
                                               -s[2] * h.grd[2] * h.grd[2] * g_grds[2]};
                    const float res_T = h.alpha < 0.999999f ? T_final / (1.0f - h.alpha) : T;
                    const float ga_dns = res_T * -T_grad;
                    float f[3], res_rad[3], run_after[3], feat_add[3];
                    for (int c = 0; c < 3; ++c) {
                        const float fv = feat[3 * (size_t)id + c];
                        f[c] = fv > 0.0f ? fv : 0.0f;
                        feat_add[c] = rgb_g[c] * w;
                        run_after[c] = rgb_run[c] + w * f[c];
                        const float rr = (Tn <= prm->min_transmittance) ? 0.0f : (rgb_final[c] - run_after[c]) / Tn;
                        res_rad[c] = rr > 0.0f ? rr : 0.0f;
                    }
                    const float G = ga_hit + ga_dns + T * (f[0] - res_rad[0]) * rgb_g[0] + T * (f[1] - res_rad[1]) * rgb_g[1] +
                                    T * (f[2] - res_rad[2]) * rgb_g[2];
                    const float d_sigma = h.resp * G;
                    const float g_resp = sigma * G;
                    const float g_d2 = kernel_response_grad(prm->kernel_degree, h.d2, h.resp, g_resp);
                    const float cr[3] = {h.grd[1] * h.gro[2] - h.grd[2] * h.gro[1], h.grd[2] * h.gro[0] - h.grd[0] * h.gro[2],
                                         h.grd[0] * h.gro[1] - h.grd[1] * h.gro[0]};
                    const float gc[3] = {2 * cr[0] * g_d2, 2 * cr[1] * g_d2, 2 * cr[2] * g_d2};
                    const float g_grd[3] = {gc[2] * h.gro[1] - gc[1] * h.gro[2], gc[0] * h.gro[2] - gc[2] * h.gro[0],
                                            gc[1] * h.gro[0] - gc[0] * h.gro[1]};
                    const float g_gro[3] = {gc[1] * h.grd[2] - gc[2] * h.grd[1], gc[2] * h.grd[0] - gc[0] * h.grd[2],
                                            gc[0] * h.grd[1] - gc[1] * h.grd[0]};
                    float gs_gro[3], g_gposcr[3];
                    for (int c = 0; c < 3; ++c) {
                        gs_gro[c] = (-h.gposcr[c] / (s[c] * s[c])) * (g_gro[c] + ggro_hit[c]);
                        g_gposcr[c] = (1.0f / s[c]) * (g_gro[c] + ggro_hit[c]);
                    }
                    float g_gposc[3];
                    for (int c = 0; c < 3; ++c) g_gposc[c] = g_gposcr[0] * rows[0][c] + g_gposcr[1] * rows[1][c] + g_gposcr[2] * rows[2][c];
                    float q_a[4];
                    matmul_bw_quat(h.gposc, g_gposcr, q, q_a);
                    /* safe_normalize_bw(grdu, g_grd + ggrd_hit), mathUtils.cuh:420-430 */
                    const float gin[3] = {g_grd[0] + ggrd_hit[0], g_grd[1] + ggrd_hit[1], g_grd[2] + ggrd_hit[2]};
                    float g_grdu[3] = {0, 0, 0};
                    {
                        const float* v = h.grdu;
                        const float l = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
                        if (l > 0.0f) {
                            const float il = 1.0f / sqrtf(l);
                            const float il3 = il * il * il;
                            g_grdu[0] = il * gin[0] - il3 * (gin[0] * (v[0] * v[0]) + gin[1] * (v[1] * v[0]) + gin[2] * (v[2] * v[0]));
                            g_grdu[1] = il * gin[1] - il3 * (gin[0] * (v[0] * v[1]) + gin[1] * (v[1] * v[1]) + gin[2] * (v[2] * v[1]));
                            g_grdu[2] = il * gin[2] - il3 * (gin[0] * (v[0] * v[2]) + gin[1] * (v[1] * v[2]) + gin[2] * (v[2] * v[2]));
                        }
                    }
                    float g_rdr[3], d_scale[3];
                    for (int c = 0; c < 3; ++c) {
                        d_scale[c] = gs_hit[c] + gs_gro[c] + (-h.rdr[c] / (s[c] * s[c])) * g_grdu[c];
                        g_rdr[c] = (1.0f / s[c]) * g_grdu[c];
                    }
                    float q_b[4];
                    matmul_bw_quat(ray.d, g_rdr, q, q_b);
                    const double add[11] = {-g_gposc[0], -g_gposc[1], -g_gposc[2], d_sigma, q_a[0] + q_b[0], q_a[1] + q_b[1],
                                            q_a[2] + q_b[2], q_a[3] + q_b[3], d_scale[0], d_scale[1], d_scale[2]};
                    if (flip_budget) {
                        /* every geometric term of the hit's gradient is proportional to G, a SUM of terms of either sign (final
                         * transmittance, colour residuals): what moves G is measured against the sum of their magnitudes */
                        const double G_abs = fabs((double)ga_hit) + fabs((double)ga_dns) +
                                             (double)T * ((fabs((double)f[0]) + res_rad[0]) * fabs((double)rgb_g[0]) +
                                                          (fabs((double)f[1]) + res_rad[1]) * fabs((double)rgb_g[1]) +
                                                          (fabs((double)f[2]) + res_rad[2]) * fabs((double)rgb_g[2]));
                        double cond = G_abs / fmax(fabs((double)G), 1e-30 + 1e-3 * G_abs);   /* >= 1, capped at 1e3 */
                        if (!(cond >= 1.0)) cond = 1.0;
                        if (term_prone) prone = 1;
                        const double share = prone ? 1.0 : 4.0 * (double)taint;
                        const double noise = (accept && !ghost) ? 5.9604645e-08 * ((double)nu + 8.0 * cond) : 0.0;
                        /* (round 4, found with the two-evaluation experiment of tests/test_cpu_oracle.py) the reference's residual
                         * form carries an ABSOLUTE fp32 error that does not shrink with the transmittance: the final transmittance
                         * is recovered as 1 - alpha_out (alpha_out is rounded near 1: eps absolute, i.e. eps / T_final relative once
                         * a ray is opaque) and the colour behind a hit as (rgb_final - rgb_run) / T' (two O(1) numbers rounded to
                         * eps, divided by a T' down to 1e-4).  G moves by eps x R, R = (|dL/dalpha| + sum_c (|rgb_final_c| +
                         * |rgb_run_c|) |dL/drgb_c|) / (1 - alpha), whatever T is, while the hit's contribution is proportional to
                         * T: deep entries of opaque rays (T ~ 1e-3 .. 1e-4) differ between two fp32 evaluations by up to a few
                         * per cent of their (tiny) rows.  Every geometric row is linear in G: the row moves by |row| / |G| x eps R. */
                        const double R_abs = ((double)fabsf(T_grad) + ((double)fabsf(rgb_final[0]) + fabs((double)run_after[0])) * fabs((double)rgb_g[0]) +
                                              ((double)fabsf(rgb_final[1]) + fabs((double)run_after[1])) * fabs((double)rgb_g[1]) +
                                              ((double)fabsf(rgb_final[2]) + fabs((double)run_after[2])) * fabs((double)rgb_g[2])) / (1.0 - (double)h.alpha);
                        double noise_abs = (accept && !ghost && G != 0.0f) ? 5.9604645e-08 * 2.0 * R_abs / fabs((double)G) : 0.0;
                        double flip_term = 0.0;
                        if (accept && !ghost && G != 0.0f) {
                            /* chain noise of this hit's G, in eps units: its T (tn_run before this hit), the final T behind the
                             * density term, the colour still to come behind the hit */
                            double dG = fabs((double)ga_hit) * tn_run + fabs((double)ga_dns) * tn_final;
                            for (int c = 0; c < 3; ++c) {
                                const double later = nrgb_total[c] - nrgb_run[c] - (double)w * f[c] * ((double)nu + tn_run);
                                dG += fabs((double)rgb_g[c]) * ((double)T * f[c] * tn_run + (later > 0.0 ? later : 0.0) / (1.0 - (double)h.alpha));
                            }
                            noise_abs += 5.9604645e-08 * dG / fabs((double)G);
                            /* termination flip: the finals this hit's residuals are built from may be those of the other ending */
                            const double dGt = ((double)fabsf(T_grad) * term_dT + fabs((double)rgb_g[0]) * term_drgb[0] +
                                                fabs((double)rgb_g[1]) * term_drgb[1] + fabs((double)rgb_g[2]) * term_drgb[2]) / (1.0 - (double)h.alpha);
                            flip_term = dGt / fabs((double)G);
                        }
                        const double nb[5] = {sqrt(add[0] * add[0] + add[1] * add[1] + add[2] * add[2]), fabs(add[3]),
                                              sqrt(add[4] * add[4] + add[5] * add[5] + add[6] * add[6] + add[7] * add[7]),
                                              sqrt(add[8] * add[8] + add[9] * add[9] + add[10] * add[10]),
                                              sqrt((double)feat_add[0] * feat_add[0] + (double)feat_add[1] * feat_add[1] +
                                                   (double)feat_add[2] * feat_add[2])};
                        for (int c = 0; c < 5; ++c) {
                            if (share > 0.0) {
                                /* a flipped hit elsewhere on the ray moves the terms of G by its alpha share (colour block: w only) */
                                const double sh = (prone || c == 4) ? share : share * cond;
#pragma omp atomic
                                flip_budget[10 * (size_t)id + c] += sh * nb[c];
                            }
                            if (flip_term > 0.0 && c != 4 && !prone) {
#pragma omp atomic
                                flip_budget[10 * (size_t)id + c] += flip_term * nb[c];
                            }
                            if (noise > 0.0) {
                                /* positions / rotation / scale: the ray's offset from the Gaussian in units of its size, gro, of
                                 * magnitude gn (1e4 for a sub-pixel Gaussian a few units away), enters through its component
                                 * perpendicular to the ray, of magnitude d ~ 1: a cancellation that leaves eps x gn there */
                                double nz_ = (c == 0 || c == 2 || c == 3) ? noise + 5.9604645e-08 * 8.0 * (double)gn : noise;
                                if (c != 4) nz_ += noise_abs;   /* the colour row w x dL/drgb does not go through G */
#pragma omp atomic
                                flip_budget[10 * (size_t)id + 5 + c] += nz_ * nb[c];
                            }
                        }
                    }
                    if (!accept) continue;
                    if (!ghost) {
                        double* dg = density_grad + 12 * (size_t)id;
                        for (int c = 0; c < 11; ++c) {
#pragma omp atomic
                            dg[c] += add[c];
                        }
                        for (int c = 0; c < 3; ++c) {
#pragma omp atomic
                            feat_grad[3 * (size_t)id + c] += (double)feat_add[c];
                        }
                    }
                    for (int c = 0; c < 3; ++c) rgb_run[c] = run_after[c];
                    if (flip_budget) {
                        for (int c = 0; c < 3; ++c) nrgb_run[c] += (double)w * f[c] * ((double)nu + tn_run);
                        tn_run += (double)h.alpha * nu / (1.0 - (double)h.alpha) + 1.0;
                    }
                    T = Tn;
                    if (flip_budget) {
                        t_noise += h.alpha * nu / (1.0f - h.alpha) + 1.0f;   /* as in render_impl */
                        if (fabsf(T - prm->min_transmittance) / (prm->min_transmittance * 5.9604645e-08f * t_noise) < flip_bound) term_prone = 1;
                    }
                    if (T < prm->min_transmittance && ray.alive) {
                        ray.alive = 0;
                        k_end = k + 1;
                        ghost = flip_budget && term_prone;
                    }
                    if (ghost && T < 0.25f * prm->min_transmittance) ghost = 0;
                }
                if (ray.alive) k_end = k;
                if (k_end == 0) k_end = k;   /* (invalid id / end of list reached while alive) */
                if (k_end - beg > deepest) deepest = k_end - beg;
            }
        traversed_total += deepest;
        if (g_tile_traversed_out) g_tile_traversed_out[tile] = deepest;
    }
    if (traversed_out) *traversed_out = traversed_total;
}

void oracle_render_bwd(const OracleParams* prm, const OracleCamera* cam, int W, int H,
                       const float* density12, const float* feat,
                       const float* ray_ori, const float* ray_dir,
                       const uint32_t* ranges, const uint32_t* sorted_ids,
                       const float* rgba, const float* rgba_grad, const float* dist, const float* dist_grad,
                       double* density_grad, double* feat_grad, uint64_t* traversed_out) {
    render_bwd_impl(prm, cam, W, H, density12, feat, ray_ori, ray_dir, ranges, sorted_ids, rgba, rgba_grad, dist, dist_grad,
                    density_grad, feat_grad, traversed_out, NULL, 0.0f);
}

/* the same backward, additionally exporting the per-Gaussian flip budget + fp32 conditioning [N,10] (see render_bwd_impl) — tests only */
void oracle_render_bwd_budget(const OracleParams* prm, const OracleCamera* cam, int W, int H,
                              const float* density12, const float* feat,
                              const float* ray_ori, const float* ray_dir,
                              const uint32_t* ranges, const uint32_t* sorted_ids,
                              const float* rgba, const float* rgba_grad, const float* dist, const float* dist_grad,
                              double* density_grad, double* feat_grad, uint64_t* traversed_out, double* flip_budget, float flip_bound) {
    render_bwd_impl(prm, cam, W, H, density12, feat, ray_ori, ray_dir, ranges, sorted_ids, rgba, rgba_grad, dist, dist_grad,
                    density_grad, feat_grad, traversed_out, flip_budget, flip_bound);
}

/* K8: projectBackward — gutProjector.cuh:390-430, gaussianParticles.cuh:120-187.
 * sph_grad [N,48] doubles, overwritten (zero for Gaussians without tiles / above the active degree). */
void oracle_project_bwd(const OracleCamera* cam, uint32_t N, int sh_degree,
                        const float* density12, const uint32_t* tiles_count,
                        const float* feat, const double* feat_grad, double* sph_grad) {
    const PoseSet ps = make_pose_set(cam);
    memset(sph_grad, 0, sizeof(double) * 48 * (size_t)N);
    for (uint32_t i = 0; i < N; ++i) {
        if (tiles_count[i] == 0) continue;
        const float* g = density12 + (size_t)i * 12;
        const float sr[3] = {g[0] - ps.cam[0], g[1] - ps.cam[1], g[2] - ps.cam[2]};
        const float l = sqrtf(sr[0] * sr[0] + sr[1] * sr[1] + sr[2] * sr[2]);
        const float inv_l = 1.0f / l;
        const float dir[3] = {sr[0] * inv_l, sr[1] * inv_l, sr[2] * inv_l};
        float Y[16];
        sh_basis(sh_degree, dir, Y);
        const int ncoef = (sh_degree + 1) * (sh_degree + 1);
        for (int ch = 0; ch < 3; ++ch) {
            const double gmask = feat[3 * (size_t)i + ch] > 0.0f ? feat_grad[3 * (size_t)i + ch] : 0.0;
            for (int k = 0; k < ncoef; ++k) sph_grad[48 * (size_t)i + 3 * k + ch] = (double)Y[k] * gmask;
        }
    }
}

/* exported scalar probes for known-answer tests */
float oracle_det_logf(float x) { return det_logf(x); }
float oracle_det_atan2f(float y, float x) { return det_atan2f_pos(y, x); }
float oracle_tile_min_power(float tx, float ty, const float* conic4, const float* mean2) { return tile_min_power(tx, ty, conic4, mean2); }

/* Layout study (tools only, not used by any test): for the traversal oracle_render performs, count the (wave, list entry) pairs
 * in which at least one of the wave's 64 pixels is hit, for two assignments of a 16x16 tile's pixels to four waves:
 * out[0] = 16x4 strips (wave = py_in_tile / 4), out[1] = 8x8 blocks (wave = (py_in_tile / 8) * 2 + px_in_tile / 8),
 * out[2] = hit (pixel, entry) pairs, out[3] = (tile, entry) pairs walked while any pixel of the tile was alive,
 * out[4] / out[5] = (wave, entry) pairs in which the wave still had an alive ray (what it must at least look at without any
 * per-wave culling), strips / blocks;  out[6] / out[7] = 8x8-block pairs whose particle's screen rectangle (proj_pos +- extent,
 * optional) overlaps the block's pixel rectangle / the bounding rectangle of the block's ALIVE pixels — screen-space proxies for
 * the static wedge test of the kernels and for one rebuilt from the alive rays of each wave. */
void oracle_count_wave_pairs(const OracleParams* prm, const OracleCamera* cam, int W, int H, const float* density12,
                             const float* ray_ori, const float* ray_dir, const uint32_t* ranges, const uint32_t* sorted_ids,
                             uint64_t* out, const float* proj_pos, const float* extent) {
    const PoseSet ps = make_pose_set(cam);
    const int gx = (W + GUT_TILE - 1) / GUT_TILE, gy = (H + GUT_TILE - 1) / GUT_TILE;
    uint64_t c_strip = 0, c_block = 0, c_hits = 0, c_walk = 0, a_strip = 0, a_block = 0, r_static = 0, r_alive = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : c_strip, c_block, c_hits, c_walk, a_strip, a_block, r_static, r_alive)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
        Ray rays[256]; float T[256]; int alive[256];
        for (int p = 0; p < 256; ++p) {
            const int px = tx * GUT_TILE + (p & 15), py = ty * GUT_TILE + (p >> 4);
            alive[p] = 0; T[p] = 1.0f;
            if (px < W && py < H) {
                const size_t pix = (size_t)py * W + px;
                rays[p] = make_ray(&ps, ray_ori + 3 * pix, ray_dir + 3 * pix);
                alive[p] = rays[p].alive;
            }
        }
        for (uint32_t k = beg; k < end; ++k) {
            int any = 0;
            for (int p = 0; p < 256; ++p) any |= alive[p];
            if (!any) break;
            const uint32_t id = sorted_ids[k];
            if (id == INVALID_IDX) break;
            c_walk++;
            const float* g = density12 + (size_t)id * 12;
            float rows[3][3];
            quat_to_rows(g + 4, rows);
            int strip[4] = {0, 0, 0, 0}, block[4] = {0, 0, 0, 0}, as[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};
            int lo_x[4] = {99, 99, 99, 99}, hi_x[4] = {-1, -1, -1, -1}, lo_y[4] = {99, 99, 99, 99}, hi_y[4] = {-1, -1, -1, -1};
            for (int p = 0; p < 256; ++p) {
                if (!alive[p]) continue;
                as[(p >> 4) >> 2] = 1;
                const int wb = ((p >> 4) >> 3) * 2 + ((p & 15) >> 3);
                ab[wb] = 1;
                if ((p & 15) < lo_x[wb]) lo_x[wb] = p & 15;
                if ((p & 15) > hi_x[wb]) hi_x[wb] = p & 15;
                if ((p >> 4) < lo_y[wb]) lo_y[wb] = p >> 4;
                if ((p >> 4) > hi_y[wb]) hi_y[wb] = p >> 4;
                Hit h;
                eval_hit(prm, g, rows, &rays[p], &h);
                if ((h.resp > prm->min_kernel_density) && (h.alpha > prm->alpha_threshold)) {
                    c_hits++;
                    strip[(p >> 4) >> 2] = 1;
                    block[((p >> 4) >> 3) * 2 + ((p & 15) >> 3)] = 1;
                    T[p] *= (1.0f - h.alpha);
                    if (T[p] < prm->min_transmittance) alive[p] = 0;
                }
            }
            for (int w = 0; w < 4; ++w) { c_strip += strip[w]; c_block += block[w]; a_strip += as[w]; a_block += ab[w]; }
            if (proj_pos && extent) {
                const float gx0 = proj_pos[2 * id] - extent[2 * id], gx1 = proj_pos[2 * id] + extent[2 * id];
                const float gy0 = proj_pos[2 * id + 1] - extent[2 * id + 1], gy1 = proj_pos[2 * id + 1] + extent[2 * id + 1];
                for (int w = 0; w < 4; ++w) {
                    if (!ab[w]) continue;
                    const float bx0 = tx * GUT_TILE + (w & 1) * 8, by0 = ty * GUT_TILE + (w >> 1) * 8;
                    if (gx1 >= bx0 && gx0 <= bx0 + 8.0f && gy1 >= by0 && gy0 <= by0 + 8.0f) r_static++;
                    const float ax0 = tx * GUT_TILE + lo_x[w], ax1 = tx * GUT_TILE + hi_x[w] + 1.0f, ay0 = ty * GUT_TILE + lo_y[w], ay1 = ty * GUT_TILE + hi_y[w] + 1.0f;
                    if (gx1 >= ax0 && gx0 <= ax1 && gy1 >= ay0 && gy0 <= ay1) r_alive++;
                }
            }
        }
    }
    out[0] = c_strip; out[1] = c_block; out[2] = c_hits; out[3] = c_walk; out[4] = a_strip; out[5] = a_block; out[6] = r_static; out[7] = r_alive;
}
