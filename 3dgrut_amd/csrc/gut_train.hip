// gut_train.hip — per-Gaussian streaming kernels around the renderer ("next" rows N1/N2 of SURVEY §8f):
//   k_activate_pack : raw parameter rows -> the activated [N,12] particle_density rows the tracer consumes
//                     (what model.py:74-93 + tracer.py:176-178 do with normalize / exp / sigmoid / cat in torch)
//   k_adam_step     : fused Adam over an [N,C] tensor with per-column learning rates and an optional per-row
//                     visibility mask (reference: threedgrut/optimizers/optimizers.cu:47-117 SelectiveAdam, and
//                     torch.optim.Adam when bias correction is on and no mask is given)
// Both are pure HBM streams: 16-byte accesses, one row (or one float4 of a row) per lane.
#include "gut_internal.h"

namespace gut {

// raw row: pos3 | density logit | quat4 (unnormalised) | log-scale3 | unused
// act row: pos3 | sigmoid       | quat4 / |quat|       | exp3       | |quat|   (the norm rides in the pad column so
//          that the backward epilogue can chain through the normalisation without re-reading the raw row)
__global__ __launch_bounds__(kBlock) void k_activate_pack(uint32_t n, const float4* __restrict__ raw, float4* __restrict__ act) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 a = raw[3 * (size_t)i + 0];
    const float4 q = raw[3 * (size_t)i + 1];
    const float4 s = raw[3 * (size_t)i + 2];
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // torch.nn.functional.normalize eps
    act[3 * (size_t)i + 0] = make_float4(a.x, a.y, a.z, 1.0f / (1.0f + expf(-a.w)));
    act[3 * (size_t)i + 1] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    act[3 * (size_t)i + 2] = make_float4(expf(s.x), expf(s.y), expf(s.z), fmaxf(nrm, 1e-12f));
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

}  // namespace gut

extern "C" {

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

}  // extern "C"
