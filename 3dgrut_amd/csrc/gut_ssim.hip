// gut_ssim.hip — fused SSIM (forward + backward) for gfx950.
//
// "Next" row N1 of SURVEY §8f: the reference's loss calls the external CUDA-only `fused_ssim` package
// (threedgrut/model/losses.py:17-33, requirements.txt:23) right after every render
// (trainer.py:425-430): mean SSIM with an 11x11 Gaussian window (sigma 1.5), C1 = 0.01^2, C2 = 0.03^2,
// padding="valid" (the 5-pixel border of the SSIM map is excluded from the mean).
//
// Layout-agnostic: images are addressed through (channel, row, pixel) element strides so the [1,H,W,3]
// tensor the tracer produces is consumed in place (no permute/contiguous copies).
// One 256-thread workgroup per 16x16 output tile and channel; the (16+10)^2 input patch of both images is
// staged in LDS once, the separable window runs horizontally into LDS and vertically in registers.
// Forward also stores the three partial-derivative maps d(ssim)/d(mu1), d(ssim)/d(sigma1^2),
// d(ssim)/d(sigma12) the backward needs (same scheme as fused-ssim); backward convolves them with the window.
#include "gut_internal.h"

namespace gut {

constexpr int kWin = 11;
constexpr int kHalo = 5;
constexpr int kSTile = 16;
constexpr int kPatch = kSTile + 2 * kHalo;  // 26

__constant__ float c_gauss[kWin] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                    0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                    0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                    0.0075987582094967365f, 0.001028380123898387f};

struct ImgView {
    int H, W, C;
    long long sc, sh, sw;  // element strides: channel, row, pixel
};

__device__ __forceinline__ float load_px(const float* __restrict__ img, const ImgView& v, int c, int y, int x) {
    if (x < 0 || y < 0 || x >= v.W || y >= v.H) return 0.0f;
    return img[(long long)c * v.sc + (long long)y * v.sh + (long long)x * v.sw];
}

// forward: partial sums of the valid-region SSIM map per workgroup + derivative maps (planar [C,H,W])
__global__ __launch_bounds__(256) void k_ssim_fwd(ImgView v, const float* __restrict__ img1, const float* __restrict__ img2,
                                                 float* __restrict__ partial, float* __restrict__ dm_dmu1,
                                                 float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12) {
    __shared__ float s1[kPatch][kPatch + 1], s2[kPatch][kPatch + 1];
    __shared__ float h[5][kPatch][kSTile + 1];  // horizontally filtered: mu1, mu2, x^2, y^2, xy
    __shared__ float red[4];
    const int c = blockIdx.z;
    const int x0 = blockIdx.x * kSTile, y0 = blockIdx.y * kSTile;
    const int tid = threadIdx.x;
    for (int i = tid; i < kPatch * kPatch; i += 256) {
        const int py = i / kPatch, pxx = i - py * kPatch;
        s1[py][pxx] = load_px(img1, v, c, y0 + py - kHalo, x0 + pxx - kHalo);
        s2[py][pxx] = load_px(img2, v, c, y0 + py - kHalo, x0 + pxx - kHalo);
    }
    __syncthreads();
    for (int i = tid; i < kPatch * kSTile; i += 256) {
        const int py = i / kSTile, ox = i - py * kSTile;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const float w = c_gauss[k];
            const float p = s1[py][ox + k], q = s2[py][ox + k];
            a += w * p; b += w * q; aa += w * p * p; bb += w * q * q; ab += w * p * q;
        }
        h[0][py][ox] = a; h[1][py][ox] = b; h[2][py][ox] = aa; h[3][py][ox] = bb; h[4][py][ox] = ab;
    }
    __syncthreads();
    const int ox = tid & 15, oy = tid >> 4;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
        const float w = c_gauss[k];
        mu1 += w * h[0][oy + k][ox]; mu2 += w * h[1][oy + k][ox];
        e11 += w * h[2][oy + k][ox]; e22 += w * h[3][oy + k][ox]; e12 += w * h[4][oy + k][ox];
    }
    const int x = x0 + ox, y = y0 + oy;
    float val = 0.0f;
    if (x < v.W && y < v.H) {
        const float C1 = 0.0001f, C2 = 0.0009f;
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float sg1 = e11 - mu1_sq, sg2 = e22 - mu2_sq, sg12 = e12 - mu12;
        const float A = mu1_sq + mu2_sq + C1, B = sg1 + sg2 + C2, Cc = 2.0f * mu12 + C1, D = 2.0f * sg12 + C2;
        const float m = (Cc * D) / (A * B);
        const size_t o = ((size_t)c * v.H + y) * v.W + x;
        dm_dmu1[o] = (mu2 * 2.0f * D) / (A * B) - (mu2 * 2.0f * Cc) / (A * B) - (mu1 * 2.0f * Cc * D) / (A * A * B) +
                     (mu1 * 2.0f * Cc * D) / (A * B * B);
        dm_dsigma1_sq[o] = (-Cc * D) / (A * B * B);
        dm_dsigma12[o] = (2.0f * Cc) / (A * B);
        const bool valid = x >= kHalo && y >= kHalo && x < v.W - kHalo && y < v.H - kHalo;
        val = valid ? m : 0.0f;
    }
    for (int mk = 32; mk >= 1; mk >>= 1) val += __shfl_xor(val, mk);
    if ((tid & 63) == 0) red[tid >> 6] = val;
    __syncthreads();
    if (tid == 0) partial[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// deterministic final sum of the per-workgroup partials -> mean SSIM
__global__ __launch_bounds__(256) void k_ssim_finish(const float* __restrict__ partial, int n, float inv_count,
                                                    float* __restrict__ out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)partial[i];
    for (int mk = 32; mk >= 1; mk >>= 1) acc += __shfl_xor(acc, mk);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) * (double)inv_count);
}

// backward: d(mean ssim)/d(img1) * upstream, written through the same strides as img1
__global__ __launch_bounds__(256) void k_ssim_bwd(ImgView v, const float* __restrict__ img1, const float* __restrict__ img2,
                                                 const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
                                                 const float* __restrict__ dm_dsigma12, const float* __restrict__ upstream,
                                                 float inv_count, float* __restrict__ grad) {
    __shared__ float s[3][kPatch][kPatch + 1];
    __shared__ float h[3][kPatch][kSTile + 1];
    const int c = blockIdx.z;
    const int x0 = blockIdx.x * kSTile, y0 = blockIdx.y * kSTile;
    const int tid = threadIdx.x;
    const float scale = upstream[0] * inv_count;
    for (int i = tid; i < kPatch * kPatch; i += 256) {
        const int py = i / kPatch, pxx = i - py * kPatch;
        const int x = x0 + pxx - kHalo, y = y0 + py - kHalo;
        float a = 0.f, b = 0.f, d = 0.f;
        if (x >= kHalo && y >= kHalo && x < v.W - kHalo && y < v.H - kHalo) {  // d(mean)/d(map) is zero on the border
            const size_t o = ((size_t)c * v.H + y) * v.W + x;
            a = dm_dmu1[o]; b = dm_dsigma1_sq[o]; d = dm_dsigma12[o];
        }
        s[0][py][pxx] = a; s[1][py][pxx] = b; s[2][py][pxx] = d;
    }
    __syncthreads();
    for (int i = tid; i < kPatch * kSTile; i += 256) {
        const int py = i / kSTile, ox = i - py * kSTile;
        float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
            const float w = c_gauss[k];
            a += w * s[0][py][ox + k]; b += w * s[1][py][ox + k]; d += w * s[2][py][ox + k];
        }
        h[0][py][ox] = a; h[1][py][ox] = b; h[2][py][ox] = d;
    }
    __syncthreads();
    const int ox = tid & 15, oy = tid >> 4;
    float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
        const float w = c_gauss[k];
        a += w * h[0][oy + k][ox]; b += w * h[1][oy + k][ox]; d += w * h[2][oy + k][ox];
    }
    const int x = x0 + ox, y = y0 + oy;
    if (x < v.W && y < v.H) {
        const long long o = (long long)c * v.sc + (long long)y * v.sh + (long long)x * v.sw;
        const float p = img1[o], q = img2[o];
        grad[o] = scale * (a + 2.0f * p * b + q * d);
    }
}

}  // namespace gut

extern "C" {

size_t gut_ssim_workspace_bytes(int32_t channels, int32_t height, int32_t width) {
    const size_t maps = (size_t)3 * channels * height * width * sizeof(float);
    const size_t tiles = (size_t)((width + 15) / 16) * ((height + 15) / 16) * channels * sizeof(float);
    return maps + tiles + 256;
}

static gut::ImgView make_view(int32_t C, int32_t H, int32_t W, int64_t sc, int64_t sh, int64_t sw) {
    gut::ImgView v;
    v.C = C; v.H = H; v.W = W; v.sc = sc; v.sh = sh; v.sw = sw;
    return v;
}

int gut_ssim_forward(void* stream, int32_t channels, int32_t height, int32_t width, int64_t stride_c, int64_t stride_h,
                     int64_t stride_w, const float* d_img1, const float* d_img2, void* d_workspace, float* d_mean_ssim) {
    if (!d_img1 || !d_img2 || !d_workspace || !d_mean_ssim || channels <= 0) return 1;
    if (height <= 2 * gut::kHalo || width <= 2 * gut::kHalo) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t plane = (size_t)channels * height * width;
    float* maps = static_cast<float*>(d_workspace);
    float* partial = maps + 3 * plane;
    const dim3 grid((width + 15) / 16, (height + 15) / 16, channels);
    const gut::ImgView v = make_view(channels, height, width, stride_c, stride_h, stride_w);
    hipLaunchKernelGGL(gut::k_ssim_fwd, grid, dim3(256), 0, s, v, d_img1, d_img2, partial, maps, maps + plane, maps + 2 * plane);
    const double count = (double)channels * (height - 2 * gut::kHalo) * (width - 2 * gut::kHalo);
    hipLaunchKernelGGL(gut::k_ssim_finish, dim3(1), dim3(256), 0, s, partial, (int)(grid.x * grid.y * grid.z), (float)(1.0 / count),
                       d_mean_ssim);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int gut_ssim_backward(void* stream, int32_t channels, int32_t height, int32_t width, int64_t stride_c, int64_t stride_h,
                      int64_t stride_w, const float* d_img1, const float* d_img2, const void* d_workspace,
                      const float* d_upstream /* 1 float */, float* d_grad_img1) {
    if (!d_img1 || !d_img2 || !d_workspace || !d_upstream || !d_grad_img1) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t plane = (size_t)channels * height * width;
    const float* maps = static_cast<const float*>(d_workspace);
    const dim3 grid((width + 15) / 16, (height + 15) / 16, channels);
    const gut::ImgView v = make_view(channels, height, width, stride_c, stride_h, stride_w);
    const double count = (double)channels * (height - 2 * gut::kHalo) * (width - 2 * gut::kHalo);
    hipLaunchKernelGGL(gut::k_ssim_bwd, grid, dim3(256), 0, s, v, d_img1, d_img2, maps, maps + plane, maps + 2 * plane, d_upstream,
                       (float)(1.0 / count), d_grad_img1);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
