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
constexpr int kRowStride = 48;              // LDS row stride of a patch, see k_ssim_fwd

// Tile of workgroup b: the eight XCDs take workgroups round-robin (b % 8), so neighbouring tiles — whose 26x26 input patches
// overlap by 10 pixels — would sit in eight different L2s and every halo would be fetched from memory again (measured: 2.5x the
// algorithmic bytes).  Each XCD gets one contiguous band of tile rows instead.
__device__ __forceinline__ void xcd_tile(uint32_t b, uint32_t gx, uint32_t gy, int* tx, int* ty) {
    const uint32_t n = gx * gy, full = n >> 3;            // tiles per band; the up to seven tiles beyond 8 * full keep t = b
    const uint32_t t = b < 8u * full ? (b & 7u) * full + (b >> 3) : b;
    *tx = (int)(t % gx); *ty = (int)(t / gx);
}

__constant__ float c_gauss[kWin] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                    0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                    0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                    0.0075987582094967365f, 0.001028380123898387f};

struct ImgView {
    int H, W, C;
    long long sc, sh, sw;  // element strides: channel, row, pixel
    // fused photometric loss only: img1 is the tracer's [H,W,4] rgba and the compared image is
    // rgb + background * (1 - alpha) (BackgroundColor.forward, model/background.py:78-93); alpha_offset < 0 = plain image
    long long alpha_offset;  // element offset of alpha relative to channel 0 of the same pixel
    float background;        // constant background colour (0 black, 1 white)
};

__device__ __forceinline__ float load_px(const float* __restrict__ img, const ImgView& v, int c, int y, int x) {
    if (x < 0 || y < 0 || x >= v.W || y >= v.H) return 0.0f;
    const long long o = (long long)y * v.sh + (long long)x * v.sw;
    float p = img[o + (long long)c * v.sc];
    if (v.alpha_offset >= 0 && v.background != 0.0f) p += v.background * (1.0f - img[o + v.alpha_offset]);
    return p;
}

// forward: partial sums of the valid-region SSIM map per workgroup + derivative maps (planar [C,H,W])
__global__ __launch_bounds__(256) void k_ssim_fwd(ImgView v, ImgView v2, const float* __restrict__ img1,
                                                 const float* __restrict__ img2, float* __restrict__ partial,
                                                 float* __restrict__ partial_l1, float* __restrict__ dm_dmu1,
                                                 float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12,
                                                 uint32_t gx, uint32_t gy) {
    // row strides chosen for the two 16-lane rows a 32-lane LDS access group covers: 48 = 16 mod 32 for the patches (row r and
    // r + 1 fall on disjoint halves of the 32 banks while the 11-tap window slides), 16 for the filtered rows (ditto for the
    // column pass).  With the earlier 27 / 17 the window passes lost 46 % of their LDS cycles to 2-way conflicts.
    __shared__ float s1[kPatch][kRowStride], s2[kPatch][kRowStride];
    __shared__ float h[5][kPatch][kSTile];  // horizontally filtered: mu1, mu2, x^2, y^2, xy
    __shared__ float red[4];
    const int c = blockIdx.z;
    int tx, ty;
    xcd_tile(blockIdx.x, gx, gy, &tx, &ty);
    const int x0 = tx * kSTile, y0 = ty * kSTile;
    const int tid = threadIdx.x;
    for (int i = tid; i < kPatch * kPatch; i += 256) {
        const int py = i / kPatch, pxx = i - py * kPatch;
        s1[py][pxx] = load_px(img1, v, c, y0 + py - kHalo, x0 + pxx - kHalo);
        s2[py][pxx] = load_px(img2, v2, c, y0 + py - kHalo, x0 + pxx - kHalo);
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
    float val = 0.0f, l1 = 0.0f;
    if (x < v.W && y < v.H) {
        l1 = fabsf(s1[oy + kHalo][ox + kHalo] - s2[oy + kHalo][ox + kHalo]);
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
    const int slot = blockIdx.z * gridDim.x + blockIdx.x;
    if (tid == 0) partial[slot] = red[0] + red[1] + red[2] + red[3];
    if (partial_l1) {  // block-uniform
        __syncthreads();
        for (int mk = 32; mk >= 1; mk >>= 1) l1 += __shfl_xor(l1, mk);
        if ((tid & 63) == 0) red[tid >> 6] = l1;
        __syncthreads();
        if (tid == 0) partial_l1[slot] = red[0] + red[1] + red[2] + red[3];
    }
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

// fused photometric loss: out3 = { lambda_l1 * L1 + lambda_ssim * (1 - SSIM), L1, SSIM } from the forward kernel's per-workgroup
// partial sums (fixed summation order: the value does not depend on which workgroup finished when).  Not a launch of its own:
// workgroup 0 of the BACKWARD kernel does it (the partials are complete when that kernel starts, and its other 12 k workgroups
// run meanwhile) — a 16 us single-workgroup launch less on the step's critical path.
struct FinishArgs {
    const float* partial = nullptr;
    const float* partial_l1 = nullptr;
    int n = 0;
    float inv_count_ssim = 0.f, inv_count_l1 = 0.f, lambda_l1 = 0.f, lambda_ssim = 0.f;
    float* out3 = nullptr;   // nullptr: nothing to finish (gut_ssim_backward)
};

__device__ __forceinline__ void photometric_finish(const float* __restrict__ partial, const float* __restrict__ partial_l1,
                                                   int n, float inv_count_ssim, float inv_count_l1, float lambda_l1,
                                                   float lambda_ssim, float* __restrict__ out3) {
    __shared__ double red[2][4];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        a += (double)partial[i];
        b += (double)partial_l1[i];
    }
    for (int mk = 32; mk >= 1; mk >>= 1) {
        a += __shfl_xor(a, mk);
        b += __shfl_xor(b, mk);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ssim = (float)((red[0][0] + red[0][1] + red[0][2] + red[0][3]) * (double)inv_count_ssim);
        const float l1 = (float)((red[1][0] + red[1][1] + red[1][2] + red[1][3]) * (double)inv_count_l1);
        out3[0] = lambda_l1 * l1 + lambda_ssim * (1.0f - ssim);
        out3[1] = l1;
        out3[2] = ssim;
    }
}

// backward: d(mean ssim)/d(img1) * upstream, written through the same strides as img1
// fused photometric loss (upstream == nullptr): grad = -lambda_ssim * d(mean ssim) + lambda_l1 * sign(p - q) / numel, and
// channel 0's workgroups also write the alpha gradient slot (-background * sum of the colour gradients is added by
// k_alpha_grad for a non-black background; zero for black).
__global__ __launch_bounds__(256) void k_ssim_bwd(ImgView v, ImgView v2, const float* __restrict__ img1,
                                                 const float* __restrict__ img2, const float* __restrict__ dm_dmu1,
                                                 const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12,
                                                 const float* __restrict__ upstream, float inv_count, float ssim_weight,
                                                 float l1_weight, float* __restrict__ grad, uint32_t gx, uint32_t gy, FinishArgs fin) {
    __shared__ float s[3][kPatch][kRowStride];
    __shared__ float h[3][kPatch][kSTile];
    if (fin.out3 && blockIdx.x == 0 && blockIdx.z == 0)   // (block-uniform)
        photometric_finish(fin.partial, fin.partial_l1, fin.n, fin.inv_count_ssim, fin.inv_count_l1, fin.lambda_l1, fin.lambda_ssim, fin.out3);
    const int c = blockIdx.z;
    int tx, ty;
    xcd_tile(blockIdx.x, gx, gy, &tx, &ty);
    const int x0 = tx * kSTile, y0 = ty * kSTile;
    const int tid = threadIdx.x;
    const float scale = (upstream ? upstream[0] : ssim_weight) * inv_count;
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
        const float p = load_px(img1, v, c, y, x), q = load_px(img2, v2, c, y, x);
        float g = scale * (a + 2.0f * p * b + q * d);
        if (l1_weight != 0.0f) g += l1_weight * (float)((p > q) - (p < q));
        grad[o] = g;
        if (v.alpha_offset >= 0 && c == 0) grad[o + v.alpha_offset] = 0.0f;
    }
}

// d(loss)/d(alpha) = -background * (g_r + g_g + g_b) for rgb_out = rgb + background * (1 - alpha)
__global__ __launch_bounds__(256) void k_alpha_grad(int pixels, float background, float* __restrict__ rgba_grad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < pixels) {
        float4* g = reinterpret_cast<float4*>(rgba_grad) + i;
        float4 t = *g;
        t.w = -background * (t.x + t.y + t.z);
        *g = t;
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
    v.alpha_offset = -1;
    v.background = 0.0f;
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
    const uint32_t gx = (width + 15) / 16, gy = (height + 15) / 16;
    const dim3 grid(gx * gy, 1, channels);
    const gut::ImgView v = make_view(channels, height, width, stride_c, stride_h, stride_w);
    hipLaunchKernelGGL(gut::k_ssim_fwd, grid, dim3(256), 0, s, v, v, d_img1, d_img2, partial, (float*)nullptr, maps, maps + plane,
                       maps + 2 * plane, gx, gy);
    const double count = (double)channels * (height - 2 * gut::kHalo) * (width - 2 * gut::kHalo);
    hipLaunchKernelGGL(gut::k_ssim_finish, dim3(1), dim3(256), 0, s, partial, (int)(grid.x * grid.z), (float)(1.0 / count),
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
    const uint32_t gx = (width + 15) / 16, gy = (height + 15) / 16;
    const dim3 grid(gx * gy, 1, channels);
    const gut::ImgView v = make_view(channels, height, width, stride_c, stride_h, stride_w);
    const double count = (double)channels * (height - 2 * gut::kHalo) * (width - 2 * gut::kHalo);
    hipLaunchKernelGGL(gut::k_ssim_bwd, grid, dim3(256), 0, s, v, v, d_img1, d_img2, maps, maps + plane, maps + 2 * plane, d_upstream,
                       (float)(1.0 / count), 0.0f, 0.0f, d_grad_img1, gx, gy, gut::FinishArgs());
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

size_t gut_photometric_workspace_bytes(int32_t height, int32_t width) {
    const size_t tiles = (size_t)((width + 15) / 16) * ((height + 15) / 16) * 3 * sizeof(float);
    return gut_ssim_workspace_bytes(3, height, width) + tiles;
}

int gut_photometric_loss(void* stream, int32_t height, int32_t width, const float* d_rgba, const float* d_gt_rgb, float background,
                         float lambda_l1, float lambda_ssim, void* d_workspace, float* d_loss3, float* d_rgba_grad) {
    if (!d_rgba || !d_gt_rgb || !d_workspace || !d_loss3 || !d_rgba_grad) return 1;
    if (height <= 2 * gut::kHalo || width <= 2 * gut::kHalo) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t plane = (size_t)3 * height * width;
    float* maps = static_cast<float*>(d_workspace);
    float* partial = maps + 3 * plane;
    const uint32_t gx = (width + 15) / 16, gy = (height + 15) / 16;
    const dim3 grid(gx * gy, 1, 3);
    const int nblocks = (int)(grid.x * grid.z);
    float* partial_l1 = partial + nblocks + 64;
    gut::ImgView v = make_view(3, height, width, 1, 4 * (int64_t)width, 4);   // rgba, interleaved
    v.alpha_offset = 3;
    v.background = background;
    const gut::ImgView g = make_view(3, height, width, 1, 3 * (int64_t)width, 3);  // ground truth, interleaved rgb
    hipLaunchKernelGGL(gut::k_ssim_fwd, grid, dim3(256), 0, s, v, g, d_rgba, d_gt_rgb, partial, partial_l1, maps, maps + plane,
                       maps + 2 * plane, gx, gy);
    const double count = 3.0 * (height - 2 * gut::kHalo) * (width - 2 * gut::kHalo);
    const double numel = 3.0 * height * width;
    gut::FinishArgs fin;
    fin.partial = partial; fin.partial_l1 = partial_l1; fin.n = nblocks;
    fin.inv_count_ssim = (float)(1.0 / count); fin.inv_count_l1 = (float)(1.0 / numel);
    fin.lambda_l1 = lambda_l1; fin.lambda_ssim = lambda_ssim; fin.out3 = d_loss3;
    hipLaunchKernelGGL(gut::k_ssim_bwd, grid, dim3(256), 0, s, v, g, d_rgba, d_gt_rgb, maps, maps + plane, maps + 2 * plane,
                       (const float*)nullptr, (float)(1.0 / count), -lambda_ssim, (float)(lambda_l1 / numel), d_rgba_grad, gx, gy, fin);
    if (background != 0.0f) {
        const int pixels = height * width;
        hipLaunchKernelGGL(gut::k_alpha_grad, dim3((pixels + 255) / 256), dim3(256), 0, s, pixels, background, d_rgba_grad);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
