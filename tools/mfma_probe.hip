// Dev probe: does v_mfma_f32_16x16x4_f32 pay for the per-(pixel, entry) product u = M d of the forward
// compositor?  u = M d over the 64 pixels of a wave and the staged entries IS a [entries*3 x 4] . [4 x pixels] product (M rows
// wave-uniform, [d | 0] per pixel), and the matrix pipe co-issues with VALU.  Two kernels with the SAME loop body as K6's
// centred-ray path (gut_render.hip: k_render, lambda `entry`) — cross product, |u|^2, rcp, d2, exp, thresholds, compositing,
// termination — that differ only in where u comes from:
//   VALU:  9 multiply/FMAs per (lane, entry) from the three rows of M read from LDS as wave-uniform ds_read_b128
//   MFMA:  per 5 entries (15 rows + 1 spare) four v_mfma_f32_16x16x4_f32 — one per 16-pixel group, B operands ([d | 0] of the
//          wave's own pixels) loaded once before the loop, the A operand one ds_read_b32 per lane per 5 entries — and a 4x4 block
//          transpose across the 16-lane rows (8 v_permlane32_swap + 8 v_permlane16_swap) that puts all 15 values of a pixel into
//          its own lane.
// Reports ns per (wave, entry) per SIMD at 1..5 workgroups per CU and checks that the two forms composite the same image.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/bin/mfma_probe tools/mfma_probe.hip && tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256, kEntries = 320;   // entries staged per workgroup (a multiple of 5 and of 64)

struct Entry { float4 mu_sigma, m0, m1, m2, feat; };   // FwdEntry of gut_render_common.h

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

struct Pix { float T, cr, cg, cb, dsum; unsigned nhits; bool alive; };

// the rest of K6's per-pair body, given u (gut_render.hip:178-205, centred rays: o = oc)
__device__ __forceinline__ void composite(Pix& p, const float4& cs, float u0, float u1, float u2, float s0, float s1, float s2,
                                          const float4& fid) {
    if (p.alive) {
        const float o0 = cs.x, o1 = cs.y, o2 = cs.z;
        const float x0 = u1 * o2 - u2 * o1, x1 = u2 * o0 - u0 * o2, x2 = u0 * o1 - u1 * o0;
        const float l2 = u0 * u0 + u1 * u1 + u2 * u2;
        const float il2 = fast_rcp(l2);
        const float d2 = (x0 * x0 + x1 * x1 + x2 * x2) * il2;
        if (d2 < 8.966f) {
            const float resp = fast_exp(-0.5f * d2);
            const float alpha = fminf(0.99f, resp * cs.w);
            if ((resp > 0.0113f) && (alpha > 1.0f / 255.0f)) {
                const float proj = -(u0 * o0 + u1 * o1 + u2 * o2) * il2;
                const float h0 = s0 * u0 * proj, h1 = s1 * u1 * proj, h2 = s2 * u2 * proj;
                const float hit_t = fast_sqrt(h0 * h0 + h1 * h1 + h2 * h2);
                if ((hit_t > 0.0f) && (hit_t < 1e6f)) {
                    const float w = alpha * p.T;
                    p.dsum += hit_t * w;
                    p.T *= (1.0f - alpha);
                    if (w > 0.0f) { p.cr += fid.x * w; p.cg += fid.y * w; p.cb += fid.z * w; p.nhits++; }
                    if (p.T < 1e-4f) p.alive = false;
                }
            }
        }
    }
}

__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

template <bool kMfma>
__global__ __launch_bounds__(kBlock, 5) void k_probe(const Entry* __restrict__ entries, const float* __restrict__ dirs, int rounds,
                                                     float4* __restrict__ out, float* __restrict__ ticks) {
    __shared__ Entry stage[kEntries];
    __shared__ float a_op[kEntries / 5][64];   // MFMA A operand of each 5-entry group: [k = lane >> 4][row = lane & 15]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t e = tid; e < kEntries; e += kBlock) {
        const Entry en = entries[e];
        stage[e] = en;
        const uint32_t grp = e / 5, r0 = 3 * (e % 5);
        const float4 rows[3] = {en.m0, en.m1, en.m2};
        for (int i = 0; i < 3; ++i) {
            a_op[grp][0 * 16 + r0 + i] = rows[i].x; a_op[grp][1 * 16 + r0 + i] = rows[i].y;
            a_op[grp][2 * 16 + r0 + i] = rows[i].z; a_op[grp][3 * 16 + r0 + i] = 0.0f;
        }
        if (e % 5 == 0) for (int k = 0; k < 4; ++k) a_op[grp][k * 16 + 15] = 0.0f;
    }
    const size_t pix = (size_t)blockIdx.x * kBlock + tid;
    const float dx = dirs[3 * pix], dy = dirs[3 * pix + 1], dz = dirs[3 * pix + 2];
    // B operands: lane l of pixel group p supplies B[k = l >> 4][n = l & 15] = component k of the direction of pixel 16 p + (l & 15)
    float bop[4];
    {
        const float comp[3] = {dx, dy, dz};
        __shared__ float d_all[4][3][64];
        for (int k = 0; k < 3; ++k) d_all[wave][k][lane] = comp[k];
        __syncthreads();
        for (int p = 0; p < 4; ++p) {
            const uint32_t k = lane >> 4, n = lane & 15;
            bop[p] = k < 3 ? d_all[wave][k][16 * p + n] : 0.0f;
        }
    }
    __syncthreads();
    Pix px{1.0f, 0.f, 0.f, 0.f, 0.f, 0u, true};
    const long long t0 = clock64();
    for (int round = 0; round < rounds; ++round) {
        px.T = 1.0f; px.alive = true;   // every round walks the whole list again (keeps the hit statistics of the first)
        if (!kMfma) {
#pragma unroll 2
            for (uint32_t j = 0; j < kEntries; ++j) {
                if (__ballot(px.alive) == 0ull) break;
                const float4 cs = stage[j].mu_sigma, c0 = stage[j].m0, c1 = stage[j].m1, c2 = stage[j].m2;
                const float u0 = c0.x * dx + c0.y * dy + c0.z * dz;
                const float u1 = c1.x * dx + c1.y * dy + c1.z * dz;
                const float u2 = c2.x * dx + c2.y * dy + c2.z * dz;
                composite(px, cs, u0, u1, u2, c0.w, c1.w, c2.w, stage[j].feat);
            }
        } else {
            for (uint32_t g = 0; g < kEntries / 5; ++g) {
                if (__ballot(px.alive) == 0ull) break;
                const float a = a_op[g][lane];
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                f32x4 acc[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bop[p], z, 0, 0, 0);
                // acc[p][r] of lane (gq = lane >> 4, c = lane & 15) = row 4 gq + r of pixel 16 p + c; wanted: every row of pixel
                // 16 gq + c, i.e. t[g'][r] = lane(g', c).acc[gq][r]: a 4x4 block transpose over the 16-lane rows
                float t[4][4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[p][r] = acc[p][r];
#pragma unroll
                for (int r = 0; r < 4; ++r) { swap32(t[0][r], t[2][r]); swap32(t[1][r], t[3][r]); }
#pragma unroll
                for (int r = 0; r < 4; ++r) { swap16(t[0][r], t[1][r]); swap16(t[2][r], t[3][r]); }
                const float* u = &t[0][0];   // u[3 e + i], e = 0..4
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    const uint32_t j = 5 * g + e;
                    const float4 cs = stage[j].mu_sigma;
                    composite(px, cs, u[3 * e], u[3 * e + 1], u[3 * e + 2], stage[j].m0.w, stage[j].m1.w, stage[j].m2.w, stage[j].feat);
                }
            }
        }
    }
    const long long t1 = clock64();
    out[pix] = make_float4(px.cr, px.cg, px.cb + px.dsum * 1e-3f, (float)px.nhits);
    if (tid == 0) ticks[blockIdx.x] = (float)(t1 - t0);
}

int main() {
    const int max_blocks = 256 * 5;
    std::vector<Entry> h(kEntries);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX; };
    for (auto& e : h) {   // Gaussians of 0.5 .. 2 pixel-footprints around a ray bundle looking down +z from the origin
        const float s[3] = {0.02f + 0.05f * rnd(), 0.02f + 0.05f * rnd(), 0.02f + 0.05f * rnd()};
        float q[4] = {rnd() - .5f, rnd() - .5f, rnd() - .5f, rnd() - .5f};
        const float qn = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (float& v : q) v *= qn;
        const float w = q[0], x = q[1], y = q[2], z = q[3];
        const float R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)},
                               {2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)},
                               {2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)}};
        const float mu[3] = {0.3f * (rnd() - .5f), 0.3f * (rnd() - .5f), 3.0f + rnd()};
        float m[3][3], oc[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) m[i][j] = R[i][j] / s[i];
            oc[i] = -(m[i][0] * mu[0] + m[i][1] * mu[1] + m[i][2] * mu[2]);
        }
        e.mu_sigma = make_float4(oc[0], oc[1], oc[2], 0.004f + 0.03f * rnd());   // faint: the rays stay alive, every listed entry is evaluated
        e.m0 = make_float4(m[0][0], m[0][1], m[0][2], s[0]);
        e.m1 = make_float4(m[1][0], m[1][1], m[1][2], s[1]);
        e.m2 = make_float4(m[2][0], m[2][1], m[2][2], s[2]);
        e.feat = make_float4(rnd(), rnd(), rnd(), 0.f);
    }
    std::vector<float> dirs((size_t)max_blocks * kBlock * 3);
    for (size_t p = 0; p < dirs.size() / 3; ++p) {   // an 8x8 pixel block per wave, blocks scattered over a 0.1 rad field
        const size_t wv = p / 64, l = p % 64;
        const float cx = 0.1f * ((float)((wv * 37) % 101) / 101.0f - 0.5f), cy = 0.1f * ((float)((wv * 53) % 89) / 89.0f - 0.5f);
        float d[3] = {cx + 0.001f * (float)(l & 7), cy + 0.001f * (float)(l >> 3), 1.0f};
        const float n = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + 1.0f);
        for (int k = 0; k < 3; ++k) dirs[3 * p + k] = d[k] * n;
    }
    Entry* d_e; float* d_d; float4 *d_o0, *d_o1; float* d_t;
    hipMalloc(&d_e, sizeof(Entry) * kEntries); hipMalloc(&d_d, dirs.size() * 4);
    hipMalloc(&d_o0, sizeof(float4) * max_blocks * kBlock); hipMalloc(&d_o1, sizeof(float4) * max_blocks * kBlock); hipMalloc(&d_t, 4 * max_blocks);
    hipMemcpy(d_e, h.data(), sizeof(Entry) * kEntries, hipMemcpyHostToDevice);
    hipMemcpy(d_d, dirs.data(), dirs.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int rounds = 400;
    for (int per_cu : {1, 2, 3, 4, 5}) {
        const int blocks = 256 * per_cu;
        float ms[2];
        for (int mode = 0; mode < 2; ++mode) {
            float4* o = mode ? d_o1 : d_o0;
            if (mode) k_probe<true><<<blocks, kBlock>>>(d_e, d_d, 4, o, d_t); else k_probe<false><<<blocks, kBlock>>>(d_e, d_d, 4, o, d_t);
            hipEventRecord(a);
            if (mode) k_probe<true><<<blocks, kBlock>>>(d_e, d_d, rounds, o, d_t); else k_probe<false><<<blocks, kBlock>>>(d_e, d_d, rounds, o, d_t);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms[mode], a, b);
        }
        std::vector<float4> o0((size_t)blocks * kBlock), o1(o0.size());
        hipMemcpy(o0.data(), d_o0, o0.size() * 16, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), d_o1, o1.size() * 16, hipMemcpyDeviceToHost);
        double maxd = 0, hits = 0; size_t hit_mismatch = 0;
        for (size_t i = 0; i < o0.size(); ++i) {
            maxd = fmax(maxd, fmax(fabs(o0[i].x - o1[i].x), fmax(fabs(o0[i].y - o1[i].y), fabs(o0[i].z - o1[i].z))));
            hits += o0[i].w; hit_mismatch += o0[i].w != o1[i].w;
        }
        // (wave, entry) pairs per SIMD: per_cu waves per SIMD, each walking `rounds` x (entries until its 64 rays are dead) — the
        // ticks say how long; report time per LISTED entry (the whole list), as K6's statistics count them
        const double pairs_per_simd = (double)per_cu * rounds * kEntries;
        printf("%d WG/CU: VALU %.3f ms = %.2f ns per (wave, entry) per SIMD | MFMA %.3f ms = %.2f ns | MFMA/VALU %.3f | max colour diff %.2e, "
               "hit-count mismatches %zu of %zu, mean hits/pixel %.1f\n", per_cu, ms[0], ms[0] * 1e6 / pairs_per_simd, ms[1],
               ms[1] * 1e6 / pairs_per_simd, ms[1] / ms[0], maxd, hit_mismatch, o0.size(), hits / o0.size());
    }
    return 0;
}
