// Dev probe: issue rate of v_fma_f32 vs v_pk_fma_f32 / v_pk_mul_f32 on gfx950, 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i; p[i] = v2f{seed + i, seed - i}; }
    const float m = 1.0001f, c = 0.5f; const v2f pm = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            } else if (MODE == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            } else if (MODE == 5) {  // pk with op_sel broadcast of the low half of src1
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(pm), "v"(pc));
            }
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(iters * 32);
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
    float* d; hipMalloc(&d, 256 * 256 * 16 * 4 * 4);
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * blocks_per_cu, 256>>>(d, 100, 1.0f);
    hipEventRecord(a); k<MODE><<<256 * blocks_per_cu, 256>>>(d, iters, 1.0f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    float ticks; hipMemcpy(&ticks, d, 4, hipMemcpyDeviceToHost);
    // wave-instructions issued per SIMD: blocks_per_cu waves per SIMD x iters x 32
    const double inst = (double)blocks_per_cu * iters * 32;
    printf("%-26s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz), clock64 ticks/instr %.2f\n", name,
           blocks_per_cu, ms, ms * 1e6 / inst, ms * 1e6 / inst * 2.4, ticks);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<5>("v_pk_fma_f32 op_sel bcast", w); run<2>("v_pk_mul_f32", w);
        run<3>("v_mul_f32", w); run<4>("v_pk_add_f32", w);
    }
    return 0;
}
