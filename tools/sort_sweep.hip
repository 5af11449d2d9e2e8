// Dev tool: time rocprim::radix_sort_pairs<u64,u32> on a bench-sized key set under several onesweep configurations.
#include <cstring>
#include <cstdio>
#include <vector>
#include <random>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

template <class Config>
float run(const char* name, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t m, unsigned end_bit) {
    size_t bytes = 0;
    rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, m, 0u, end_bit, (hipStream_t)0);
    void* tmp; hipMalloc(&tmp, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, m, 0u, end_bit, (hipStream_t)0);
    hipEventRecord(a, 0);
    const int it = 20;
    for (int i = 0; i < it; ++i) rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, m, 0u, end_bit, (hipStream_t)0);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %.4f ms  (temp %.1f MB)\n", name, ms / it, bytes / 1e6);
    fflush(stdout);
    hipFree(tmp);
    return ms / it;
}

template <unsigned HB, unsigned HI, unsigned SB, unsigned SI, unsigned BITS, rocprim::block_radix_rank_algorithm A>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<HB, HI>, rocprim::kernel_config<SB, SI>, BITS, A>>;

int main() {
    const size_t m = 9360000;
    const unsigned end_bit = 45;
    std::vector<uint64_t> k(m); std::vector<uint32_t> v(m);
    std::mt19937_64 g(1);
    for (size_t i = 0; i < m; ++i) {
        float d = 0.5f + 20.f * (float)((g() >> 11) * (1.0 / 9007199254740992.0));
        uint32_t bits; memcpy(&bits, &d, 4);
        k[i] = ((uint64_t)(g() % 4056) << 32) | bits; v[i] = (uint32_t)(g() % 6000000);
    }
    uint64_t *kin, *kout; uint32_t *vin, *vout;
    hipMalloc(&kin, m * 8); hipMalloc(&kout, m * 8); hipMalloc(&vin, m * 4); hipMalloc(&vout, m * 4);
    hipMemcpy(kin, k.data(), m * 8, hipMemcpyHostToDevice); hipMemcpy(vin, v.data(), m * 4, hipMemcpyHostToDevice);
    using R = rocprim::block_radix_rank_algorithm;
    run<rocprim::default_config>("default", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 512, 16, 8, R::match>>("512x16 8b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 1024, 8, 8, R::match>>("1024x8 8b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 256, 16, 8, R::match>>("256x16 8b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 256, 24, 8, R::match>>("256x24 8b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 512, 12, 8, R::match>>("512x12 8b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 512, 22, 8, R::match>>("512x22 8b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 512, 8, 8, R::match>>("512x8 8b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 1024, 12, 8, R::match>>("1024x12 8b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 512, 16, 7, R::match>>("512x16 7b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 512, 16, 6, R::match>>("512x16 6b match", kin, kout, vin, vout, m, end_bit);
    run<cfg<512, 16, 512, 16, 5, R::match>>("512x16 5b match (9 passes)", kin, kout, vin, vout, m, end_bit);
    // depth-only and tile-only sorts, to price the two-stage alternative
    run<rocprim::default_config>("default, 13 bits @32", kin, kout, vin, vout, m, 13);
    {
        size_t n = 6000000, bytes = 0;
        uint32_t *a = (uint32_t*)kin, *b = (uint32_t*)kout;
        rocprim::radix_sort_pairs(nullptr, bytes, a, b, vin, vout, n, 0u, 32u, (hipStream_t)0);
        void* tmp; hipMalloc(&tmp, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        rocprim::radix_sort_pairs(tmp, bytes, a, b, vin, vout, n, 0u, 32u, (hipStream_t)0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs(tmp, bytes, a, b, vin, vout, n, 0u, 32u, (hipStream_t)0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("u32/u32 6M 32 bits           %.4f ms\n", ms / 20);
    }
    return 0;
}
