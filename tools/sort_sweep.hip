// Dev tool: time rocprim::radix_sort_pairs<u64,u32> on a bench-sized key set under several onesweep configurations.
#include <cstring>
#include <cstdio>
#include <vector>
#include <random>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

template <class Config>
float run(const char* name, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t m, unsigned end_bit) {
    if (getenv("ONLY") && !strstr(name, getenv("ONLY"))) return 0.f;
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
    run<cfg<512, 16, 1024, 8, 9, R::match>>("1024x8 9b 44bits", kin, kout, vin, vout, m, 44);
    run<cfg<512, 32, 1024, 8, 9, R::match>>("h512x32 1024x8 9b 44bits", kin, kout, vin, vout, m, 44);
    run<cfg<512, 32, 1024, 9, 9, R::match>>("h512x32 1024x9 9b 44bits", kin, kout, vin, vout, m, 44);
    run<cfg<512, 32, 1024, 10, 9, R::match>>("h512x32 1024x10 9b 44bits", kin, kout, vin, vout, m, 44);
    run<cfg<512, 32, 1024, 12, 9, R::match>>("h512x32 1024x12 9b 44bits", kin, kout, vin, vout, m, 44);
    run<cfg<1024, 16, 1024, 8, 9, R::match>>("h1024x16 1024x8 9b 44bits", kin, kout, vin, vout, m, 44);
    run<cfg<512, 32, 1024, 8, 9, R::match>>("h512x32 1024x8 9b 47bits", kin, kout, vin, vout, m, 47);
    run<cfg<512, 16, 512, 12, 8, R::match>>("512x12 8b 47bits", kin, kout, vin, vout, m, 47);
    return 0;
}
