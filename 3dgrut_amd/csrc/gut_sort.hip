// gut_sort.hip — K2 (inclusive scan of tile counts) and K4 (stable radix sort of the (tile|depth) keys).
// Reference: cub::DeviceScan::InclusiveSum / cub::DeviceRadixSort::SortPairs (src/gutRenderer.cu:302-310,
// 356-365).  rocPRIM's device-wide primitives are the ROCm counterparts; the sort is an LSD radix sort and
// therefore stable, restricted to key bits [0, 32 + bit_width(T)) exactly like the reference.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gut_internal.h"

namespace gut {

size_t scan_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n,
                                  rocprim::plus<uint32_t>(), (hipStream_t)0);
    return bytes;
}

hipError_t run_scan(hipStream_t s, void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, uint32_t n) {
    return rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), s);
}

// rocPRIM's tuned gfx950 entry for (8-byte key, 4-byte value) is 512 threads x 16 items with 8-bit digits: 6 passes over
// the 44 key bits of a 4056-tile frame.  9-bit digits cover them in 5 passes; with 1024 x 8 sort blocks and a 512 x 32
// histogram the bench's 9.4 M (tile|depth) pairs sort in 0.446 ms instead of 0.612 ms (tools/sort_sweep.hip; 10-bit
// digits lose again to LDS pressure, 11-bit digits do not fit).
using SortConfig = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 32>, rocprim::kernel_config<1024, 8>, 9,
                                        rocprim::block_radix_rank_algorithm::match>>;

size_t sort_temp_bytes(uint32_t m, int end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (size_t)m, 0u, (unsigned int)end_bit, (hipStream_t)0);
    return bytes;
}

hipError_t run_sort(hipStream_t s, void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                    const uint32_t* vals_in, uint32_t* vals_out, uint32_t m, int end_bit) {
    return rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)m, 0u,
                                     (unsigned int)end_bit, s);
}

// Tile-only grouping for the lazy per-tile depth order (gut_render_common.h: LazyOrder): the same keys, sorted on the tile
// bits [32, end_bit) alone — two radix passes instead of five; stable, so each tile keeps its entries in particle order.
// The digit width is chosen so that two passes cover the tile bits exactly (measured on the bench's 12 tile bits,
// tools: 1024 x 6 items with 6-bit digits 0.185 ms, rocPRIM's default configuration 0.227 ms).
template <unsigned kBits>
using TileSortConfig = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 32>, rocprim::kernel_config<1024, 6>, kBits,
                                        rocprim::block_radix_rank_algorithm::match>>;

template <class Config>
static hipError_t sort_tiles_with(hipStream_t s, void* temp, size_t& temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                                  const uint32_t* vals_in, uint32_t* vals_out, uint32_t m, int end_bit) {
    return rocprim::radix_sort_pairs<Config>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)m, 32u,
                                             (unsigned int)end_bit, s);
}

static hipError_t sort_tiles_dispatch(hipStream_t s, void* temp, size_t& temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                                      const uint32_t* vals_in, uint32_t* vals_out, uint32_t m, int end_bit) {
    const int bits = end_bit - 32;
    if (bits <= 12) return sort_tiles_with<TileSortConfig<6>>(s, temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, m, end_bit);
    if (bits <= 14) return sort_tiles_with<TileSortConfig<7>>(s, temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, m, end_bit);
    if (bits <= 16) return sort_tiles_with<TileSortConfig<8>>(s, temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, m, end_bit);
    if (bits <= 18) return sort_tiles_with<TileSortConfig<9>>(s, temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, m, end_bit);
    return sort_tiles_with<rocprim::default_config>(s, temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, m, end_bit);
}

size_t sort_tiles_temp_bytes(uint32_t m, int end_bit) {
    size_t bytes = 0;
    (void)sort_tiles_dispatch((hipStream_t)0, nullptr, bytes, nullptr, nullptr, nullptr, nullptr, m, end_bit);
    return bytes;
}

hipError_t run_sort_tiles(hipStream_t s, void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out,
                          const uint32_t* vals_in, uint32_t* vals_out, uint32_t m, int end_bit) {
    return sort_tiles_dispatch(s, temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, m, end_bit);
}

}  // namespace gut
