// query_sort.hip -- orders a pattern batch by pattern SUFFIX before the backward search.
//
// Backward search consumes a pattern from its last symbol, so patterns that share a suffix visit
// the same rows, buckets and segments for their first steps.  Sorting the batch by the last eight
// symbols puts such patterns in neighbouring lanes: their loads fall on the same cache lines (or the
// very same address inside a wavefront) instead of being issued 64 times.  This is the GPU form of
// the reference's own central idea -- "sort requests by block and row" (src/main/server.h:930-971,
// doc/femto.pdf 3.1) -- applied to lanes instead of disk seeks.  Results are written back in the
// caller's order, so the API contract is unchanged.
//
// The 64-bit key/value radix sort is rocPRIM's device primitive (a plain library sort, as in
// suffix_sort.hip); the key extraction kernel is ours.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdint>

namespace femto_amd {

// key = symbols len-1-skip, len-2-skip, ... (eight of them), the first one most significant.
// idx_in == nullptr: identity order; otherwise keys are produced for the order idx_in (second pass of
// an LSD sort: sort by symbols 8..15 first, then stably by symbols 0..7).
__global__ void suffix_key_kernel(const int64_t npats, const int32_t* __restrict__ plen,
                                  const uint16_t* __restrict__ pats, const int64_t* __restrict__ starts,
                                  const int skip, const uint32_t* __restrict__ idx_in,
                                  uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= npats) return;
  const int64_t q = idx_in ? int64_t(idx_in[j]) : j;
  const int len = plen[q];
  const uint16_t* p = pats + starts[q];
  uint64_t key = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int pos = len - 1 - skip - k;
    const uint64_t c = pos >= 0 ? uint64_t(p[pos] & 0xffu) + 1 : 0;  // 0 = "pattern exhausted"
    key = (key << 8) | (c > 255 ? 255 : c);
  }
  keys[j] = key;
  idx[j] = uint32_t(q);
}

size_t query_sort_temp_bytes(int64_t npats) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                                  static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), size_t(npats), 0, 64,
                                  nullptr);
  return bytes;
}

// keys/keys2: npats u64 each; idx/idx2: npats u32 each; tmp: query_sort_temp_bytes(npats).
// On return idx2 holds the processing order (a permutation of 0..npats-1).
hipError_t query_sort(int64_t npats, const int32_t* d_plen, const uint16_t* d_pats, const int64_t* d_starts,
                      uint64_t* keys, uint64_t* keys2, uint32_t* idx, uint32_t* idx2, void* tmp, size_t tmp_bytes,
                      hipStream_t stream, int levels) {
  const dim3 grd(uint32_t((npats + 255) / 256)), blk(256);
  hipError_t e;
  if (levels >= 2) {  // LSD: symbols 8..15 first
    hipLaunchKernelGGL(suffix_key_kernel, grd, blk, 0, stream, npats, d_plen, d_pats, d_starts, 8,
                       static_cast<const uint32_t*>(nullptr), keys, idx);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, idx, idx2, size_t(npats), 0, 64, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(suffix_key_kernel, grd, blk, 0, stream, npats, d_plen, d_pats, d_starts, 0, idx2, keys, idx);
  } else {
    hipLaunchKernelGGL(suffix_key_kernel, grd, blk, 0, stream, npats, d_plen, d_pats, d_starts, 0,
                       static_cast<const uint32_t*>(nullptr), keys, idx);
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, idx, idx2, size_t(npats), 0, 64, stream);
}

}  // namespace femto_amd
