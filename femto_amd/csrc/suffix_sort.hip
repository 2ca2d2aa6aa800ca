// suffix_sort.hip -- GPU suffix array construction for the index builder (not the query hot path).
//
// Prefix doubling on the device: the prepared text (alpha_t codes) is mapped to dense symbol
// ranks, the first round sorts every suffix by its first K symbols packed into one 64-bit key,
// and each following round sorts by (rank[i], rank[i+h]) until every suffix has a unique rank.
// A virtual end marker smaller than every symbol terminates the text, which is exactly the order
// the reference's test sorter produces (src/main/bwt_qsufsort.c:176-240).  The 64-bit key/value
// radix sorts and the scans are rocPRIM device primitives (plain library calls); memory is sized
// for 288 GB HBM: ~36 bytes per suffix.
#include <cstring>  // rocprim's texture iterator needs ::memset declared on the host side

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <string>
#include <vector>

#include "index_builder.hpp"

namespace femto_amd {
namespace {

#define SS_TRY(expr)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      if (err) { err->code = e_ == hipErrorOutOfMemory ? 1 : 6; err->msg = std::string(#expr) + ": " + hipGetErrorString(e_); } \
      return err ? err->code : 6;                                                               \
    }                                                                                           \
  } while (0)

__global__ void pack_keys_kernel(const uint16_t* __restrict__ text, const uint8_t* __restrict__ dense, int64_t n,
                                 int bits, int k, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0;
  for (int j = 0; j < k; j++) {
    const int64_t p = i + j;
    const uint64_t c = p < n ? dense[text[p]] : 0;  // 0 = the virtual end marker
    key = (key << bits) | c;
  }
  keys[i] = key;
  idx[i] = uint32_t(i);
}

__global__ void head_flags_kernel(const uint64_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ head) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n) return;
  head[j] = (j == 0 || keys[j] != keys[j - 1]) ? uint32_t(j + 1) : 0u;  // 1-based group head position
}

__global__ void scatter_rank_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ grp, int64_t n,
                                    uint32_t* __restrict__ rank) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < n) rank[idx[j]] = grp[j];
}

__global__ void pair_keys_kernel(const uint32_t* __restrict__ rank, int64_t n, int64_t h, uint64_t* __restrict__ keys,
                                 uint32_t* __restrict__ idx) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t r2 = i + h < n ? rank[i + h] : 0;
  keys[i] = (uint64_t(rank[i]) << 32) | r2;
  idx[i] = uint32_t(i);
}

__global__ void count_heads_kernel(const uint32_t* __restrict__ head, int64_t n, unsigned long long* __restrict__ out) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool h = j < n && head[j] != 0;
  const unsigned long long m = __ballot(h);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

__global__ void widen_kernel(const uint32_t* __restrict__ idx, int64_t n, int64_t* __restrict__ out) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < n) out[j] = int64_t(idx[j]);
}

struct DevMem {
  void* p = nullptr;
  ~DevMem() { if (p) (void)hipFree(p); }
};

}  // namespace

int gpu_suffix_sort(const std::vector<uint16_t>& text, int device, std::vector<int64_t>* sa_out, Error* err) {
  const int64_t n = int64_t(text.size());
  sa_out->resize(size_t(n));
  if (n == 0) return 0;
  if (n >= (int64_t(1) << 32) - 2) {
    if (err) { err->code = 3; err->msg = "GPU suffix sorter handles texts below 2^32 symbols"; }
    return 3;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    if (err) { err->code = 6; err->msg = "no usable HIP device for the suffix sorter (no CPU fallback)"; }
    return 6;
  }
  SS_TRY(hipSetDevice(device));

  // dense symbol ranks: 0 is reserved for the end marker
  std::vector<uint8_t> present(512, 0), dense(512, 0);
  for (uint16_t c : text) present[c] = 1;  // alpha codes are < 261
  int sigma = 0;
  for (int c = 0; c < 512; c++) if (present[size_t(c)]) dense[size_t(c)] = uint8_t(++sigma);
  int bits = 1;
  while ((1 << bits) <= sigma) bits++;
  const int k = 64 / bits;

  DevMem d_text, d_dense, d_keys, d_keys2, d_idx, d_idx2, d_rank, d_head, d_tmp, d_cnt;
  SS_TRY(hipMalloc(&d_text.p, size_t(n) * 2));
  SS_TRY(hipMalloc(&d_dense.p, 512));
  SS_TRY(hipMalloc(&d_keys.p, size_t(n) * 8));
  SS_TRY(hipMalloc(&d_keys2.p, size_t(n) * 8));
  SS_TRY(hipMalloc(&d_idx.p, size_t(n) * 4));
  SS_TRY(hipMalloc(&d_idx2.p, size_t(n) * 4));
  SS_TRY(hipMalloc(&d_rank.p, size_t(n) * 4));
  SS_TRY(hipMalloc(&d_head.p, size_t(n) * 4));
  SS_TRY(hipMalloc(&d_cnt.p, 8));
  SS_TRY(hipMemcpy(d_text.p, text.data(), size_t(n) * 2, hipMemcpyHostToDevice));
  SS_TRY(hipMemcpy(d_dense.p, dense.data(), 512, hipMemcpyHostToDevice));

  uint64_t* keys = static_cast<uint64_t*>(d_keys.p);
  uint64_t* keys2 = static_cast<uint64_t*>(d_keys2.p);
  uint32_t* idx = static_cast<uint32_t*>(d_idx.p);
  uint32_t* idx2 = static_cast<uint32_t*>(d_idx2.p);
  uint32_t* rank = static_cast<uint32_t*>(d_rank.p);
  uint32_t* head = static_cast<uint32_t*>(d_head.p);

  size_t tmp_sort = 0, tmp_scan = 0;
  SS_TRY(rocprim::radix_sort_pairs(nullptr, tmp_sort, keys, keys2, idx, idx2, size_t(n), 0, 64, nullptr));
  SS_TRY(rocprim::inclusive_scan(nullptr, tmp_scan, head, head, size_t(n), rocprim::maximum<uint32_t>(), nullptr));
  const size_t tmp_bytes = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
  SS_TRY(hipMalloc(&d_tmp.p, tmp_bytes ? tmp_bytes : 16));

  const dim3 blk(256), grd(uint32_t((n + 255) / 256));
  hipLaunchKernelGGL(pack_keys_kernel, grd, blk, 0, nullptr, static_cast<const uint16_t*>(d_text.p),
                     static_cast<const uint8_t*>(d_dense.p), n, bits, k, keys, idx);
  int64_t h = k;
  for (int round = 0; round < 40; round++) {
    size_t tb = tmp_bytes;
    const unsigned end_bit = round == 0 ? unsigned(k * bits) : 64u;
    SS_TRY(rocprim::radix_sort_pairs(d_tmp.p, tb, keys, keys2, idx, idx2, size_t(n), 0, end_bit, nullptr));
    hipLaunchKernelGGL(head_flags_kernel, grd, blk, 0, nullptr, keys2, n, head);
    SS_TRY(hipMemsetAsync(d_cnt.p, 0, 8, nullptr));
    hipLaunchKernelGGL(count_heads_kernel, grd, blk, 0, nullptr, head, n, static_cast<unsigned long long*>(d_cnt.p));
    unsigned long long groups = 0;
    SS_TRY(hipMemcpy(&groups, d_cnt.p, 8, hipMemcpyDeviceToHost));
    if (int64_t(groups) == n) break;  // idx2 is the suffix array
    tb = tmp_bytes;
    SS_TRY(rocprim::inclusive_scan(d_tmp.p, tb, head, head, size_t(n), rocprim::maximum<uint32_t>(), nullptr));
    hipLaunchKernelGGL(scatter_rank_kernel, grd, blk, 0, nullptr, idx2, head, n, rank);
    hipLaunchKernelGGL(pair_keys_kernel, grd, blk, 0, nullptr, rank, n, h, keys, idx);
    h *= 2;
    if (round == 39) {
      if (err) { err->code = 6; err->msg = "suffix sort did not converge"; }
      return 6;
    }
  }
  SS_TRY(hipGetLastError());
  // widen to int64 through the (now free) key buffer and copy out
  hipLaunchKernelGGL(widen_kernel, grd, blk, 0, nullptr, idx2, n, reinterpret_cast<int64_t*>(keys));
  SS_TRY(hipMemcpy(sa_out->data(), keys, size_t(n) * 8, hipMemcpyDeviceToHost));
  return 0;
}

}  // namespace femto_amd
