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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "index_builder.hpp"

namespace femto_amd {
int gpu_suffix_sort_large(const std::vector<uint16_t>& text, int device, int64_t part_cap, std::vector<int64_t>* sa_out, Error* err);
namespace {

#define SS_TRY(expr)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      if (err) { err->code = e_ == hipErrorOutOfMemory ? 1 : 6; err->msg = std::string(#expr) + ": " + hipGetErrorString(e_); } \
      return err ? err->code : 6;                                                               \
    }                                                                                           \
  } while (0)

__global__ void pack_keys_kernel(const uint16_t* __restrict__ text, const uint16_t* __restrict__ dense, int64_t n,
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
  {
    const char* force = getenv("FEMTO_AMD_LARGE_SORT_CAP");   // test hook: exercise the large-text path on small inputs
    if (n >= (int64_t(1) << 32) - 2 || force) {
      const int64_t cap = force ? atoll(force) : (int64_t(1) << 30) + (int64_t(1) << 28);
      return gpu_suffix_sort_large(text, device, cap, sa_out, err);
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    if (err) { err->code = 6; err->msg = "no usable HIP device for the suffix sorter (no CPU fallback)"; }
    return 6;
  }
  SS_TRY(hipSetDevice(device));

  // dense symbol ranks: 0 is reserved for the end marker.  A text with all 256 byte values plus SEOF has 257 symbols,
  // so the ranks need 16 bits (an 8-bit table wrapped rank 256 onto the end marker: wrong order for such texts).
  std::vector<uint8_t> present(512, 0);
  std::vector<uint16_t> dense(512, 0);
  for (uint16_t c : text) present[c] = 1;  // alpha codes are < 261
  int sigma = 0;
  for (int c = 0; c < 512; c++) if (present[size_t(c)]) dense[size_t(c)] = uint16_t(++sigma);
  int bits = 1;
  while ((1 << bits) <= sigma) bits++;
  const int k = 64 / bits;

  DevMem d_text, d_dense, d_keys, d_keys2, d_idx, d_idx2, d_rank, d_head, d_tmp, d_cnt;
  SS_TRY(hipMalloc(&d_text.p, size_t(n) * 2));
  SS_TRY(hipMalloc(&d_dense.p, 1024));
  SS_TRY(hipMalloc(&d_keys.p, size_t(n) * 8));
  SS_TRY(hipMalloc(&d_keys2.p, size_t(n) * 8));
  SS_TRY(hipMalloc(&d_idx.p, size_t(n) * 4));
  SS_TRY(hipMalloc(&d_idx2.p, size_t(n) * 4));
  SS_TRY(hipMalloc(&d_rank.p, size_t(n) * 4));
  SS_TRY(hipMalloc(&d_head.p, size_t(n) * 4));
  SS_TRY(hipMalloc(&d_cnt.p, 8));
  SS_TRY(hipMemcpy(d_text.p, text.data(), size_t(n) * 2, hipMemcpyHostToDevice));
  SS_TRY(hipMemcpy(d_dense.p, dense.data(), 1024, hipMemcpyHostToDevice));

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
                     static_cast<const uint16_t*>(d_dense.p), n, bits, k, keys, idx);
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


// =====================================================================================
// Large texts (n >= 2^32 - 2, e.g. the 8 GiB configuration): 64-bit suffix indexes and ranks.
// A double-buffered 16-byte key/value radix sort of all n suffixes would not fit next to SA and ISA in
// 288 GB, so round 1 is done in PARTS: suffixes are partitioned by the top 12 bits of their K-symbol key
// (histogram -> contiguous key ranges of at most `part_cap` suffixes), each part is compacted, sorted and
// appended to SA.  Refinement then touches only suffixes whose K-symbol prefix is not unique
// (Larsson-Sadakane style prefix doubling on the tied groups: ~0.2 % of a random 8 GiB ACGT text).
// Limits (reported as errors, never silently wrong): one key bin larger than part_cap, or more tied
// suffixes than part_cap (highly repetitive texts) -- those need the external-memory sorter class of the
// reference's src/dcx_cc, which is out of scope.
// =====================================================================================
namespace {

__device__ __forceinline__ uint64_t key_at(const uint16_t* __restrict__ D, int64_t n, int64_t i, int bits, int k) {
  uint64_t key = 0;
  for (int j = 0; j < k; j++) {
    const int64_t p = i + j;
    key = (key << bits) | (p < n ? uint64_t(D[p]) : 0ull);
  }
  return key;
}

__global__ void hist12_kernel(const uint16_t* __restrict__ D, int64_t n, int bits, int k, int keybits,
                              unsigned long long* __restrict__ hist) {
  __shared__ unsigned int h[4096];
  for (int t = threadIdx.x; t < 4096; t += blockDim.x) h[t] = 0;
  __syncthreads();
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t key = key_at(D, n, i, bits, k);
    atomicAdd(&h[uint32_t(key >> (keybits - 12))], 1u);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 4096; t += blockDim.x)
    if (h[t]) atomicAdd(&hist[t], (unsigned long long)h[t]);
}

__global__ void compact_part_kernel(const uint16_t* __restrict__ D, int64_t n, int bits, int k, int keybits, uint32_t lo_bin,
                                    uint32_t hi_bin, uint64_t* __restrict__ keys, uint64_t* __restrict__ vals,
                                    unsigned long long* __restrict__ counter) {
  // grid-stride: an AQL dispatch carries at most 2^32 - 1 work-items per dimension, fewer than the suffixes
  // of an 8 GiB text.  Every lane of a wavefront runs the same number of iterations (the ballot needs that).
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  const int lane = threadIdx.x & 63;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i - lane < n; i += stride) {
    bool take = false;
    uint64_t key = 0;
    if (i < n) {
      key = key_at(D, n, i, bits, k);
      const uint32_t bin = uint32_t(key >> (keybits - 12));
      take = bin >= lo_bin && bin < hi_bin;
    }
    const unsigned long long m = __ballot(take);
    if (!m) continue;
    const int leader = __ffsll((long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = (unsigned long long)__shfl((long long)base, leader, 64);
    if (take) {
      const unsigned long long pos = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1));
      keys[pos] = key;
      vals[pos] = uint64_t(i);
    }
  }
}

// head[j] = global position of the first element of j's group (1-based so that 0 can mean "not a head")
__global__ void part_heads_kernel(const uint64_t* __restrict__ keys, int64_t cnt, uint64_t off, uint64_t* __restrict__ head) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  head[j] = (j == 0 || keys[j] != keys[j - 1]) ? off + uint64_t(j) + 1 : 0;
}

__global__ void part_store_kernel(const uint64_t* __restrict__ vals, const uint64_t* __restrict__ head, int64_t cnt, uint64_t off,
                                  uint64_t* __restrict__ SA, uint64_t* __restrict__ ISA) {
  const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  SA[off + uint64_t(j)] = vals[j];
  ISA[vals[j]] = head[j] - 1;  // rank = position of the group head
}

// positions whose group (by ISA[SA[.]]) has more than one member, appended in increasing order PER BLOCK RANGE:
// the caller launches it over consecutive chunks so that the list stays sorted.
__global__ void tied_flags_kernel(const uint64_t* __restrict__ SA, const uint64_t* __restrict__ ISA, int64_t n, int64_t base,
                                  int64_t cnt, uint8_t* __restrict__ flag) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= cnt) return;
  const int64_t j = base + t;
  const uint64_t g = ISA[SA[j]];
  const bool tied = (j > 0 && ISA[SA[j - 1]] == g) || (j + 1 < n && ISA[SA[j + 1]] == g);
  flag[t] = tied ? 1 : 0;
}

__global__ void tied_keys_kernel(const uint64_t* __restrict__ pos, int64_t T, const uint64_t* __restrict__ SA,
                                 const uint64_t* __restrict__ ISA, int64_t n, int64_t h, uint64_t* __restrict__ grp,
                                 uint64_t* __restrict__ key2, uint64_t* __restrict__ val) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const uint64_t s = SA[pos[t]];
  grp[t] = ISA[s];
  key2[t] = (int64_t(s) + h < n) ? ISA[s + uint64_t(h)] + 1 : 0;
  val[t] = s;
}

// after sorting the tied suffixes by (group, key2): new heads
__global__ void tied_heads_kernel(const uint64_t* __restrict__ grp, const uint64_t* __restrict__ key2, const uint64_t* __restrict__ pos,
                                  int64_t T, uint64_t* __restrict__ head) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= T) return;
  head[t] = (t == 0 || grp[t] != grp[t - 1] || key2[t] != key2[t - 1]) ? pos[t] + 1 : 0;
}

__global__ void tied_store_kernel(const uint64_t* __restrict__ pos, const uint64_t* __restrict__ val, const uint64_t* __restrict__ head,
                                  int64_t T, uint64_t* __restrict__ SA, uint64_t* __restrict__ ISA) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= T) return;
  SA[pos[t]] = val[t];
  ISA[val[t]] = head[t] - 1;
}

__global__ void gather_u64_kernel(const uint64_t* __restrict__ src, const uint64_t* __restrict__ idx, int64_t T, uint64_t* __restrict__ dst) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t < T) dst[t] = src[idx[t]];
}

__global__ void iota_u64_kernel(int64_t T, uint64_t* __restrict__ dst) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t < T) dst[t] = uint64_t(t);
}

inline dim3 grid_for(int64_t n) { return dim3(uint32_t((n + 255) / 256)); }

}  // namespace

int gpu_suffix_sort_large(const std::vector<uint16_t>& text, int device, int64_t part_cap, std::vector<int64_t>* sa_out, Error* err) {
  const int64_t n = int64_t(text.size());
  sa_out->resize(size_t(n));
  if (n == 0) return 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    if (err) { err->code = 6; err->msg = "no usable HIP device for the suffix sorter (no CPU fallback)"; }
    return 6;
  }
  SS_TRY(hipSetDevice(device));
  std::vector<uint8_t> present(512, 0);
  std::vector<uint16_t> dense(512, 0);   // up to 257 symbols: 16-bit ranks (see gpu_suffix_sort)
  for (uint16_t c : text) present[c] = 1;
  int sigma = 0;
  for (int c = 0; c < 512; c++) if (present[size_t(c)]) dense[size_t(c)] = uint16_t(++sigma);
  int bits = 1;
  while ((1 << bits) <= sigma) bits++;
  const int k = 64 / bits;
  const int keybits = k * bits;
  if (keybits < 12) { if (err) { err->code = 6; err->msg = "key too short"; } return 6; }

  DevMem d_D, d_SA, d_ISA, d_keys, d_keys2, d_vals, d_vals2, d_head, d_tmp, d_cnt, d_hist, d_flag, d_pos, d_grp;
  {
    std::vector<uint16_t> Dh(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++) Dh[size_t(i)] = dense[text[size_t(i)]];
    SS_TRY(hipMalloc(&d_D.p, size_t(n) * 2));
    SS_TRY(hipMemcpy(d_D.p, Dh.data(), size_t(n) * 2, hipMemcpyHostToDevice));
  }
  SS_TRY(hipMalloc(&d_SA.p, size_t(n) * 8));
  SS_TRY(hipMalloc(&d_ISA.p, size_t(n) * 8));
  SS_TRY(hipMalloc(&d_cnt.p, 8));
  SS_TRY(hipMalloc(&d_hist.p, 4096 * 8));
  SS_TRY(hipMemset(d_hist.p, 0, 4096 * 8));
  const uint16_t* D = static_cast<const uint16_t*>(d_D.p);
  uint64_t* SA = static_cast<uint64_t*>(d_SA.p);
  uint64_t* ISA = static_cast<uint64_t*>(d_ISA.p);

  hipLaunchKernelGGL(hist12_kernel, dim3(4096), dim3(256), 0, nullptr, D, n, bits, k, keybits, static_cast<unsigned long long*>(d_hist.p));
  std::vector<unsigned long long> hist(4096);
  SS_TRY(hipMemcpy(hist.data(), d_hist.p, 4096 * 8, hipMemcpyDeviceToHost));
  // contiguous key ranges of at most part_cap suffixes
  std::vector<std::pair<uint32_t, uint32_t>> parts;
  std::vector<int64_t> part_size;
  {
    uint32_t lo = 0;
    int64_t acc = 0;
    for (uint32_t b = 0; b < 4096; b++) {
      if (int64_t(hist[b]) > part_cap) {
        if (err) { err->code = 3; err->msg = "suffix sorter: one 12-bit key bin exceeds the part capacity (text too repetitive for the large-text sorter)"; }
        return 3;
      }
      if (acc + int64_t(hist[b]) > part_cap) {
        parts.push_back({lo, b});
        part_size.push_back(acc);
        lo = b;
        acc = 0;
      }
      acc += int64_t(hist[b]);
    }
    parts.push_back({lo, 4096});
    part_size.push_back(acc);
  }
  const bool dbg = getenv("FEMTO_AMD_SORT_DEBUG") != nullptr;
  if (dbg) {
    fprintf(stderr, "[sort] n=%lld sigma=%d bits=%d k=%d parts=%zu:", (long long)n, sigma, bits, k, parts.size());
    for (size_t p = 0; p < parts.size(); p++) fprintf(stderr, " [%u,%u)=%lld", parts[p].first, parts[p].second, (long long)part_size[p]);
    fprintf(stderr, "\n");
  }
  int64_t cap = 0;
  for (int64_t sz : part_size) cap = std::max(cap, sz);
  cap = std::max<int64_t>(cap, 1);
  SS_TRY(hipMalloc(&d_keys.p, size_t(cap) * 8));
  SS_TRY(hipMalloc(&d_keys2.p, size_t(cap) * 8));
  SS_TRY(hipMalloc(&d_vals.p, size_t(cap) * 8));
  SS_TRY(hipMalloc(&d_vals2.p, size_t(cap) * 8));
  SS_TRY(hipMalloc(&d_head.p, size_t(cap) * 8));
  uint64_t* keys = static_cast<uint64_t*>(d_keys.p);
  uint64_t* keys2 = static_cast<uint64_t*>(d_keys2.p);
  uint64_t* vals = static_cast<uint64_t*>(d_vals.p);
  uint64_t* vals2 = static_cast<uint64_t*>(d_vals2.p);
  uint64_t* head = static_cast<uint64_t*>(d_head.p);
  size_t tmp_sort = 0, tmp_scan = 0, tmp_sel = 0;
  SS_TRY(rocprim::radix_sort_pairs(nullptr, tmp_sort, keys, keys2, vals, vals2, size_t(cap), 0, 64, nullptr));
  SS_TRY(rocprim::inclusive_scan(nullptr, tmp_scan, head, head, size_t(cap), rocprim::maximum<uint64_t>(), nullptr));
  const int64_t sel_chunk = int64_t(1) << 28;
  SS_TRY(hipMalloc(&d_flag.p, size_t(sel_chunk)));
  SS_TRY(rocprim::select(nullptr, tmp_sel, rocprim::counting_iterator<uint64_t>(0), static_cast<uint8_t*>(d_flag.p), keys,
                         static_cast<unsigned long long*>(d_cnt.p), size_t(sel_chunk), static_cast<hipStream_t>(nullptr)));
  const size_t tmp_bytes = std::max(tmp_sort, std::max(tmp_scan, tmp_sel));
  SS_TRY(hipMalloc(&d_tmp.p, tmp_bytes ? tmp_bytes : 16));

  // ---- round 1, part by part
  uint64_t off = 0;
  for (size_t p = 0; p < parts.size(); p++) {
    const int64_t cnt = part_size[p];
    if (cnt == 0) continue;
    SS_TRY(hipMemset(d_cnt.p, 0, 8));
    hipLaunchKernelGGL(compact_part_kernel, dim3(1u << 20), dim3(256), 0, nullptr, D, n, bits, k, keybits, parts[p].first, parts[p].second,
                       keys, vals, static_cast<unsigned long long*>(d_cnt.p));
    if (dbg) {
      unsigned long long got = 0;
      SS_TRY(hipMemcpy(&got, d_cnt.p, 8, hipMemcpyDeviceToHost));
      fprintf(stderr, "[sort] part %zu compacted %llu (expected %lld)\n", p, got, (long long)cnt);
    }
    size_t tb = tmp_bytes;
    SS_TRY(rocprim::radix_sort_pairs(d_tmp.p, tb, keys, keys2, vals, vals2, size_t(cnt), 0, unsigned(keybits), nullptr));
    hipLaunchKernelGGL(part_heads_kernel, grid_for(cnt), dim3(256), 0, nullptr, keys2, cnt, off, head);
    tb = tmp_bytes;
    SS_TRY(rocprim::inclusive_scan(d_tmp.p, tb, head, head, size_t(cnt), rocprim::maximum<uint64_t>(), nullptr));
    hipLaunchKernelGGL(part_store_kernel, grid_for(cnt), dim3(256), 0, nullptr, vals2, head, cnt, off, SA, ISA);
    off += uint64_t(cnt);
  }
  SS_TRY(hipDeviceSynchronize());
  if (int64_t(off) != n) { if (err) { err->code = 6; err->msg = "suffix sorter: partition sizes do not add up"; } return 6; }

  // ---- refinement of tied groups (prefix doubling restricted to them)
  SS_TRY(hipMalloc(&d_pos.p, size_t(cap) * 8));
  SS_TRY(hipMalloc(&d_grp.p, size_t(cap) * 8));
  uint64_t* pos = static_cast<uint64_t*>(d_pos.p);
  uint64_t* grp = static_cast<uint64_t*>(d_grp.p);
  int64_t h = k;
  for (int round = 0; round < 64; round++) {
    // collect tied positions in increasing order, chunk by chunk
    int64_t T = 0;
    for (int64_t base = 0; base < n; base += sel_chunk) {
      const int64_t cnt = std::min<int64_t>(sel_chunk, n - base);
      hipLaunchKernelGGL(tied_flags_kernel, grid_for(cnt), dim3(256), 0, nullptr, SA, ISA, n, base, cnt, static_cast<uint8_t*>(d_flag.p));
      size_t tb = tmp_bytes;
      // selected values written to keys2 (scratch), then appended to pos
      SS_TRY(rocprim::select(d_tmp.p, tb, rocprim::counting_iterator<uint64_t>(uint64_t(base)), static_cast<uint8_t*>(d_flag.p), keys2,
                             static_cast<unsigned long long*>(d_cnt.p), size_t(cnt), static_cast<hipStream_t>(nullptr)));
      unsigned long long got = 0;
      SS_TRY(hipMemcpy(&got, d_cnt.p, 8, hipMemcpyDeviceToHost));
      if (dbg) fprintf(stderr, "[sort] round %d chunk base %lld: tied %llu\n", round, (long long)base, got);
      if (T + int64_t(got) > cap) {
        if (err) { err->code = 3; err->msg = "suffix sorter: too many tied suffixes for the large-text sorter (text too repetitive)"; }
        return 3;
      }
      if (got) SS_TRY(hipMemcpy(pos + T, keys2, size_t(got) * 8, hipMemcpyDeviceToDevice));
      T += int64_t(got);
    }
    if (T == 0) break;
    hipLaunchKernelGGL(tied_keys_kernel, grid_for(T), dim3(256), 0, nullptr, pos, T, SA, ISA, n, h, grp, keys, vals);
    // stable LSD sort by (grp, key2): first by key2 carrying an index, then by grp
    hipLaunchKernelGGL(iota_u64_kernel, grid_for(T), dim3(256), 0, nullptr, T, head);
    size_t tb = tmp_bytes;
    SS_TRY(rocprim::radix_sort_pairs(d_tmp.p, tb, keys, keys2, head, vals2, size_t(T), 0, 64, nullptr));  // vals2 = order by key2
    hipLaunchKernelGGL(gather_u64_kernel, grid_for(T), dim3(256), 0, nullptr, grp, vals2, T, keys);        // keys = grp in that order
    tb = tmp_bytes;
    SS_TRY(rocprim::radix_sort_pairs(d_tmp.p, tb, keys, keys2, vals2, head, size_t(T), 0, 64, nullptr));   // head = final order, keys2 = grp sorted
    // materialise sorted (grp, key2, val)
    hipLaunchKernelGGL(tied_keys_kernel, grid_for(T), dim3(256), 0, nullptr, pos, T, SA, ISA, n, h, grp, keys, vals);  // recompute originals
    hipLaunchKernelGGL(gather_u64_kernel, grid_for(T), dim3(256), 0, nullptr, keys, head, T, vals2);   // vals2 = key2 sorted
    hipLaunchKernelGGL(gather_u64_kernel, grid_for(T), dim3(256), 0, nullptr, vals, head, T, keys);    // keys  = val sorted
    // keys2 = grp sorted, vals2 = key2 sorted, keys = val sorted
    hipLaunchKernelGGL(tied_heads_kernel, grid_for(T), dim3(256), 0, nullptr, keys2, vals2, pos, T, head);
    tb = tmp_bytes;
    SS_TRY(rocprim::inclusive_scan(d_tmp.p, tb, head, head, size_t(T), rocprim::maximum<uint64_t>(), nullptr));
    hipLaunchKernelGGL(tied_store_kernel, grid_for(T), dim3(256), 0, nullptr, pos, keys, head, T, SA, ISA);
    SS_TRY(hipDeviceSynchronize());
    h *= 2;
    if (round == 63) { if (err) { err->code = 6; err->msg = "suffix sort did not converge"; } return 6; }
  }
  SS_TRY(hipGetLastError());
  SS_TRY(hipMemcpy(sa_out->data(), SA, size_t(n) * 8, hipMemcpyDeviceToHost));
  return 0;
}

}  // namespace femto_amd
