// query_sort.hip -- orders a pattern batch by pattern SUFFIX before the backward search.
//
// Backward search consumes a pattern from its last symbol, so patterns that share a suffix visit
// the same rows, buckets and segments for their first steps.  Sorting the batch by its last
// symbols (as many as fit a 64-bit key at ceil(log2(sigma+1)) bits each) puts such patterns in neighbouring lanes: their loads fall on the same cache lines (or the
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

// key = the pattern's last `nsym` = 63/bits symbols, the last one in the most significant field, each mapped through
// `dense` to `bits` bits: dense[ch] = 1 + rank of ch among the characters that occur in the indexed text, 0 for
// "pattern exhausted" and for characters that do not occur (such patterns die at once anyway).  The fields sit at
// the top of the key; bit 0 is a flag: 1 = the key describes the WHOLE pattern (it has at most nsym symbols, all of
// them characters of the text), so a kernel that works on dense codes never has to read the pattern itself.  For
// ACGT texts bits = 3: a whole 20-mer fits the key and the batch is fully suffix-sorted.
__global__ __launch_bounds__(256) void suffix_key_kernel(const int64_t npats, const int32_t* __restrict__ plen,
                                                         const uint16_t* __restrict__ pats, const int64_t* __restrict__ starts,
                                                         const uint8_t* __restrict__ dense, const int bits, const int nsym,
                                                         uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
  __shared__ uint8_t s_dense[264];
  for (int i = threadIdx.x; i < 261; i += blockDim.x) s_dense[i] = dense[i];
  __syncthreads();
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= npats) return;
  const int len = plen[q];
  const int take = len < nsym ? len : nsym;
  uint64_t key = 0;
  bool whole = len <= nsym;
  if (take > 0) {
    // the pattern's tail is read as aligned 8-byte words, last word first (4 symbols per load instead of one);
    // a word never crosses a page, so touching the few bytes around the pattern inside its first/last word is safe
    const uintptr_t last_sym = reinterpret_cast<uintptr_t>(pats + starts[q] + len) - 2;
    uintptr_t wa = last_sym & ~uintptr_t(7);
    int s_in_word = int((last_sym - wa) >> 1);
    int got = 0;
    while (got < take) {
      const uint64_t w = *reinterpret_cast<const uint64_t*>(wa);
      for (int s = s_in_word; s >= 0 && got < take; s--, got++) {
        const uint32_t ch = uint32_t(w >> (16 * s)) & 0xffffu;
        const uint64_t c = ch < 261u ? s_dense[ch] : 0;
        whole = whole && c != 0;
        key = (key << bits) | c;
      }
      wa -= 8;
      s_in_word = 3;
    }
  }
  key <<= bits * (nsym - take);
  keys[q] = (key << (64 - nsym * bits)) | (whole ? 1u : 0u);
  idx[q] = uint32_t(q);
}

static int clamp_sort_syms(int bits, int sort_syms) {
  const int nsym = 63 / bits;
  return (sort_syms > nsym || sort_syms <= 0) ? nsym : sort_syms;
}

// first key bit that takes part in the sort.  Batches of up to 2^20 patterns go through rocPRIM's merge-sort
// path, which returned a non-permutation for a partial bit range on this ROCm (7.2.0, observed on gfx950 with
// 4096..100000 keys and begin_bit 37..43): they are sorted on whole keys -- at that size the passes cost nothing.
static unsigned sort_begin_bit(int64_t npats, int bits, int sort_syms) {
  if (npats <= (int64_t(1) << 20)) return 0;
  int nbits = clamp_sort_syms(bits, sort_syms) * bits;
  if (nbits > 8 && (nbits & 7) != 0 && (nbits & 7) <= 4) nbits &= ~7;  // a radix pass for <= 4 bits of the last symbols is not worth it
  return unsigned(64 - nbits);
}

// temporary storage of the sort for exactly the bit range query_sort() will use (rocPRIM picks its algorithm,
// and with it the storage it needs, from the size and the bit range)
size_t query_sort_temp_bytes(int64_t npats, int bits, int sort_syms) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, static_cast<uint64_t*>(nullptr), static_cast<uint64_t*>(nullptr),
                                  static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), size_t(npats),
                                  sort_begin_bit(npats, bits, sort_syms), 64, nullptr);
  return bytes;
}

// keys/keys2: npats u64 each; idx/idx2: npats u32 each; tmp: query_sort_temp_bytes(npats).
// On return idx2 holds the processing order (a permutation of 0..npats-1) and keys2 the keys in that order.
// Only the top `sort_syms` symbols take part in the sort: once sigma^s exceeds the batch size, patterns that agree
// on s symbols are (almost) alone, and deeper order buys no locality -- fewer radix passes.
hipError_t query_sort(int64_t npats, const int32_t* d_plen, const uint16_t* d_pats, const int64_t* d_starts,
                      const uint8_t* d_dense, int bits, int sort_syms, uint64_t* keys, uint64_t* keys2, uint32_t* idx,
                      uint32_t* idx2, void* tmp, size_t tmp_bytes, hipStream_t stream) {
  const int nsym = 63 / bits;
  hipLaunchKernelGGL(suffix_key_kernel, dim3(uint32_t((npats + 255) / 256)), dim3(256), 0, stream, npats, d_plen, d_pats,
                     d_starts, d_dense, bits, nsym, keys, idx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, idx, idx2, size_t(npats), sort_begin_bit(npats, bits, sort_syms), 64,
                                   stream);
}

}  // namespace femto_amd
