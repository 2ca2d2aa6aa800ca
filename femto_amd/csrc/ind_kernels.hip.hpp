// ind_kernels.hip.hpp -- per-character rank lines for byte alphabets: ONE memory line per Occ of a backward-search step.
//
// The two-level lines of pack2_kernels.hip.hpp answer Occ(c,row) with two DEPENDENT lines (rank of the high digit, then
// the count of the low digit): two memory round trips per search step, the structural cost of cfg 3 in round 1.  A
// backward-search step, however, knows its character: it only needs rank_c(row) of ONE character's indicator bit vector.
// When HBM allows, open therefore also derives, for every character c of the text, the plain bit vector
// B_c[row] = (L[row] == c) cut into 128-byte lines of 960 rows:
//
//   qword 0       bits 0..39: C[c] + Occ(c, rows before this line); bits 40..49 / 50..59: set bits in qwords 1..5 /
//                 1..10 of this line (two sub-block counts: a rank popcounts at most five words)
//   qword 1..15   bit i of qword 1+k = B_c[960*line + 64k + i]
//
// sigma * rows / 7.5 bytes (13.7 GB for a 2^30-row text with 96 characters -- "size everything for 288 GB").  A search
// step is then two INDEPENDENT line reads (one when both range ends fall into the same 960 rows): half the lines and half
// the latency of the two-level layout.  LF steps (locate walks, the text build) do not know their character in advance
// and keep using the two-level lines; with the full suffix array resident there are none left on the query path.
// Same results as every other layout (tests compare them all against the reference's goldens).
#pragma once

namespace femto_amd {

constexpr int kIndRows = 960;

// C[c] + Occ(c, row) for row = 960*line + b.  Only what the rank needs is loaded: the head word and the five words of
// b's sub-block (the 128-byte line is one memory request either way, but 48 instead of 128 bytes reach the registers and a
// rank costs ~45 VALU instructions instead of ~150 -- the round-2 profile showed the whole-line version VALU-bound).
__device__ __forceinline__ int64_t ind_rank(const uint32_t* __restrict__ ind, uint64_t line, uint32_t b) {
  const uint64_t* lp = reinterpret_cast<const uint64_t*>(ind + line * 32);
  const uint64_t head = lp[0];
  const uint32_t w = b >> 6;                 // data word 0..14
  const uint32_t sblk = w / 5u;              // sub-block 0..2
  const uint64_t* dp = lp + 1 + 5u * sblk;
  uint64_t d[5];
#pragma unroll
  for (int k = 0; k < 5; k++) d[k] = dp[k];
  const int nb = int(b - 320u * sblk) + 1;   // bits of the sub-block to count: 1..320
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const int bits = nb - 64 * k;
    const uint64_t m = bits >= 64 ? ~0ull : (bits <= 0 ? 0ull : ((1ull << bits) - 1ull));
    cnt += uint32_t(__popcll(d[k] & m));
  }
  const uint32_t sub = sblk == 0 ? 0u : (sblk == 1 ? uint32_t(head >> 40) & 0x3ffu : uint32_t(head >> 50) & 0x3ffu);
  return int64_t(head & ((1ull << 40) - 1ull)) + int64_t(sub + cnt);
}

__device__ __forceinline__ void ind_split(int64_t row, uint64_t* line, uint32_t* b) {
  const uint32_t q = uint32_t(uint64_t(row) >> 6);   // rows < 2^38
  const uint32_t l = q / 15u;
  *line = l;
  *b = uint32_t(uint64_t(row) - uint64_t(l) * kIndRows);
}

// one step of the backward search with dense code `code` (server.c:909-936)
__device__ __forceinline__ void ind_search_step(const DevIndex& ix, int j, uint32_t code, int64_t& first, int64_t& last) {
  if (j == 0) {
    first = ix.p2_c[code];
    last = ix.p2_c[256 + code];
    return;
  }
  const uint64_t base = uint64_t(code) * uint64_t(ix.ind_stride);
  uint64_t lineL, lineF = 0;
  uint32_t bL, bF = 0;
  ind_split(last, &lineL, &bL);
  const bool haveF = first != 0;
  if (haveF) ind_split(first - 1, &lineF, &bF);
  trace_touch(ix, kTraceInd, base + lineL);
  if (haveF && lineF != lineL) trace_touch(ix, kTraceInd, base + lineF);
  const int64_t nl = ind_rank(ix.ind, base + lineL, bL);      // the two ranks are independent: their loads overlap
  const int64_t rf = ind_rank(ix.ind, base + lineF, bF);      // (first == 0: line 0 of the character, result unused)
  first = haveF ? rf : ix.p2_c[code];
  last = nl - 1;
}

// construction: one block per 960 rows; thread c builds character c's line from the block's symbols (LDS) and takes
// the count before the line from the two-level lines
inline __global__ __launch_bounds__(256) void ind_build_kernel(const DevIndex ix, const int64_t nrows, const uint16_t* __restrict__ sym,
                                                        uint32_t* __restrict__ ind, const int64_t stride, const int64_t group0) {
  __shared__ uint16_t s_sym[kIndRows];
  const int64_t g = group0 + int64_t(blockIdx.x);
  const int64_t row0 = g * kIndRows;
  for (int i = threadIdx.x; i < kIndRows; i += blockDim.x) s_sym[i] = row0 + i < nrows ? uint16_t(sym[row0 + i] & 0x7fffu) : uint16_t(0xffffu);
  __syncthreads();
  const uint32_t c = threadIdx.x;
  if (int(c) >= ix.p2_sigma) return;
  const int64_t before = row0 == 0 ? ix.p2_c[c] : p2_c_plus_occ(ix, c, row0 - 1);
  uint64_t* dst = reinterpret_cast<uint64_t*>(ind + (uint64_t(c) * uint64_t(stride) + uint64_t(g)) * 32);
  uint32_t sub1 = 0, sub2 = 0, run = 0;
#pragma unroll 1
  for (int k = 0; k < 15; k++) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) v |= uint64_t(uint32_t(s_sym[64 * k + i]) == c ? 1u : 0u) << i;
    dst[1 + k] = v;
    run += uint32_t(__popcll(v));
    if (k == 4) sub1 = run;
    if (k == 9) sub2 = run;
  }
  dst[0] = (uint64_t(before) & ((1ull << 40) - 1ull)) | (uint64_t(sub1) << 40) | (uint64_t(sub2) << 50);
}

}  // namespace femto_amd
