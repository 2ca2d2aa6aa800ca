// ind_kernels.hip.hpp -- per-character rank lines for byte alphabets: ONE memory line per Occ of a backward-search step.
//
// The two-level lines of pack2_kernels.hip.hpp answer Occ(c,row) with two DEPENDENT lines (rank of the high digit, then
// the count of the low digit): two memory round trips per search step, the structural cost of cfg 3 in round 1.  A
// backward-search step, however, knows its character: it only needs rank_c(row) of ONE character's indicator bit vector.
// When HBM allows, open therefore also derives, for every character c of the text, the plain bit vector
// B_c[row] = (L[row] == c) cut into 128-byte lines of 896 rows = a 16-byte header + seven 16-byte blocks of 128 rows:
//
//   qword 0       bits 0..39: C[c] + Occ(c, rows before this line)
//   qword 1       six 10-bit fields: set bits in blocks 0..k-1 of this line, k = 1..6
//   qword 2 + 2k  block k: bit i of its 128 bits = B_c[896*line + 128k + i]
//
// A rank is TWO 16-byte loads from ONE line -- the header and the block of the row -- and two masked popcounts.
// (Round 3's first layout had 960 rows per line, an 8-byte head and three sub-blocks of five words: six 8-byte loads per
// rank, twelve per search step.  The count kernel on the sigma~96 workload turned out to be bound by the CU's vector-memory
// ADDRESS rate -- a 64-lane load whose lanes hit 64 different lines occupies the texture addresser for ~64 cycles whatever
// its width: 134 load instructions per wavefront x 64 lanes x 156 k wavefronts / 256 CUs = 5.2 M cycles = the kernel's
// 2 ms -- so the number of load instructions per step matters, not the bytes.)
//
// sigma * rows / 7 bytes (14.7 GB for a 2^30-row text with 96 characters -- "size everything for 288 GB").  A search
// step is then two INDEPENDENT line reads (one when both range ends fall into the same 896 rows): half the lines and half
// the latency of the two-level layout.  LF steps (locate walks, the text build) do not know their character in advance
// and keep using the two-level lines; with the full suffix array resident there are none left on the query path.
// Same results as every other layout (tests compare them all against the reference's goldens).
#pragma once

namespace femto_amd {

constexpr int kIndRows = 896;

// C[c] + Occ(c, row) for row = 896*line + b, from the line's header h and the row's block v
__device__ __forceinline__ int64_t ind_rank_of(const uint4 h, const uint4 v, uint32_t b) {
  const uint32_t blk = b >> 7;               // block 0..6
  const int nb = int(b & 127u) + 1;          // bits of the block to count: 1..128
  const uint64_t lo = uint64_t(v.x) | (uint64_t(v.y) << 32), hi = uint64_t(v.z) | (uint64_t(v.w) << 32);
  const uint64_t mlo = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
  const uint64_t mhi = nb <= 64 ? 0ull : (nb >= 128 ? ~0ull : ((1ull << (nb - 64)) - 1ull));
  const uint32_t cnt = uint32_t(__popcll(lo & mlo)) + uint32_t(__popcll(hi & mhi));
  const uint32_t fields = blk <= 3u ? h.z : ((h.z >> 30) | (h.w << 2));      // the 10-bit fields of blocks 1..3 / from block 4 on
  const uint32_t sub = blk == 0u ? 0u : (fields >> (10u * (blk <= 3u ? blk - 1u : blk - 4u))) & 0x3ffu;
  const uint64_t base = (uint64_t(h.x) | (uint64_t(h.y) << 32)) & ((1ull << 40) - 1ull);
  return int64_t(base) + int64_t(sub + cnt);
}
__device__ __forceinline__ int64_t ind_rank(const uint32_t* __restrict__ ind, uint64_t line, uint32_t b) {
  const uint4* lp = reinterpret_cast<const uint4*>(ind + line * 32);
  return ind_rank_of(lp[0], lp[1 + (b >> 7)], b);
}

__device__ __forceinline__ void ind_split(int64_t row, uint64_t* line, uint32_t* b) {
  const uint32_t q = uint32_t(uint64_t(row) >> 7);   // rows < 2^38
  const uint32_t l = q / 7u;
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
  // four independent 16-byte loads at most -- and only the ones that differ: the two ends of a narrow range share the
  // line (its header) and often the block, and a load that a lane does not issue costs the address unit nothing.  ALL of
  // them are issued before any is waited for: the first end's header / block registers are defined by their own loads only
  // and the three possible ranks (same block, other block of the line, other line) are selected as numbers.  (Written as
  // `hF = hL; if (other_line) hF = load;` the compiler waited for hL in order to copy it -- up to three dependent round trips
  // per step instead of one; found in round 4 on the rank units, whose first version did the same.)
  const uint4* const lpL = reinterpret_cast<const uint4*>(ix.ind + (base + lineL) * 32);
  const bool other_line = haveF && lineF != lineL;
  const bool other_blk = haveF && (other_line || (bF >> 7) != (bL >> 7));
  const uint4* const lpF = reinterpret_cast<const uint4*>(ix.ind + (base + (other_line ? lineF : lineL)) * 32);
  // (the compiler closes every conditional block that loads with a wait for everything in flight: so the unconditional loads
  // go first and ONE conditional block issues whatever else the lane needs -- its header is loaded again when only the block
  // differs: the same address as hL, an L1 hit, cheaper than a second round trip)
  const uint4 hL = lpL[0], vL = lpL[1 + (bL >> 7)];
  uint4 hF, vF;
  if (other_blk) {
    vF = lpF[1 + (bF >> 7)];
    hF = lpF[0];
  }
  const int64_t nl = ind_rank_of(hL, vL, bL);
  int64_t rf = 0;                                   // (first == 0: unused)
  if (other_blk) rf = ind_rank_of(hF, vF, bF);
  else if (haveF) rf = ind_rank_of(hL, vL, bF);
  first = haveF ? rf : ix.p2_c[code];
  last = nl - 1;
}

// construction: one workgroup per 896 rows; thread c builds character c's line from the rows' symbols (LDS) and takes
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
  uint64_t fields = 0;
  uint32_t run = 0;
#pragma unroll 1
  for (int k = 0; k < 14; k++) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) v |= uint64_t(uint32_t(s_sym[64 * k + i]) == c ? 1u : 0u) << i;
    dst[2 + k] = v;
    run += uint32_t(__popcll(v));
    if ((k & 1) && k < 13) fields |= uint64_t(run) << (10 * (k >> 1));      // set bits in blocks 0 .. k/2
  }
  dst[0] = uint64_t(before) & ((1ull << 40) - 1ull);
  dst[1] = fields;
}

}  // namespace femto_amd
