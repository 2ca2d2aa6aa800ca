// ind_kernels.hip.hpp -- per-character rank lines for byte alphabets: ONE memory line per Occ of a backward-search step.
//
// The two-level lines of pack2_kernels.hip.hpp answer Occ(c,row) with two DEPENDENT lines (rank of the high digit, then
// the count of the low digit): two memory round trips per search step, the structural cost of cfg 3 in round 1.  A
// backward-search step, however, knows its character: it only needs rank_c(row) of ONE character's indicator bit vector.
// When HBM allows, open therefore also derives, for every character c of the text, the plain bit vector
// B_c[row] = (L[row] == c) cut into 128-byte lines of 960 rows:
//
//   dword 0,1     C[c] + Occ(c, rows before this line)        (40 bits)
//   dword 2..31   bit i of dword 2+k = B_c[960*line + 32k + i]
//
// sigma * rows / 7.5 bytes (13.7 GB for a 2^30-row text with 96 characters -- "size everything for 288 GB").  A search
// step is then two INDEPENDENT line reads (one when both range ends fall into the same 960 rows): half the lines and half
// the latency of the two-level layout.  LF steps (locate walks, the text build) do not know their character in advance
// and keep using the two-level lines; with the full suffix array resident there are none left on the query path.
// Same results as every other layout (tests compare them all against the reference's goldens).
#pragma once

namespace femto_amd {

constexpr int kIndRows = 960;

struct IndLine { uint32_t w[32]; };

__device__ __forceinline__ void ind_load(const uint32_t* __restrict__ ind, uint64_t line, IndLine& L) {
  const uint4* lp = reinterpret_cast<const uint4*>(ind + line * 32);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint4 v = lp[k];
    L.w[4 * k] = v.x; L.w[4 * k + 1] = v.y; L.w[4 * k + 2] = v.z; L.w[4 * k + 3] = v.w;
  }
}

// C[c] + Occ(c, row) for row = 960*line + b: the line's count + set bits among its first b+1
__device__ __forceinline__ int64_t ind_rank(const IndLine& L, uint32_t b) {
  uint32_t cnt = 0;
  const int nb = int(b) + 1;
#pragma unroll
  for (int k = 0; k < 30; k++) {
    const int bits = nb - 32 * k;
    const uint32_t m = bits >= 32 ? ~0u : (bits <= 0 ? 0u : ((1u << bits) - 1u));
    cnt += uint32_t(__popc(L.w[2 + k] & m));
  }
  return int64_t((uint64_t(L.w[1] & 0xffu) << 32) | L.w[0]) + int64_t(cnt);
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
  const bool other = haveF && lineF != lineL;
  IndLine LL, LF;
  ind_load(ix.ind, base + lineL, LL);
  trace_touch(ix, kTraceInd, base + lineL);
  if (other) {
    ind_load(ix.ind, base + lineF, LF);
    trace_touch(ix, kTraceInd, base + lineF);
  }
  const int64_t nl = ind_rank(LL, bL);
  int64_t nf;
  if (!haveF) nf = ix.p2_c[code];
  else if (other) nf = ind_rank(LF, bF);
  else nf = ind_rank(LL, bF);
  first = nf;
  last = nl - 1;
}

// construction: one block per 960 rows; thread c builds character c's line from the block's symbols (LDS) and takes
// the count before the line from the two-level lines
__global__ __launch_bounds__(256) void ind_build_kernel(const DevIndex ix, const int64_t nrows, const uint16_t* __restrict__ sym,
                                                        uint32_t* __restrict__ ind, const int64_t stride, const int64_t group0) {
  __shared__ uint16_t s_sym[kIndRows];
  const int64_t g = group0 + int64_t(blockIdx.x);
  const int64_t row0 = g * kIndRows;
  for (int i = threadIdx.x; i < kIndRows; i += blockDim.x) s_sym[i] = row0 + i < nrows ? uint16_t(sym[row0 + i] & 0x7fffu) : uint16_t(0xffffu);
  __syncthreads();
  const uint32_t c = threadIdx.x;
  if (int(c) >= ix.p2_sigma) return;
  uint32_t w[32];
  const int64_t before = row0 == 0 ? ix.p2_c[c] : p2_c_plus_occ(ix, c, row0 - 1);
  w[0] = uint32_t(uint64_t(before));
  w[1] = uint32_t(uint64_t(before) >> 32) & 0xffu;
#pragma unroll 1
  for (int k = 0; k < 30; k++) {
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) v |= (uint32_t(s_sym[32 * k + i]) == c ? 1u : 0u) << i;
    w[2 + k] = v;
  }
  uint4* dst = reinterpret_cast<uint4*>(ind + (uint64_t(c) * uint64_t(stride) + uint64_t(g)) * 32);
#pragma unroll
  for (int k = 0; k < 8; k++) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}

}  // namespace femto_amd
