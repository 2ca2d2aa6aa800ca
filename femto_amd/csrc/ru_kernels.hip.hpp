// ru_kernels.hip.hpp -- rank units: ONE 16-byte load per Occ of a backward-search step on a small alphabet.
//
// A backward-search step (do_string_query, src/main/server.c:909-936) knows its character c: it needs rank_c at the two
// ends of the range and nothing else.  The packed lines of pack_kernels.hip.hpp answer that from one 128-byte line, but
// with SIX load instructions per range end (four plane pieces + a count word + its high byte), and a scattered load
// costs the CU's address unit ~64 cycles per instruction whatever its width, and one address translation per lane.
// For every TABLE character c (a character of the text that is not <= SEOF) open therefore also derives the plain bit
// vector B_c[row] = (L[row] == c), cut into self-contained 16-byte units of 88 rows:
//
//   bits   0.. 39   C[c] + Occ(c, rows before this unit)          (40 bits: the format's 2^39-row limit)
//   bits  40..127   bit 40 + i = B_c[88 * unit + i]
//
// so that  C[c] + Occ(c, row) = count + popcount(bits 40 .. 40 + row % 88)  is ONE load and two masked popcounts, and
// the two ends of a narrow range are usually the SAME unit (one load per step).  Eight units share a 128-byte line
// (704 rows).  (table characters) x rows / 5.5 bytes: 0.78 GB for a 2^30-row DNA index -- less than the packed lines,
// which stay: an LF step (locate walk, text build) does not know its character and needs L[row] and the mark plane.
// Patterns' stop characters (<= SEOF inside a pattern) are answered from a sorted list of their rows.  Same results as every other
// layout (the tests compare them all with the reference's goldens); measured in profiles/r04_*.
#pragma once

namespace femto_amd {

constexpr int kRuRows = 88;

__device__ __forceinline__ void ru_split(int64_t row, uint64_t* unit, uint32_t* r) {
  const uint32_t q = uint32_t(uint64_t(row) >> 3);   // rows < 2^35 (checked at open)
  const uint32_t u = q / 11u;
  *unit = u;
  *r = uint32_t(uint64_t(row) - uint64_t(u) * kRuRows);
}

// C[c] + Occ(c, row) for row = 88 * unit + r
__device__ __forceinline__ int64_t ru_rank_of(const uint4 v, uint32_t r) {
  const uint64_t lo = uint64_t(v.x) | (uint64_t(v.y) << 32), hi = uint64_t(v.z) | (uint64_t(v.w) << 32);
  const int nb = int(r) + 1;                       // bits to count: 1..88; the first 24 live in lo's top bits
  const uint64_t mlo = nb >= 24 ? 0xffffffull : ((1ull << nb) - 1ull);
  const uint64_t mhi = nb <= 24 ? 0ull : (nb >= 88 ? ~0ull : ((1ull << (nb - 24)) - 1ull));
  const uint32_t cnt = uint32_t(__popcll((lo >> 40) & mlo)) + uint32_t(__popcll(hi & mhi));
  return int64_t(lo & ((1ull << 40) - 1ull)) + int64_t(cnt);
}

// A STOP character inside a pattern (a character <= SEOF: one row per document holds it) has no unit vector; the rows
// whose L is that character are listed in row order (ru_stop_rows), and C[c] + Occ(c,row) is C[c] + the number of listed
// rows <= row: a binary search.  (Falling back to the packed lines here put their sixteen plane registers into the
// count kernel: 70 instead of 43 VGPRs, seven instead of eight waves per SIMD for a case that hardly ever runs.)
__device__ __forceinline__ void ru_stop_step(const DevIndex& ix, uint32_t code, int64_t& first, int64_t& last) {
  const int64_t* const rows = ix.ru_stop_rows + ix.ru_stop_off[code];
  const int64_t n = int64_t(ix.ru_stop_off[code + 1]) - int64_t(ix.ru_stop_off[code]);
  auto count_le = [&](int64_t row) -> int64_t {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (rows[mid] <= row) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int64_t c = ix.pack_c[code];
  const int64_t nl = c + count_le(last);
  first = first == 0 ? c : c + count_le(first - 1);
  last = nl - 1;
}

// one step of the backward search with dense code `code` (server.c:909-936): [first,last] -> rows preceded by code
__device__ __forceinline__ void ru_search_step(const DevIndex& ix, int j, uint32_t code, int64_t& first, int64_t& last) {
  if (j == 0) {
    first = ix.pack_c[code];
    last = ix.pack_c[8 + code];
    return;
  }
  if (code < uint32_t(ix.ru_nstop)) {    // a stop character inside a pattern: no unit vector, the packed lines answer
    ru_stop_step(ix, code, first, last);
    return;
  }
  const uint4* const ub = reinterpret_cast<const uint4*>(ix.ru);
  const uint4* const up = ub + uint64_t(code - uint32_t(ix.ru_nstop)) * uint64_t(ix.ru_stride);
  uint64_t uL, uF = 0;
  uint32_t rL, rF = 0;
  ru_split(last, &uL, &rL);
  const bool haveF = first != 0;
  if (haveF) ru_split(first - 1, &uF, &rF);
  // Both ends of a narrow range usually share the unit: the second load is issued only by the lanes that need it -- and
  // BEFORE the first one is waited for (the compiler closes a conditional block that loads with a wait for everything in
  // flight: the unconditional load goes first).  (Written as `vF = vL; if (other) vF = up[uF];` the compiler waited for vL to copy
  // it, so every wavefront with one such lane -- half of them at 64 lanes x 1/88 -- paid two dependent round trips per step,
  // and split the second load in two: 124 M instead of 87 M memory requests per launch on 10 M sampled 20-mers.)
  const bool other = haveF && uF != uL;
  const uint4 vL = up[uL];
  uint4 vF;
  if (other) vF = up[uF];
  trace_touch(ix, kTraceRu, uint64_t(up + uL - ub) >> 3);
  if (other) trace_touch(ix, kTraceRu, uint64_t(up + uF - ub) >> 3);
  const int64_t nl = ru_rank_of(vL, rL);
  const int64_t nfL = ru_rank_of(vL, rF), nfF = other ? ru_rank_of(vF, rF) : 0;
  first = haveF ? (other ? nfF : nfL) : ix.pack_c[code];
  last = nl - 1;
}

// ---- MARKED rank units (handles that do not hold the suffix array: every located row otherwise costs a walk to a mark) ----
//
// The same bit vector B_c, cut into 16-byte units of 64 rows that also say which of the rows HOLDING c are marked:
//
//   bits   0.. 39   C[c] + Occ(c, rows before this unit)
//   bits  40.. 63   bit 40 + i = the i-th row of this unit with B_c set is a marked row (i < 24; later ones: not recorded)
//   bits  64..127   bit 64 + i = B_c[64 * unit + i]
//
// Why: a pattern that occurs spends the last steps of its search on ONE row, and those rows are consecutive text positions
// (step j's row is the suffix one symbol longer).  The step loads the unit of (c, row) anyway; the unit now also says
// whether that row is marked.  A marked row met s symbols before the end fixes the pattern's text position --
// SA[final row] = SA[marked row] - s -- so the row expansion starts its "walk" on a marked row (one packed line + one offset)
// instead of walking ~period / 2 lines from the final row.  With marks every 5th position five one-row steps always meet
// one.  Measured with the rows' own packed lines doing this (profiles/tuning_history.md, round 5 "mark spotting"): the
// saving was real but the 128-byte line steps cost twice a unit step; here the search step stays ONE 16-byte load.
// 64 instead of 88 rows per unit: (table characters) x rows / 4 bytes, 1.07 GB for a 2^30-row DNA index.
constexpr int kRumRows = 64;

__device__ __forceinline__ int64_t rum_rank_of(const uint4 v, uint32_t r) {     // C[c] + Occ(c, 64 * unit + r)
  const uint64_t lo = uint64_t(v.x) | (uint64_t(v.y) << 32), hi = uint64_t(v.z) | (uint64_t(v.w) << 32);
  const uint64_t m = r >= 63u ? ~0ull : ((2ull << r) - 1ull);
  return int64_t(lo & ((1ull << 40) - 1ull)) + int64_t(__popcll(hi & m));
}
// is row 64 * unit + r a row holding c AND marked?
__device__ __forceinline__ bool rum_marked(const uint4 v, uint32_t r) {
  const uint64_t hi = uint64_t(v.z) | (uint64_t(v.w) << 32);
  if (!((hi >> r) & 1ull)) return false;
  const uint32_t idx = uint32_t(__popcll(hi & ((1ull << r) - 1ull)));
  return idx < 24u && ((v.y >> (8u + idx)) & 1u);
}

// ru_search_step on marked units; *spotted: the range was ONE row on entry, that row holds `code` and is marked
__device__ __forceinline__ void rum_search_step(const DevIndex& ix, int j, uint32_t code, int64_t& first, int64_t& last, bool* spotted) {
  *spotted = false;
  if (j == 0) {
    first = ix.pack_c[code];
    last = ix.pack_c[8 + code];
    return;
  }
  if (code < uint32_t(ix.ru_nstop)) {
    ru_stop_step(ix, code, first, last);
    return;
  }
  const uint4* const ub = reinterpret_cast<const uint4*>(ix.ru);
  const uint4* const up = ub + uint64_t(code - uint32_t(ix.ru_nstop)) * uint64_t(ix.ru_stride);
  const uint64_t uL = uint64_t(last) >> 6;
  const uint32_t rL = uint32_t(last) & 63u;
  const bool haveF = first != 0;
  const uint64_t uF = haveF ? uint64_t(first - 1) >> 6 : 0;
  const uint32_t rF = haveF ? uint32_t(first - 1) & 63u : 0u;
  const bool other = haveF && uF != uL;       // (the load order: see ru_search_step)
  const bool single = first == last;
  const uint4 vL = up[uL];
  uint4 vF;
  if (other) vF = up[uF];
  trace_touch(ix, kTraceRu, uint64_t(up + uL - ub) >> 3);
  if (other) trace_touch(ix, kTraceRu, uint64_t(up + uF - ub) >> 3);
  const int64_t nl = rum_rank_of(vL, rL);
  const int64_t nfL = rum_rank_of(vL, rF), nfF = other ? rum_rank_of(vF, rF) : 0;
  *spotted = single && rum_marked(vL, rL);
  first = haveF ? (other ? nfF : nfL) : ix.pack_c[code];
  last = nl - 1;
}

// construction of the marked units: one thread per unit, all table characters; the packed lines are final (marks densified)
inline __global__ __launch_bounds__(256) void rum_build_kernel(const DevIndex ix, const int64_t nrows, const uint8_t* __restrict__ sym,
                                                        uint4* __restrict__ ru, const int64_t stride, const int nstop, const int ntab) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= stride) return;
  const int64_t row0 = u * kRumRows;
  uint64_t bits[8], marks[8];
  uint32_t seen[8];
#pragma unroll
  for (int t = 0; t < 8; t++) { bits[t] = marks[t] = 0; seen[t] = 0; }
  for (int i = 0; i < kRumRows; i++) {
    const int64_t row = row0 + i;
    if (row >= nrows) break;
    uint64_t ln;
    uint32_t r;
    pack_split(row, &ln, &r);
    const bool marked = (ix.pack[ln * kPackLineWords + 15u + (r >> 5)] >> (r & 31u)) & 1u;     // the line's mark plane
    const int t = int(sym[row] & 0x7fu) - nstop;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (t == k) {
        bits[k] |= 1ull << i;
        if (marked && seen[k] < 24u) marks[k] |= 1ull << (40u + seen[k]);
        seen[k]++;
      }
  }
  uint64_t bline = 0;
  uint32_t br = 0;
  PackPlanes P;
  if (row0 > 0 && row0 <= nrows) {
    pack_split(row0 - 1, &bline, &br);
    pack_load_planes(ix.pack, bline, P);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k >= ntab) break;
    const uint32_t code = uint32_t(k + nstop);
    int64_t before;
    if (row0 == 0) before = ix.pack_c[code];
    else if (row0 > nrows) before = ix.pack_c[8 + code] + 1;
    else before = pack_base(ix.pack, bline, code) + int64_t(pack_match(P, code, br + 1));
    const uint64_t lo = (uint64_t(before) & ((1ull << 40) - 1ull)) | marks[k];
    ru[uint64_t(k) * uint64_t(stride) + uint64_t(u)] = make_uint4(uint32_t(lo), uint32_t(lo >> 32), uint32_t(bits[k]), uint32_t(bits[k] >> 32));
  }
}

// construction: one thread per unit and table character, from the rows' dense codes (sym, as pack_extract_kernel left
// them) and the packed lines (the count before the unit)
inline __global__ __launch_bounds__(256) void ru_build_kernel(const DevIndex ix, const int64_t nrows, const uint8_t* __restrict__ sym,
                                                       uint4* __restrict__ ru, const int64_t stride, const int nstop, const int ntab) {
  const int64_t u = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= stride) return;
  const int64_t row0 = u * kRuRows;
  // the unit's 88 codes: 5.5 aligned 16-byte pieces of sym
  uint64_t blo[8], bhi[8];
#pragma unroll
  for (int t = 0; t < 8; t++) blo[t] = bhi[t] = 0;
  for (int i = 0; i < kRuRows; i++) {
    const int64_t row = row0 + i;
    if (row >= nrows) break;
    const int t = int(sym[row] & 0x7fu) - nstop;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (t == k) {
        if (i < 24) blo[k] |= 1ull << (40 + i);
        else bhi[k] |= 1ull << (i - 24);
      }
  }
  uint64_t line = 0;
  uint32_t r = 0;
  PackPlanes P;
  if (row0 > 0 && row0 <= nrows) {
    pack_split(row0 - 1, &line, &r);
    pack_load_planes(ix.pack, line, P);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k >= ntab) break;
    const uint32_t code = uint32_t(k + nstop);
    int64_t before;
    if (row0 == 0) before = ix.pack_c[code];
    else if (row0 > nrows) before = ix.pack_c[8 + code] + 1;     // past the end: C[c] + all occurrences of c (never read)
    else before = pack_base(ix.pack, line, code) + int64_t(pack_match(P, code, r + 1));
    const uint64_t lo = (uint64_t(before) & ((1ull << 40) - 1ull)) | blo[k];
    ru[uint64_t(k) * uint64_t(stride) + uint64_t(u)] = make_uint4(uint32_t(lo), uint32_t(lo >> 32), uint32_t(bhi[k]), uint32_t(bhi[k] >> 32));
  }
}

// the rows whose L is a stop character, per character in row order: one thread per packed line; scans[c * stride + line] =
// rows with code c before the line (the exclusive scans build_pack made for the line counts)
inline __global__ __launch_bounds__(256) void ru_stop_rows_kernel(const DevIndex ix, const int64_t nlines, const int64_t nrows,
                                                            const uint8_t* __restrict__ sym, const int64_t* __restrict__ scans,
                                                            const int64_t stride, const int nstop, int64_t* __restrict__ out) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  int64_t at[3];
  for (int c = 0; c < 3; c++) at[c] = c < nstop ? int64_t(ix.ru_stop_off[c]) + scans[int64_t(c) * stride + line] : 0;
  for (int i = 0; i < kPackRows; i++) {
    const int64_t row = line * kPackRows + i;
    if (row >= nrows) break;
    const int c = int(sym[row] & 0x7fu);
    if (c < nstop) {
      const int64_t slot = c == 0 ? at[0]++ : (c == 1 ? at[1]++ : at[2]++);
      if (slot >= int64_t(ix.ru_stop_off[c]) && slot < int64_t(ix.ru_stop_off[c + 1])) out[slot] = row;   // (a damaged index: L disagrees with C)
    }
  }
}

}  // namespace femto_amd
