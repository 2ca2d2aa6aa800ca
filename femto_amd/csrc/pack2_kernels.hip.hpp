// pack2_kernels.hip.hpp -- fast path for byte alphabets (mode 4, "pack2"): up to 256 distinct characters.
//
// The generalisation of pack_kernels.hip.hpp's one-line-per-step layout to alphabets that do not fit three bits:
// a two-level 16-ary decomposition of the dense character code c = 16*h + l (a wavelet tree of arity 16 and
// depth 2 in place of femto's binary Huffman-shaped tree of RLE/literal sequences, src/main/wtree.c):
//
//   level 1, one 128-byte line per 64 rows:
//     dword  0.. 7  four bit planes of h for rows 64*line + [0,64)      (plane p = dwords 2p, 2p+1)
//     dword  8.. 9  1 = the row is marked (its offset is in pack_sa)
//     dword 10..25  low 32 bits of  #rows before this line whose h == k            k = 0..15
//     dword 26..29  bits 32..39 of those counts (four per dword)
//     dword 30, 31  number of marked rows before this line (low 32 bits, bits 32..39)
//   level 2: for every h the l-digits of the rows with that h, in row order, each h starting on a line
//   boundary (p2_base[h] = its first line); one 128-byte line per 96 positions:
//     dword  0..11  four bit planes of l                                   (plane p = dwords 3p..3p+2)
//     dword 12..27  low 32 bits of  C[ch(16h+k)] + #earlier positions of this h whose l == k   k = 0..15
//     dword 28..31  bits 32..39 of those
//
// Round 5, a FREQUENCY-SHAPED first level: the split is not c = 16 h + l for every code.  The S codes behind the stop characters
// -- the text's most frequent characters: open assigns the dense codes of a byte alphabet by falling frequency -- get a level-1
// class of their own (h = code - nstop, no level 2 at all: C[c] + the class's rank IS C + Occ), the others share the remaining
// 16 - S classes sixteen to a class (p2_hl).  S is the largest number for which all characters still fit: 10 for 96 characters,
// 0 for 256 (then this is the plain 16 x 16 split).  On an English-like text three quarters of all search steps and LF steps are
// then ONE memory line instead of two dependent ones, and level 2 shrinks to the rare characters' share (3.4 -> 2.5 GB at 1 GiB).
//
// Occ(c,row): rank of h among rows <= row from ONE level-1 line, then the l-count from ONE level-2 line -- two
// memory lines instead of ~4.5 wavelet levels of one or two lines each plus Elias-gamma decoding; an LF step of
// locate (character of the row, its Occ, the mark test) is the same two lines.  Derived at open on the GPU from the
// uploaded femto blocks (lane kernels); results identical to every other mode (tests compare all of them).
#pragma once

namespace femto_amd {

constexpr int kP2Rows1 = 64;
constexpr int kP2Rows2 = 96;

// dense code -> (level-1 class, digit inside it); S = DevIndex::p2_single, nstop = p2_stop_below
__host__ __device__ inline void p2_hl(uint32_t code, uint32_t S, uint32_t nstop, uint32_t* h, uint32_t* l) {
  if (code - nstop < S) {      // (unsigned: false for code < nstop)
    *h = code - nstop;
    *l = 0;
    return;
  }
  const uint32_t q = code < nstop ? code : code - S;
  *h = S + (q >> 4);
  *l = q & 15u;
}
__host__ __device__ inline uint32_t p2_code_of_hl(uint32_t h, uint32_t l, uint32_t S, uint32_t nstop) {
  if (h < S) return nstop + h;
  const uint32_t q = ((h - S) << 4) | l;
  return q < nstop ? q : q + S;
}
// the largest S with S + ceil((sigma - S) / 16) <= 16 and S <= table characters
__host__ __device__ inline uint32_t p2_singles_for(uint32_t sigma, uint32_t nstop) {
  uint32_t best = 0;
  for (uint32_t S = 0; S <= 15 && S + nstop <= sigma; S++)
    if (S + (sigma - S + 15) / 16 <= 16) best = S;
  return best;
}

__device__ __forceinline__ uint64_t p2_below_incl(uint32_t r) {  // bits 0..r
  return r >= 63u ? ~0ull : ((1ull << (r + 1)) - 1ull);
}

// rows of the level-1 line, among its first r+1, whose h equals `h`  (+ the count before the line): global rank of h
__device__ __forceinline__ int64_t p2_rank_h(const uint32_t* __restrict__ l1, uint64_t line, uint32_t r, uint32_t h) {
  const uint32_t* lp = l1 + line * 32;
  const uint4 a = reinterpret_cast<const uint4*>(lp)[0], b = reinterpret_cast<const uint4*>(lp)[1];
  const uint32_t lo = lp[10 + h];
  const uint32_t hi = (lp[26 + (h >> 2)] >> (8 * (h & 3u))) & 0xffu;
  const uint64_t p0 = (uint64_t(a.y) << 32) | a.x, p1 = (uint64_t(a.w) << 32) | a.z;
  const uint64_t p2 = (uint64_t(b.y) << 32) | b.x, p3 = (uint64_t(b.w) << 32) | b.z;
  const uint64_t eq = (p0 ^ ((h & 1u) ? 0ull : ~0ull)) & (p1 ^ ((h & 2u) ? 0ull : ~0ull)) & (p2 ^ ((h & 4u) ? 0ull : ~0ull)) &
                      (p3 ^ ((h & 8u) ? 0ull : ~0ull));
  return int64_t((uint64_t(hi) << 32) | lo) + int64_t(__popcll(eq & p2_below_incl(r)));
}

struct P2Planes2 { uint32_t w[12]; };

__device__ __forceinline__ void p2_load2(const uint32_t* __restrict__ l2, uint64_t line, P2Planes2& P) {
  const uint4* lp = reinterpret_cast<const uint4*>(l2 + line * 32);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint4 v = lp[k];
    P.w[4 * k] = v.x; P.w[4 * k + 1] = v.y; P.w[4 * k + 2] = v.z; P.w[4 * k + 3] = v.w;
  }
}

// positions among the first n (1..96) of a level-2 line whose l equals `l`
__device__ __forceinline__ uint32_t p2_match2(const P2Planes2& P, uint32_t l, uint32_t n) {
  const uint32_t c0 = (l & 1u) ? 0u : ~0u, c1 = (l & 2u) ? 0u : ~0u, c2 = (l & 4u) ? 0u : ~0u, c3 = (l & 8u) ? 0u : ~0u;
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t eq = (P.w[k] ^ c0) & (P.w[3 + k] ^ c1) & (P.w[6 + k] ^ c2) & (P.w[9 + k] ^ c3);
    const int bits = int(n) - 32 * k;
    const uint32_t m = bits >= 32 ? ~0u : (bits <= 0 ? 0u : ((1u << bits) - 1u));
    cnt += uint32_t(__popc(eq & m));
  }
  return cnt;
}

__device__ __forceinline__ void p2_split2(int64_t p, uint64_t* line_in_h, uint32_t* r) {
  const uint32_t q = uint32_t(uint64_t(p) >> 5);
  const uint32_t l = q / 3u;
  *line_in_h = l;
  *r = uint32_t(uint64_t(p) - uint64_t(l) * kP2Rows2);
}

// C[ch] + Occ(code, row) (row included), row >= 0
__device__ __forceinline__ int64_t p2_c_plus_occ(const DevIndex& ix, uint32_t code, int64_t row) {
  uint32_t h, l;
  p2_hl(code, ix.p2_single, ix.p2_stop_below, &h, &l);
  const int64_t rank = p2_rank_h(ix.p2_l1, uint64_t(row) >> 6, uint32_t(row) & 63u, h);
  if (h < ix.p2_single) return rank;      // a class of its own: its level-1 counts start at C[ch], level 1 is the whole answer
  if (rank == 0) return ix.p2_c[code];
  uint64_t lh;
  uint32_t r2;
  p2_split2(rank - 1, &lh, &r2);
  const uint64_t line2 = uint64_t(ix.p2_base[h]) + lh;
  P2Planes2 P;
  p2_load2(ix.p2_l2, line2, P);
  const uint32_t* lp = ix.p2_l2 + line2 * 32;
  const uint32_t lo = lp[12 + l];
  const uint32_t hi = (lp[28 + (l >> 2)] >> (8 * (l & 3u))) & 0xffu;
  return int64_t((uint64_t(hi) << 32) | lo) + int64_t(p2_match2(P, l, r2 + 1));
}

// one step of the backward search (server.c:909-936): the two ends are two independent two-line chains
__device__ __forceinline__ void p2_search_step(const DevIndex& ix, int j, uint32_t code, int64_t& first, int64_t& last) {
  if (j == 0) {
    first = ix.p2_c[code];
    last = ix.p2_c[256 + code];
    return;
  }
  uint32_t h, l;
  p2_hl(code, ix.p2_single, ix.p2_stop_below, &h, &l);
  const uint32_t* __restrict__ l1 = ix.p2_l1;
  const uint32_t* __restrict__ l2 = ix.p2_l2;
  const int64_t base = ix.p2_base[h];
  const bool haveF = first != 0;
  // level 1: both ends usually fall into the same 64-row line once the range is narrow -- load it once
  const uint64_t l1L = uint64_t(last) >> 6, l1F = haveF ? uint64_t(first - 1) >> 6 : l1L;
  int64_t rkL, rkF = 0;
  {
    const uint32_t* lp = l1 + l1L * 32;
    const uint4 a = reinterpret_cast<const uint4*>(lp)[0], b = reinterpret_cast<const uint4*>(lp)[1];
    const uint32_t lo = lp[10 + h];
    const uint32_t hi = (lp[26 + (h >> 2)] >> (8 * (h & 3u))) & 0xffu;
    const uint64_t p0 = (uint64_t(a.y) << 32) | a.x, p1 = (uint64_t(a.w) << 32) | a.z;
    const uint64_t p2 = (uint64_t(b.y) << 32) | b.x, p3 = (uint64_t(b.w) << 32) | b.z;
    const uint64_t eq = (p0 ^ ((h & 1u) ? 0ull : ~0ull)) & (p1 ^ ((h & 2u) ? 0ull : ~0ull)) & (p2 ^ ((h & 4u) ? 0ull : ~0ull)) &
                        (p3 ^ ((h & 8u) ? 0ull : ~0ull));
    const int64_t before = int64_t((uint64_t(hi) << 32) | lo);
    rkL = before + int64_t(__popcll(eq & p2_below_incl(uint32_t(last) & 63u)));
    if (haveF && l1F == l1L) rkF = before + int64_t(__popcll(eq & p2_below_incl(uint32_t(first - 1) & 63u)));
  }
  if (haveF && l1F != l1L) {
    rkF = p2_rank_h(l1, l1F, uint32_t(first - 1) & 63u, h);
    trace_touch(ix, kTraceL1, l1F);
  }
  trace_touch(ix, kTraceL1, l1L);
  const int64_t c0 = ix.p2_c[code];
  if (h < ix.p2_single) {      // a class of its own: the level-1 ranks are C + Occ
    first = haveF ? rkF : c0;
    last = rkL - 1;
    return;
  }
  int64_t nl = c0, nf = c0;
  uint64_t lineL = 0, lineF = 0;
  uint32_t rL = 0, rF = 0;
  if (rkL) {
    uint64_t lh;
    p2_split2(rkL - 1, &lh, &rL);
    lineL = uint64_t(base) + lh;
  }
  if (rkF) {
    uint64_t lh;
    p2_split2(rkF - 1, &lh, &rF);
    lineF = uint64_t(base) + lh;
  }
  P2Planes2 PL, PF;
  uint32_t loL = 0, hiL = 0, loF = 0, hiF = 0;
  const bool other = rkF && (!rkL || lineF != lineL);
  if (rkL) {
    trace_touch(ix, kTraceL2, lineL);
    p2_load2(l2, lineL, PL);
    loL = l2[lineL * 32 + 12 + l];
    hiL = l2[lineL * 32 + 28 + (l >> 2)];
  }
  if (other) {
    trace_touch(ix, kTraceL2, lineF);
    p2_load2(l2, lineF, PF);
    loF = l2[lineF * 32 + 12 + l];
    hiF = l2[lineF * 32 + 28 + (l >> 2)];
  }
  if (rkL) nl = int64_t((uint64_t((hiL >> (8 * (l & 3u))) & 0xffu) << 32) | loL) + int64_t(p2_match2(PL, l, rL + 1));
  if (rkF) {
    if (other) nf = int64_t((uint64_t((hiF >> (8 * (l & 3u))) & 0xffu) << 32) | loF) + int64_t(p2_match2(PF, l, rF + 1));
    else nf = int64_t((uint64_t((hiL >> (8 * (l & 3u))) & 0xffu) << 32) | loL) + int64_t(p2_match2(PL, l, rF + 1));
  }
  first = nf;
  last = nl - 1;
}

// what an LF step / leaf request needs from a row: its character, C+Occ of it, the mark test
struct P2Step {
  uint32_t code;
  int64_t c_plus_occ;
  bool marked;
  int64_t sa_index;
};

__device__ __forceinline__ P2Step p2_step(const DevIndex& ix, int64_t row) {
  P2Step s;
  const uint64_t line1 = uint64_t(row) >> 6;
  const uint32_t r = uint32_t(row) & 63u;
  const uint32_t* lp = ix.p2_l1 + line1 * 32;
  uint32_t w[32];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint4 v = reinterpret_cast<const uint4*>(lp)[k];
    w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
  }
  const uint64_t p0 = (uint64_t(w[1]) << 32) | w[0], p1 = (uint64_t(w[3]) << 32) | w[2];
  const uint64_t p2 = (uint64_t(w[5]) << 32) | w[4], p3 = (uint64_t(w[7]) << 32) | w[6];
  const uint64_t pm = (uint64_t(w[9]) << 32) | w[8];
  const uint32_t h = uint32_t((p0 >> r) & 1u) | (uint32_t((p1 >> r) & 1u) << 1) | (uint32_t((p2 >> r) & 1u) << 2) |
                     (uint32_t((p3 >> r) & 1u) << 3);
  s.marked = (pm >> r) & 1u;
  const uint64_t below = r ? ((1ull << r) - 1ull) : 0ull;
  s.sa_index = int64_t((uint64_t(w[31] & 0xffu) << 32) | w[30]) + int64_t(__popcll(pm & below));
  uint32_t lo = 0, hb = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t is = h == uint32_t(k) ? ~0u : 0u;
    lo |= w[10 + k] & is;
    hb |= ((w[26 + (k >> 2)] >> (8 * (k & 3))) & 0xffu) & is;
  }
  const uint64_t eq = (p0 ^ ((h & 1u) ? 0ull : ~0ull)) & (p1 ^ ((h & 2u) ? 0ull : ~0ull)) & (p2 ^ ((h & 4u) ? 0ull : ~0ull)) &
                      (p3 ^ ((h & 8u) ? 0ull : ~0ull));
  const int64_t rank = int64_t((uint64_t(hb) << 32) | lo) + int64_t(__popcll(eq & p2_below_incl(r)));  // >= 1: the row itself
  if (h < ix.p2_single) {      // a class of its own: no level 2
    trace_touch(ix, kTraceL1, line1);
    s.code = ix.p2_stop_below + h;
    s.c_plus_occ = rank;      // (its counts start at C[ch])
    return s;
  }
  uint64_t lh;
  uint32_t r2;
  p2_split2(rank - 1, &lh, &r2);
  const uint64_t line2 = uint64_t(ix.p2_base[h]) + lh;
  trace_touch(ix, kTraceL1, line1);
  trace_touch(ix, kTraceL2, line2);
  const uint32_t* lq = ix.p2_l2 + line2 * 32;
  uint32_t v[32];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint4 t = reinterpret_cast<const uint4*>(lq)[k];
    v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
  }
  uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t d = r2 - 32u * uint32_t(k);
    const uint32_t hot = d < 32u ? (1u << d) : 0u;
    x0 |= v[k] & hot; x1 |= v[3 + k] & hot; x2 |= v[6 + k] & hot; x3 |= v[9 + k] & hot;
  }
  const uint32_t l = (x0 ? 1u : 0u) | (x1 ? 2u : 0u) | (x2 ? 4u : 0u) | (x3 ? 8u : 0u);
  s.code = p2_code_of_hl(h, l, ix.p2_single, ix.p2_stop_below);
  P2Planes2 P;
#pragma unroll
  for (int k = 0; k < 12; k++) P.w[k] = v[k];
  uint32_t lo2 = 0, hb2 = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t is = l == uint32_t(k) ? ~0u : 0u;
    lo2 |= v[12 + k] & is;
    hb2 |= ((v[28 + (k >> 2)] >> (8 * (k & 3))) & 0xffu) & is;
  }
  s.c_plus_occ = int64_t((uint64_t(hb2) << 32) | lo2) + int64_t(p2_match2(P, l, r2 + 1));
  return s;
}

inline __global__ __launch_bounds__(256) void locate_kernel_pack2(const DevIndex ix, const int64_t total, int64_t* __restrict__ offsets) {
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= total) return;
  int64_t row = offsets[item];
  int64_t steps = 0, result = -1;
  while (row >= 0 && steps <= int64_t(ix.walk_limit)) {
    const P2Step s = p2_step(ix, row);
    if (s.marked) {
      result = mark_offset_at(ix, s.sa_index) + steps;
      break;
    }
    if (s.code < ix.p2_stop_below) break;                  // cannot walk past a document start (server.c:2336-2342)
    row = s.c_plus_occ - 1;                                // LF (server.c:2279-2282)
    steps++;
  }
  offsets[item] = result;
}

inline __global__ __launch_bounds__(256) void block_request_kernel_pack2(const DevIndex ix, const int64_t n, const int64_t* __restrict__ rows,
                                                                  const uint16_t* __restrict__ ch_in, uint16_t* __restrict__ ch_out,
                                                                  int64_t* __restrict__ occ_out, int64_t* __restrict__ off_out) {
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= n) return;
  const int64_t row = rows[item];
  const P2Step s = p2_step(ix, row);
  int64_t occ = s.c_plus_occ;
  if (ch_in) {
    const uint32_t ch = ch_in[item];
    const uint32_t code = ix.p2_code[ch];
    occ = code > 255u ? ix.C[ch] : p2_c_plus_occ(ix, code, row);
  }
  if (ch_out) ch_out[item] = ix.p2_alpha[s.code];
  if (occ_out) occ_out[item] = occ;
  if (off_out) off_out[item] = lane_mark_offset(ix, row);   // femto's own marks (the derived lines may mark more rows)
}

// ---- construction (at open, on the GPU, from the lane tables) -------------------------------------------------------

// phase A: sym[row] = dense code | 0x8000 if the row is marked
inline __global__ __launch_bounds__(256) void p2_extract_kernel(const DevIndex ix, const int64_t row0, const int64_t n, uint16_t* __restrict__ sym) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= row0 + n) return;
  uint32_t idx1;
  const int64_t gb = bucket_of(ix, row, &idx1);
  const DevBucket bk = ix.buckets[gb];
  int seq;
  uint32_t cnt;
  wt_rank_lane(ix, bk.node_base, idx1, &seq, &cnt);
  const LaneSeq sq = ix.lseqs[bk.seq_base + seq_in_bucket(bk, seq)];
  const RankResult m = bseq_rank_lane(ix, sq.mark_table, cnt);
  sym[row] = uint16_t(ix.p2_code[sq.ch] | (m.bit ? 0x8000u : 0u));
}

// phase B1: level-1 planes of one line (64 rows) + its 16 h-counts and mark count (SoA counts[k*stride + line], k=16: marks)
inline __global__ __launch_bounds__(256) void p2_l1_planes_kernel(const int64_t nlines, const int64_t nrows, const uint16_t* __restrict__ sym,
                                                           uint32_t* __restrict__ l1, int64_t* __restrict__ counts, const int64_t stride,
                                                           const uint32_t singles, const uint32_t nstop) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  uint64_t pl[5] = {0, 0, 0, 0, 0};
  uint32_t cnt[17];
#pragma unroll
  for (int k = 0; k < 17; k++) cnt[k] = 0;
  const int64_t r0 = line * kP2Rows1;
  for (int i = 0; i < kP2Rows1; i++) {
    if (r0 + i >= nrows) break;
    const uint32_t s = sym[r0 + i];
    uint32_t h, l;
    p2_hl(s & 0x7fffu, singles, nstop, &h, &l);
    pl[0] |= uint64_t(h & 1u) << i;
    pl[1] |= uint64_t((h >> 1) & 1u) << i;
    pl[2] |= uint64_t((h >> 2) & 1u) << i;
    pl[3] |= uint64_t((h >> 3) & 1u) << i;
    pl[4] |= uint64_t(s >> 15) << i;
  }
  const int valid = int(nrows - r0 < kP2Rows1 ? nrows - r0 : kP2Rows1);
  const uint64_t vm = valid >= 64 ? ~0ull : ((1ull << valid) - 1ull);
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) {
    const uint64_t eq = (pl[0] ^ ((k & 1u) ? 0ull : ~0ull)) & (pl[1] ^ ((k & 2u) ? 0ull : ~0ull)) & (pl[2] ^ ((k & 4u) ? 0ull : ~0ull)) &
                        (pl[3] ^ ((k & 8u) ? 0ull : ~0ull));
    cnt[k] = uint32_t(__popcll(eq & vm));
  }
  cnt[16] = uint32_t(__popcll(pl[4]));
  uint32_t* lp = l1 + line * 32;
#pragma unroll
  for (int p = 0; p < 5; p++) {
    lp[2 * p] = uint32_t(pl[p]);
    lp[2 * p + 1] = uint32_t(pl[p] >> 32);
  }
#pragma unroll
  for (int k = 0; k < 17; k++) counts[int64_t(k) * stride + line] = int64_t(cnt[k]);
}

// phase B3: counts before every line -> dwords 10..31
// (a class of its own stores C[ch] + the rows before the line: its level-1 rank IS C + Occ, as a level-2 count is for the others)
inline __global__ __launch_bounds__(256) void p2_l1_counts_kernel(const int64_t nlines, uint32_t* __restrict__ l1, const int64_t* __restrict__ scans,
                                                           const int64_t stride, const int64_t* __restrict__ p2_c, const uint32_t singles,
                                                           const uint32_t nstop) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  uint32_t* lp = l1 + line * 32;
  uint32_t hi[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint64_t v = uint64_t(scans[int64_t(k) * stride + line] + (uint32_t(k) < singles ? p2_c[nstop + uint32_t(k)] : 0));
    lp[10 + k] = uint32_t(v);
    hi[k >> 2] |= (uint32_t(v >> 32) & 0xffu) << (8 * (k & 3));
  }
#pragma unroll
  for (int k = 0; k < 4; k++) lp[26 + k] = hi[k];
  const uint64_t m = uint64_t(scans[16 * stride + line]);
  lp[30] = uint32_t(m);
  lp[31] = uint32_t(m >> 32) & 0xffu;
}

// phase C: scatter every row's l digit to its level-2 position: lo2[p2_base[h]*96 + rank_h(row) - 1]
inline __global__ __launch_bounds__(256) void p2_scatter_kernel(const DevIndex ix, const int64_t row0, const int64_t n, const uint16_t* __restrict__ sym,
                                                         uint8_t* __restrict__ lo2) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= row0 + n) return;
  const uint32_t s = sym[row];
  uint32_t h, l;
  p2_hl(s & 0x7fffu, ix.p2_single, ix.p2_stop_below, &h, &l);
  if (h < ix.p2_single) return;      // (a class of its own has no level 2)
  const int64_t rank = p2_rank_h(ix.p2_l1, uint64_t(row) >> 6, uint32_t(row) & 63u, h);
  lo2[ix.p2_base[h] * kP2Rows2 + rank - 1] = uint8_t(l);
}

// phase D1: level-2 planes of one line (96 positions) + its 16 l-counts
inline __global__ __launch_bounds__(256) void p2_l2_planes_kernel(const int64_t nlines, const uint8_t* __restrict__ lo2, uint32_t* __restrict__ l2,
                                                           int64_t* __restrict__ counts, const int64_t stride) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  uint32_t pl[4][3];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int k = 0; k < 3; k++) pl[p][k] = 0;
  const uint4* sp = reinterpret_cast<const uint4*>(lo2 + line * kP2Rows2);
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int hlf = 0; hlf < 2; hlf++) {
      const uint4 v = sp[2 * k + hlf];
      const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint32_t by = (d[j] >> (8 * b)) & 0xffu;
          const int pos = 16 * hlf + 4 * j + b;
          pl[0][k] |= (by & 1u) << pos;
          pl[1][k] |= ((by >> 1) & 1u) << pos;
          pl[2][k] |= ((by >> 2) & 1u) << pos;
          pl[3][k] |= ((by >> 3) & 1u) << pos;
        }
    }
  uint32_t* lp = l2 + line * 32;
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int k = 0; k < 3; k++) lp[3 * p + k] = pl[p][k];
#pragma unroll
  for (uint32_t c = 0; c < 16; c++) {
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 3; k++)
      n += uint32_t(__popc((pl[0][k] ^ ((c & 1u) ? 0u : ~0u)) & (pl[1][k] ^ ((c & 2u) ? 0u : ~0u)) & (pl[2][k] ^ ((c & 4u) ? 0u : ~0u)) &
                           (pl[3][k] ^ ((c & 8u) ? 0u : ~0u))));
    counts[int64_t(c) * stride + line] = int64_t(n);
  }
}

// phase D3: C[ch(16h+k)] + earlier positions of the same h with l == k -> dwords 12..31.  Padding positions (value 0)
// sit at the end of an h's last line, i.e. after every real position of that h, and the next h subtracts the scan
// value at its own first line, so they are never counted.
inline __global__ __launch_bounds__(256) void p2_l2_counts_kernel(const DevIndex ix, const int64_t nlines, uint32_t* __restrict__ l2,
                                                           const int64_t* __restrict__ scans, const int64_t stride) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  int h = 0;
#pragma unroll
  for (int k = 1; k < 16; k++) if (line >= ix.p2_base[k]) h = k;
  const int64_t first_line = ix.p2_base[h];
  uint32_t* lp = l2 + line * 32;
  uint32_t hi[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t code = p2_code_of_hl(uint32_t(h), uint32_t(k), ix.p2_single, ix.p2_stop_below);
    int64_t v = scans[int64_t(k) * stride + line] - scans[int64_t(k) * stride + first_line];
    if (int(code) < ix.p2_sigma) v += ix.p2_c[code];
    lp[12 + k] = uint32_t(uint64_t(v));
    hi[k >> 2] |= (uint32_t(uint64_t(v) >> 32) & 0xffu) << (8 * (k & 3));
  }
#pragma unroll
  for (int k = 0; k < 4; k++) lp[28 + k] = hi[k];
}

// phase E: offsets of the marked rows in row order
inline __global__ __launch_bounds__(256) void p2_sa_kernel(const DevIndex ix, const int64_t row0, const int64_t n, const uint16_t* __restrict__ sym,
                                                    int64_t* __restrict__ sa) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= row0 + n) return;
  if (!(sym[row] & 0x8000u)) return;
  uint32_t idx1;
  const int64_t gb = bucket_of(ix, row, &idx1);
  const DevBucket bk = ix.buckets[gb];
  int seq;
  uint32_t cnt;
  wt_rank_lane(ix, bk.node_base, idx1, &seq, &cnt);
  const LaneSeq sq = ix.lseqs[bk.seq_base + seq_in_bucket(bk, seq)];
  const RankResult m = bseq_rank_lane(ix, sq.mark_table, cnt);
  const int64_t off = int64_t(read_bits_ptr(wrap_ptr(ix.image, sq.mark_array), mark_rec(ix, m.o1) * uint64_t(ix.text_size_bits), ix.text_size_bits));
  const uint32_t* lp = ix.p2_l1 + (uint64_t(row) >> 6) * 32;
  const uint32_t r = uint32_t(row) & 63u;
  const uint64_t pm = (uint64_t(lp[9]) << 32) | lp[8];
  const uint64_t below = r ? ((1ull << r) - 1ull) : 0ull;
  mark_offset_store(ix, sa, int64_t((uint64_t(lp[31] & 0xffu) << 32) | lp[30]) + int64_t(__popcll(pm & below)), off);
}

// ---- denser marks (derived), see pack_kernels.hip.hpp ---------------------------------------------------------------
__device__ __forceinline__ int64_t p2_mark_rank(const uint32_t* __restrict__ l1, int64_t row) {
  const uint32_t* lp = l1 + (uint64_t(row) >> 6) * 32;
  const uint32_t r = uint32_t(row) & 63u;
  const uint64_t pm = (uint64_t(lp[9]) << 32) | lp[8];
  const uint64_t below = r ? ((1ull << r) - 1ull) : 0ull;
  return int64_t((uint64_t(lp[31] & 0xffu) << 32) | lp[30]) + int64_t(__popcll(pm & below));
}

template <bool kStore>
inline __global__ __launch_bounds__(256) void p2_densify_kernel(const DevIndex ix, uint32_t* __restrict__ l1, const int64_t row0, const int64_t n,
                                                         const uint16_t* __restrict__ sym, const int every, const int period,
                                                         int64_t* __restrict__ sa) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= row0 + n) return;
  if (!(sym[row] & 0x8000u)) return;
  int64_t off = 0;
  if (kStore) {
    off = lane_mark_offset(ix, row);
    mark_offset_store(ix, sa, p2_mark_rank(l1, row), off);
  }
  int64_t r = row;
  for (int j = 1; j < period; j++) {
    const P2Step s = p2_step(ix, r);
    if (s.code < ix.p2_stop_below) break;
    r = s.c_plus_occ - 1;
    if (j % every == 0) {
      if (kStore) mark_offset_store(ix, sa, p2_mark_rank(l1, r), off - j);
      else atomicOr(l1 + (uint64_t(r) >> 6) * 32 + 8 + ((uint32_t(r) & 63u) >> 5), 1u << (uint32_t(r) & 31u));
    }
  }
}

inline __global__ __launch_bounds__(256) void p2_recount_marks_kernel(const int64_t nlines, const uint32_t* __restrict__ l1, int64_t* __restrict__ counts) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  counts[line] = int64_t(__popc(l1[line * 32 + 8]) + __popc(l1[line * 32 + 9]));
}

inline __global__ __launch_bounds__(256) void p2_markcount_kernel(const int64_t nlines, uint32_t* __restrict__ l1, const int64_t* __restrict__ scan) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  const uint64_t m = uint64_t(scan[line]);
  l1[line * 32 + 30] = uint32_t(m);
  l1[line * 32 + 31] = uint32_t(m >> 32) & 0xffu;
}

}  // namespace femto_amd
