// pack_kernels.hip.hpp -- small-alphabet fast path (mode 3, "pack").
//
// femto stores L as a Huffman-shaped wavelet tree of RLE/literal bit sequences with one mark table per
// character (src/main/wtree.c, src/main/index.c:645-720): a DNA index needs 2.25 sequence ranks per Occ and two
// more for the mark test of a locate step, each a separate 128-byte line somewhere in HBM.  When the whole index
// uses at most 8 distinct characters (DNA: A C G T + SEOF [+ N, newline]) the loader derives -- on the GPU, from
// the uploaded femto blocks, with the lane kernels of kernels.hip.hpp -- ONE self-contained 128-byte line per
// 160 rows, so that Occ(ch,row), L[row] and "is row marked" are answered from a single memory line:
//
//   dword  0.. 4  bit 0 of the dense character code of rows 160*line + [0,160)   (bit i of dword k = row 32k+i)
//   dword  5.. 9  bit 1,   dword 10..14  bit 2
//   dword 15..19  1 = the row is marked (its offset is in pack_sa)
//   dword 20..27  low 32 bits of  C[ch(code)] + Occ(code, rows before this line)   for code 0..7
//   dword 28      low 32 bits of  number of marked rows before this line
//   dword 29,30   bits 32..39 of the eight counts (code c: byte c&3 of dword 29 + (c>>2))
//   dword 31      bits 32..39 of the mark count (byte 0)
//   pack_sa[k]    offset of the k-th marked row (row order): the mark arrays (index.c:2102-2140), re-ordered
//
// Results are identical to the wavelet path (every leaf request of every fixture row is compared in the tests);
// 40-bit counts cover the format's 2^39 text limit (src/dcx_cc/index_tool.cc:45-46).
#pragma once

namespace femto_amd {

constexpr int kPackRows = 160;
constexpr int kPackPlaneWords = 5;
constexpr int kPackLineWords = 32;

// compulsory-traffic trace (femto_amd_trace_lines): trace_kernels.hip compiles these same sources a second time with
// FEMTO_AMD_TRACE defined, in a namespace of their own; there every 128-byte line loaded from a traced array sets its
// bit.  The production kernels (this header compiled without the macro) contain no trace code at all.
__device__ __forceinline__ void trace_touch(const DevIndex& ix, int region, uint64_t line) {
#ifdef FEMTO_AMD_TRACE
  const uint64_t b = uint64_t(ix.trace_off[region]) + line;
  atomicOr(ix.trace + (b >> 5), 1u << (b & 31u));
  if (ix.trace_reads) atomicAdd(ix.trace_reads + region, 1ull);
#else
  (void)ix; (void)region; (void)line;
#endif
}

// offset of the i-th marked row (pack_sa: 8-byte entries, or 4-byte ones when the index has fewer than 2^32 rows)
__device__ __forceinline__ int64_t mark_offset_at(const DevIndex& ix, int64_t i) {
  trace_touch(ix, kTraceSa, uint64_t(i) >> (ix.pack_sa32 ? 5 : 4));
  if (!ix.pack_sa32) return ix.pack_sa[i];
  const uint32_t v = reinterpret_cast<const uint32_t*>(ix.pack_sa)[i];
  return v == 0xffffffffu ? int64_t(-1) : int64_t(v);      // (an index of < 2^32 rows has no offset 2^32 - 1)
}
// A derived offset can only be negative on a DAMAGED index (a mark offset smaller than the steps walked back from it); the
// 4-byte form keeps it negative (-1) instead of wrapping it into a large valid-looking offset.
__device__ __forceinline__ void mark_offset_store(const DevIndex& ix, int64_t* sa, int64_t i, int64_t off) {
  if (ix.pack_sa32) reinterpret_cast<uint32_t*>(sa)[i] = off < 0 ? 0xffffffffu : uint32_t(uint64_t(off));
  else sa[i] = off;
}

// a load of bytes [b0, b1] of a traced array: one line, or two when it straddles a boundary
__device__ __forceinline__ void trace_touch_span(const DevIndex& ix, int region, uint64_t b0, uint64_t b1) {
  trace_touch(ix, region, b0 >> 7);
  if ((b1 >> 7) != (b0 >> 7)) trace_touch(ix, region, b1 >> 7);
}

__device__ __forceinline__ void pack_split(int64_t row, uint64_t* line, uint32_t* r) {
  const uint32_t q = uint32_t(uint64_t(row) >> 5);  // rows < 2^37
  const uint32_t l = q / 5u;
  *line = l;
  *r = uint32_t(uint64_t(row) - uint64_t(l) * kPackRows);
}

struct PackPlanes { uint32_t w[16]; };  // dwords 0..15 of a line: the three code planes (+ mark word 0, unused)

__device__ __forceinline__ void pack_load_planes(const uint32_t* __restrict__ pack, uint64_t line, PackPlanes& P) {
  const uint4* lp = reinterpret_cast<const uint4*>(pack + line * kPackLineWords);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint4 v = lp[k];
    P.w[4 * k] = v.x; P.w[4 * k + 1] = v.y; P.w[4 * k + 2] = v.z; P.w[4 * k + 3] = v.w;
  }
}

// C[ch] + Occ(code, rows before the line)
__device__ __forceinline__ int64_t pack_base(const uint32_t* __restrict__ pack, uint64_t line, uint32_t code) {
  const uint32_t* lp = pack + line * kPackLineWords;
  const uint32_t lo = lp[20 + code];
  const uint32_t hi = (lp[29 + (code >> 2)] >> (8 * (code & 3))) & 0xffu;
  return int64_t((uint64_t(hi) << 32) | lo);
}

// rows among the first n (1..160) of the line whose code equals `code`
__device__ __forceinline__ uint32_t pack_match(const PackPlanes& P, uint32_t code, uint32_t n) {
  const uint32_t c0 = (code & 1u) ? 0u : ~0u, c1 = (code & 2u) ? 0u : ~0u, c2 = (code & 4u) ? 0u : ~0u;  // xor -> 1 where equal
  uint32_t cnt = 0;
#pragma unroll
  for (int k = 0; k < kPackPlaneWords; k++) {
    const uint32_t eq = (P.w[k] ^ c0) & (P.w[kPackPlaneWords + k] ^ c1) & (P.w[2 * kPackPlaneWords + k] ^ c2);
    const int bits = int(n) - 32 * k;
    const uint32_t m = bits >= 32 ? ~0u : (bits <= 0 ? 0u : ((1u << bits) - 1u));
    cnt += uint32_t(__popc(eq & m));
  }
  return cnt;
}

// one step of the backward search with dense code `code` (server.c:909-936): [first,last] -> rows preceded by code
__device__ __forceinline__ void pack_search_step(const DevIndex& ix, const uint32_t* __restrict__ pack, int j, uint32_t code,
                                                 int64_t& first, int64_t& last) {
  if (j == 0) {
    first = ix.pack_c[code];
    last = ix.pack_c[8 + code];
    return;
  }
  uint64_t lineL, lineF = 0;
  uint32_t rL, rF = 0;
  pack_split(last, &lineL, &rL);
  PackPlanes PL, PF;
  pack_load_planes(pack, lineL, PL);
  const int64_t bL = pack_base(pack, lineL, code);
  int64_t bF = 0;
  bool other = false;
  if (first != 0) {
    pack_split(first - 1, &lineF, &rF);
    other = lineF != lineL;
  }
  if (other) {  // both ends of a narrow range usually share the line
    pack_load_planes(pack, lineF, PF);
    bF = pack_base(pack, lineF, code);
    trace_touch(ix, kTracePack, lineF);
  }
  trace_touch(ix, kTracePack, lineL);
  const int64_t nl = bL + int64_t(pack_match(PL, code, rL + 1));
  int64_t nf;
  if (first == 0) nf = ix.pack_c[code];
  else if (other) nf = bF + int64_t(pack_match(PF, code, rF + 1));
  else nf = bL + int64_t(pack_match(PL, code, rF + 1));
  first = nf;
  last = nl - 1;
}

// level-table entries (direct_kernels.hip.hpp): x = first, y = (last + 1) | level << 48
constexpr uint64_t kKtabLastMask = (uint64_t(1) << 48) - 1;

// hand a pattern whose range is down to one row, with many symbols to go, over to count_tail_kernel
// (text_kernels.hip.hpp): one atomic per wavefront, entries {slot, symbols done, row}
__device__ __forceinline__ void tail_append(const DevIndex& ix, int64_t slot, int done, int64_t row) {
  const unsigned long long m = __ballot(1);
  const int lane = int(threadIdx.x & 63u);
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(ix.tail_count, __popcll(m));
  base = __shfl(base, leader, 64);
  const int at = base + __popcll(m & ((1ull << lane) - 1ull));
  int4* dst = reinterpret_cast<int4*>(ix.tail_items) + at;
  *dst = make_int4(int(uint32_t(slot)), done, int(uint32_t(uint64_t(row))), int(uint32_t(uint64_t(row) >> 32)));
}

struct PackLine { uint32_t w[kPackLineWords]; };

__device__ __forceinline__ void pack_load_line(const uint32_t* __restrict__ pack, uint64_t line, PackLine& L) {
  const uint4* lp = reinterpret_cast<const uint4*>(pack + line * kPackLineWords);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint4 v = lp[k];
    L.w[4 * k] = v.x; L.w[4 * k + 1] = v.y; L.w[4 * k + 2] = v.z; L.w[4 * k + 3] = v.w;
  }
}

// Everything a locate step / leaf request needs from one line: L[row] (dense code), C+Occ(L[row],row) and the
// mark test with the index of the row's offset in pack_sa.
struct PackStep {
  uint32_t code;
  int64_t c_plus_occ;   // C[ch] + Occ(ch, row), row included
  bool marked;
  int64_t sa_index;
};

__device__ __forceinline__ PackStep pack_step(const PackLine& L, uint32_t r) {
  PackStep s;
  // bit r of each 160-bit plane through one-hot masks (no register indexing: that would go through scratch)
  uint32_t x0 = 0, x1 = 0, x2 = 0, xm = 0, mb = 0;
#pragma unroll
  for (int k = 0; k < kPackPlaneWords; k++) {
    const uint32_t d = r - 32u * uint32_t(k);
    const uint32_t hot = d < 32u ? (1u << d) : 0u;
    x0 |= L.w[k] & hot;
    x1 |= L.w[5 + k] & hot;
    x2 |= L.w[10 + k] & hot;
    xm |= L.w[15 + k] & hot;
    const int bits = int(r) - 32 * k;
    const uint32_t below = bits >= 32 ? ~0u : (bits <= 0 ? 0u : ((1u << bits) - 1u));
    mb += uint32_t(__popc(L.w[15 + k] & below));  // marked rows of this line before row r
  }
  s.code = (x0 ? 1u : 0u) | (x1 ? 2u : 0u) | (x2 ? 4u : 0u);
  s.marked = xm != 0;
  PackPlanes P;
#pragma unroll
  for (int j = 0; j < 15; j++) P.w[j] = L.w[j];
  uint32_t lo = 0, hw = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t is = s.code == uint32_t(j) ? ~0u : 0u;
    lo |= L.w[20 + j] & is;
    hw |= ((L.w[29 + (j >> 2)] >> (8 * (j & 3))) & 0xffu) & is;
  }
  s.c_plus_occ = int64_t((uint64_t(hw) << 32) | lo) + int64_t(pack_match(P, s.code, r + 1));
  s.sa_index = int64_t((uint64_t(L.w[31] & 0xffu) << 32) | L.w[28]) + int64_t(mb);
  return s;
}

// SA[row] of a row the caller KNOWS to be marked (count_direct_kernel met it during a search: the marked rank units of
// ru_kernels.hip.hpp): only the line's mark plane and mark count are read -- three of its eight 16-byte pieces -- then the
// offset.  A row that is not marked after all (a damaged index): -1.
__device__ __forceinline__ int64_t pack_marked_offset(const DevIndex& ix, int64_t row) {
  uint64_t line;
  uint32_t r;
  pack_split(row, &line, &r);
  const uint4* lp = reinterpret_cast<const uint4*>(ix.pack + line * kPackLineWords);
  const uint4 a = lp[3], b = lp[4], c = lp[7];     // dwords 12..15, 16..19, 28..31
  trace_touch(ix, kTracePack, line);
  const uint32_t m[kPackPlaneWords] = {a.w, b.x, b.y, b.z, b.w};
  uint32_t xm = 0, mb = 0;
#pragma unroll
  for (int k = 0; k < kPackPlaneWords; k++) {
    const uint32_t d = r - 32u * uint32_t(k);
    xm |= m[k] & (d < 32u ? (1u << d) : 0u);
    const int bits = int(r) - 32 * k;
    const uint32_t below = bits >= 32 ? ~0u : (bits <= 0 ? 0u : ((1u << bits) - 1u));
    mb += uint32_t(__popc(m[k] & below));
  }
  if (!xm) return -1;
  return mark_offset_at(ix, int64_t((uint64_t(c.w & 0xffu) << 32) | c.x) + int64_t(mb));
}

// rows to locate, written where their offsets will go: pattern q's rows first[q] .. first[q]+noccs-1 at
// offsets[out_starts[q] ..] (setup_locate_range, src/main/server.c:4047).  One thread per pattern; the walk kernel
// then needs no search for "which pattern owns output slot i".
constexpr int64_t kExpandSerialMax = 4096;   // a longer range (e.g. the empty pattern's) is filled by one thread per row

inline __global__ __launch_bounds__(256) void expand_rows_kernel(const int64_t npats, const int64_t* __restrict__ first,
                                                          const int64_t* __restrict__ out_starts, int64_t* __restrict__ offsets,
                                                          int* __restrict__ big_flag) {
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= npats) return;
  const int64_t base = out_starts[q], n = out_starts[q + 1] - base, f = first[q];
  if (n > kExpandSerialMax) {
    atomicOr(big_flag, 1);
    return;
  }
  for (int64_t j = 0; j < n; j++) offsets[base + j] = f + j;
}

// the long ranges left over by expand_rows_kernel: one thread per output slot (idle unless the flag is set)
inline __global__ __launch_bounds__(256) void expand_big_rows_kernel(const int64_t npats, const int64_t* __restrict__ first,
                                                              const int64_t* __restrict__ out_starts, const int64_t total,
                                                              int64_t* __restrict__ offsets, const int* __restrict__ big_flag) {
  if (!*big_flag) return;
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= total) return;
  int64_t lo = 0, hi = npats;   // largest q with out_starts[q] <= item
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (out_starts[mid] <= item) lo = mid; else hi = mid;
  }
  if (out_starts[lo + 1] - out_starts[lo] > kExpandSerialMax) offsets[item] = first[lo] + (item - out_starts[lo]);
}

// locate walk (do_back_query / do_context_query, src/main/server.c:2228-2359, :2627-2795): one line per LF step;
// offsets[item] holds the row on entry and the row's text offset on return
// (A persistent-lane variant -- finished lanes pull the next row instead of idling until the slowest lane of the
// wavefront is done -- and the removal of the per-row owner search both measured neutral: 105 M random line
// fetches per 10 M rows run at 30 G lines/s = 4 TB/s either way, the rate of purely random 128-byte reads.)
inline __global__ __launch_bounds__(256) void locate_kernel_pack(const DevIndex ix, const int64_t total, int64_t* __restrict__ offsets) {
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= total) return;
  int64_t row = offsets[item];
  int64_t steps = 0, result = -1;
  while (row >= 0 && steps <= int64_t(ix.walk_limit)) {
    uint64_t line;
    uint32_t r;
    pack_split(row, &line, &r);
    PackLine L;
    pack_load_line(ix.pack, line, L);
    trace_touch(ix, kTracePack, line);
    const PackStep s = pack_step(L, r);
    if (s.marked) {
      result = mark_offset_at(ix, s.sa_index) + steps;
      break;
    }
    if ((ix.pack_stop >> s.code) & 1u) break;              // cannot walk past a document start (server.c:2336-2342)
    row = s.c_plus_occ - 1;                                // LF (server.c:2279-2282)
    steps++;
  }
  offsets[item] = result;
}

// BLOCK_REQUEST_LOCATION exactly as the reference answers it: the row's entry in its character's mark table / array
__device__ __forceinline__ int64_t lane_mark_offset(const DevIndex& ix, int64_t row) {
  uint32_t idx1;
  const int64_t gb = bucket_of(ix, row, &idx1);
  const DevBucket bk = ix.buckets[gb];
  int seq;
  uint32_t cnt;
  wt_rank_lane(ix, bk.node_base, idx1, &seq, &cnt);
  const LaneSeq sq = ix.lseqs[bk.seq_base + seq_in_bucket(bk, seq)];
  const RankResult m = bseq_rank_lane(ix, sq.mark_table, cnt);
  if (!m.bit) return -1;
  return int64_t(read_bits_ptr(wrap_ptr(ix.image, sq.mark_array), mark_rec(ix, m.o1) * uint64_t(ix.text_size_bits), ix.text_size_bits));
}

// leaf requests (block_request CHAR|OCCS|LOCATION, src/main/index.c:1973-2144) from the packed lines
inline __global__ __launch_bounds__(256) void block_request_kernel_pack(const DevIndex ix, const int64_t n, const int64_t* __restrict__ rows,
                                                                 const uint16_t* __restrict__ ch_in, uint16_t* __restrict__ ch_out,
                                                                 int64_t* __restrict__ occ_out, int64_t* __restrict__ off_out) {
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= n) return;
  const int64_t row = rows[item];
  uint64_t line;
  uint32_t r;
  pack_split(row, &line, &r);
  PackLine L;
  pack_load_line(ix.pack, line, L);
  const PackStep s = pack_step(L, r);
  int64_t occ = s.c_plus_occ;
  if (ch_in) {
    const uint32_t ch = ch_in[item];
    const uint32_t code = ix.pack_code[ch];
    if (code > 7u) occ = ix.C[ch];
    else {
      PackPlanes P;
#pragma unroll
      for (int j = 0; j < 15; j++) P.w[j] = L.w[j];
      occ = pack_base(ix.pack, line, code) + int64_t(pack_match(P, code, r + 1));
    }
  }
  if (ch_out) {
    uint32_t a = ix.pack_alpha[0];
#pragma unroll
    for (int j = 1; j < 8; j++) a = s.code == uint32_t(j) ? uint32_t(ix.pack_alpha[j]) : a;
    ch_out[item] = uint16_t(a);
  }
  if (occ_out) occ_out[item] = occ;
  if (off_out) off_out[item] = lane_mark_offset(ix, row);   // femto's own marks (the derived lines may mark more rows)
}

// ---- construction of the packed lines (at open, on the GPU, from the lane tables) ------------------------------

// phase A: sym[row] = dense code of L[row] | 0x80 if the row is marked
inline __global__ __launch_bounds__(256) void pack_extract_kernel(const DevIndex ix, const int64_t row0, const int64_t n, uint8_t* __restrict__ sym) {
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
  sym[row] = uint8_t(ix.pack_code[sq.ch] | (m.bit ? 0x80u : 0u));
}

// phase B1: one thread per line: planes from 160 sym bytes, per-line counts (SoA: counts[c * stride + line], c = 8: marks)
inline __global__ __launch_bounds__(256) void pack_planes_kernel(const int64_t nlines, const uint8_t* __restrict__ sym, uint32_t* __restrict__ pack,
                                                          int64_t* __restrict__ counts, const int64_t stride) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  const uint4* sp = reinterpret_cast<const uint4*>(sym + line * kPackRows);
  uint32_t pl[4][kPackPlaneWords];
  uint32_t cnt[9];
#pragma unroll
  for (int c = 0; c < 9; c++) cnt[c] = 0;
#pragma unroll
  for (int k = 0; k < kPackPlaneWords; k++) {
    uint32_t w0 = 0, w1 = 0, w2 = 0, wm = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint4 v = sp[2 * k + h];
      const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint32_t by = (d[j] >> (8 * b)) & 0xffu;
          const int pos = 16 * h + 4 * j + b;
          w0 |= (by & 1u) << pos;
          w1 |= ((by >> 1) & 1u) << pos;
          w2 |= ((by >> 2) & 1u) << pos;
          wm |= (by >> 7) << pos;
        }
    }
    pl[0][k] = w0; pl[1][k] = w1; pl[2][k] = w2; pl[3][k] = wm;
    cnt[8] += uint32_t(__popc(wm));
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
      const uint32_t eq = (w0 ^ ((c & 1u) ? 0u : ~0u)) & (w1 ^ ((c & 2u) ? 0u : ~0u)) & (w2 ^ ((c & 4u) ? 0u : ~0u));
      cnt[c] += uint32_t(__popc(eq));
    }
  }
  uint32_t* lp = pack + line * kPackLineWords;
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int k = 0; k < kPackPlaneWords; k++) lp[p * kPackPlaneWords + k] = pl[p][k];
#pragma unroll
  for (int c = 0; c < 9; c++) counts[int64_t(c) * stride + line] = int64_t(cnt[c]);
}

// phase B3: counts before every line (exclusive scans, SoA) + C[ch] -> dwords 20..31
// NOTE: padding rows past total_length carry code 0; they lie after every real row, so no count that is ever read includes them.
inline __global__ __launch_bounds__(256) void pack_counts_kernel(const DevIndex ix, const int64_t nlines, uint32_t* __restrict__ pack,
                                                          const int64_t* __restrict__ scans, const int64_t stride) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  uint32_t* lp = pack + line * kPackLineWords;
  uint32_t hi[3] = {0, 0, 0};
#pragma unroll
  for (int c = 0; c < 8; c++) {
    int64_t v = scans[int64_t(c) * stride + line];
    if (c < ix.pack_sigma) v += ix.C[ix.pack_alpha[c]];
    lp[20 + c] = uint32_t(uint64_t(v));
    hi[c >> 2] |= (uint32_t(uint64_t(v) >> 32) & 0xffu) << (8 * (c & 3));
  }
  const int64_t m = scans[8 * stride + line];
  lp[28] = uint32_t(uint64_t(m));
  hi[2] = uint32_t(uint64_t(m) >> 32) & 0xffu;
  lp[29] = hi[0];
  lp[30] = hi[1];
  lp[31] = hi[2];
}

// phase C: the offsets of the marked rows, in row order
inline __global__ __launch_bounds__(256) void pack_sa_kernel(const DevIndex ix, const int64_t row0, const int64_t n, const uint8_t* __restrict__ sym,
                                                      const uint32_t* __restrict__ pack, int64_t* __restrict__ sa) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= row0 + n) return;
  if (!(sym[row] & 0x80u)) return;
  uint32_t idx1;
  const int64_t gb = bucket_of(ix, row, &idx1);
  const DevBucket bk = ix.buckets[gb];
  int seq;
  uint32_t cnt;
  wt_rank_lane(ix, bk.node_base, idx1, &seq, &cnt);
  const LaneSeq sq = ix.lseqs[bk.seq_base + seq_in_bucket(bk, seq)];
  const RankResult m = bseq_rank_lane(ix, sq.mark_table, cnt);
  const int64_t off = int64_t(read_bits_ptr(wrap_ptr(ix.image, sq.mark_array), mark_rec(ix, m.o1) * uint64_t(ix.text_size_bits), ix.text_size_bits));
  uint64_t line;
  uint32_t r;
  pack_split(row, &line, &r);
  const uint32_t* lp = pack + line * kPackLineWords;
  uint32_t mb = 0;
  for (int j = 0; j < kPackPlaneWords; j++) {
    const int bits = int(r) - 32 * j;
    const uint32_t msk = bits >= 32 ? ~0u : (bits <= 0 ? 0u : ((1u << bits) - 1u));
    mb += uint32_t(__popc(lp[15 + j] & msk));
  }
  mark_offset_store(ix, sa, int64_t((uint64_t(lp[31] & 0xffu) << 32) | lp[28]) + int64_t(mb), off);
}

// ---- denser marks (derived) ---------------------------------------------------------------------------------------
// femto marks every mark_period-th text position (default 20), so a locate walks ~mark_period/2 LF steps -- each a
// random memory line.  The index files stay as they are, but the DERIVED lines may mark more rows: from every
// original mark (offset o) the LF walk visits the rows of offsets o-1, o-2, ...; those at distance `every`,
// 2*every, ... get a mark of their own (offset o-j), so a walk ends after at most `every`-1 steps.  Offsets returned
// are the same numbers; leaf requests keep answering from femto's own mark tables (BLOCK_REQUEST_LOCATION parity).
// pass 1 sets the extra bits (atomicOr into the mark plane; walking reads only the code planes and counts),
// then the mark counts are recomputed, then pass 2 repeats the walk and stores the offsets at their ranks.
__device__ __forceinline__ int64_t pack_mark_rank(const uint32_t* __restrict__ pack, int64_t row) {
  uint64_t line;
  uint32_t r;
  pack_split(row, &line, &r);
  const uint32_t* lp = pack + line * kPackLineWords;
  uint32_t mb = 0;
  for (int j = 0; j < kPackPlaneWords; j++) {
    const int bits = int(r) - 32 * j;
    const uint32_t msk = bits >= 32 ? ~0u : (bits <= 0 ? 0u : ((1u << bits) - 1u));
    mb += uint32_t(__popc(lp[15 + j] & msk));
  }
  return int64_t((uint64_t(lp[31] & 0xffu) << 32) | lp[28]) + int64_t(mb);
}

template <bool kStore>
inline __global__ __launch_bounds__(256) void pack_densify_kernel(const DevIndex ix, uint32_t* __restrict__ pack, const int64_t row0, const int64_t n,
                                                           const uint8_t* __restrict__ sym, const int every, const int period,
                                                           int64_t* __restrict__ sa) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= row0 + n) return;
  if (!(sym[row] & 0x80u)) return;
  int64_t off = 0;
  if (kStore) {  // the original mark's offset, from femto's mark table and array (index.c:2102-2140)
    uint32_t idx1;
    const int64_t gb = bucket_of(ix, row, &idx1);
    const DevBucket bk = ix.buckets[gb];
    int seq;
    uint32_t cnt;
    wt_rank_lane(ix, bk.node_base, idx1, &seq, &cnt);
    const LaneSeq sq = ix.lseqs[bk.seq_base + seq_in_bucket(bk, seq)];
    const RankResult m = bseq_rank_lane(ix, sq.mark_table, cnt);
    off = int64_t(read_bits_ptr(wrap_ptr(ix.image, sq.mark_array), mark_rec(ix, m.o1) * uint64_t(ix.text_size_bits), ix.text_size_bits));
    mark_offset_store(ix, sa, pack_mark_rank(pack, row), off);
  }
  int64_t r = row;
  for (int j = 1; j < period; j++) {
    uint64_t line;
    uint32_t rr;
    pack_split(r, &line, &rr);
    PackLine L;
    pack_load_line(pack, line, L);
    const PackStep s = pack_step(L, rr);
    if ((ix.pack_stop >> s.code) & 1u) break;     // a document starts here: nothing precedes it
    r = s.c_plus_occ - 1;
    if (j % every == 0) {
      if (kStore) {
        mark_offset_store(ix, sa, pack_mark_rank(pack, r), off - j);
      } else {
        pack_split(r, &line, &rr);
        atomicOr(pack + line * kPackLineWords + 15 + (rr >> 5), 1u << (rr & 31u));
      }
    }
  }
}

inline __global__ __launch_bounds__(256) void pack_recount_marks_kernel(const int64_t nlines, const uint32_t* __restrict__ pack, int64_t* __restrict__ counts) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  const uint32_t* lp = pack + line * kPackLineWords;
  uint32_t c = 0;
  for (int j = 0; j < kPackPlaneWords; j++) c += uint32_t(__popc(lp[15 + j]));
  counts[line] = int64_t(c);
}

inline __global__ __launch_bounds__(256) void pack_markcount_kernel(const int64_t nlines, uint32_t* __restrict__ pack, const int64_t* __restrict__ scan) {
  const int64_t line = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (line >= nlines) return;
  const uint64_t m = uint64_t(scan[line]);
  pack[line * kPackLineWords + 28] = uint32_t(m);
  pack[line * kPackLineWords + 31] = uint32_t(m >> 32) & 0xffu;
}

}  // namespace femto_amd
