// kernels.hip.hpp -- hand-written HIP kernels for gfx950 (CDNA4, wave64): batched Occ/rank over
// femto's wavelet-tree buckets, backward search (count) and the sampled-SA locate walk.
//
// Execution model.  A *group* of W lanes (W = 32 by default: half a wavefront) cooperates on ONE
// binary-sequence rank at a time:
//   - group search over A0+A1: one coalesced load per lane, __ballot + popcount picks the group
//     (the reference's bsearch_A0A1, src/main/wtree.c:609-629, does ~7 dependent probes);
//   - varbyte S scan: one byte per lane, terminator bits via __ballot, a lane-parallel inclusive
//     scan (ds_bpermute/DPP shuffles) of the packed (zeros, ones) contributions replaces the serial
//     decode_varbyte loop (src/main/wtree.c:657-676);
//   - the 64-byte D segment: 8 lanes x one big-endian u64 each, masked __popcll + shuffle reduce
//     for literal segments (src/main/wtree.c:713-759); gamma-run decode for RLE segments
//     (src/main/wtree.c:690-712) reads its 64-bit window by lane shuffles.
// A count query owns a *pair* of groups (one wavefront at W = 32): group 0 ranks row first-1,
// group 1 ranks row last, so one wavefront advances one backward-search step of one pattern.
// Everything is integer/bit work: no MFMA, the roofline is HBM bandwidth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_tables.h"

namespace femto_amd {

__device__ __forceinline__ uint32_t ld_be32(const uint8_t* p) {
  return __builtin_bswap32(*reinterpret_cast<const uint32_t*>(p));
}
__device__ __forceinline__ uint64_t ld_be64(const uint8_t* p) {
  return __builtin_bswap64(*reinterpret_cast<const uint64_t*>(p));
}

template <int W>
struct Grp {
  static_assert(W == 64 || W == 32 || W == 16, "group width");
  static constexpr uint64_t kMask = W == 64 ? ~0ull : ((1ull << W) - 1);
  __device__ __forceinline__ static int lane() { return int(threadIdx.x) & (W - 1); }
  __device__ __forceinline__ static uint64_t ballot(bool p) {
    uint64_t b = __ballot(p);
    if (W == 64) return b;
    int sh = (int(threadIdx.x) & 63) & ~(W - 1);
    return (b >> sh) & kMask;
  }
  __device__ __forceinline__ static uint32_t shfl(uint32_t v, int src) { return uint32_t(__shfl(int(v), src, W)); }
  __device__ __forceinline__ static uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = uint32_t(__shfl(int(uint32_t(v)), src, W));
    uint32_t hi = uint32_t(__shfl(int(uint32_t(v >> 32)), src, W));
    return (uint64_t(hi) << 32) | lo;
  }
  __device__ __forceinline__ static uint64_t shfl_up64(uint64_t v, int d) {
    uint32_t lo = uint32_t(__shfl_up(int(uint32_t(v)), d, W));
    uint32_t hi = uint32_t(__shfl_up(int(uint32_t(v >> 32)), d, W));
    return (uint64_t(hi) << 32) | lo;
  }
};

struct RankResult {
  uint32_t o0, o1;  // zeros / ones in bits [1..index]
  uint32_t bit;     // bit[index]
};

// bseq_rank (src/main/wtree.c:635-763) by one group of W lanes.  All control flow is uniform
// within the group.  index1 is 1-based.
template <int W>
__device__ __forceinline__ RankResult bseq_rank(const uint8_t* __restrict__ image, const DevBseq bs, uint32_t index1) {
  using G = Grp<W>;
  const int lane = G::lane();
  const uint32_t t = index1 - 1;
  const uint8_t* z = image + bs.off;
  const uint32_t NG = bs.num_groups;
  const uint8_t* A0 = z + 16;
  const uint8_t* A1 = A0 + 4ull * NG;
  const uint8_t* AP = A1 + 4ull * NG;
  const uint8_t* S = AP + 4ull * NG;

  // ---- step 1: group with A0[g]+A1[g] <= t < A0[g+1]+A1[g+1]
  uint32_t o0 = 0, o1 = 0, ap = 0;
  uint32_t group = 0;
  for (uint32_t base = 0; base < NG; base += W) {
    const uint32_t g = base + uint32_t(lane);
    uint32_t a0 = 0, a1 = 0, p = 0;
    const bool in = g < NG;
    if (in) {
      a0 = ld_be32(A0 + 4ull * g);
      a1 = ld_be32(A1 + 4ull * g);
      p = ld_be32(AP + 4ull * g);
    }
    const uint64_t m = G::ballot(in && (a0 + a1 <= t));
    const int cnt = __popcll(m);
    if (cnt > 0) {
      o0 = G::shfl(a0, cnt - 1);
      o1 = G::shfl(a1, cnt - 1);
      ap = G::shfl(p, cnt - 1);
      group = base + uint32_t(cnt) - 1;
    }
    if (cnt < W) break;
  }

  // ---- step 2: walk the (zeros, ones) varbyte pairs of this group until the pair holding t
  uint32_t seg = 0;
  {
    const uint8_t* sp = S + ap;
    for (int round = 0; round < 2 * kGroupSize; round++) {
      const uint32_t b = sp[lane];
      const bool term = (b & 0x80u) != 0;
      const uint64_t T = G::ballot(term);
      const uint64_t below = T & ((1ull << lane) - 1);
      const int vid = __popcll(below);                            // value index inside this chunk
      const int vstart = below ? 64 - __clzll(below) : 0;         // first byte of my value
      const int p = lane - vstart;                                // my byte's position inside the value
      uint64_t x = p < 5 ? (uint64_t(b & 0x7fu) << (7 * p)) : 0;  // decode_varbyte, wtree_funcs.h:458
      if (vid & 1) x <<= 32;                                      // odd values are the ones-counts
#pragma unroll
      for (int d = 1; d < W; d <<= 1) {
        const uint64_t y = G::shfl_up64(x, d);
        if (lane >= d) x += y;
      }
      const bool pairend = term && (vid & 1);
      const uint32_t tot = o0 + o1 + uint32_t(x) + uint32_t(x >> 32);
      const uint64_t PE = G::ballot(pairend);
      const uint64_t OKm = G::ballot(pairend && tot <= t);
      const uint64_t bad = PE & ~OKm;
      if (bad) {
        const int first_bad = __ffsll((long long)bad) - 1;
        const uint64_t good = OKm & ((1ull << first_bad) - 1);
        if (good) {
          const uint64_t acc = G::shfl64(x, 63 - __clzll(good));
          o0 += uint32_t(acc);
          o1 += uint32_t(acc >> 32);
          seg += uint32_t(__popcll(good));
        }
        break;
      }
      if (!PE) break;  // corrupt data guard (a pair is at most 10 bytes)
      const int lastpe = 63 - __clzll(PE);
      const uint64_t acc = G::shfl64(x, lastpe);
      o0 += uint32_t(acc);
      o1 += uint32_t(acc >> 32);
      seg += uint32_t(__popcll(PE));
      sp += lastpe + 1;
      if (seg >= uint32_t(kGroupSize)) break;  // corrupt data guard
    }
  }
  const uint32_t segment = seg + uint32_t(kGroupSize) * group;

  // ---- step 3: the 64-byte segment (bseq_segment, wtree_funcs.h:482-511: zero past the end)
  uint64_t w = 0;
  {
    const uint32_t wi = kSegmentWords * segment + uint32_t(lane);
    if (lane < kSegmentWords && wi < bs.total_words) w = ld_be64(z + bs.d_off + 8ull * wi);
  }
  const uint64_t w0 = G::shfl64(w, 0);
  RankResult r;
  if (w0 >> 63) {
    // RLE segment: bit 1 = value of the first run, then Elias-gamma run lengths (wtree.c:690-712)
    uint32_t bit = uint32_t(w0 >> 62) & 1u;
    int p = 2;
    uint64_t win = 0;
    int avail = 0;
    for (int it = 0; it < 512; it++) {
      int k = win ? __clzll(win) : 64;
      if (2 * k + 1 > avail) {  // refill the 64-bit window at bit p (advance_segs_reader, wtree_funcs.h:113)
        const int wi = p >> 6, sh = p & 63;
        const uint64_t a = wi < kSegmentWords ? G::shfl64(w, wi) : 0;
        const uint64_t c = wi + 1 < kSegmentWords ? G::shfl64(w, wi + 1) : 0;
        win = (a << sh) | (sh ? (c >> (64 - sh)) : 0);
        avail = 64;
        k = win ? __clzll(win) : 64;
        if (k >= 32) break;  // no gamma code here: corrupt data guard
      }
      const int nb = 2 * k + 1;
      const uint32_t v = uint32_t(win >> (64 - nb));
      win = nb < 64 ? (win << nb) : 0;
      avail -= nb;
      p += nb;
      const uint32_t tot = o0 + o1;
      if (tot + v <= t) {
        if (bit) o1 += v; else o0 += v;
        bit ^= 1u;
      } else {
        const uint32_t rem = t + 1 - tot;
        if (bit) o1 += rem; else o0 += rem;
        break;
      }
    }
    r.bit = bit;
  } else {
    // literal segment: bit 0 is the flag, bits 1..511 are data (wtree.c:713-759)
    const uint32_t nb = 1 + t - o0 - o1;  // position of bit[index] inside the segment
    const uint32_t lo = 64u * uint32_t(lane);
    uint64_t m = 0;
    if (lane < kSegmentWords && nb >= lo) m = (nb - lo >= 63) ? ~0ull : (~0ull << (63 - (nb - lo)));
    uint32_t ones = uint32_t(__popcll(w & m));
    ones += uint32_t(__shfl_xor(int(ones), 1, W));
    ones += uint32_t(__shfl_xor(int(ones), 2, W));
    ones += uint32_t(__shfl_xor(int(ones), 4, W));
    ones = G::shfl(ones, 0);
    o1 += ones;
    o0 += nb - ones;
    const uint64_t wt = G::shfl64(w, int(nb >> 6));
    r.bit = uint32_t(wt >> (63 - (nb & 63))) & 1u;
  }
  r.o0 = o0;
  r.o1 = o1;
  return r;
}

// wtree_occs (src/main/wtree.c:1081-1115): Occ of leaf `code` at 1-based index inside one bucket.
template <int W>
__device__ __forceinline__ uint32_t wt_occs(const DevIndex& ix, const DevBucket bk, uint32_t code, uint32_t idx) {
  const int len = 31 - __clz(int(code));
  int cur = 0;
  for (int i = 1; i <= len; i++) {
    const DevNode nd = ix.nodes[bk.node_base + uint32_t(cur)];
    const RankResult r = bseq_rank<W>(ix.image, nd.bs, idx);
    const uint32_t b = (code >> (len - i)) & 1u;
    idx -= b ? r.o0 : r.o1;  // index -= occs[!bit]
    if (idx == 0) break;
    cur = b ? nd.child[1] : nd.child[0];
    if (cur < 0) break;
  }
  return idx;
}

// wtree_rank (src/main/wtree.c:1117-1148): (seq of L[index], Occ(L[index], index)) in one bucket.
template <int W>
__device__ __forceinline__ void wt_rank(const DevIndex& ix, const DevBucket bk, uint32_t idx, int* seq_out, uint32_t* cnt_out) {
  int cur = 0;
  int seq = -1;
  for (int depth = 0; depth < 32; depth++) {
    const DevNode nd = ix.nodes[bk.node_base + uint32_t(cur)];
    const RankResult r = bseq_rank<W>(ix.image, nd.bs, idx);
    idx -= r.bit ? r.o0 : r.o1;
    const int c = r.bit ? nd.child[1] : nd.child[0];
    if (c < 0) { seq = -1 - c; break; }
    cur = c;
  }
  *seq_out = seq;
  *cnt_out = idx;
}

__device__ __forceinline__ int64_t bucket_of(const DevIndex& ix, int64_t row, uint32_t* idx1) {
  int64_t gb;
  if (ix.b_shift >= 0) gb = row >> ix.b_shift;
  else gb = row / ix.b_size;
  *idx1 = uint32_t(row - gb * int64_t(ix.b_size)) + 1u;
  return gb;
}

// One backward-search job per pair of groups (do_string_query, src/main/server.c:713-946).
template <int W>
inline __global__ __launch_bounds__(256) void count_kernel(const DevIndex ix, const int64_t npats,
                                                    const int32_t* __restrict__ plen,
                                                    const uint16_t* __restrict__ pats,
                                                    const int64_t* __restrict__ starts,
                                                    int64_t* __restrict__ first_out,
                                                    int64_t* __restrict__ last_out, int* __restrict__ err_flag) {
  constexpr int PW = 2 * W;  // lanes per query
  const int64_t q = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / PW;
  if (q >= npats) return;
  const int pl = int(threadIdx.x) & (PW - 1);
  const int side = pl / W;  // group 0 ranks row first-1, group 1 ranks row last
  const int len = plen[q];
  const int64_t st = starts[q];
  int64_t first, last;
  if (len == 0) {  // server.c:782-808
    first = 0;
    last = ix.total_length - 1;
  } else {
    int i = len - 1;
    int cbase = (i / PW) * PW;
    uint32_t preg = (cbase + pl < len) ? pats[st + cbase + pl] : 0u;
    uint32_t ch = uint32_t(__shfl(int(preg), i - cbase, PW));
    if (ch >= uint32_t(kAlphaSize)) {
      if (pl == 0) atomicOr(err_flag, 1);
      first = 0;
      last = -1;
    } else {
      first = ix.C[ch];          // server.c:795-829
      last = ix.C[ch + 1] - 1;
      while (first <= last && i > 0) {  // server.c:832
        const int j = i - 1;
        if (j < cbase) {
          cbase -= PW;
          preg = pats[st + cbase + pl];
        }
        ch = uint32_t(__shfl(int(preg), j - cbase, PW));
        if (ch >= uint32_t(kAlphaSize)) {
          if (pl == 0) atomicOr(err_flag, 1);
          first = 0;
          last = -1;
          break;
        }
        const int64_t row = side ? last : first - 1;
        int64_t val;
        if (row < 0) {
          val = ix.C[ch];        // first == 0: only C[ch] (server.c:838-843, :884-888)
        } else {
          uint32_t idx1;
          const int64_t gb = bucket_of(ix, row, &idx1);
          const int64_t base = ix.occ_base[gb * kAlphaSize + ch];
          const uint32_t code = ix.leaf_code[gb * kAlphaSize + ch];
          uint32_t occ = 0;
          if (code) occ = wt_occs<W>(ix, ix.buckets[gb], code, idx1);  // absent characters add 0 (index.c:2080-2089)
          val = base + int64_t(occ);
        }
        // first = C+Occ(ch, first-1); last = C+Occ(ch, last) - 1 (server.c:909-936)
        const uint32_t vlo = uint32_t(val), vhi = uint32_t(uint64_t(val) >> 32);
        const uint32_t flo = uint32_t(__shfl(int(vlo), 0, PW)), fhi = uint32_t(__shfl(int(vhi), 0, PW));
        const uint32_t llo = uint32_t(__shfl(int(vlo), W, PW)), lhi = uint32_t(__shfl(int(vhi), W, PW));
        first = int64_t((uint64_t(fhi) << 32) | flo);
        last = int64_t((uint64_t(lhi) << 32) | llo) - 1;
        i--;
      }
    }
  }
  if (pl == 0) {
    first_out[q] = first;
    if (last_out) last_out[q] = last;
    else first_out[q] = last - first + 1;  // femto.c:313-318
  }
}

// do_locate_query clamp (src/main/server.c:4405-4415): note `last-first > max_occs`.
inline __global__ void clamp_kernel(const int64_t npats, const int64_t* __restrict__ first, const int64_t* __restrict__ last,
                             const int max_occs, int32_t* __restrict__ noccs, int64_t* __restrict__ noccs64) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= npats) return;
  const int64_t f = first[i], l = last[i];
  int64_t n;
  if (f > l) n = 0;
  else if (l - f > int64_t(max_occs)) n = max_occs;
  else n = l - f + 1;
  noccs[i] = int32_t(n);
  noccs64[i] = n;
}

// ---- exclusive prefix sum of int64 (three-kernel blocked scan, recursive on block sums)
constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;  // per thread
constexpr int kScanTile = kScanBlock * kScanItems;

inline __global__ __launch_bounds__(kScanBlock) void scan_tile_kernel(const int64_t n, const int64_t* __restrict__ in,
                                                              int64_t* __restrict__ out, int64_t* __restrict__ tile_sums) {
  __shared__ int64_t warp_tot[kScanBlock / 64];
  const int64_t base = int64_t(blockIdx.x) * kScanTile + int64_t(threadIdx.x) * kScanItems;
  int64_t v[kScanItems];
  int64_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    v[k] = base + k < n ? in[base + k] : 0;
    sum += v[k];
  }
  // inclusive scan of per-thread sums across the block
  int64_t x = sum;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t lo = uint32_t(__shfl_up(int(uint32_t(uint64_t(x))), d, 64));
    uint32_t hi = uint32_t(__shfl_up(int(uint32_t(uint64_t(x) >> 32)), d, 64));
    int64_t y = int64_t((uint64_t(hi) << 32) | lo);
    if (lane >= d) x += y;
  }
  if (lane == 63) warp_tot[wave] = x;
  __syncthreads();
  int64_t woff = 0;
  for (int k = 0; k < wave; k++) woff += warp_tot[k];
  int64_t excl = woff + x - sum;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    if (base + k < n) out[base + k] = excl;
    excl += v[k];
  }
  if (threadIdx.x == kScanBlock - 1) tile_sums[blockIdx.x] = woff + x;
}

inline __global__ void scan_add_kernel(const int64_t n, int64_t* __restrict__ out, const int64_t* __restrict__ tile_offs) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] += tile_offs[i / kScanTile];
}

inline __global__ void set_total_kernel(const int64_t n, const int64_t* __restrict__ excl, const int64_t* __restrict__ in, int64_t* __restrict__ out_n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out_n = n ? excl[n - 1] + in[n - 1] : 0;
}

// MSB-first bit field of `bits` bits at absolute bit position `bitpos` of image (bsInitReadAt+bsR64,
// src/utils/buffer_funcs.h:168-195); two aligned 8-byte loads + funnel shift.
__device__ __forceinline__ uint64_t read_bits(const uint8_t* __restrict__ image, uint64_t bitpos, int bits) {
  const uint64_t byte = bitpos >> 3;
  const uint64_t al = byte & ~7ull;
  const int s = int((byte & 7) * 8 + (bitpos & 7));
  const uint64_t a = ld_be64(image + al);
  const uint64_t b = ld_be64(image + al + 8);
  const uint64_t v = (a << s) | (s ? (b >> (64 - s)) : 0);
  return v >> (64 - bits);
}

// Range-split indexes (femto_amd_open_split) keep a sequence's segment lines / a character's mark array in
// ANOTHER GPU's memory, mapped into this process (hipIpcOpenMemHandle / peer access over xGMI).  The lane tables
// then hold offsets relative to the local base that wrap modulo 2^64, so the address arithmetic is done on
// integers and the same kernels serve both layouts with no owner lookup.
__device__ __forceinline__ const uint8_t* wrap_ptr(const void* base, uint64_t byte_off) {
  return reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(base) + uintptr_t(byte_off));
}
// read_bits for an arbitrary (wrapped) base pointer: `bits` bits starting `bitpos` bits after p
__device__ __forceinline__ uint64_t read_bits_ptr(const uint8_t* p, uint64_t bitpos, int bits) {
  const uintptr_t a0 = reinterpret_cast<uintptr_t>(p) + uintptr_t(bitpos >> 3);
  const uintptr_t al = a0 & ~uintptr_t(7);
  const int s = int((a0 & 7) * 8 + (bitpos & 7));
  const uint64_t a = ld_be64(reinterpret_cast<const uint8_t*>(al));
  const uint64_t b = ld_be64(reinterpret_cast<const uint8_t*>(al + 8));
  const uint64_t v = (a << s) | (s ? (b >> (64 - s)) : 0);
  return v >> (64 - bits);
}

// Record number of a marked row in its character's mark array: o1 - 1, at most one bucket's worth (the zero slack behind
// the image covers exactly that; a damaged mark table can report any count).
__device__ __forceinline__ uint64_t mark_rec(const DevIndex& ix, uint32_t o1) {
  const uint64_t rec = uint64_t(o1) - 1;
  return rec < uint64_t(ix.b_size) ? rec : uint64_t(ix.b_size);
}

// A damaged index can leave the tree walk without a leaf (seq = -1) or at a leaf number its bucket does not have: the
// result is garbage either way (as in the reference), but the table read stays inside the bucket's sequences.
__device__ __forceinline__ uint32_t seq_in_bucket(const DevBucket& bk, int seq) {
  return uint32_t(seq) < bk.n_in_use ? uint32_t(seq) : 0u;
}

// One row per group: walk LF backwards until a marked row (do_back_query, src/main/server.c:2228-2359,
// driven as do_context_query does with LOCATE_STRONG, :2627-2795).  offset = mark + steps (server.c:2718).
template <int W>
inline __global__ __launch_bounds__(256) void locate_kernel(const DevIndex ix, const int64_t npats,
                                                     const int64_t* __restrict__ first,
                                                     const int64_t* __restrict__ out_starts, const int64_t total,
                                                     int64_t* __restrict__ offsets) {
  const int64_t item = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / W;
  if (item >= total) return;
  // pattern owning this output slot: largest q with out_starts[q] <= item
  int64_t lo = 0, hi = npats;  // out_starts has npats+1 entries; invariant out_starts[lo] <= item < out_starts[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (out_starts[mid] <= item) lo = mid; else hi = mid;
  }
  int64_t row = first[lo] + (item - out_starts[lo]);
  int64_t steps = 0, result = -1;
  while (row >= 0 && steps <= int64_t(ix.walk_limit)) {
    uint32_t idx1;
    const int64_t gb = bucket_of(ix, row, &idx1);
    const DevBucket bk = ix.buckets[gb];
    int seq;
    uint32_t cnt;
    wt_rank<W>(ix, bk, idx1, &seq, &cnt);
    if (seq < 0 || uint32_t(seq) >= bk.n_in_use) break;  // corrupt data guard
    const DevSeq sq = ix.seqs[bk.seq_base + seq_in_bucket(bk, seq)];
    const RankResult m = bseq_rank<W>(ix.image, sq.mark_table, cnt);  // index.c:2102-2140
    if (m.bit) {
      const uint64_t rec = mark_rec(ix, m.o1);
      result = int64_t(read_bits(ix.image, sq.mark_array * 8 + rec * uint64_t(ix.text_size_bits), ix.text_size_bits)) + steps;
      break;
    }
    if (sq.ch <= uint32_t(kSEOF)) break;  // cannot walk past a document start (server.c:2336-2342)
    row = ix.occ_base[gb * kAlphaSize + sq.ch] + int64_t(cnt) - 1;  // LF (server.c:2279-2282)
    steps++;
  }
  if (Grp<W>::lane() == 0) offsets[item] = result;
}

// Leaf requests for parity tests (block_request CHAR|OCCS|LOCATION, src/main/index.c:1973-2144).
// occ_out = C[ch] + block_occs + Occ-in-block (the host subtracts the header part).
template <int W>
inline __global__ __launch_bounds__(256) void block_request_kernel(const DevIndex ix, const int64_t n,
                                                            const int64_t* __restrict__ rows,
                                                            const uint16_t* __restrict__ ch_in,
                                                            uint16_t* __restrict__ ch_out,
                                                            int64_t* __restrict__ occ_out,
                                                            int64_t* __restrict__ off_out) {
  const int64_t item = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / W;
  if (item >= n) return;
  const int64_t row = rows[item];
  uint32_t idx1;
  const int64_t gb = bucket_of(ix, row, &idx1);
  const DevBucket bk = ix.buckets[gb];
  int seq;
  uint32_t cnt;
  wt_rank<W>(ix, bk, idx1, &seq, &cnt);
  const DevSeq sq = ix.seqs[bk.seq_base + seq_in_bucket(bk, seq)];
  int64_t off = -1;
  const RankResult m = bseq_rank<W>(ix.image, sq.mark_table, cnt);
  if (m.bit) off = int64_t(read_bits(ix.image, sq.mark_array * 8 + mark_rec(ix, m.o1) * uint64_t(ix.text_size_bits), ix.text_size_bits));
  int64_t occ;
  uint32_t ch = sq.ch;
  if (ch_in) {
    ch = ch_in[item];
    const uint32_t code = ix.leaf_code[gb * kAlphaSize + ch];
    occ = ix.occ_base[gb * kAlphaSize + ch] + (code ? int64_t(wt_occs<W>(ix, bk, code, idx1)) : 0);
  } else {
    occ = ix.occ_base[gb * kAlphaSize + ch] + int64_t(cnt);
  }
  if (Grp<W>::lane() == 0) {
    if (ch_out) ch_out[item] = uint16_t(sq.ch);
    if (occ_out) occ_out[item] = occ;
    if (off_out) off_out[item] = off;
  }
}


// =====================================================================================
// Lane-per-item kernels over the derived segment lines (device_tables.h: LaneBseq) -- the default.
//
// Measured on MI355X (profiles/r01_v1_*): the wavefront-cooperative walk above spends ~190 VALU
// wave-instructions per rank (cross-lane scans/shuffles) and is VALU-issue + latency bound at 5 %
// of the HBM roofline.  Here every LANE owns a whole query: a rank on an all-literal sequence is ONE
// 128-byte line (segment words + the zeros/ones before it), a few instructions per rank per lane once
// amortised over the 64 lanes of a wavefront, so the kernel is limited by the memory system
// (scattered 128-byte lines), which is what the roofline prices.
// Lanes of a wavefront start their patterns together, so the early backward-search steps (few
// distinct ranges) hit the same cache lines; measured, this beats the flattened variant below.
// =====================================================================================

__device__ __forceinline__ uint64_t sel8(const uint64_t (&w)[kSegmentWords], int i) {
  const uint64_t a = (i & 1) ? w[1] : w[0];
  const uint64_t b = (i & 1) ? w[3] : w[2];
  const uint64_t c = (i & 1) ? w[5] : w[4];
  const uint64_t d = (i & 1) ? w[7] : w[6];
  const uint64_t ab = (i & 2) ? b : a;
  const uint64_t cd = (i & 2) ? d : c;
  const uint64_t r = (i & 4) ? cd : ab;
  return (i & ~7) ? 0 : r;
}

// RLE skip table (device_tables.h): jump to the last 64-bit boundary of the gamma stream whose
// preceding runs all lie before the target bit `rel` (bits of this segment before the target).
__device__ __forceinline__ void rle_skip(const uint64_t* __restrict__ aux, uint32_t rel, uint32_t& o0, uint32_t& o1,
                                         uint32_t& bit, int& p) {
  const ulonglong2* ap = reinterpret_cast<const ulonglong2*>(aux);
  const ulonglong2 a01 = ap[0], a23 = ap[1], a45 = ap[2], a67 = ap[3];
  const uint64_t e[7] = {a01.y, a23.x, a23.y, a45.x, a45.y, a67.x, a67.y};
  uint64_t best = 0;
  int k = 0;
#pragma unroll
  for (int j = 0; j < 7; j++) {
    if (uint32_t(e[j]) <= rel) {  // totals are non-decreasing; 0xffffffff marks "no code starts here"
      best = e[j];
      k = j + 1;
    }
  }
  if (k) {
    const uint32_t total = uint32_t(best), hi = uint32_t(best >> 32);
    const uint32_t ones = hi & 0x7fffffffu;
    o0 += total - ones;
    o1 += ones;
    bit = hi >> 31;
    p = int((a01.x >> (9 * (k - 1))) & 0x1ffu);
  }
}

// bseq_rank (src/main/wtree.c:635-763) by ONE lane, in three stages so that callers may interleave the loads
// of independent ranks: (1) which segment -- t/511 for all-literal sequences, one 32-byte block-directory
// entry otherwise; (2) the segment's 128-byte line; (3) masked popcount (literal) or gamma runs (RLE, entered
// through the skip table).
struct RankJob {
  uint32_t t;          // 0-based bit position
  uint32_t seg, o0, o1;
  uint64_t slot;       // segment slot index in DevIndex::segs
  bool has_aux;        // the next slot holds the RLE skip table
};

__device__ __forceinline__ void rank_locate_segment(const DevIndex& ix, const LaneBseq bs, uint32_t index1, RankJob& j) {
  j.t = index1 - 1;
  if (bs.hint_base == kNoHint) {
    j.seg = j.t / 511u;
    j.o0 = j.o1 = 0;   // read from word 8 of the segment's line in rank_load_segment
  } else {
    const uint4* dp = reinterpret_cast<const uint4*>(ix.bdir + (uint64_t(bs.hint_base) + (j.t >> 9)));
    const uint4 d0 = dp[0];
    const uint32_t n1 = reinterpret_cast<const uint32_t*>(dp)[4];
    j.seg = d0.x;
    j.o0 = d0.y;
    j.o1 = d0.z;
    if (j.t >= d0.w + n1) {
      j.o0 = d0.w;
      j.o1 = n1;
      j.seg++;
    }
  }
  j.slot = bs.seg_base + 2ull * j.seg;
  j.has_aux = bs.hint_base != kNoHint;
}

__device__ __forceinline__ void rank_load_segment(const DevIndex& ix, RankJob& j, uint64_t (&w)[kSegmentWords]) {
  const ulonglong2* sp = reinterpret_cast<const ulonglong2*>(wrap_ptr(ix.segs, j.slot * (kSegmentWords * 8ull)));
#pragma unroll
  for (int k = 0; k < kSegmentWords / 2; k++) {
    const ulonglong2 v = sp[k];
    w[2 * k] = v.x;
    w[2 * k + 1] = v.y;
  }
  if (!j.has_aux) {
    const uint64_t c = reinterpret_cast<const uint64_t*>(sp)[kSegmentWords];
    j.o0 = uint32_t(c);
    j.o1 = uint32_t(c >> 32);
  }
}

__device__ __forceinline__ RankResult rank_finish(const DevIndex& ix, const RankJob& j, const uint64_t (&w)[kSegmentWords]) {
  uint32_t o0 = j.o0, o1 = j.o1;
  const uint32_t t = j.t;
  RankResult r;
  if (w[0] >> 63) {  // RLE segment (wtree.c:690-712)
    uint32_t bit = uint32_t(w[0] >> 62) & 1u;
    int p = 2;
    if (j.has_aux) rle_skip(reinterpret_cast<const uint64_t*>(wrap_ptr(ix.segs, (j.slot + 1) * (kSegmentWords * 8ull))), t - o0 - o1, o0, o1, bit, p);
    uint64_t win = 0;
    int avail = 0;
    for (int it = 0; it < 512; it++) {
      int k = win ? __clzll(win) : 64;
      if (2 * k + 1 > avail) {
        const int wi = p >> 6, sh = p & 63;
        const uint64_t a = sel8(w, wi);
        const uint64_t c = sel8(w, wi + 1);
        win = (a << sh) | (sh ? (c >> (64 - sh)) : 0);
        avail = 64;
        k = win ? __clzll(win) : 64;
        if (k >= 32) break;
      }
      const int nb = 2 * k + 1;
      const uint32_t v = uint32_t(win >> (64 - nb));
      win = nb < 64 ? (win << nb) : 0;
      avail -= nb;
      p += nb;
      const uint32_t tot = o0 + o1;
      if (tot + v <= t) {
        if (bit) o1 += v; else o0 += v;
        bit ^= 1u;
      } else {
        const uint32_t rem = t + 1 - tot;
        if (bit) o1 += rem; else o0 += rem;
        break;
      }
    }
    r.bit = bit;
  } else {  // literal segment (wtree.c:713-759)
    const uint32_t nb = 1 + t - o0 - o1;
    uint32_t ones = 0;
    uint64_t bw = 0;
#pragma unroll
    for (int k = 0; k < kSegmentWords; k++) {
      const uint32_t lo = 64u * uint32_t(k);
      uint64_t m = 0;
      if (nb >= lo) m = (nb - lo >= 63) ? ~0ull : (~0ull << (63 - (nb - lo)));
      ones += uint32_t(__popcll(w[k] & m));
      if ((nb >> 6) == uint32_t(k)) bw = w[k];
    }
    o1 += ones;
    o0 += nb - ones;
    r.bit = uint32_t(bw >> (63 - (nb & 63))) & 1u;
  }
  r.o0 = o0;
  r.o1 = o1;
  return r;
}

__device__ __forceinline__ RankResult bseq_rank_lane(const DevIndex& ix, const LaneBseq bs, uint32_t index1) {
  RankJob j;
  uint64_t w[kSegmentWords];
  rank_locate_segment(ix, bs, index1, j);
  rank_load_segment(ix, j, w);
  return rank_finish(ix, j, w);
}

// wtree_occs (src/main/wtree.c:1081-1115), one lane
__device__ __forceinline__ uint32_t wt_occs_lane(const DevIndex& ix, const uint32_t node_base, uint32_t code, uint32_t idx) {
  const int len = 31 - __clz(int(code));
  int cur = 0;
  for (int i = 1; i <= len; i++) {
    const LaneNode nd = ix.lnodes[node_base + uint32_t(cur)];
    const RankResult r = bseq_rank_lane(ix, nd.bs, idx);
    const uint32_t b = (code >> (len - i)) & 1u;
    idx -= b ? r.o0 : r.o1;
    if (idx == 0) break;
    cur = b ? nd.child[1] : nd.child[0];
    if (cur < 0) break;
  }
  return idx;
}

// wtree_rank (src/main/wtree.c:1117-1148), one lane
__device__ __forceinline__ void wt_rank_lane(const DevIndex& ix, const uint32_t node_base, uint32_t idx, int* seq_out, uint32_t* cnt_out) {
  int cur = 0, seq = -1;
  for (int depth = 0; depth < 32; depth++) {
    const LaneNode nd = ix.lnodes[node_base + uint32_t(cur)];
    const RankResult r = bseq_rank_lane(ix, nd.bs, idx);
    idx -= r.bit ? r.o0 : r.o1;
    const int c = r.bit ? nd.child[1] : nd.child[0];
    if (c < 0) { seq = -1 - c; break; }
    cur = c;
  }
  *seq_out = seq;
  *cnt_out = idx;
}

// C[ch] + Occ(ch,row): header + block + bucket bases and the wavelet walk, one lane
__device__ __forceinline__ int64_t c_plus_occ_lane(const DevIndex& ix, uint32_t ch, int64_t row) {
  uint32_t idx1;
  const int64_t gb = bucket_of(ix, row, &idx1);
  const OccEntry oe = ix.occ[gb * kAlphaSize + ch];
  uint32_t occ = 0;
  if (oe.code) occ = wt_occs_lane(ix, oe.node_base, oe.code, idx1);
  return oe.base + int64_t(occ);
}

// One backward-search step for MANY (row range, character) pairs: the fan-out of do_regexp_query (src/main/server.c:1656,
// states 0x200-0x411: "first = C[ch] + Occ(ch, first-1); last = C[ch] + Occ(ch, last) - 1" for every reachable character
// of every range in flight), one lane per pair on femto's own wavelet tree.
inline __global__ __launch_bounds__(256) void ranges_step_kernel(const DevIndex ix, const int64_t n, const int64_t* __restrict__ first,
                                                          const int64_t* __restrict__ last, const uint16_t* __restrict__ ch,
                                                          int64_t* __restrict__ first_out, int64_t* __restrict__ last_out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t f = first[i], l = last[i];
  const uint32_t c = ch[i];
  first_out[i] = f == 0 ? ix.C[c] : c_plus_occ_lane(ix, c, f - 1);
  last_out[i] = c_plus_occ_lane(ix, c, l) - 1;
}

// do_string_query (src/main/server.c:713-946): one LANE per pattern
// `perm` (optional): lane j processes pattern perm[j] -- the batch ordered by pattern suffix
// (query_sort.hip) so that neighbouring lanes share the rows of their first steps.
inline __global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void count_kernel_lane(const DevIndex ix, const int64_t npats,
                                                         const int32_t* __restrict__ plen,
                                                         const uint16_t* __restrict__ pats,
                                                         const int64_t* __restrict__ starts,
                                                         int64_t* __restrict__ first_out,
                                                         int64_t* __restrict__ last_out, int* __restrict__ err_flag,
                                                         const uint32_t* __restrict__ perm) {
  const int64_t slot = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (slot >= npats) return;
  const int64_t q = perm ? int64_t(perm[slot]) : slot;
  const int len = plen[q];
  const uint16_t* pat = pats + starts[q];
  int64_t first, last;
  if (len == 0) {
    first = 0;
    last = ix.total_length - 1;
  } else {
    int i = len - 1;
    uint32_t ch = pat[i];
    if (ch >= uint32_t(kAlphaSize)) {
      atomicOr(err_flag, 1);
      first = 0;
      last = -1;
    } else {
      first = ix.C[ch];
      last = ix.C[ch + 1] - 1;
      while (first <= last && i > 0) {
        ch = pat[i - 1];
        if (ch >= uint32_t(kAlphaSize)) {
          atomicOr(err_flag, 1);
          first = 0;
          last = -1;
          break;
        }
        const int64_t nf = first == 0 ? ix.C[ch] : c_plus_occ_lane(ix, ch, first - 1);
        const int64_t nl = c_plus_occ_lane(ix, ch, last);
        first = nf;
        last = nl - 1;
        i--;
      }
    }
  }
  first_out[q] = first;
  if (last_out) last_out[q] = last;
  else first_out[q] = last - first + 1;
}

// locate walk (do_back_query / do_context_query), one LANE per located row
inline __global__ __launch_bounds__(256) void locate_kernel_lane(const DevIndex ix, const int64_t npats,
                                                          const int64_t* __restrict__ first,
                                                          const int64_t* __restrict__ out_starts, const int64_t total,
                                                          int64_t* __restrict__ offsets) {
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= total) return;
  int64_t lo = 0, hi = npats;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (out_starts[mid] <= item) lo = mid; else hi = mid;
  }
  int64_t row = first[lo] + (item - out_starts[lo]);
  int64_t steps = 0, result = -1;
  while (row >= 0 && steps <= int64_t(ix.walk_limit)) {
    uint32_t idx1;
    const int64_t gb = bucket_of(ix, row, &idx1);
    const DevBucket bk = ix.buckets[gb];
    int seq;
    uint32_t cnt;
    wt_rank_lane(ix, bk.node_base, idx1, &seq, &cnt);
    if (seq < 0 || uint32_t(seq) >= bk.n_in_use) break;
    const LaneSeq sq = ix.lseqs[bk.seq_base + seq_in_bucket(bk, seq)];
    const RankResult m = bseq_rank_lane(ix, sq.mark_table, cnt);
    if (m.bit) {
      const uint64_t rec = mark_rec(ix, m.o1);
      result = int64_t(read_bits_ptr(wrap_ptr(ix.image, sq.mark_array), rec * uint64_t(ix.text_size_bits), ix.text_size_bits)) + steps;
      break;
    }
    if (sq.ch <= uint32_t(kSEOF)) break;
    row = ix.occ_base[gb * kAlphaSize + sq.ch] + int64_t(cnt) - 1;
    steps++;
  }
  offsets[item] = result;
}

inline __global__ __launch_bounds__(256) void block_request_kernel_lane(const DevIndex ix, const int64_t n,
                                                                 const int64_t* __restrict__ rows,
                                                                 const uint16_t* __restrict__ ch_in,
                                                                 uint16_t* __restrict__ ch_out,
                                                                 int64_t* __restrict__ occ_out,
                                                                 int64_t* __restrict__ off_out) {
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= n) return;
  const int64_t row = rows[item];
  uint32_t idx1;
  const int64_t gb = bucket_of(ix, row, &idx1);
  const DevBucket bk = ix.buckets[gb];
  int seq;
  uint32_t cnt;
  wt_rank_lane(ix, bk.node_base, idx1, &seq, &cnt);
  const LaneSeq sq = ix.lseqs[bk.seq_base + seq_in_bucket(bk, seq)];
  int64_t off = -1;
  const RankResult m = bseq_rank_lane(ix, sq.mark_table, cnt);
  if (m.bit) off = int64_t(read_bits_ptr(wrap_ptr(ix.image, sq.mark_array), mark_rec(ix, m.o1) * uint64_t(ix.text_size_bits), ix.text_size_bits));
  int64_t occ;
  uint32_t ch = sq.ch;
  if (ch_in) {
    ch = ch_in[item];
    occ = c_plus_occ_lane(ix, ch, row);
  } else {
    occ = ix.occ_base[gb * kAlphaSize + ch] + int64_t(cnt);
  }
  if (ch_out) ch_out[item] = uint16_t(sq.ch);
  if (occ_out) occ_out[item] = occ;
  if (off_out) off_out[item] = off;
}

// =====================================================================================
// LF^-1 (do_forward_query, src/main/server.c:2424-2565): not needed for exact locate results (the
// backward walk always meets a mark first) but part of the reference's leaf interface
// (BLOCK_REQUEST_ROW, wtree_select, bseq_select).  One lane per row, on femto's RAW tables
// (A0/A1/AP group arrays, varbyte S sums, D segments in `image`), following the reference's own
// select algorithm step for step.
// =====================================================================================

// bseq_select (src/main/wtree.c:770-885): occs[] such that occs[0]+occs[1] is the 1-based index of the
// rank1'th occurrence of `bit`.
__device__ __forceinline__ void bseq_select_lane(const uint8_t* __restrict__ image, const DevBseq bs, const uint32_t bit,
                                                 const uint32_t rank1, uint32_t& out0, uint32_t& out1) {
  const uint32_t rank = rank1 - 1;
  const uint8_t* z = image + bs.off;
  const uint32_t NG = bs.num_groups;
  const uint8_t* A0 = z + 16;
  const uint8_t* A1 = A0 + 4ull * NG;
  const uint8_t* AP = A1 + 4ull * NG;
  const uint8_t* S = AP + 4ull * NG;
  const uint8_t* Ab = bit ? A1 : A0;
  // bsearch_A0A1 on one array (wtree.c:609-629)
  uint32_t a = 0, b = NG - 1, group;
  if (rank >= ld_be32(Ab + 4ull * b)) group = b;
  else {
    while (b - a > 1) {
      const uint32_t m = (a + b) / 2;
      if (rank < ld_be32(Ab + 4ull * m)) b = m; else a = m;
    }
    group = a;
  }
  uint32_t o[2] = {ld_be32(A0 + 4ull * group), ld_be32(A1 + 4ull * group)};
  const uint8_t* sp = S + ld_be32(AP + 4ull * group);
  uint32_t segment = 0;
  for (int j = 0; j < kGroupSize; j++) {  // decode_varbyte pairs (wtree_funcs.h:458)
    uint32_t s[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      uint32_t val = 0;
      for (int i = 0; i < 5; i++) {
        const uint32_t w = *sp++;
        val |= (w & 0x7fu) << (7 * i);
        if (w & 0x80u) break;
      }
      s[k] = val;
    }
    const uint32_t ob = bit ? o[1] : o[0], sb = bit ? s[1] : s[0];
    if (ob + sb <= rank) { o[0] += s[0]; o[1] += s[1]; segment++; }
    else break;
  }
  segment += uint32_t(kGroupSize) * group;
  uint64_t w[kSegmentWords];
#pragma unroll
  for (int k = 0; k < kSegmentWords; k++) {
    const uint32_t wi = kSegmentWords * segment + uint32_t(k);
    w[k] = wi < bs.total_words ? ld_be64(z + bs.d_off + 8ull * wi) : 0;
  }
  uint32_t o0 = o[0], o1 = o[1];
  if (w[0] >> 63) {  // RLE: wtree.c:818-832
    uint32_t rb = uint32_t(w[0] >> 62) & 1u;
    int p = 2;
    for (int it = 0; it < 512; it++) {
      const int wi = p >> 6, sh = p & 63;
      const uint64_t x = sel8(w, wi), y = sel8(w, wi + 1);
      const uint64_t win = (x << sh) | (sh ? (y >> (64 - sh)) : 0);
      if (!win) break;
      const int k = __clzll(win);
      if (k >= 32) break;
      const int nb = 2 * k + 1;
      const uint32_t v = uint32_t(win >> (64 - nb));
      p += nb;
      const uint32_t ob = bit ? o1 : o0;
      if (bit != rb || ob + v <= rank) {
        if (rb) o1 += v; else o0 += v;
        rb ^= 1u;
      } else {
        if (bit) o1 = rank + 1; else o0 = rank + 1;  // o[bit] += 1 + rank - o[bit]
        break;
      }
    }
  } else {  // literal: wtree.c:833-880
    int word_idx = 0;
    for (; word_idx < kSegmentWords; word_idx++) {
      const uint32_t c1 = uint32_t(__popcll(sel8(w, word_idx)));
      uint32_t c0 = 64 - c1;
      if (word_idx == 0) c0--;  // the is_rle flag bit
      const uint32_t ob = bit ? o1 : o0, cb = bit ? c1 : c0;
      if (ob + cb <= rank) { o0 += c0; o1 += c1; }
      else break;
    }
    uint64_t tmp = sel8(w, word_idx);
    if (word_idx == 0) tmp <<= 1;
    for (int k = 0; k < 64; k++) {
      const uint32_t rb = uint32_t(tmp >> 63);
      const uint32_t ob = bit ? o1 : o0;
      if (bit != rb || ob + 1 <= rank) {
        if (rb) o1++; else o0++;
      } else {
        if (bit) o1 = rank + 1; else o0 = rank + 1;
        break;
      }
      tmp <<= 1;
    }
  }
  out0 = o0;
  out1 = o1;
}

// do_forward_query for one row per lane: chr = F[row] (bsearch_C, index.c:1522), the bucket holding the
// (row+1-C[chr])'th occurrence of chr (bsearch_block_occs + bsearch_bucket_occs, index.c:1571,1847, here one
// search over the combined Occ bases), wtree_select (wtree.c:1150-1178) and the mark lookup at the row found.
inline __global__ __launch_bounds__(256) void forward_kernel(const DevIndex ix, const int64_t n, const int64_t* __restrict__ rows,
                                                      uint16_t* __restrict__ ch_out, int64_t* __restrict__ row_out,
                                                      int64_t* __restrict__ off_out) {
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= n) return;
  const int64_t row = rows[item];
  int lo = 0, hi = kAlphaSize;  // largest ch with C[ch] <= row (C[0] == 0)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ix.C[mid] <= row) lo = mid; else hi = mid;
  }
  const uint32_t ch = uint32_t(lo);
  int64_t new_row = -1, off = -1;
  if (ch > uint32_t(kSEOF)) {
    int64_t a = 0, b = ix.total_buckets;  // largest gb with base(gb, ch) <= row
    while (b - a > 1) {
      const int64_t mid = (a + b) >> 1;
      if (ix.occ[mid * kAlphaSize + ch].base <= row) a = mid; else b = mid;
    }
    const int64_t gb = a;
    const OccEntry oe = ix.occ[gb * kAlphaSize + ch];
    const DevBucket bk = ix.buckets[gb];
    uint32_t rank = uint32_t(row + 1 - oe.base);  // which occurrence of ch inside the bucket
    const uint32_t count = rank;
    const uint32_t code = oe.code;
    if (code) {
      const int len = 31 - __clz(int(code));
      // root-to-leaf node indexes of ch's code, then select bottom-up
      int path[20];
      int cur = 0, seq = -1;
      for (int i = 1; i <= len; i++) {
        path[i - 1] = cur;
        const DevNode nd = ix.nodes[bk.node_base + uint32_t(cur)];
        const uint32_t bbit = (code >> (len - i)) & 1u;
        const int c = bbit ? nd.child[1] : nd.child[0];
        if (c < 0) { seq = -1 - c; break; }
        cur = c;
      }
      for (int i = len; i >= 1; i--) {
        const DevNode nd = ix.nodes[bk.node_base + uint32_t(path[i - 1])];
        const uint32_t bbit = (code >> (len - i)) & 1u;
        uint32_t s0, s1;
        bseq_select_lane(ix.image, nd.bs, bbit, rank, s0, s1);
        rank = s0 + s1;
      }
      new_row = gb * int64_t(ix.b_size) + int64_t(rank) - 1;
      if (seq >= 0 && uint32_t(seq) < bk.n_in_use) {  // BLOCK_REQUEST_LOCATION at the row found
        const LaneSeq sq = ix.lseqs[bk.seq_base + seq_in_bucket(bk, seq)];
        const RankResult m = bseq_rank_lane(ix, sq.mark_table, count);
        if (m.bit)
          off = int64_t(read_bits(ix.image, sq.mark_array * 8 + mark_rec(ix, m.o1) * uint64_t(ix.text_size_bits), ix.text_size_bits));
      }
    }
  }
  ch_out[item] = uint16_t(ch);
  row_out[item] = new_row;
  off_out[item] = off;
}

}  // namespace femto_amd
