// direct_kernels.hip.hpp -- the batch pipeline of the packed layouts (modes 3 and 4) WITHOUT a batch sort.
//
// Round 1 ordered every batch by pattern suffix (key kernel + four radix-sort passes), searched from the sort keys
// and scattered 16-byte results back to the caller's order: 35 % of the headline step was not the search kernel.
// This pipeline processes the batch in the CALLER's order with one lane per pattern:
//
//   count_direct_kernel   pattern symbols are read straight from the caller's buffer (lane q reads pattern q: the
//                         wavefront's loads cover one contiguous span, so they are coalesced -- the line-per-symbol cost
//                         that made round 1 search from sort keys only exists when a batch is processed out of order);
//                         the first steps come from the LEVEL TABLE below; the remaining steps walk the packed lines;
//                         (first,last) and the clamped row count are stored coalesced, and the block's row-count sum
//                         goes to block_sums[] (do_locate_query's clamp, src/main/server.c:4405-4415, fused);
//                         every block also adds its sum to its group-of-64's word (PlanSums::super, one atomic): there is
//                         no scan kernel on the step's critical path;
//   count_tail_kernel     (text_kernels.hip.hpp) long patterns whose range is one row: compared against the text;
//   plan_rows_kernel      block offset = the group sums before the block's group + the block sums before it inside the
//                         group; out_starts[] = block offset + in-block scan; the rows to locate are written where their
//                         offsets will go (setup_locate_range, src/main/server.c:4047); the last block publishes the total;
//   locate_walk_kernel    persistent grid, reads the total from the device word: no host round trip inside a step.
//
// LEVEL TABLE (ktab2).  The first steps of a backward search depend only on the pattern's last symbols and are shared by
// huge numbers of patterns, so they are precomputed at open, on the GPU, from the packed lines themselves: for every
// string s of at most K "table characters" (the characters of the text that are not <= SEOF; digit = dense code -
// nstop, base t) the entry at heap position pos(s) -- pos(empty) = 0, pos(s.c) = pos(s)*t + 1 + digit(c), so level m
// occupies [ (t^m-1)/(t-1), (t^(m+1)-1)/(t-1) ) -- holds exactly the values do_string_query's loop
// (src/main/server.c:832-936) holds after searching s, including an early death:  x = first,  y = (last + 1) | m << 48.
// A pattern shorter than K, or one that meets a character outside the table, leaves the table at its level and
// continues symbol by symbol.  K: the deepest level has at most four entries per row (t^K <= 4 rows; K = 16 for a 2^30-row
// DNA index) and the table at most a quarter of the free HBM; the deepest level is stored in 8 bytes per entry (see
// ktab2_deep_kernel): 22.9 + 34.4 GB there.  FEMTO_AMD_KTAB_SYMS / FEMTO_AMD_KTAB_MB override.
#pragma once
#include <type_traits>

namespace femto_amd {

// ---- level table construction: one launch per level, entries of level m+1 from their parents in level m -----------
template <class P>
inline __global__ __launch_bounds__(256) void ktab2_level_kernel(const DevIndex ix, const int level /* m+1 >= 1 */, const int64_t lo, const int64_t n,
                                                          longlong2* __restrict__ tab) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t pos = lo + i;
  const int64_t t = ix.kt2_base;
  const int64_t parent = (pos - 1) / t;
  const uint32_t digit = uint32_t((pos - 1) - parent * t);
  const longlong2 e = tab[parent];
  int64_t first = e.x, last = int64_t(uint64_t(e.y) & kKtabLastMask) - 1;
  if (first <= last) P::search_step(ix, level - 1, digit + uint32_t(ix.kt2_nstop), first, last);
  tab[pos] = make_longlong2(first, int64_t(uint64_t(last + 1) | (uint64_t(level) << 48)));
}

inline __global__ void ktab2_root_kernel(const DevIndex ix, longlong2* __restrict__ tab) {
  if (threadIdx.x == 0 && blockIdx.x == 0) tab[0] = make_longlong2(0, ix.total_length);   // empty pattern: [0, n-1] (server.c:782-808)
}

// The deepest levels hold nearly all the entries and their ranges are short, so the last TWO levels (from kt2_cfrom on) are
// stored compactly: 8 bytes = first (40 bits: the format's 2^39 rows) | number of rows (24 bits).  A dead range has
// last == first - 1 by construction (first = C+Occ(c,first-1), last = C+Occ(c,last)-1 with equal Occs), i.e. 0 rows; a range
// of 2^24 - 1 rows or more (highly repetitive text) stores 0xffffff and is recomputed from its parent with one ordinary
// step.  (DevIndex::kt2_deep_big is that bound; a test lowers it -- FEMTO_AMD_KTAB_DEEP_BIG -- so that the recomputation
// runs on every fixture.)  Heap positions of consecutive levels are adjacent, so the compact array is indexed by
// pos - kt2_deep_off across both levels.  (Round 3 kept only the deepest level compact: 16 instead of 8 bytes for a quarter
// as many entries again -- 0.13 GB of a K = 13 table, 8.6 GB of the K = 16 one.)
//
// ONE-ROW entries with their position (DevIndex::kt2_sa1; round 6).  An entry that holds one row spends 24 bits on the number 1.
// On an index of at most 2^31 rows whose suffix array is resident when the table is built such an entry is stored as
// bit 63 | SA[first] << 31 | first instead: a pattern that leaves the table on one row -- four in five of the 20-mers sampled
// from a 2^30-row DNA text leave a K = 16 table that way -- then knows its row's text position and goes from the table line
// straight to the text (count_direct_kernel's sa_hint), without the suffix-array read in between; a row-free locate is then two
// requests per pattern, table and text.  Bit 63 is free for this: the "recompute" bound is capped at 2^23 rows with kt2_sa1
// (0x800000 .. 0xfffffe never appear as row counts), and 0xffffff itself has bits 40 .. 62 all set, which SA < 2^31 never has.
constexpr uint64_t kDeepBig = 0xffffffu;
constexpr uint64_t kDeepFirstMask = (uint64_t(1) << 40) - 1;
constexpr uint64_t kDeepOneRow = uint64_t(1) << 63;

// the table entry of heap position `pos` at level m: the (first,last) after searching those m symbols; *sa1 = SA[first] where the
// entry carries it (else untouched)
template <class P>
__device__ __forceinline__ void ktab2_lookup(const DevIndex& ix, int64_t pos, int m, int64_t& first, int64_t& last, int64_t* sa1 = nullptr) {
  const int64_t t = ix.kt2_base;
  uint32_t digit[2];
  int up = 0;               // compact entries that said "recompute": their digits, to be stepped again from the ancestor
  while (ix.kt2_deep && m >= ix.kt2_cfrom) {
    const int64_t i = pos - ix.kt2_deep_off;
    const uint64_t e = ix.kt2_deep[i];
    trace_touch(ix, kTraceKtab, uint64_t(ix.kt2_deep_off >> 3) + 1 + (uint64_t(i) >> 4));
    const uint64_t rows = e >> 40;
    if (ix.kt2_sa1 && (e & kDeepOneRow) && rows != kDeepBig) {
      first = last = int64_t(e & 0x7fffffffu);
      if (sa1 && up == 0) *sa1 = int64_t((e >> 31) & 0xffffffffu);
      break;
    }
    if (rows != kDeepBig) {
      first = int64_t(e & kDeepFirstMask);
      last = first + int64_t(rows) - 1;
      break;
    }
    const int64_t parent = (pos - 1) / t;
    digit[up++] = uint32_t((pos - 1) - parent * t);
    pos = parent;
    m--;
  }
  if (!(ix.kt2_deep && m >= ix.kt2_cfrom)) {
    const longlong2 e = reinterpret_cast<const longlong2*>(ix.ktab2)[pos];
    trace_touch(ix, kTraceKtab, uint64_t(pos) >> 3);
    first = e.x;
    last = int64_t(uint64_t(e.y) & kKtabLastMask) - 1;
  }
  while (up > 0) {          // (a big range is never empty: the steps below always run on rows)
    up--;
    P::search_step(ix, m, digit[up] + uint32_t(ix.kt2_nstop), first, last);
    m++;
  }
}

template <class P>
inline __global__ __launch_bounds__(256) void ktab2_deep_kernel(const DevIndex ix, const int level, const int64_t lo, const int64_t n,
                                                         uint64_t* __restrict__ deep) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t pos = lo + i;
  const int64_t t = ix.kt2_base;
  const int64_t parent = (pos - 1) / t;
  const uint32_t digit = uint32_t((pos - 1) - parent * t);
  int64_t first, last;
  ktab2_lookup<P>(ix, parent, level - 1, first, last);      // (the parent's level is complete: one launch per level)
  if (first <= last) P::search_step(ix, level - 1, digit + uint32_t(ix.kt2_nstop), first, last);
  const uint64_t rows = first <= last ? uint64_t(last - first + 1) : 0;
  uint64_t e = (uint64_t(first) & kDeepFirstMask) | ((rows < uint64_t(ix.kt2_deep_big) ? rows : kDeepBig) << 40);
  if (ix.kt2_sa1 && rows == 1) {
    const int64_t p = sa_at(ix, first);
    if (p >= 0 && p < (int64_t(1) << 31)) e = kDeepOneRow | (uint64_t(p) << 31) | uint64_t(first);      // (-1: a row a damaged index could not locate)
  }
  deep[pos - ix.kt2_deep_off] = e;
}

// sum of `v` over the 256-thread block (all threads must call); valid in thread 0
__device__ __forceinline__ int64_t block_sum_256(int64_t v, int64_t* s_w /* [4] */) {
  uint32_t lo = uint32_t(uint64_t(v)), hi = uint32_t(uint64_t(v) >> 32);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t lo2 = uint32_t(__shfl_down(int(lo), d, 64)), hi2 = uint32_t(__shfl_down(int(hi), d, 64));
    const uint64_t s = ((uint64_t(hi) << 32) | lo) + ((uint64_t(hi2) << 32) | lo2);
    lo = uint32_t(s);
    hi = uint32_t(s >> 32);
  }
  if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = int64_t((uint64_t(hi) << 32) | lo);
  __syncthreads();
  return s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t lo = uint32_t(__shfl_down(int(uint32_t(v)), d, 64)), hi = uint32_t(__shfl_down(int(uint32_t(v >> 32)), d, 64));
    v += (uint64_t(hi) << 32) | lo;
  }
  return v;   // valid in lane 0
}

// The row counts of a count launch, summed per 256-pattern block, per group of 64 blocks (super) and per 64 groups
// (super2).  Every block with rows adds its sum to its group's two words with device-scope atomics (64 / 4096 blocks per
// word: no contention to speak of), so no scan kernel sits between the count and the row expansion; plan_rows_kernel
// adds up at most 63 words of each level.  The words must be zero when the count kernel starts: the two (super, super2)
// sets alternate between launches and plan_rows_kernel clears the set the NEXT launch will use.
// (A "last block of the group adds the 64 sums up" scheme needs a __threadfence() per block, which on this part -- eight
// XCDs, one L2 each -- writes the XCD's L2 back: the count kernel ran 5x slower, measured.)
// acc == 0: the sums are not final when the count kernel ends (count_tail_kernel still adds to them) and
// plan_super_kernel computes the group sums afterwards.
struct PlanSums {
  int64_t* sums;        // [nblocks]
  int64_t* super;       // [ns = ceil(nblocks / 64)]
  int64_t* super2;      // [ceil(ns / 64)]
  int64_t* next_super;  // the other set: super and super2 contiguous, cleared by plan_rows_kernel
  int64_t next_words;
  int64_t nblocks;
  int32_t acc;          // 1: the count kernel accumulates super / super2
};
__host__ __device__ inline int64_t plan_set_words(int64_t nblocks) {
  const int64_t ns = (nblocks + 63) / 64;
  return ((ns + (ns + 63) / 64 + 7) & ~int64_t(7)) + 8;
}
__host__ __device__ inline size_t plan_sums_bytes(int64_t nblocks) {
  return size_t(((nblocks + 63) & ~int64_t(63)) + 2 * plan_set_words(nblocks)) * 8;
}
// `parity` selects the set this launch accumulates into
__host__ __device__ inline PlanSums plan_sums_at(void* base, int64_t nblocks, bool fold, int parity) {
  const int64_t ns = (nblocks + 63) / 64, w = plan_set_words(nblocks);
  int64_t* sums = static_cast<int64_t*>(base);
  int64_t* sets = sums + ((nblocks + 63) & ~int64_t(63));
  int64_t* mine = sets + (parity & 1) * w;
  return PlanSums{sums, mine, mine + ns, sets + ((parity & 1) ^ 1) * w, w, nblocks, fold ? 1 : 0};
}

// Does count_direct_kernel compare text tails inline (kDense)?  With the suffix array of every row, the text and the inverse suffix
// array of every position -- or, on a ROW-FREE launch, without the latter: such a launch never turns a position back into a row
// (a tail that would have to -- the text's first positions, a symbol the step itself must look at -- is not taken: the steps go on).
__host__ __device__ inline bool inline_tail_applies(const DevIndex& d, bool row_free) {
  return d.txt && d.sa_full && ((d.isa8 && d.isa_shift == 0) || row_free);
}

// do_string_query (src/main/server.c:713-946), one lane per pattern in the caller's order.
// kPlan: also do_locate_query's clamp (server.c:4405-4415): noccs[q] and the block's sum of them.
// kDense: the full suffix array and the full inverse suffix array are resident: once the range is ONE row with a long
// tail to go, the tail is compared with the text right here -- position = SA[row] (one read), compare, row of the
// last matching position = ISA[..] (one read) -- instead of being handed to count_tail_kernel (whose LF walks only
// exist because the sampled arrays need them).  Same result as stepping on: see text_kernels.hip.hpp.
template <class P, bool kPlan, bool kDense>
inline __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(P::kDirectWaves, P::kDirectWaves))) void count_direct_kernel(
    const DevIndex ix, const int64_t npats, const int32_t* __restrict__ plen, const uint16_t* __restrict__ pats,
    const int64_t* __restrict__ starts, int64_t* __restrict__ first_out, int64_t* __restrict__ last_out, int* __restrict__ err_flag,
    const int max_occs, int32_t* __restrict__ noccs, const PlanSums ps, int* __restrict__ big_flag, int64_t* __restrict__ sa_out) {
  constexpr int kWinDw = 18, kWin = 4 * kWinDw;   // the window: 72 symbols per lane
  __shared__ uint16_t s_code[264];
  __shared__ uint8_t s_byte[264];
  __shared__ int64_t s_w[4];
  __shared__ uint32_t s_win[kWinDw * 256];
  // (the lane's length and start are requested BEFORE the alphabet map is built: the map's loads and the barrier behind them
  // then overlap the first memory round trip of the pattern instead of preceding it -- a workgroup lives for three or four
  // round trips on a batch of random patterns, and this was one of them)
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int len_q = q < npats ? plen[q] : 0;
  const int64_t start_q = q < npats ? starts[q] : 0;
  for (int i = threadIdx.x; i < 264; i += blockDim.x) {
    const uint32_t c = i < kAlphaSize ? uint32_t(P::code_of(ix, uint32_t(i))) : 0xffffu;
    s_code[i] = uint16_t(c);
    s_byte[i] = uint8_t((c < 255u && !P::is_stop(ix, c)) ? c : 0xFFu);     // the window's byte for this symbol
  }
  __syncthreads();
  // (Lanes take the workgroup's patterns in the caller's order.  Taking them in order of LENGTH -- a counting sort in LDS, so
  // that a wavefront's lanes run chains of similar length -- was tried on the mixed-length sigma~96 batch: 1.70 instead of
  // 1.64 ms; neighbouring lanes then no longer read neighbouring patterns.)
  int64_t nocc = 0;
  if (q < npats) {
    const int len = len_q;
    const uint16_t* pat = pats + start_q;
    // The lane's window on its pattern: the DENSE codes of symbols j = w0 .. w0 + 71 (j counts from the pattern's END, the
    // order a backward search reads them in), one byte each, in LDS (byte k of lane t in dword (k >> 2) * 256 + t: bank =
    // t mod 32, the minimum for 64 lanes).  Each symbol is translated ONCE (alpha code -> window byte through s_byte);
    // 0xFF = "look at the alpha symbol": a code >= ALPHA_SIZE, a character the text lacks, a character <= SEOF, or dense
    // code 255 -- the fast paths stop there and the ordinary step below reads the symbol itself.  Every later use --
    // table digits, context keys, search steps, the comparison with the text, eight symbols per iteration -- reads bytes
    // of the window.  (Round 2 kept the raw 64-byte chunk and translated every symbol at every use: 5 800 VALU
    // wave-instructions per wavefront on the sigma~96 workload.)
    //
    // w0 is chosen so that every PAIR of window dwords is one ALIGNED 16-byte piece of the caller's symbols: symbol j lives
    // at pat + (len - 1 - j), so with w0 = (pat / 2 + len) mod 8 (mod 8) the eight symbols of dwords 2e, 2e+1 are the piece
    // at pat + (len - 8 - w0) - 8 e, highest address = lowest j = byte 0.  One 16-byte load, eight table reads, two LDS
    // writes per eight symbols, no per-symbol range test, and as few load INSTRUCTIONS as the pattern allows (a 20-mer: three
    // or four) -- a load whose 64 lanes hit different lines keeps the CU's address unit busy for ~64 cycles whatever its
    // width, and that unit is what bounds this kernel (ind_kernels.hip.hpp has the numbers).  The first version tested and
    // stored byte by byte: 18 VALU instructions per symbol.  A piece at either end of the pattern may hold symbols of the
    // neighbours (at most 7 = 14 bytes before / behind the pattern, never across a 16-byte boundary, so never across a
    // page); their window bytes are never read.
    auto phase = [&]() -> int { return int(((reinterpret_cast<uintptr_t>(pat) >> 1) + uintptr_t(uint32_t(len))) & 7u); };   // (recomputed: one register less across the search loops)
    int w0 = -(1 << 30);         // no window yet
    uint8_t* const win = reinterpret_cast<uint8_t*>(s_win);
    auto waddr = [&](uint32_t k) -> uint32_t { return (((k >> 2) * 256u + threadIdx.x) << 2) + (k & 3u); };
    auto tr4 = [&](const uint32_t a, const uint32_t b) -> uint32_t {     // symbols (a lo, a hi, b lo, b hi) at rising addresses = falling j
      auto T = [&](uint32_t sym) -> uint32_t { return s_byte[sym < 263u ? sym : 263u]; };
      return T(b >> 16) | (T(b & 0xffffu) << 8) | (T(a >> 16) << 16) | (T(a & 0xffffu) << 24);
    };
    // window at w0 = nw0 (== phase mod 8): kB pieces are loaded together, then translated
    auto fill_at = [&](const int nw0, auto batch) {
      constexpr int kB = decltype(batch)::value;
      w0 = nw0;
      int emax = (len - 1 - nw0) >> 3;
      emax = emax < kWinDw / 2 - 1 ? emax : kWinDw / 2 - 1;
      const uint16_t* const gp0 = pat + (len - 8 - nw0);
#pragma unroll 1
      for (int e0 = 0; e0 <= emax; e0 += kB) {
        uint4 v[kB];
#pragma unroll
        for (int i = 0; i < kB; i++)
          if (e0 + i <= emax) v[i] = *reinterpret_cast<const uint4*>(gp0 - 8 * (e0 + i));
#pragma unroll
        for (int i = 0; i < kB; i++)
          if (e0 + i <= emax) {
            s_win[uint32_t(2 * (e0 + i)) * 256u + threadIdx.x] = tr4(v[i].z, v[i].w);
            s_win[uint32_t(2 * (e0 + i) + 1) * 256u + threadIdx.x] = tr4(v[i].x, v[i].y);
          }
      }
    };
    // The FIRST window holds all of a pattern of up to 64 symbols; its loads are issued four pieces = 64 bytes at a time,
    // so a line of the caller's symbols is fetched while it is hot.  (32 wavefronts per CU each hold 64 x 200 bytes of
    // 100-mers in flight: far more than the L1, more than the CU's share of the L2.  Visiting every line four times,
    // iterations apart, re-fetched it: 3.4x the compulsory traffic on 100-mers, measured.)  Refills (patterns of more than
    // 64 symbols) sit inside the search loops, where four pieces in registers spill: two at a time there.
    if (len > 0) fill_at(-((8 - phase()) & 7), std::integral_constant<int, 4>{});
    auto refill = [&](int j) { fill_at(j - ((j - phase()) & 7), std::integral_constant<int, 2>{}); };
    auto wcode = [&](int j) -> uint32_t {    // dense code of the j-th symbol from the end, or 0xFF
      if (j < w0 || j >= w0 + kWin) refill(j);
      return win[waddr(uint32_t(j - w0))];
    };
    auto alpha = [&](int j) -> uint32_t { return pat[len - 1 - j]; };     // the symbol itself (0xFF cases only)
    int64_t first = 0, last = ix.total_length - 1;
    int64_t spot = -1;          // P::kSpotMarks: the last marked row a one-row step stood on | symbols it had to go << 40
    int64_t sa_hint = -1;       // SA[first] where it is known: as the wide context table delivered it, or behind a text tail that left one row;
                                // forgotten with the next search step
    int j = 0;
    if (ix.ctx && len >= ix.ctx_syms) {     // byte alphabets: the last H symbols as one hashed read (ctx_kernels.hip.hpp)
      const int H1 = ix.ctx_syms, H2 = (ix.ctx2 && len >= ix.ctx2_syms) ? ix.ctx2_syms : 0;
      const int HM = (ix.ctxm && len >= ix.ctxm_syms) ? ix.ctxm_syms : 0;     // the table in between: patterns of H1 < len < H2 symbols
      const int HH = H2 ? H2 : (HM ? HM : H1);
      const uint32_t nstop = uint32_t(ix.ctx_nstop);
      const int bits = ix.ctx_bits;
      uint64_t key1 = 0;
      CtxKey2 key2{0, 0};
      int okn = 0;                           // leading symbols (from the end) that are table characters
      for (; okn < HH; okn++) {
        const uint32_t code = wcode(okn);
        if (code == 0xFFu) break;
        const uint64_t field = uint64_t(code - nstop + 1u);
        if (okn < H1) key1 |= field << (bits * okn);
        if (H2 | HM) ctx_key2_or(key2, field, bits * okn);
      }
      // A miss means the range dies within these symbols; the shorter table / the level table / the steps below then find
      // where, because the reference's (first, last) of an empty range are those of the step that emptied it.
      if (H2 && okn == H2 && ctx2_lookup(ix, key2, first, last, &sa_hint) == 1) j = H2;
      else if (HM && okn >= HM && ctxm_lookup(ix, key2, first, last, &sa_hint) == 1) j = HM;
      else if (okn >= H1 && ctx_lookup(ix, key1, first, last) == 1) j = H1;
      if (j == 0 || j == H1) sa_hint = -1;        // (a wide-table miss may have left a value behind)
    }
    if (j == 0 && ix.ktab2) {
      const int kmax = len < ix.kt2_syms ? len : ix.kt2_syms;
      const uint32_t nstop = uint32_t(ix.kt2_nstop);
      const int64_t t = ix.kt2_base;
      int64_t pos = 0;
      for (; j < kmax; j++) {
        const uint32_t code = wcode(j);
        if (code == 0xFFu) break;            // not a table character: the ordinary step below deals with it
        pos = pos * t + 1 + int64_t(code - nstop);
      }
      if (kDense) ktab2_lookup<P>(ix, pos, j, first, last, &sa_hint);
      else ktab2_lookup<P>(ix, pos, j, first, last);
      if (first > last) j = len;
    }
    bool handed = false;
    // kDense: the stepping loop below is LEFT when the range is one row with a long tail to go; the wavefront's lanes
    // reconverge behind it, so the lanes that compare their tails with the text do so together, in lockstep, while the
    // others have finished (taking the shortcut inside the stepping loop would serialise it lane by lane: 19 ms instead
    // of 5.7 + 2.8 ms handed over, measured on the sigma~96 workload).
    bool tried = false, finished = false;
    int ones = 0;      // consecutive steps that left exactly one row
    // ROW-FREE locate (ix.row_free: the caller wants parallel_locate's results -- noccs and offsets, src/main/femto.c:331-400 -- and
    // no rows).  A one-row range whose text tail consumes the pattern is then LOCATED by the compare itself (offset = where the
    // compared text starts): the inverse-suffix-array read that would turn that position back into a row is skipped; and a tail
    // that meets a text character other than the pattern's (an ordinary character, not one the step below has to look at) is an
    // empty range whose (first, last) nobody asked for: no read, no further step.  rf: 1 located, 2 empty.
    const bool row_free = kPlan && kDense && ix.row_free != 0;
    int rf = 0;
    // ... unless the wavefront says the batch is made of patterns that occur: when at least three quarters of its patterns
    // still have rows after the table, waiting for two more steps only costs lines (10 M sampled DNA 20-mers: 1.61 ms per step
    // waiting, 1.45 ms jumping at once; random 20-mers leave the table with 22 % alive and keep waiting).
    int need_ones = ix.tail_ones;
    if (kDense && need_ones > 0) {
      const unsigned long long act = __ballot(1), alive = __ballot(first <= last && j < len);
      if (__popcll(alive) * 4 >= __popcll(act) * 3) need_ones = 0;
    }
    const bool fast_cmp = P::max_code(ix) < 255u;     // a text byte is never 0xFF: four symbols compare as one word
    for (;;) {
      for (; j < len; j++) {
        if (kDense) {
          // tail-ready.  ix.tail_ones: a small alphabet's RANDOM pattern often has one row left after the table and dies
          // on the next step or two (one line each), which is cheaper than the three dependent lines of the tail; a row
          // that survived tail_ones steps belongs to a pattern that occurs (measured on 10 M random DNA 20-mers: 0.86 ms
          // jumping at once, 0.65 ms stepping on).
          if (!tried && last - first < int64_t(P::kTailRows) && last - first < int64_t(ix.tail_rows) && j > 0 && ones >= need_ones &&
              len - j >= ix.tail_min + ix.tail_row_cost * int(last - first)) break;
          // (row-free, one row whose position a table delivered: the compare is ONE request and ends the pattern either way --
          // never more than the step it replaces, and the row expansion gets the position for nothing)
          if (row_free && !tried && sa_hint >= 0 && first == last) break;
        } else if (ix.txt && first == last && j > 0 && len - j >= ix.tail_min) {
          tail_append(ix, q, j, first);   // one row left, a long tail to go: count_tail_kernel compares it with the text
          handed = true;
          finished = true;
          break;
        }
        uint32_t code = wcode(j);
        if (code == 0xFFu) {            // the alpha symbol decides (server.c:832-936 on the symbol itself)
          const uint32_t ch = alpha(j);
          if (ch >= uint32_t(kAlphaSize)) {
            atomicOr(err_flag, 1);
            first = 0;
            last = -1;
            finished = true;
            break;
          }
          code = s_code[ch];
          if (code == 0xffffu) {  // the character does not occur in the text: Occ == 0 (index.c:2080-2089)
            first = ix.C[ch];
            last = first - 1;
            finished = true;
            break;
          }
        }
        if constexpr (P::kSpotMarks && kPlan && !kDense) {
          // MARK SPOTTING (handles without the suffix array: every located row costs a walk to the next mark).  A pattern that
          // occurs spends its last steps on ONE row, and the marked rank unit a step loads anyway says whether that row is
          // marked.  A marked row met with len - j symbols to go fixes the pattern's text position: SA[final row] =
          // SA[this row] - (len - j); plan_rows_kernel then reads that row's mark instead of walking from the final row.
          // The hint travels as row | symbols-to-go << 40 (one register pair); a step with a stop character forgets it (a
          // walk does not cross a document start, server.c:2336-2342).
          bool spotted;
          const int64_t row = last;
          P::search_step_spot(ix, j, code, first, last, &spotted);
          if (P::is_stop(ix, code)) spot = -1;
          else if (spotted && len - j < (1 << 22)) spot = row | (int64_t(len - j) << 40);
        } else {
          P::search_step(ix, j, code, first, last);
        }
        sa_hint = -1;
        if (first > last) { finished = true; break; }
        tried = false;
        ones = first == last ? ones + 1 : 0;
      }
      if (!kDense || finished || j >= len) break;
      // ---- tail-ready: position of the row, compare with the text, row of the last matching position
      tried = true;
      const int remaining = len - j;
      int best = 0;                      // most symbols any row of the range matches; [qmin, qmax] = the rows of those that do
      int64_t qmin = 0, qmax = -1;
      int64_t pbest = -1;                // text position behind qmin (the match of `best` symbols starts there: SA[qmin])
      for (int64_t row = first; row <= last; row++) {
        int64_t p = sa_hint;
        if (!(kDense && row == first && sa_hint >= 0)) {
          p = sa_at(ix, row);
          trace_touch(ix, kTraceSa, sa_line_of(ix, row));
        }
        if (p <= 0 || p >= ix.total_length) continue;   // (p = -1: the row could not be located, see text_isa_build_kernel)
        const int lim = p < int64_t(remaining) ? int(p) : remaining;
        const uint8_t* const tp = ix.txt + (p - 1);       // txt[p - 1 - m] = tp[-m]
        int m = 0;                       // symbols matched
        bool stopped = false;
        if (fast_cmp) {
          while (m + 16 <= lim) {         // sixteen symbols per round, both loads in flight together: a round is one dependent
                                           // read of the lane's chain, and the wavefront waits for its longest chain
            const int jj = j + m;
            if (jj < w0 || jj + 16 > w0 + kWin) refill(jj);
            const uint32_t k = uint32_t(jj - w0), d = k >> 2, sh = k & 3u;
            const uint32_t wa = s_win[d * 256u + threadIdx.x], wb = s_win[(d + 1u) * 256u + threadIdx.x];
            const uint32_t wc = s_win[(d + 2u) * 256u + threadIdx.x], wd = s_win[(d + 3u) * 256u + threadIdx.x];
            const uint32_t we = sh ? s_win[(d + 4u) * 256u + threadIdx.x] : 0u;
            const uint64_t pa = uint64_t(__builtin_amdgcn_alignbyte(wb, wa, sh)) | (uint64_t(__builtin_amdgcn_alignbyte(wc, wb, sh)) << 32);
            const uint64_t pb = uint64_t(__builtin_amdgcn_alignbyte(wd, wc, sh)) | (uint64_t(__builtin_amdgcn_alignbyte(we, wd, sh)) << 32);
            uint64_t ta, tb;
            __builtin_memcpy(&ta, tp - m - 7, 8);                                  // txt[p-8-m .. p-1-m]
            __builtin_memcpy(&tb, tp - m - 15, 8);                                 // txt[p-16-m .. p-9-m]
            trace_touch_span(ix, kTraceTxt, uint64_t(tp - m - 15 - ix.txt), uint64_t(tp - m - ix.txt));
            const uint64_t xa = pa ^ __builtin_bswap64(ta), xb = pb ^ __builtin_bswap64(tb);
            if (xa | xb) {
              m += xa ? (__ffsll(static_cast<long long>(xa)) - 1) >> 3 : 8 + ((__ffsll(static_cast<long long>(xb)) - 1) >> 3);
              stopped = true;
              break;
            }
            m += 16;
          }
          while (!stopped && m + 8 <= lim) {          // eight symbols per load
            const int jj = j + m;
            if (jj < w0 || jj + 8 > w0 + kWin) refill(jj);
            const uint32_t k = uint32_t(jj - w0), d = k >> 2, sh = k & 3u;
            const uint32_t wa = s_win[d * 256u + threadIdx.x], wb = s_win[(d + 1u) * 256u + threadIdx.x];
            const uint32_t wc = sh ? s_win[(d + 2u) * 256u + threadIdx.x] : 0u;
            const uint64_t pw = uint64_t(__builtin_amdgcn_alignbyte(wb, wa, sh)) | (uint64_t(__builtin_amdgcn_alignbyte(wc, wb, sh)) << 32);
            uint64_t tw;
            __builtin_memcpy(&tw, tp - m - 7, 8);                                  // txt[p-8-m .. p-1-m]
            trace_touch_span(ix, kTraceTxt, uint64_t(tp - m - 7 - ix.txt), uint64_t(tp - m - ix.txt));
            const uint64_t x = pw ^ __builtin_bswap64(tw);                          // byte 0: symbol j+m against txt[p-1-m]
            if (x) {
              m += (__ffsll(static_cast<long long>(x)) - 1) >> 3;
              stopped = true;
              break;
            }
            m += 8;
          }
          while (!stopped && m + 4 <= lim) {
            const int jj = j + m;
            if (jj < w0 || jj + 4 > w0 + kWin) refill(jj);
            const uint32_t k = uint32_t(jj - w0), d = k >> 2, sh = k & 3u;
            const uint32_t lo = s_win[d * 256u + threadIdx.x];
            const uint32_t hi = sh ? s_win[(d + 1u) * 256u + threadIdx.x] : 0u;
            const uint32_t pw = __builtin_amdgcn_alignbyte(hi, lo, sh);           // window bytes k .. k+3, k in the low byte
            uint32_t tw;
            __builtin_memcpy(&tw, tp - m - 3, 4);                                  // txt[p-4-m .. p-1-m]
            trace_touch_span(ix, kTraceTxt, uint64_t(tp - m - 3 - ix.txt), uint64_t(tp - m - ix.txt));
            const uint32_t x = pw ^ __builtin_bswap32(tw);                          // byte 0: symbol j+m against txt[p-1-m]
            if (x) {
              m += (__ffs(int(x)) - 1) >> 3;
              stopped = true;
              break;
            }
            m += 4;
          }
        }
        if (!stopped)
          for (; m < lim; m++) {
            const uint32_t code = wcode(j + m);
            if (code == 0xFFu) break;                       // anything unusual is left to the ordinary step
            trace_touch(ix, kTraceTxt, uint64_t(tp - m - ix.txt) >> 7);
            if (uint32_t(tp[-m]) != code) break;
          }
        if (row_free && first == last) {
          if (m == remaining) {              // (lim == remaining: the text holds all of them)
            rf = 1;
            pbest = p - m;
            break;
          }
          if (m < lim) {                     // stopped at symbol j + m: a mismatch for good unless the step has to look at the symbol itself
            const uint32_t code = wcode(j + m);
            if (code != 0xFFu && uint32_t(tp[-m]) != code) {
              rf = 2;
              break;
            }
          }
        }
        if (m > 0 && m >= best && ix.isa_shift == 0) {      // (a row-free launch on the sampled inverse suffix array: no way back to a row, the steps go on)
          const int64_t q2 = isa_at(ix, p - m);   // isa_shift == 0: the row of every text position
          trace_touch(ix, kTraceIsa, sa_line_of(ix, p - m));
          if (m > best) {
            best = m;
            qmin = qmax = q2;
            pbest = p - m;
          } else {
            qmin = q2 < qmin ? q2 : qmin;
            qmax = q2 > qmax ? q2 : qmax;
          }
        }
      }
      sa_hint = -1;        // (it described the range this tail started from)
      if (rf == 1) {       // located: first == last stays the row the tail started from (not the pattern's row: nobody reads it)
        sa_hint = pbest;
        break;
      }
      if (rf == 2) {
        first = 0;
        last = -1;
        break;
      }
      if (best > 0) {      // the rows whose text goes on with `best` more pattern symbols: one contiguous range again
        first = qmin;
        last = qmax;
        j += best;
        if (qmin == qmax) sa_hint = pbest;     // (rows are distinct positions: one row left means pbest is ITS position)
      }
      if (j >= len) break;
    }
    if (!handed) {
      if (kPlan && ix.row_free) {
        // (rows are not returned; first_out is the launch layer's own array, and only what plan_rows_kernel will read of it is written)
      } else if (last_out) {
        first_out[q] = first;
        last_out[q] = last;
      } else {
        first_out[q] = last - first + 1;   // femto.c:313-318
      }
      if (kPlan) {
        if (first > last) nocc = 0;
        else if (last - first > int64_t(max_occs)) nocc = max_occs;
        else nocc = last - first + 1;
        noccs[q] = int32_t(nocc);
        if (ix.row_free && nocc > 0 && rf != 1) first_out[q] = first;
        // kDense: a search that ended in the text tail (or in the wide context table) on ONE row knows that row's text
        // position already -- it is where the compared text starts.  Handing it to plan_rows_kernel saves that pattern's
        // suffix-array read there: one scattered request less per located pattern.  Written by the wavefronts that hold
        // a one-row pattern only (plan_rows_kernel reads sa_out[q] only where noccs[q] == 1); -1 = not known.
        if (kDense && sa_out) {
          if (__ballot(nocc == 1)) sa_out[q] = first == last ? sa_hint : -1;
        }
        // ... or, without the suffix array, a marked row the search stood on: <= -2 encodes the hint (see "MARK SPOTTING")
        // (a row-free launch on the sampled arrays hands count_tail_kernel's positions over the same way: -1 where there is nothing to say)
        if (!kDense && sa_out) {
          if (__ballot(nocc == 1)) sa_out[q] = (P::kSpotMarks && first == last && spot >= 0) ? -2 - spot : -1;
        }
      }
    } else if (kPlan) {
      noccs[q] = 0;    // count_tail_kernel stores the real value and adds it to the block's sum
      if (!kDense && sa_out) sa_out[q] = -1;      // (mark spotting, or a row-free launch: count_tail_kernel may know better)
    }
  }
  if (kPlan) {
    const int64_t s = block_sum_256(nocc, s_w);
    if (threadIdx.x == 0) {
      ps.sums[blockIdx.x] = s;
      if (ps.acc && s) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&ps.super[blockIdx.x >> 6]), static_cast<unsigned long long>(s));
        atomicAdd(reinterpret_cast<unsigned long long*>(&ps.super2[blockIdx.x >> 12]), static_cast<unsigned long long>(s));
      }
      if (blockIdx.x == 0 && big_flag) *big_flag = 0;      // plan_rows_kernel's "long ranges" flag
    }
  }
}

// Host-pointer batches whose patterns fit a 64-bit key travel as KEYS: the staging threads pack every pattern's dense
// codes (1 + rank of the character among the text's characters, `bits` bits each, last symbol in the top field, 0 = end
// of pattern) into 8 bytes instead of 2 bytes per symbol + 12 bytes of length / start, and the ranges come back as 32-bit
// pairs when the index has fewer than 2^31 - 1 rows: 16 instead of 68 bytes per 20-mer over PCIe.  Same search, same
// results; a chunk holding any pattern a key cannot describe (longer, or a character outside the text) travels as symbols.
// kPlan: also the clamped row counts and their block / group sums, as count_direct_kernel<.., kPlan = true> leaves them:
// the device-pointer form (femto_amd_locate_keys_device) then runs plan_rows_kernel on 28 bytes per pattern all told.
template <class P, bool kPlan>
inline __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(P::kWaves, P::kWaves))) void count_keys_kernel(
    const DevIndex ix, const int64_t npats, const uint64_t* __restrict__ keys, const int bits, const int nsym,
    int2* __restrict__ out32, int64_t* __restrict__ first_out, int64_t* __restrict__ last_out, const int max_occs,
    int32_t* __restrict__ noccs, const PlanSums ps, int* __restrict__ big_flag, int64_t* __restrict__ sa_out /* or NULL: "MARK SPOTTING" as in count_direct_kernel */) {
  __shared__ int64_t s_w[4];
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  int64_t nocc = 0;
  if (q < npats) {
    const uint64_t key = keys[q];
    int64_t spot = -1;
    const uint32_t fmask = (1u << bits) - 1u;
    auto field = [&](int j) -> uint32_t { return uint32_t(key >> (64 - bits * (j + 1))) & fmask; };
    int64_t first = 0, last = ix.total_length - 1;
    int j = 0;
    bool ended = false;
    if (ix.ktab2) {
      const int kmax = nsym < ix.kt2_syms ? nsym : ix.kt2_syms;
      const uint32_t nstop = uint32_t(ix.kt2_nstop);
      const int64_t t = ix.kt2_base;
      int64_t pos = 0;
      for (; j < kmax; j++) {
        const uint32_t f = field(j);
        if (f == 0) { ended = true; break; }
        if (f - 1 < nstop) break;          // a character <= SEOF: not a table character, stepped below
        pos = pos * t + 1 + int64_t(f - 1 - nstop);
      }
      ktab2_lookup<P>(ix, pos, j, first, last);
      if (first > last) ended = true;
    }
    int len = j;      // (symbols of the key: counted on as the steps run)
    if (!ended)
      for (; j < nsym; j++) {
        const uint32_t f = field(j);
        if (f == 0) break;
        if constexpr (P::kSpotMarks && kPlan) {
          bool spotted;
          const int64_t row = last;
          P::search_step_spot(ix, j, f - 1, first, last, &spotted);
          if (P::is_stop(ix, f - 1)) spot = -1;
          else if (spotted) spot = row | (int64_t(j) << 40);      // (symbols DONE before this step; turned into symbols to go below)
        } else {
          P::search_step(ix, j, f - 1, first, last);
        }
        if (first > last) break;
      }
    len = j;
    if (out32) {
      out32[q] = make_int2(int(first), int(last));
    } else {
      first_out[q] = first;
      last_out[q] = last;
    }
    if (kPlan) {
      if (first > last) nocc = 0;
      else if (last - first > int64_t(max_occs)) nocc = max_occs;      // server.c:4411 (">": see Appendix C of SURVEY.md)
      else nocc = last - first + 1;
      noccs[q] = int32_t(nocc);
      if (P::kSpotMarks && sa_out) {      // a marked row the search stood on with (len - done) symbols to go: plan_rows_kernel starts there
        int64_t hint = -1;
        if (first == last && spot >= 0) hint = -2 - ((spot & ((int64_t(1) << 40) - 1)) | (int64_t(len - int(spot >> 40)) << 40));
        if (__ballot(nocc == 1)) sa_out[q] = hint;
      }
    }
  }
  if (kPlan) {
    const int64_t s = block_sum_256(nocc, s_w);
    if (threadIdx.x == 0) {
      ps.sums[blockIdx.x] = s;
      if (s) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&ps.super[blockIdx.x >> 6]), static_cast<unsigned long long>(s));
        atomicAdd(reinterpret_cast<unsigned long long*>(&ps.super2[blockIdx.x >> 12]), static_cast<unsigned long long>(s));
      }
      if (blockIdx.x == 0 && big_flag) *big_flag = 0;
    }
  }
}

// u16 symbols -> keys on the device (femto_amd_pack_keys_device): key 0 and *bad += 1 for a pattern a key cannot describe
// (longer than nsym symbols, or a character that does not occur in the text -- its empty range has values of its own)
inline __global__ __launch_bounds__(256) void pack_keys_kernel(const int64_t npats, const int32_t* __restrict__ plen, const uint16_t* __restrict__ pats,
                                                        const int64_t* __restrict__ starts, const uint8_t* __restrict__ dense, const int bits,
                                                        const int nsym, uint64_t* __restrict__ keys, unsigned long long* __restrict__ bad) {
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= npats) return;
  const int l = plen[q];
  uint64_t key = 0;
  bool ok = l >= 0 && l <= nsym;
  if (ok && l) {
    const uint16_t* pat = pats + starts[q];
    for (int s = l - 1; s >= 0; s--) {       // last symbol first: it lands in the top field
      const uint32_t ch = pat[s];
      const uint32_t c = ch < uint32_t(kAlphaSize) ? dense[ch] : 0u;
      if (c == 0) { ok = false; break; }
      key = (key << bits) | c;
    }
    key = ok ? key << (64 - l * bits) : 0;
  }
  keys[q] = ok ? key : 0;
  if (!ok) atomicAdd(bad, 1ull);
}

// super[g] = sum of the block sums of group g, for count launches whose sums are only final after count_tail_kernel
// (sampled suffix arrays): one wavefront per group.  Launches with the dense arrays never run this -- the count kernel's
// blocks have accumulated the group sums already.
inline __global__ __launch_bounds__(256) void plan_super_kernel(const PlanSums ps) {
  const int64_t g = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int64_t ns = (ps.nblocks + 63) >> 6;
  if (g >= ns) return;
  const int64_t b = (g << 6) + int64_t(threadIdx.x & 63u);
  const uint64_t t = wave_sum_u64(b < ps.nblocks ? uint64_t(ps.sums[b]) : 0);
  if ((threadIdx.x & 63u) == 0) {
    ps.super[g] = int64_t(t);
    if (t) atomicAdd(reinterpret_cast<unsigned long long*>(&ps.super2[g >> 6]), static_cast<unsigned long long>(t));   // cleared by the host
  }
}

// The match counts of a batch in the form they travel to another GPU (the result gather of a multi-GPU run): ONE byte per
// pattern -- last - first + 1, 0 when there is no match, 255 = "255 or more: see the list" -- and the patterns with 255
// matches or more appended to a list of (pattern, count) pairs (order unspecified; *big_n counts them even beyond
// big_cap, so the receiver sees an overflow).  Lossless, 1 byte instead of 16 per pattern on the links.
inline __global__ __launch_bounds__(256) void pack_counts_kernel(const int64_t n, const int64_t* __restrict__ first, const int64_t* __restrict__ last,
                                                          uint8_t* __restrict__ counts8, int64_t* __restrict__ big, const int64_t big_cap,
                                                          unsigned long long* __restrict__ big_n) {
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int64_t c = last[q] - first[q] + 1;
  if (c < 0) c = 0;
  counts8[q] = uint8_t(c < 255 ? c : 255);
  if (c >= 255) {
    const unsigned long long k = atomicAdd(big_n, 1ull);
    if (int64_t(k) < big_cap) {
      big[2 * k] = q;
      big[2 * k + 1] = c;
    }
  }
}

// set bits of trace words [w0, w1) added to *out (femto_amd_trace_lines)
inline __global__ __launch_bounds__(256) void trace_popcount_kernel(const uint32_t* __restrict__ bitmap, const int64_t w0, const int64_t w1,
                                                             unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (int64_t i = w0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < w1; i += int64_t(gridDim.x) * blockDim.x) c += uint32_t(__popc(bitmap[i]));
  for (int d = 32; d >= 1; d >>= 1) c += (unsigned long long)__shfl_down((long long)c, d, 64);
  if ((threadIdx.x & 63u) == 0 && c) atomicAdd(out, c);
}

// paths that scanned the row counts themselves: publish the total the same way plan_scan_kernel does
inline __global__ void copy_total_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ total_out, const int64_t capacity) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    total_out[0] = *src;
    total_out[1] = *src > capacity ? 1 : 0;
  }
}

// the text offset of `row` by the locate walk (do_back_query / do_context_query, src/main/server.c:2228-2359, :2627-2795):
// LF steps until a marked row; -1 when a document start or the walk limit comes first
template <class P>
__device__ __forceinline__ int64_t walk_row(const DevIndex& ix, int64_t row) {
  int64_t steps = 0;
  while (row >= 0 && steps <= int64_t(ix.walk_limit)) {
    uint32_t code;
    bool marked;
    int64_t sa_index, next;
    P::lf(ix, row, code, marked, sa_index, next);
    if (marked) return mark_offset_at(ix, sa_index) + steps;
    if (P::is_stop(ix, code)) break;              // cannot walk past a document start (server.c:2336-2342)
    row = next;                                    // LF (server.c:2279-2282)
    steps++;
  }
  return -1;
}

// out_starts[q] = rows located for the patterns before q; pattern q's rows first[q] .. first[q]+noccs-1 are written at
// offsets[out_starts[q] ..] (the walk replaces each row by its text offset).  Ranges longer than kExpandSerialMax rows
// (the empty pattern with a huge max_occs) are left to expand_big_rows_kernel.
// kMode 1: the full suffix array is resident -- the offsets themselves are written (offsets[..] = SA[first + k], consecutive
// reads) and no walk follows.  kMode 2: sampled marks -- the offsets themselves again, each by its walk (walk_row<P>), right
// here: the rows are never written and read back and no walk kernel is launched behind this one (a step of the
// footprint-bounded handle: 1.116 -> ... ms, profiles/r04_*).  kMode 0: the rows (two-call API; femto_amd_locate_walk_device walks them).
constexpr int kRowsOnly = 0, kRowsSa = 1, kRowsWalk = 2;
constexpr int64_t kKnownPos = INT64_MIN;      // kRowsWalk: kKnownPos + position (< -3 x 2^61) in place of a row; spotted rows are -2 - (row | to go << 40) >= -2^62 - 1
template <int kMode, class P>
inline __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(P::kPlanWaves, 8))) void plan_rows_kernel(const int64_t npats, const int32_t* __restrict__ noccs, const int64_t* __restrict__ first,
                                                        const int2* __restrict__ first32 /* or NULL: (first,last) pairs instead of first[] */,
                                                        const PlanSums ps, int64_t* __restrict__ out_starts,
                                                        int64_t* __restrict__ offsets, const int64_t capacity, int* __restrict__ big_flag,
                                                        const DevIndex ix, int64_t* __restrict__ total_out, int64_t* __restrict__ total_user,
                                                        const int64_t* __restrict__ sa_known /* or NULL: count_direct_kernel's sa_out */) {
  __shared__ int64_t s_w[4];
  __shared__ int64_t s_boff;
  const int64_t q = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t n = q < npats ? int64_t(noccs[q]) : 0;
  // inclusive scan inside the wavefront, then across the four wavefronts (a block's rows: <= 256 * (2^31 - 1): 64 bits)
  uint64_t incl = uint64_t(n);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0) {   // rows before this block: the groups before its group + the blocks before it inside the group
    const int64_t b = int64_t(blockIdx.x), g = b >> 6;
    // three independent loads per lane: blocks before b in its group, groups before g in its group of groups, and those
    uint64_t acc = lane < int(b & 63) ? uint64_t(ps.sums[(b & ~int64_t(63)) + lane]) : 0;
    if (lane < int(g & 63)) acc += uint64_t(ps.super[(g & ~int64_t(63)) + lane]);
    for (int64_t k = lane; k < (g >> 6); k += 64) acc += uint64_t(ps.super2[k]);    // (one round up to 2^18 blocks = 67 M patterns)
    const uint64_t part = wave_sum_u64(acc);
    // the set of group sums the NEXT launch accumulates into starts at zero
    for (int64_t k = int64_t(blockIdx.x) * 64 + lane; k < ps.next_words; k += int64_t(gridDim.x) * 64) ps.next_super[k] = 0;
    if (lane == 0) {
      s_boff = int64_t(part);
      if (b == ps.nblocks - 1) {      // the last block knows the total: everything before it + its own rows
        const int64_t total = int64_t(part) + ps.sums[b];
        total_out[0] = total;
        total_out[1] = total > capacity ? 1 : 0;
        if (total_user) {
          total_user[0] = total;
          total_user[1] = total > capacity ? 1 : 0;
        }
        out_starts[npats] = total;
      }
    }
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t lo = uint32_t(__shfl_up(int(uint32_t(incl)), d, 64)), hi = uint32_t(__shfl_up(int(uint32_t(incl >> 32)), d, 64));
    if (lane >= d) incl += (uint64_t(hi) << 32) | lo;
  }
  if (lane == 63) s_w[wave] = int64_t(incl);
  __syncthreads();
  int64_t woff = 0;
  for (int k = 0; k < wave; k++) woff += s_w[k];
  const int64_t base = q < npats ? s_boff + woff + int64_t(incl) - n : 0;
  if (q < npats) out_starts[q] = base;
  if (!offsets) return;
  if (kMode != kRowsOnly) {
    // The wavefront writes its patterns' offsets TOGETHER: output slot s of the wavefront's span belongs to the lane whose
    // inclusive in-wave count first exceeds it, so consecutive lanes write consecutive slots (coalesced) and read
    // consecutive suffix-array entries of a pattern's range.  Long ranges go to plan_big_rows_kernel as before.
    __shared__ uint32_t s_incl[4][64];
    __shared__ int64_t s_first[4][64], s_lbase[4][64];
    const bool big = n > kExpandSerialMax;
    if (big) atomicOr(big_flag, 1);
    const uint32_t mine = (q < npats && !big) ? uint32_t(n) : 0u;     // rows this lane contributes to the cooperative part
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = uint32_t(__shfl_up(int(inc), d, 64));
      if (lane >= d) inc += y;
    }
    s_incl[wave][lane] = inc;
    int64_t f0 = 0;
    bool have = false;
    if (kMode == kRowsSa && sa_known && mine == 1u) {      // the count kernel may know this row's position: ~position < 0 in place of the row
      const int64_t known = sa_known[q];
      if (known >= 0) {
        f0 = ~known;
        have = true;
      }
    }
    if (kMode == kRowsWalk && sa_known && mine == 1u) {    // ... or a MARKED row its search stood on (<= -2: marked row | symbols to go << 40),
      const int64_t known = sa_known[q];                  // or the position itself (a row-free launch's text tail): kKnownPos + position
      if (known <= -2) {
        f0 = known;
        have = true;
      } else if (known >= 0) {
        f0 = kKnownPos + known;
        have = true;
      }
    }
    if (mine && !have) f0 = first32 ? int64_t(first32[q].x) : first[q];      // (only ranges with rows: most lines of first[] are never touched on a random batch)
    s_first[wave][lane] = f0;
    s_lbase[wave][lane] = base;      // slots are addressed per lane: a long range keeps its slots but is not written here
    const uint32_t total = uint32_t(__shfl(int(inc), 63, 64));
    __syncthreads();
    for (uint32_t s0 = 0; s0 < total; s0 += 64) {
      const uint32_t sidx = s0 + uint32_t(lane);
      if (sidx < total) {
        int lo = 0, hi = 63;                 // first lane whose inclusive count exceeds sidx
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s_incl[wave][mid] > sidx) hi = mid; else lo = mid + 1;
        }
        const uint32_t before = lo ? s_incl[wave][lo - 1] : 0u;
        const int64_t k = int64_t(sidx - before);
        const int64_t slot = s_lbase[wave][lo] + k;
        if (slot < capacity) {
          const int64_t f = s_first[wave][lo], row = f + k;
          if (kMode == kRowsSa) {
            if (f < 0) {
              offsets[slot] = ~f;
            } else {
              offsets[slot] = sa_at(ix, row);
              trace_touch(ix, kTraceSa, sa_line_of(ix, row));
            }
          } else if (f < -(int64_t(3) << 61)) {      // the position itself (kKnownPos + position)
            offsets[slot] = f - kKnownPos;
          } else if (f <= -2) {      // spotted during the search: the marked row's offset minus the symbols searched after it
            const int64_t h = -2 - f;
            const int64_t off = P::marked_offset(ix, h & ((int64_t(1) << 40) - 1));
            offsets[slot] = off < 0 ? -1 : off - (h >> 40);
          } else {
            offsets[slot] = walk_row<P>(ix, row);
          }
        }
      }
    }
    return;
  }
  if (q >= npats || n == 0) return;
  if (n > kExpandSerialMax) {
    atomicOr(big_flag, 1);
    return;
  }
  const int64_t f = first32 ? int64_t(first32[q].x) : first[q];
  const int64_t lim = base + n <= capacity ? n : (capacity > base ? capacity - base : 0);
  for (int64_t k = 0; k < lim; k++) offsets[base + k] = f + k;
}

// the long ranges left over by plan_rows_kernel: grid-stride, one thread per output slot (idle unless the flag is set)
template <int kMode, class P>
inline __global__ __launch_bounds__(256) void plan_big_rows_kernel(const int64_t npats, const int64_t* __restrict__ first, const int2* __restrict__ first32,
                                                            const int64_t* __restrict__ out_starts, const int64_t* __restrict__ total_ptr,
                                                            const int64_t capacity, int64_t* __restrict__ offsets, const int* __restrict__ big_flag,
                                                            const DevIndex ix) {
  if (!*big_flag) return;
  const int64_t total = *total_ptr < capacity ? *total_ptr : capacity;
  for (int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; item < total; item += int64_t(gridDim.x) * blockDim.x) {
    int64_t lo = 0, hi = npats;   // largest q with out_starts[q] <= item
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (out_starts[mid] <= item) lo = mid; else hi = mid;
    }
    const int64_t end = lo + 1 < npats ? out_starts[lo + 1] : *total_ptr;
    if (end - out_starts[lo] > kExpandSerialMax) {
      const int64_t row = (first32 ? int64_t(first32[lo].x) : first[lo]) + (item - out_starts[lo]);
      if (kMode == kRowsSa) {
        offsets[item] = sa_at(ix, row);
        trace_touch(ix, kTraceSa, sa_line_of(ix, row));
      } else if (kMode == kRowsWalk) {
        offsets[item] = walk_row<P>(ix, row);
      } else {
        offsets[item] = row;
      }
    }
  }
}

// the other row expansions (two-call API, host paths): rows already in offsets[] -> their text offsets, full suffix array
inline __global__ __launch_bounds__(256) void gather_sa_kernel(const DevIndex ix, const int64_t total, int64_t* __restrict__ offsets) {
  const int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (item >= total) return;
  const int64_t row = offsets[item];
  offsets[item] = sa_at(ix, row);
  trace_touch(ix, kTraceSa, sa_line_of(ix, row));
}

// ONE step of the locate walk for a batch of rows (femto_amd_lf_steps_device): the unit of work a range-split index exchanges
// walkers in (every round a GPU advances the walkers it owns by one LF step and sends each to the owner of its
// next row, src/main/index.c:1613-1617).  off[i] = the row's text offset when it is marked (the walk ends), else -1 and
// next[i] = LF(row) -- -1 when L[row] is a stop character (the walk cannot cross a document start, server.c:2336-2342).
template <class P>
inline __global__ __launch_bounds__(256) void lf_steps_kernel(const DevIndex ix, const int64_t n, const int64_t* __restrict__ rows,
                                                       int64_t* __restrict__ next, int64_t* __restrict__ off) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = rows[i];
  if (row < 0 || row >= ix.total_length) {
    next[i] = -1;
    off[i] = -1;
    return;
  }
  uint32_t code;
  bool marked;
  int64_t sa_index, nx;
  P::lf(ix, row, code, marked, sa_index, nx);
  off[i] = marked ? mark_offset_at(ix, sa_index) : int64_t(-1);
  next[i] = (marked || P::is_stop(ix, code)) ? int64_t(-1) : nx;
}
// ... from the leaf answers of femto's own tables (modes 0 / 1: block_request_kernel_lane's character, C + Occ and mark offset)
inline __global__ __launch_bounds__(256) void lf_from_leaf_kernel(const int64_t n, const uint16_t* __restrict__ ch, const int64_t* __restrict__ occ,
                                                           int64_t* __restrict__ next, int64_t* __restrict__ off) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  next[i] = (off[i] >= 0 || ch[i] <= uint16_t(kSEOF)) ? int64_t(-1) : occ[i] - 1;      // LF (server.c:2279-2282)
}

// locate walk (do_back_query / do_context_query, src/main/server.c:2228-2359, :2627-2795), persistent grid: the number
// of rows is read from the device word the plan wrote, so count -> plan -> walk is one stream-ordered chain.
// offsets[item] holds the row on entry and the row's text offset on return.
template <class P>
inline __global__ __launch_bounds__(256) void locate_walk_kernel(const DevIndex ix, const int64_t* __restrict__ total_ptr, const int64_t capacity,
                                                          int64_t* __restrict__ offsets) {
  const int64_t total = *total_ptr < capacity ? *total_ptr : capacity;
  for (int64_t item = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; item < total; item += int64_t(gridDim.x) * blockDim.x) {
    offsets[item] = walk_row<P>(ix, offsets[item]);
  }
}

}  // namespace femto_amd
