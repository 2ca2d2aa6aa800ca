// text_kernels.hip.hpp -- long patterns: once the range of a backward search is a SINGLE row, the rest of the
// pattern is compared against the text itself instead of being searched symbol by symbol.
//
// A search step costs one (mode 3) or two (mode 4) scattered memory lines.  A 100-symbol read, or a 40-symbol phrase,
// is down to one row after ~16 / ~8 symbols; every further step can only keep that row or empty the range.  With the
// text (dense codes, one byte per position, rebuilt at open from the index by locating every row:
// txt[SA[row]-1] = L[row]) and a sampled inverse suffix array (isa8[i] = row of the suffix at position 8i) the tail is
//   1. p = SA[row]                                   (LF walk to the next derived mark: <= 4 steps)
//   2. compare the remaining symbols with txt[p-1], txt[p-2], ...      (one or two lines)
//   3. the row of the last matching position from isa8 + <= 7 LF steps; if a symbol mismatched, ONE ordinary search
//      step with that symbol yields exactly the (first, last) the symbol-by-symbol search dies with.
// The count kernels only hand such patterns over (slot, row, symbols done) to count_tail_kernel; patterns that meet a
// document boundary, a character <= SEOF, or the last few text positions simply continue symbol by symbol there.
#pragma once

namespace femto_amd {

constexpr int kIsaShift = 3;          // sampled isa8: every 8th text position (DevIndex::isa_shift; 0 = every position)

// The suffix array of every row and the inverse suffix array hold 4-BYTE entries on an index of fewer than 2^32 - 1 rows
// (DevIndex::sa32; round 6): 4.3 + 4.3 GB instead of 8.6 + 8.6 GB at 1 GiB of text -- what lets a handle under the default
// bound (8 x text) keep the suffix array at all, and twice the rows per line where a range's rows are read together.
// 0xffffffff = -1 ("this row could not be located": text_isa_build_kernel on a damaged index).
__device__ __forceinline__ int64_t sa_at(const DevIndex& ix, int64_t row) {
  if (ix.sa32) {
    const uint32_t v = reinterpret_cast<const uint32_t*>(ix.sa_full)[row];
    return v == 0xffffffffu ? int64_t(-1) : int64_t(v);
  }
  return ix.sa_full[row];
}
__device__ __forceinline__ int64_t isa_at(const DevIndex& ix, int64_t i) {
  if (ix.sa32) return int64_t(reinterpret_cast<const uint32_t*>(ix.isa8)[i]);
  return ix.isa8[i];
}
// 128-byte line of entry i of either array (trace regions)
__device__ __forceinline__ uint64_t sa_line_of(const DevIndex& ix, int64_t i) { return uint64_t(i) >> (ix.sa32 ? 5 : 4); }

struct TailItem {
  uint32_t slot;      // position in the sorted batch
  int32_t done;       // symbols already searched (j)
  int64_t row;        // first == last
};

// nfa_search_kernel's "spec" (regexp_search.hip): a policy whose search step is ONE load per range end may split it into the
// loads (issued as soon as the next entry's range is known) and the arithmetic (done when that entry is popped)
struct NoSpec {
  static constexpr bool kNfaSpec = false;
  struct Spec {};
  static __device__ __forceinline__ void spec_load(const DevIndex&, uint32_t, int64_t, int64_t, Spec&) {}
};
struct PackPolicy : NoSpec {
  static constexpr bool kSpotMarks = false;   // search steps do not report marked rows (RumPolicy does)
  static __device__ __forceinline__ int64_t marked_offset(const DevIndex& ix, int64_t row) { return pack_marked_offset(ix, row); }
  // one word of the memory line search_step(code, .. row) will read: issued early by a caller that knows the row ahead of time
  // (nfa_search_kernel: the entry it pops next), so that the line is on its way while other work runs
  static __device__ __forceinline__ uint32_t touch(const DevIndex& ix, uint32_t, int64_t row) {
    uint64_t line;
    uint32_t r;
    pack_split(row, &line, &r);
    return ix.pack[line * kPackLineWords];
  }
  static constexpr int kWaves = 8;     // waves per SIMD the count kernel is built for (64 VGPRs)
  static constexpr int kPlanWaves = 7; // ... plan_rows_kernel at least (the walk inside it: 66 VGPRs)
  static constexpr int kNfaWaves = 4;  // ... nfa_search_kernel (regexp_search.hip): what it reaches without scratch
  static constexpr int kDirectWaves = 7;   // ... count_direct_kernel: 72 VGPRs (at 64 it spills 20 bytes per lane)
  static constexpr int kTailRows = 1;  // ranges of up to this many rows take the text tail (direct_kernels.hip.hpp)
  static __device__ __forceinline__ void search_step(const DevIndex& ix, int j, uint32_t code, int64_t& f, int64_t& l) {
    pack_search_step(ix, ix.pack, j, code, f, l);
  }
  // one LF step from `row`: its character, mark state and the row of the preceding position
  static __device__ __forceinline__ void lf(const DevIndex& ix, int64_t row, uint32_t& code, bool& marked, int64_t& sa_index, int64_t& next) {
    uint64_t line;
    uint32_t r;
    pack_split(row, &line, &r);
    PackLine L;
    pack_load_line(ix.pack, line, L);
    trace_touch(ix, kTracePack, line);
    const PackStep s = pack_step(L, r);
    code = s.code;
    marked = s.marked;
    sa_index = s.sa_index;
    next = s.c_plus_occ - 1;
  }
  static __device__ __forceinline__ bool is_stop(const DevIndex& ix, uint32_t code) { return (ix.pack_stop >> code) & 1u; }
  static __device__ __forceinline__ uint32_t max_code(const DevIndex&) { return 7u; }
  static __device__ __forceinline__ uint32_t stop_info(const DevIndex& ix) { return ix.pack_stop; }
  static __device__ __forceinline__ bool is_stop_by(uint32_t info, uint32_t code) { return (info >> code) & 1u; }
  static __device__ __forceinline__ uint32_t code_of(const DevIndex& ix, uint32_t ch) {
    const uint32_t c = ix.pack_code[ch];
    return c > 7u ? 0xffffu : c;
  }
};

// small alphabets with the rank units resident (ru_kernels.hip.hpp): a search step is ONE 16-byte load per range end;
// LF steps (locate walks) do not know their character and stay on the packed lines
#ifndef FEMTO_AMD_EXP_RU_WAVES
#define FEMTO_AMD_EXP_RU_WAVES 8       // (experiments: tools/ab_bench.sh builds a second library with another value)
#endif
struct RuPolicy : PackPolicy {
  static constexpr int kNfaWaves = 4;  // (no scratch: a spill reload would sit in the chain of dependent steps that IS this kernel's time, and the batch ends with its longest search: occupancy buys little)
  static constexpr int kDirectWaves = FEMTO_AMD_EXP_RU_WAVES;   // 43-64 VGPRs: eight waves per SIMD without spilling
  static __device__ __forceinline__ void search_step(const DevIndex& ix, int j, uint32_t code, int64_t& f, int64_t& l) {
    ru_search_step(ix, j, code, f, l);
  }
  static __device__ __forceinline__ uint32_t touch(const DevIndex& ix, uint32_t code, int64_t row) {
    const uint32_t tc = code < uint32_t(ix.ru_nstop) ? 0u : code - uint32_t(ix.ru_nstop);      // (a stop character has no units: any valid address, unconditionally)
    uint64_t u;
    uint32_t r;
    ru_split(row, &u, &r);
    return reinterpret_cast<const uint32_t*>(ix.ru)[(uint64_t(tc) * uint64_t(ix.ru_stride) + u) * 4];
  }
  static constexpr bool kNfaSpec = true;
  struct Spec { uint4 vL, vF; };
  // the two units search_step(code, [f, l]) reads, loaded unconditionally (a stop character: any valid units, never used)
  static __device__ __forceinline__ void spec_load(const DevIndex& ix, uint32_t code, int64_t f, int64_t l, Spec& sp) {
    const uint32_t tc = code < uint32_t(ix.ru_nstop) ? 0u : code - uint32_t(ix.ru_nstop);
    const uint4* const up = reinterpret_cast<const uint4*>(ix.ru) + uint64_t(tc) * uint64_t(ix.ru_stride);
    uint64_t uL, uF;
    uint32_t r;
    ru_split(l, &uL, &r);
    ru_split(f > 0 ? f - 1 : 0, &uF, &r);
    sp.vL = up[uL];
    sp.vF = up[uF];
  }
  // ... and the step itself from them (j > 0)
  static __device__ __forceinline__ void spec_step(const DevIndex& ix, uint32_t code, int64_t& f, int64_t& l, const Spec& sp) {
    if (code < uint32_t(ix.ru_nstop)) {
      ru_stop_step(ix, code, f, l);
      return;
    }
    uint64_t u;
    uint32_t rL, rF = 0;
    ru_split(l, &u, &rL);
    if (f > 0) ru_split(f - 1, &u, &rF);
    const int64_t nl = ru_rank_of(sp.vL, rL);
    f = f > 0 ? ru_rank_of(sp.vF, rF) : ix.pack_c[code];
    l = nl - 1;
  }
};

// ... with the MARKED rank units (handles without the suffix array): a one-row step also says whether its row is marked
struct RumPolicy : PackPolicy {
  static constexpr int kNfaWaves = 4;
  static constexpr int kDirectWaves = FEMTO_AMD_EXP_RU_WAVES;
  static constexpr bool kSpotMarks = true;
  static __device__ __forceinline__ void search_step(const DevIndex& ix, int j, uint32_t code, int64_t& f, int64_t& l) {
    bool s;
    rum_search_step(ix, j, code, f, l, &s);
  }
  static __device__ __forceinline__ void search_step_spot(const DevIndex& ix, int j, uint32_t code, int64_t& f, int64_t& l, bool* spotted) {
    rum_search_step(ix, j, code, f, l, spotted);
  }
  static __device__ __forceinline__ uint32_t touch(const DevIndex& ix, uint32_t code, int64_t row) {
    const uint32_t tc = code < uint32_t(ix.ru_nstop) ? 0u : code - uint32_t(ix.ru_nstop);
    return reinterpret_cast<const uint32_t*>(ix.ru)[(uint64_t(tc) * uint64_t(ix.ru_stride) + (uint64_t(row) >> 6)) * 4];
  }
  static constexpr bool kNfaSpec = true;
  struct Spec { uint4 vL, vF; };
  static __device__ __forceinline__ void spec_load(const DevIndex& ix, uint32_t code, int64_t f, int64_t l, Spec& sp) {
    const uint32_t tc = code < uint32_t(ix.ru_nstop) ? 0u : code - uint32_t(ix.ru_nstop);
    const uint4* const up = reinterpret_cast<const uint4*>(ix.ru) + uint64_t(tc) * uint64_t(ix.ru_stride);
    sp.vL = up[uint64_t(l) >> 6];
    sp.vF = up[uint64_t(f > 0 ? f - 1 : 0) >> 6];
  }
  static __device__ __forceinline__ void spec_step(const DevIndex& ix, uint32_t code, int64_t& f, int64_t& l, const Spec& sp) {
    if (code < uint32_t(ix.ru_nstop)) {
      ru_stop_step(ix, code, f, l);
      return;
    }
    const int64_t nl = rum_rank_of(sp.vL, uint32_t(l) & 63u);
    f = f > 0 ? rum_rank_of(sp.vF, uint32_t(f - 1) & 63u) : ix.pack_c[code];
    l = nl - 1;
  }
};

struct Pack2Policy : NoSpec {
  static __device__ __forceinline__ uint32_t touch(const DevIndex&, uint32_t, int64_t) { return 0; }     // (two dependent lines: not worth a guess)
  static constexpr int kNfaWaves = 3;
  static constexpr int kPlanWaves = 6; // (the walk's LF step on the two-level lines: 79 VGPRs without scratch; left alone the compiler takes 82 and five waves)
  static constexpr bool kSpotMarks = false;
  static __device__ __forceinline__ int64_t marked_offset(const DevIndex&, int64_t) { return -1; }
  static constexpr int kWaves = 8;
  static constexpr int kDirectWaves = 7;   // (72 VGPRs: at 64 the search step's second exit -- a character with a level-1 class of its own -- spills 20 bytes)
  static constexpr int kTailRows = 4;  // repeated phrases of a byte text: a few rows with tens of symbols to go
  static __device__ __forceinline__ void search_step(const DevIndex& ix, int j, uint32_t code, int64_t& f, int64_t& l) {
    p2_search_step(ix, j, code, f, l);
  }
  static __device__ __forceinline__ void lf(const DevIndex& ix, int64_t row, uint32_t& code, bool& marked, int64_t& sa_index, int64_t& next) {
    const P2Step s = p2_step(ix, row);
    code = s.code;
    marked = s.marked;
    sa_index = s.sa_index;
    next = s.c_plus_occ - 1;
  }
  static __device__ __forceinline__ bool is_stop(const DevIndex& ix, uint32_t code) { return code < ix.p2_stop_below; }
  static __device__ __forceinline__ uint32_t max_code(const DevIndex& ix) { return uint32_t(ix.p2_sigma) - 1u; }
  static __device__ __forceinline__ uint32_t stop_info(const DevIndex& ix) { return ix.p2_stop_below; }
  static __device__ __forceinline__ bool is_stop_by(uint32_t info, uint32_t code) { return code < info; }
  static __device__ __forceinline__ uint32_t code_of(const DevIndex& ix, uint32_t ch) {
    const uint32_t c = ix.p2_code[ch];
    return c > 255u ? 0xffffu : c;
  }
};

// byte alphabets with the per-character rank lines resident (ind_kernels.hip.hpp): search steps read one line per range
// end; everything that does not know its character in advance (LF steps) stays on the two-level lines
struct IndPolicy : Pack2Policy {
  static constexpr int kDirectWaves = 8;
  static constexpr int kNfaWaves = 4;
  static __device__ __forceinline__ uint32_t touch(const DevIndex& ix, uint32_t code, int64_t row) {
    uint64_t line;
    uint32_t b;
    ind_split(row, &line, &b);
    return ix.ind[(uint64_t(code) * uint64_t(ix.ind_stride) + line) * 32];
  }
  static constexpr int kWaves = 8;
  static __device__ __forceinline__ void search_step(const DevIndex& ix, int j, uint32_t code, int64_t& f, int64_t& l) {
    ind_search_step(ix, j, code, f, l);
  }
};

// SA[row]: one read of the full suffix array when it is resident (kSaFull), else the locate walk; false when the walk
// cannot finish (never for a well-formed index)
template <class P, bool kSaFull = false>
__device__ __forceinline__ bool tail_locate(const DevIndex& ix, int64_t row, int64_t* pos, uint32_t* first_code) {
  if (kSaFull) {
    *pos = sa_at(ix, row);
    trace_touch(ix, kTraceSa, sa_line_of(ix, row));
    return true;
  }
  int64_t steps = 0;
  for (;;) {
    uint32_t code;
    bool marked;
    int64_t sa_index, next;
    P::lf(ix, row, code, marked, sa_index, next);
    if (steps == 0 && first_code) *first_code = code;
    if (marked) {
      *pos = mark_offset_at(ix, sa_index) + steps;
      return true;
    }
    if (P::is_stop(ix, code) || steps > int64_t(ix.walk_limit)) return false;
    row = next;
    steps++;
  }
}

// build: txt[SA[row] - 1] = L[row] (position -1 wraps to the last one), isa[SA[row] >> shift] = row for the sampled
// positions (shift 0: all of them) and, kSa, sa_full[row] = SA[row]
template <class P, bool kSa>
inline __global__ __launch_bounds__(256) void text_isa_build_kernel(const DevIndex ix, const int64_t row0, const int64_t n, uint8_t* __restrict__ txt,
                                                             int64_t* __restrict__ isa, const int isa_shift, int64_t* __restrict__ sa_full,
                                                             const int w32 /* 1: both arrays hold 4-byte entries */) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= row0 + n) return;
  int64_t pos;
  uint32_t code = 0;
  if (!tail_locate<P>(ix, row, &pos, &code) || pos < 0 || pos >= ix.total_length) {
    // an inconsistent index (an LF cycle without marks, a damaged mark array holding an offset outside the text): the row
    // reports -1 as the walk does, and nothing is written outside the arrays
    if (kSa) {
      if (w32) reinterpret_cast<uint32_t*>(sa_full)[row] = 0xffffffffu;
      else sa_full[row] = -1;
    }
    return;
  }
  txt[pos == 0 ? ix.total_length - 1 : pos - 1] = uint8_t(code);
  if ((pos & ((int64_t(1) << isa_shift) - 1)) == 0) {
    if (w32) reinterpret_cast<uint32_t*>(isa)[pos >> isa_shift] = uint32_t(row);
    else isa[pos >> isa_shift] = row;
  }
  if (kSa) {
    if (w32) reinterpret_cast<uint32_t*>(sa_full)[row] = uint32_t(pos);
    else sa_full[row] = pos;
  }
}

// row of the suffix starting at text position x (sampled ISA + LF walk); false near the text end or a document end
template <class P>
__device__ __forceinline__ bool tail_row_of(const DevIndex& ix, int64_t x, int64_t* row_out) {
  const int64_t K = int64_t(1) << ix.isa_shift;
  const int64_t s = (x + K - 1) & ~(K - 1);
  if (s >= ix.total_length) return false;
  int64_t row = isa_at(ix, s >> ix.isa_shift);
  trace_touch(ix, kTraceIsa, sa_line_of(ix, s >> ix.isa_shift));
  for (int64_t k = s; k > x; k--) {
    uint32_t code;
    bool marked;
    int64_t sa_index, next;
    P::lf(ix, row, code, marked, sa_index, next);
    if (P::is_stop(ix, code)) return false;
    row = next;
  }
  *row_out = row;
  return true;
}

// One pattern symbol, (j)-th from the end: from the sort key when it holds it, else from the pattern (four symbols per
// aligned load).  Returns 0: `code` valid; 1: pattern exhausted (key says so); 2: symbol >= 261; 3: character not in the
// text (`ch` set).
struct SymbolReader {
  uint64_t key;
  bool whole;
  int bits, nsym, len;
  const uint16_t* pat;
  uint64_t word;
  uintptr_t word_addr;
};

template <class P>
__device__ __forceinline__ int tail_symbol(const DevIndex& ix, SymbolReader& R, int j, uint32_t* code, uint32_t* ch_out) {
  uint32_t c = 0;
  if (j < R.nsym) c = uint32_t(R.key >> (64 - R.bits * (j + 1))) & ((1u << R.bits) - 1u);
  if (c != 0) {
    *code = c - 1;
    return 0;
  }
  if (R.whole) return 1;
  const uintptr_t sa = reinterpret_cast<uintptr_t>(R.pat + (R.len - 1 - j));
  const uintptr_t wa = sa & ~uintptr_t(7);
  if (wa != R.word_addr) {
    R.word = *reinterpret_cast<const uint64_t*>(wa);
    R.word_addr = wa;
  }
  const uint32_t ch = uint32_t(R.word >> (8 * (sa - wa))) & 0xffffu;
  *ch_out = ch;
  if (ch >= uint32_t(kAlphaSize)) return 2;
  const uint32_t cc = P::code_of(ix, ch);
  if (cc == 0xffffu) return 3;
  *code = cc;
  return 0;
}

// Direct batches (direct_kernels.hip.hpp: perm == NULL, slot == pattern index, no sort keys) store into the caller's
// arrays themselves and, when a locate plan is being made (noccs != NULL), the clamped row count too -- added to the
// sum of the pattern's 256-pattern block, which plan_scan_kernel scans afterwards.
struct TailOut {
  longlong2* pair_out;      // sorted batches: (first,last) pairs, split later
  int64_t* first_out;       // direct batches
  int64_t* last_out;        // NULL: first_out receives the count
  int32_t* noccs;           // locate plan, or NULL
  int64_t* block_sums;
  int max_occs;
  int64_t* sa_out;          // or NULL: count_direct_kernel's sa_out (text positions of one-row patterns the search already knows)
  int row_free;             // 1: no rows are returned (DevIndex::row_free): a tail that consumes the pattern leaves its POSITION in sa_out
                            // and skips the way back to a row (the sampled inverse suffix array + up to 7 LF steps); first_out gets the
                            // rows plan_rows_kernel still needs, last_out nothing
};

template <class P, bool kSaFull>
inline __global__ __launch_bounds__(256) void count_tail_kernel(const DevIndex ix, const TailItem* __restrict__ items, const int* __restrict__ n_items,
                                                         const int32_t* __restrict__ plen, const uint16_t* __restrict__ pats,
                                                         const int64_t* __restrict__ starts, const uint32_t* __restrict__ perm,
                                                         const uint64_t* __restrict__ keys, const int bits, const int nsym,
                                                         const TailOut out, int* __restrict__ err_flag) {
  const int64_t n_it = int64_t(*n_items);
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n_it; t += int64_t(gridDim.x) * blockDim.x) {
  const TailItem it = items[t];
  const int64_t q = perm ? int64_t(perm[it.slot]) : int64_t(it.slot);
  SymbolReader R;
  R.key = keys ? keys[it.slot] : 0;
  R.whole = false;
  R.bits = bits;
  R.nsym = keys ? nsym : 0;
  R.len = plen[q];
  R.pat = pats + starts[q];
  R.word = 0;
  R.word_addr = 0;
  const int len = R.len;
  int j = it.done;
  int64_t first = it.row, last = it.row;

  // ---- the shortcut: position of the row, compare against the text, row of the last matching position
  int64_t p;
  if (tail_locate<P, kSaFull>(ix, first, &p, nullptr) && p >= int64_t(len - j) && p < ix.total_length) {
    const int remaining = len - j;
    int m = 0;                       // symbols matched
    uint64_t tw = 0;                 // aligned 8-byte word of txt holding the byte being compared
    uintptr_t tw_addr = 0;
    bool differs = false;            // the compare ended on a text character other than the pattern's (an ordinary one)
    for (; m < remaining; m++) {
      uint32_t code = 0, ch = 0;
      if (tail_symbol<P>(ix, R, j + m, &code, &ch) != 0) break;   // leave anything unusual to the ordinary step below
      if (P::is_stop(ix, code)) break;
      const uintptr_t ta = reinterpret_cast<uintptr_t>(ix.txt + (p - 1 - m));
      const uintptr_t wa = ta & ~uintptr_t(7);
      if (wa != tw_addr) {
        tw = *reinterpret_cast<const uint64_t*>(wa);
        tw_addr = wa;
        trace_touch(ix, kTraceTxt, uint64_t(wa - reinterpret_cast<uintptr_t>(ix.txt)) >> 7);
      }
      const uint32_t tc = uint32_t(tw >> (8 * (ta - wa))) & 0xffu;
      if (tc != code) {
        differs = true;
        break;
      }
    }
    if (out.row_free && out.noccs && (m == remaining || differs)) {
      // row-free locate (direct_kernels.hip.hpp): located by the compare itself, or empty -- no way back to a row
      const int64_t c = m == remaining ? 1 : 0;
      out.noccs[q] = int32_t(c);
      if (c) {
        atomicAdd(reinterpret_cast<unsigned long long*>(out.block_sums + (q >> 8)), static_cast<unsigned long long>(c));
        out.sa_out[q] = p - m;
      }
      continue;
    }
    if (m > 0) {
      int64_t row;
      if (tail_row_of<P>(ix, p - m, &row)) {
        first = last = row;
        j += m;
      }
    }
  }
  // ---- whatever is left (the mismatching symbol, or everything when the shortcut did not apply): ordinary steps
  for (; j < len; j++) {
    uint32_t code = 0, ch = 0;
    const int st = tail_symbol<P>(ix, R, j, &code, &ch);
    if (st == 1) break;
    if (st == 2) {
      atomicOr(err_flag, 1);
      first = 0;
      last = -1;
      break;
    }
    if (st == 3) {  // the character does not occur in the text: Occ == 0 (index.c:2080-2089)
      first = ix.C[ch];
      last = first - 1;
      break;
    }
    P::search_step(ix, j, code, first, last);
    if (first > last) break;
  }
  if (out.pair_out) {
    out.pair_out[q] = make_longlong2(first, last);
  } else {
    if (out.row_free && out.noccs) {
      if (first <= last) out.first_out[q] = first;
    } else if (out.last_out) {
      out.first_out[q] = first;
      out.last_out[q] = last;
    } else {
      out.first_out[q] = last - first + 1;
    }
    if (out.noccs) {   // do_locate_query's clamp (server.c:4405-4415)
      int64_t c;
      if (first > last) c = 0;
      else if (last - first > int64_t(out.max_occs)) c = out.max_occs;
      else c = last - first + 1;
      out.noccs[q] = int32_t(c);
      if (c) atomicAdd(reinterpret_cast<unsigned long long*>(out.block_sums + (q >> 8)), static_cast<unsigned long long>(c));
    }
  }
  }
}

}  // namespace femto_amd
