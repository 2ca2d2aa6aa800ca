// ctx_kernels.hip.hpp -- the CONTEXT TABLE of byte alphabets (mode 4): the first H backward-search steps as ONE hashed read.
//
// The level table (direct_kernels.hip.hpp) is a full t-ary heap, so with t ~ 96 table characters it stops at K = 4
// symbols (96^5 entries would not fit); an English-like 1 GiB text then needs ~8 more steps of two rank lines each before
// a sampled pattern's range is down to one row.  But only the strings that OCCUR matter: the rows whose suffixes share
// their first H characters are contiguous in suffix order, so one pass over the resident suffix array + text (both dense
// arrays of build_text) finds every distinct H-gram of the text with its row range, and a pattern's last H symbols are
// then looked up in an open-addressing hash table instead of being stepped:
//
//   key   = H fields of `bits` bits, first character in the top field: 1 + the character's rank among the table characters
//           (bits = ceil(log2(t + 1)), H <= min(12, 64 / bits): 9 symbols for t ~ 96; no character <= SEOF: such windows
//           are not keys)
//   value = first row (40 bits) | rows (24 bits; 0xffffff = "too many": the pattern takes the level table instead)
//   slot  = 16 bytes {key, value}, key 0 = empty (every field of a key is >= 1);
//           slots = power of two >= 2 x distinct H-grams; linear probing.
//
// A second, WIDE table of the same kind (two-word keys, 32-byte slots {key lo, key hi, value, 0}, H2 <= min(16, 128 / bits),
// 1.4 x the distinct H2-grams slots) answers patterns of at least H2 symbols: 16 symbols for t ~ 96, after which most
// sampled patterns of an English-like 1 GiB text are down to a row or two (measured: H2 = 11 / 13 / 16 -> 3.29 / 2.99 /
// 2.78 ms per 10 M-pattern step before the dense pattern window, 2.27 ms with it).  A pattern tries the wide table, then the narrow one, then the level table.
//
// The table is COMPLETE for windows without stop characters, so a miss means the range is empty -- exactly what
// do_string_query (src/main/server.c:832-936) finds after those H steps.  H is the largest of 12, 11, ... whose table fits
// the budget (a quarter of the free HBM at most); FEMTO_AMD_CTX=0 disables, FEMTO_AMD_CTX_SYMS=h forces.
#pragma once

namespace femto_amd {

constexpr uint64_t kCtxBig = 0xffffffu;
constexpr uint64_t kCtxFirstMask = (uint64_t(1) << 40) - 1;

__device__ __forceinline__ uint64_t ctx_hash(uint64_t key, int log2_slots) {
  return (key * 0x9E3779B97F4A7C15ull) >> (64 - log2_slots);
}

// H-gram starting at text position p as a key (0: not a key -- it runs past the text or holds a stop character)
__device__ __forceinline__ uint64_t ctx_gram(const DevIndex& ix, int64_t p, int H, uint32_t nstop) {
  if (p < 0 || p + H > ix.total_length) return 0;
  const uintptr_t a = reinterpret_cast<uintptr_t>(ix.txt + p);
  const uintptr_t al = a & ~uintptr_t(7);
  const uint64_t w0 = *reinterpret_cast<const uint64_t*>(al);            // txt has 64 bytes of slack behind it
  const uint64_t w1 = *reinterpret_cast<const uint64_t*>(al + 8);
  const uint64_t w2 = H > 9 ? *reinterpret_cast<const uint64_t*>(al + 16) : 0;
  const uint32_t sh = uint32_t(a - al) * 8u;
  const uint64_t lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;         // bytes p .. p+7, p in the low byte
  const uint64_t hi = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;         // bytes p+8 .. p+15
  const int bits = ix.ctx_bits;
  uint64_t key = 0;
  for (int i = 0; i < H; i++) {
    const uint32_t c = uint32_t((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8))) & 0xffu);
    if (c < nstop || c - nstop + 1u >= (1u << bits)) return 0;     // (a code beyond the alphabet: an unwritten byte of a damaged index)
    key = (key << bits) | uint64_t(c - nstop + 1u);
  }
  return key;
}

__device__ __forceinline__ uint64_t ctx_gram_of_row(const DevIndex& ix, int64_t row, int H, uint32_t nstop) {
  if (row < 0 || row >= ix.total_length) return 0;
  return ctx_gram(ix, sa_at(ix, row), H, nstop);
}

// number of rows that start a group of equal keys (= distinct H-grams), added to *count
inline __global__ __launch_bounds__(256) void ctx_count_kernel(const DevIndex ix, const int64_t row0, const int64_t n, const int H, const uint32_t nstop,
                                                        unsigned long long* __restrict__ count) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = row < row0 + n;
  const uint64_t g = in ? ctx_gram_of_row(ix, row, H, nstop) : 0;
  uint32_t plo = uint32_t(__shfl_up(int(uint32_t(g)), 1, 64)), phi = uint32_t(__shfl_up(int(uint32_t(g >> 32)), 1, 64));
  uint64_t prev = (uint64_t(phi) << 32) | plo;
  if ((threadIdx.x & 63u) == 0) prev = in ? ctx_gram_of_row(ix, row - 1, H, nstop) : 0;
  const bool start = in && g != 0 && g != prev;
  const unsigned long long b = __ballot(start);
  if ((threadIdx.x & 63u) == 0 && b) atomicAdd(count, static_cast<unsigned long long>(__popcll(b)));
}

// every group start claims a slot and stores its first row
inline __global__ __launch_bounds__(256) void ctx_insert_kernel(const DevIndex ix, const int64_t row0, const int64_t n, const int H, const uint32_t nstop,
                                                         unsigned long long* __restrict__ slots, const int log2_slots) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = row < row0 + n;
  const uint64_t g = in ? ctx_gram_of_row(ix, row, H, nstop) : 0;
  uint32_t plo = uint32_t(__shfl_up(int(uint32_t(g)), 1, 64)), phi = uint32_t(__shfl_up(int(uint32_t(g >> 32)), 1, 64));
  uint64_t prev = (uint64_t(phi) << 32) | plo;
  if ((threadIdx.x & 63u) == 0) prev = in ? ctx_gram_of_row(ix, row - 1, H, nstop) : 0;
  if (!in || g == 0 || g == prev) return;
  const uint64_t mask = (uint64_t(1) << log2_slots) - 1;
  uint64_t s = ctx_hash(g, log2_slots);
  for (uint64_t probes = 0; probes <= mask; probes++, s = (s + 1) & mask) {
    const unsigned long long old = atomicCAS(slots + 2 * s, 0ull, static_cast<unsigned long long>(g));
    if (old == 0ull) {
      slots[2 * s + 1] = uint64_t(row) & kCtxFirstMask;      // rows are added by ctx_ends_kernel
      return;
    }
    // (old == g cannot happen for a well-formed index: equal keys are contiguous; on a damaged one the later group is lost)
    if (old == g) return;
  }
}

// every group end finds its slot and completes the value
inline __global__ __launch_bounds__(256) void ctx_ends_kernel(const DevIndex ix, const int64_t row0, const int64_t n, const int H, const uint32_t nstop,
                                                       unsigned long long* __restrict__ slots, const int log2_slots) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = row < row0 + n;
  const uint64_t g = in ? ctx_gram_of_row(ix, row, H, nstop) : 0;
  uint32_t nlo = uint32_t(__shfl_down(int(uint32_t(g)), 1, 64)), nhi = uint32_t(__shfl_down(int(uint32_t(g >> 32)), 1, 64));
  uint64_t next = (uint64_t(nhi) << 32) | nlo;
  if ((threadIdx.x & 63u) == 63u || row + 1 >= row0 + n) next = in ? ctx_gram_of_row(ix, row + 1, H, nstop) : 0;
  if (!in || g == 0 || g == next) return;
  const uint64_t mask = (uint64_t(1) << log2_slots) - 1;
  uint64_t s = ctx_hash(g, log2_slots);
  for (uint64_t probes = 0; probes <= mask; probes++, s = (s + 1) & mask) {
    const uint64_t k = slots[2 * s];
    if (k == g) {
      const uint64_t first = slots[2 * s + 1] & kCtxFirstMask;
      const uint64_t rows = uint64_t(row) >= first ? uint64_t(row) - first + 1 : 0;   // (0 only on a damaged index)
      slots[2 * s + 1] = first | ((rows < kCtxBig ? rows : kCtxBig) << 40);
      return;
    }
    if (k == 0) return;
  }
}

// 1: found (first, last set); 0: the H-gram does not occur (range empty); -1: too many rows for the value field
__device__ __forceinline__ int ctx_lookup(const DevIndex& ix, uint64_t key, int64_t& first, int64_t& last) {
  const int lg = ix.ctx_log2;
  const uint64_t mask = (uint64_t(1) << lg) - 1;
  const ulonglong2* const tab = reinterpret_cast<const ulonglong2*>(ix.ctx);
  uint64_t s = ctx_hash(key, lg);
  auto check = [&](const ulonglong2& e) -> int {      // 2: probe on
    if (e.x == key) {
      const uint64_t rows = e.y >> 40;
      if (rows == kCtxBig) return -1;
      first = int64_t(e.y & kCtxFirstMask);
      last = first + int64_t(rows) - 1;
      return 1;
    }
    return e.x == 0 ? 0 : 2;
  };
  // four slots are loaded together: the wavefront waits for its slowest lane's probes (see ctx2_lookup)
  for (uint64_t probes = 0; probes <= mask; probes += 4, s = (s + 4) & mask) {
    const uint64_t s3 = (s + 3) & mask;
    const ulonglong2 e0 = tab[s], e1 = tab[(s + 1) & mask], e2 = tab[(s + 2) & mask], e3 = tab[s3];
    trace_touch(ix, kTraceCtx, s >> 3);
    if ((s3 >> 3) != (s >> 3)) trace_touch(ix, kTraceCtx, s3 >> 3);
    int r;
    if ((r = check(e0)) != 2) return r;
    if ((r = check(e1)) != 2) return r;
    if ((r = check(e2)) != 2) return r;
    if ((r = check(e3)) != 2) return r;
  }
  return 0;
}

// ---- the wide table -------------------------------------------------------------------------------------------------
struct CtxKey2 {
  uint64_t lo, hi;
};
__device__ __forceinline__ bool operator==(const CtxKey2& a, const CtxKey2& b) { return a.lo == b.lo && a.hi == b.hi; }
__device__ __forceinline__ bool operator!=(const CtxKey2& a, const CtxKey2& b) { return !(a == b); }

// key |= field << sh  (0 <= sh < 128, the field may straddle the two words)
__device__ __forceinline__ void ctx_key2_or(CtxKey2& k, uint64_t field, int sh) {
  if (sh < 64) {
    k.lo |= field << sh;
    if (sh) k.hi |= field >> (64 - sh);
  } else {
    k.hi |= field << (sh - 64);
  }
}

// First slot to try for a key in a table of `nslots` slots (any multiple of four, not a power of two: the table is 1.4x
// the distinct keys, not up to 4x): the first slot of a BUCKET of four 32-byte slots = one 128-byte line (bucket = high
// half of hash x buckets).  Probing runs on linearly from there, so a key sits in its own line unless that line's four
// slots were taken -- at a load of 0.7 nine look-ups in ten read ONE line; starting anywhere in a line (plain linear
// probing) the same load made the sigma~96 count kernel 2.16 instead of 1.84 ms, measured.
__device__ __forceinline__ uint64_t ctx_hash2(const CtxKey2& k, uint64_t nslots) {
  const uint64_t h = ((k.lo * 0x9E3779B97F4A7C15ull) ^ (k.hi * 0xC2B2AE3D27D4EB4Full)) * 0xD6E8FEB86659FD93ull;
  return __umul64hi(h, nslots >> 2) << 2;
}

// H-gram (H <= 16) starting at text position p as a wide key ({0,0}: not a key); field i (text order) at bits * (H-1-i)
__device__ __forceinline__ CtxKey2 ctx_gram2(const DevIndex& ix, int64_t p, int H, uint32_t nstop) {
  CtxKey2 key{0, 0};
  if (p < 0 || p + H > ix.total_length) return key;
  const uintptr_t a = reinterpret_cast<uintptr_t>(ix.txt + p);
  const uintptr_t al = a & ~uintptr_t(7);
  const uint64_t w0 = *reinterpret_cast<const uint64_t*>(al);            // txt has 64 bytes of slack behind it
  const uint64_t w1 = *reinterpret_cast<const uint64_t*>(al + 8);
  const uint64_t w2 = *reinterpret_cast<const uint64_t*>(al + 16);
  const uint32_t sh = uint32_t(a - al) * 8u;
  const uint64_t lo = sh ? (w0 >> sh) | (w1 << (64u - sh)) : w0;         // bytes p .. p+7, p in the low byte
  const uint64_t hi = sh ? (w1 >> sh) | (w2 << (64u - sh)) : w1;         // bytes p+8 .. p+15
  const int bits = ix.ctx_bits;
  for (int i = 0; i < H; i++) {
    const uint32_t c = uint32_t((i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8))) & 0xffu);
    if (c < nstop || c - nstop + 1u >= (1u << bits)) return CtxKey2{0, 0};
    ctx_key2_or(key, uint64_t(c - nstop + 1u), bits * (H - 1 - i));
  }
  return key;
}

__device__ __forceinline__ CtxKey2 ctx_gram2_of_row(const DevIndex& ix, int64_t row, int H, uint32_t nstop) {
  if (row < 0 || row >= ix.total_length) return CtxKey2{0, 0};
  return ctx_gram2(ix, sa_at(ix, row), H, nstop);
}

__device__ __forceinline__ CtxKey2 ctx_shfl2(const CtxKey2& g, int delta, bool up) {
  CtxKey2 r;
  uint32_t a = uint32_t(g.lo), b = uint32_t(g.lo >> 32), c = uint32_t(g.hi), d = uint32_t(g.hi >> 32);
  if (up) {
    a = uint32_t(__shfl_up(int(a), delta, 64)); b = uint32_t(__shfl_up(int(b), delta, 64));
    c = uint32_t(__shfl_up(int(c), delta, 64)); d = uint32_t(__shfl_up(int(d), delta, 64));
  } else {
    a = uint32_t(__shfl_down(int(a), delta, 64)); b = uint32_t(__shfl_down(int(b), delta, 64));
    c = uint32_t(__shfl_down(int(c), delta, 64)); d = uint32_t(__shfl_down(int(d), delta, 64));
  }
  r.lo = (uint64_t(b) << 32) | a;
  r.hi = (uint64_t(d) << 32) | c;
  return r;
}

// pass 0: count the group starts; pass 1: every group start claims a slot; pass 2: every group end completes its value
inline __global__ __launch_bounds__(256) void ctx2_build_kernel(const DevIndex ix, const int64_t row0, const int64_t n, const int H, const uint32_t nstop,
                                                         const int pass, unsigned long long* __restrict__ count,
                                                         unsigned long long* __restrict__ slots, const uint64_t nslots) {
  const int64_t row = row0 + int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool in = row < row0 + n;
  const CtxKey2 zero{0, 0};
  const CtxKey2 g = in ? ctx_gram2_of_row(ix, row, H, nstop) : zero;
  if (pass < 2) {
    CtxKey2 prev = ctx_shfl2(g, 1, true);
    if ((threadIdx.x & 63u) == 0) prev = in ? ctx_gram2_of_row(ix, row - 1, H, nstop) : zero;
    const bool start = in && g != zero && g != prev;
    if (pass == 0) {
      const unsigned long long b = __ballot(start);
      if ((threadIdx.x & 63u) == 0 && b) atomicAdd(count, static_cast<unsigned long long>(__popcll(b)));
      return;
    }
    if (!start) return;
    uint64_t s = ctx_hash2(g, nslots);
    for (uint64_t probes = 0; probes < nslots; probes++, s = s + 1 == nslots ? 0 : s + 1) {
      // every distinct key is inserted exactly once (equal keys are contiguous), so a taken slot is simply passed over
      if (atomicCAS(slots + 4 * s, 0ull, static_cast<unsigned long long>(g.lo)) == 0ull) {
        slots[4 * s + 1] = g.hi;
        slots[4 * s + 2] = uint64_t(row) & kCtxFirstMask;
        slots[4 * s + 3] = uint64_t(sa_at(ix, row));      // the text position of the range's FIRST row (ctx2_lookup's sa_first)
        return;
      }
    }
    return;
  }
  CtxKey2 next = ctx_shfl2(g, 1, false);
  if ((threadIdx.x & 63u) == 63u || row + 1 >= row0 + n) next = in ? ctx_gram2_of_row(ix, row + 1, H, nstop) : zero;
  if (!in || g == zero || g == next) return;
  uint64_t s = ctx_hash2(g, nslots);
  for (uint64_t probes = 0; probes < nslots; probes++, s = s + 1 == nslots ? 0 : s + 1) {
    const uint64_t klo = slots[4 * s];
    if (klo == 0) return;
    if (klo == g.lo && slots[4 * s + 1] == g.hi) {
      const uint64_t first = slots[4 * s + 2] & kCtxFirstMask;
      const uint64_t rows = uint64_t(row) >= first ? uint64_t(row) - first + 1 : 0;   // (0 only on a damaged index)
      slots[4 * s + 2] = first | ((rows < kCtxBig ? rows : kCtxBig) << 40);
      return;
    }
  }
}

// 1: found (first, last set); 0: the H-gram does not occur; -1: too many rows for the value field
// The keys of a whole bucket (four slots = one 128-byte line; the probe sequence starts at a bucket and runs on bucket by
// bucket) are loaded TOGETHER: at a load of 0.5 a key sits 2.1 slots into its sequence on average, but the slowest of a
// wavefront's 64 lanes 6.8 -- probing slot by slot, every wavefront waited for seven dependent reads; bucket by bucket it
// waits for two.
constexpr int kCtx2Gang = 4;
// *sa_first = SA[first], stored in the slot's fourth word: a pattern whose sixteen last symbols occur ONCE -- most of a sampled
// batch on a large text -- goes from this line straight to the text, without the suffix-array read in between.
__device__ __forceinline__ int ctx2_lookup_in(const DevIndex& ix, const uint64_t* __restrict__ table, const uint64_t nslots, const int64_t trace_off,
                                              const CtxKey2& key, int64_t& first, int64_t& last, int64_t* sa_first) {
  uint64_t s = ctx_hash2(key, nslots);
  for (uint64_t probes = 0; probes < nslots; probes += kCtx2Gang) {
    ulonglong2 e[kCtx2Gang];
#pragma unroll
    for (int i = 0; i < kCtx2Gang; i++) e[i] = reinterpret_cast<const ulonglong2*>(table)[2 * (s + uint64_t(i))];
    trace_touch(ix, kTraceCtx, uint64_t(trace_off) + (s >> 2));
#pragma unroll
    for (int i = 0; i < kCtx2Gang; i++) {
      if (e[i].x == 0) return 0;
      if (e[i].x == key.lo && e[i].y == key.hi) {
        const ulonglong2 vv = reinterpret_cast<const ulonglong2*>(table)[2 * (s + uint64_t(i)) + 1];
        const uint64_t v = vv.x;
        if (sa_first) *sa_first = int64_t(vv.y);
        const uint64_t rows = v >> 40;
        if (rows == kCtxBig) return -1;
        first = int64_t(v & kCtxFirstMask);
        last = first + int64_t(rows) - 1;
        return 1;
      }
    }
    s += kCtx2Gang;
    if (s >= nslots) s = 0;
  }
  return 0;
}
__device__ __forceinline__ int ctx2_lookup(const DevIndex& ix, const CtxKey2& key, int64_t& first, int64_t& last, int64_t* sa_first = nullptr) {
  return ctx2_lookup_in(ix, ix.ctx2, ix.ctx2_slots, ix.ctx2_trace_off, key, first, last, sa_first);
}
// the table of the middle length: its key is the wide key cut down to the last ctxm_syms symbols (field j from the end sits at
// bits * j in both)
__device__ __forceinline__ int ctxm_lookup(const DevIndex& ix, const CtxKey2& key2, int64_t& first, int64_t& last, int64_t* sa_first = nullptr) {
  const int nb = ix.ctx_bits * ix.ctxm_syms;
  CtxKey2 key;
  key.lo = nb >= 64 ? key2.lo : (key2.lo & ((1ull << nb) - 1ull));
  key.hi = nb <= 64 ? 0ull : (nb >= 128 ? key2.hi : (key2.hi & ((1ull << (nb - 64)) - 1ull)));
  return ctx2_lookup_in(ix, ix.ctxm, ix.ctxm_slots, ix.ctxm_trace_off, key, first, last, sa_first);
}

}  // namespace femto_amd
