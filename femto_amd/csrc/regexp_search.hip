// regexp_search.hip -- batched NFA search over the index: do_regexp_query (src/main/server.c:1656-2163) for
// MANY automata at once, entirely on the GPU.
//
// The reference simulates one nfa_description_t (src/main/nfa.h:62-88) backwards over the index: a map from row range
// [first,last] to the set of NFA states (one error count per state) the strings with that range have reached, kept in a
// STACK with a hash on the range (queue_map, src/utils/queue_map.c -- "THIS ... IS NOT ACTUALLY A QUEUE; IT IS A STACK").
// Pop an entry; if a final state is alive it is a result and is not extended; otherwise, for every character some alive
// state can read (every character of the alphabet while errors are left), new_first = C[ch] + Occ(ch, first-1),
// new_last = C[ch] + Occ(ch, last) - 1 -- the 261-way fan-out -- and the child (new range -> states after reading ch,
// merged with the substitution and insertion states) is pushed, or min-merged into the pending entry that already has
// that range (add_mapping, server.c:1558-1652).  The results are sorted, de-duplicated and ranges inside other ranges
// dropped (regexp_result_list_sort, server.c:1528-1573).
//
// Here ONE 64-lane workgroup runs one automaton's whole search, and a batch of automata fills the GPU:
//   * the stack (ranges, match lengths, one cost byte per node and entry) lives in a per-workgroup arena in HBM;
//   * the popped entry's cost vector, the per-character transition lists (the automaton's transitions sorted by
//     character at upload: reading character ch touches exactly its own entries) and the children live in LDS;
//   * lane k computes the k-th reachable character's new range -- the fan-out is one wavefront-wide step on the index's
//     fastest rank layout (packed lines / per-character rank lines / femto's wavelet tree);
//   * the lanes stride over transition entries (LDS atomicMin), over the nodes of a cost vector, and over the pending
//     entries when a child looks for an entry with its range.
// The order of every push, merge and pop is the reference's (it decides which match length and cost a merged entry
// reports), so the result lists are IDENTICAL to do_regexp_query's for the same nfa_description_t: pinned by
// tests/golden/*_regexp.npz (generated through setup_regexp_query_take_nfa, oracle/ref_tool.c) and tests/test_regexp.py.
#include <algorithm>
#include <climits>
#include <cstring>

#include "api_internal.hpp"
#include "kernels.hip.hpp"
#include "pack_kernels.hip.hpp"
#include "ru_kernels.hip.hpp"
#include "pack2_kernels.hip.hpp"
#include "ind_kernels.hip.hpp"
#include "text_kernels.hip.hpp"
#include "regexp_nfa.hpp"
#include "query_parser.hpp"

using namespace femto_amd;

namespace femto_amd {

constexpr int kNfaMaxNodes = 2048;        // cost vectors of one automaton live in LDS
constexpr int kNfaOffset = 5;             // CHARACTER_OFFSET (src/main/index_types.h): alpha codes below it are EOF/SEOF/...
constexpr int kNfaDead = 255;             // MAX_NFA_ERRCNT (src/main/nfa.h:75)
constexpr int kNfaStatusFull = FEMTO_AMD_ERR_FULL;
constexpr int kNfaStatusOverworked = FEMTO_AMD_ERR_OVERWORKED;

struct NfaQueryDev {
  int64_t node_off;       // first node's flag byte (bit 0: start node, bit 1: final node)
  int64_t ent_off;        // first transition entry (sorted by character)
  int64_t bychar_off;     // this automaton's 262 by-character starts
  int32_t num_nodes, num_ents;
  int32_t cost_bound, subst, del, ins;
};
struct NfaResultDev {
  int64_t first, last;
  int32_t query, len, cost, pass;   // pass: the attempt that produced it (a search that ran out of stack is run again)
};
struct NfaBatchDev {
  const NfaQueryDev* queries;
  const int32_t* order;      // queries[order[i]] is the i-th to be taken (NULL: identity)
  int32_t nq;
  const uint8_t* node_flags;
  const uint32_t* ent_sd;    // source node | destination node << 16
  const uint16_t* ent_ch;    // the entry's character
  const int32_t* bychar;
  uint8_t* arena;            // per workgroup: first[cap] i64 | last[cap] i64 | len[cap] i32 | cost[cap][cost_stride] u8
  int64_t arena_bytes;
  int32_t cap, cost_stride;
  int32_t hash_size;          // heads of the pending-range hash: a power of two >= 2 * cap
  int32_t* next;             // work counter
  NfaResultDev* results;
  int64_t result_cap;
  unsigned long long* result_count;
  int32_t* status;           // per query
  int32_t* iters_out;        // NULL, or per query: entries popped (FEMTO_AMD_NFA_STATS=1 prints their distribution)
  int64_t max_iterations;    // MAX_REGEXP_ITERATIONS (src/main/server.c:40)
  int32_t pass;
  int32_t lds_nodes, lds_children;   // sizes of the workgroup's dynamic LDS arrays (nfa_lds_bytes)
  int32_t lds_ents;                  // > 0: every automaton's transitions and node flags are copied to LDS (room for this many entries)
};

// mode 1: femto's own wavelet tree through the derived segment lines (alphabets of more than 256 characters, range-split
// indexes); "code" is the alpha code itself
struct WavePolicy {
  static constexpr int kNfaWaves = 3;
  static __device__ __forceinline__ void search_step(const DevIndex& ix, int, uint32_t ch, int64_t& f, int64_t& l) {
    const int64_t nf = f == 0 ? ix.C[ch] : c_plus_occ_lane(ix, ch, f - 1);
    const int64_t nl = c_plus_occ_lane(ix, ch, l) - 1;
    f = nf;
    l = nl;
  }
  static __device__ __forceinline__ uint32_t code_of(const DevIndex& ix, uint32_t ch) { return ix.C[ch + 1] > ix.C[ch] ? ch : 0xffffu; }
};

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int o = __shfl_xor(v, d, 64);
    v = o < v ? o : v;
  }
  return v;
}
// a value every lane of the wavefront holds (read from one address, or the result of a wavefront reduction), moved to a
// scalar register: the search loop below is one wavefront working on ONE automaton, and most of what it carries -- the
// popped range, counts, slots -- is uniform; left in vector registers it cost the kernel its occupancy (105 VGPRs)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uint64_t(v))))), hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uint64_t(v) >> 32))));
  return int64_t((uint64_t(hi) << 32) | lo);
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Dynamic LDS of one workgroup (sized per launch from the LARGEST automaton of the batch and the number of characters the
// text holds, so that typical automata -- a few dozen nodes on a small alphabet -- leave room for a CU's full complement of
// wavefronts; until round 5 the arrays were static for 2048 nodes and 264 children: 19 KB, 8 workgroups per CU):
//   s_tmp u32[nn] | s_child_f i64[cc] | s_child_l i64[cc] | s_child_found i32[cc] | s_live i32[cc] | s_child_h u32[cc] |
//   s_child_head i32[cc] | s_child_slot i32[cc] | s_child_ch u16[cc] | s_cur u8[nn] | s_sub u8[nn] | s_top u8[nn]
//                                           nn = lds_nodes (multiple of 8), cc = lds_children (multiple of 4)
//   kLds (the batch's largest automaton has at most kNfaLdsEnts transitions): + s_ent_sd u32[ne] | s_ent_ch u16[ne] | s_flags u8[nn] --
//   the automaton itself.  A pop reads the transition list five or six times (deletions, reachable characters,
//   substitutions, one slice per live child) and the node flags once; from global memory each of those is a dependent
//   round trip of a kernel that waits two thirds of its cycles (profiles/r05_regexp_stats.txt: SQ_WAIT_ANY 68 % of the
//   wave cycles at 4 waves per SIMD); the copy is made once per automaton.
constexpr int kNfaLdsEnts = 2048;
constexpr int kNfaPend = 1024;            // counters of pending entries by hash bucket, in LDS
__host__ __device__ inline size_t nfa_lds_bytes(int nn, int cc, int ne) {
  return size_t(nn) * 4 + size_t(cc) * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 2) + size_t(nn) * 3 + 16 + size_t(ne) * 6 + (ne ? size_t(nn) + 16 : 0);
}

template <class P, bool kLds>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(P::kNfaWaves, P::kNfaWaves))) void nfa_search_kernel(const DevIndex ix, const NfaBatchDev B) {
  extern __shared__ __align__(16) uint8_t s_dyn[];
  __shared__ int32_t s_bychar[264];
  __shared__ uint32_t s_rc[9];                  // r_c: reachable characters
  __shared__ uint32_t s_text[9];                // characters that occur in the text (a range stepped with any other is empty)
  __shared__ int32_t s_q;
  const int nn = B.lds_nodes, cc = B.lds_children;
  uint32_t* const s_tmp = reinterpret_cast<uint32_t*>(s_dyn);                    // tmp_states being accumulated (atomicMin)
  int64_t* const s_child_f = reinterpret_cast<int64_t*>(s_tmp + nn);
  int64_t* const s_child_l = s_child_f + cc;
  int32_t* const s_child_found = reinterpret_cast<int32_t*>(s_child_l + cc);
  int32_t* const s_live = s_child_found + cc;
  uint32_t* const s_child_h = reinterpret_cast<uint32_t*>(s_live + cc);          // the child's hash bucket,
  int32_t* const s_child_head = reinterpret_cast<int32_t*>(s_child_h + cc);      // that bucket's head as the fan-out saw it,
  int32_t* const s_child_slot = s_child_head + cc;                               // and the slot the child was pushed into (-1: merged / not yet)
  uint16_t* const s_child_ch = reinterpret_cast<uint16_t*>(s_child_slot + cc);
  uint8_t* const s_cur = reinterpret_cast<uint8_t*>(s_child_ch + cc);            // nfa_states: the popped entry's costs, deletions merged in
  uint8_t* const s_sub = s_cur + nn;                                             // states after one substitution error (any character)
  uint8_t* const s_top = s_sub + nn;                                             // the costs of the entry on top of the stack (see "top")
  __shared__ uint8_t s_pend[kNfaPend];                                           // pending entries per hash bucket mod kNfaPend (see "pend")
  uint32_t* const s_ent_sd = reinterpret_cast<uint32_t*>(s_dyn + ((nfa_lds_bytes(nn, cc, 0) + 3) & ~size_t(3)));   // kLds: the automaton's transitions ...
  uint16_t* const s_ent_ch = reinterpret_cast<uint16_t*>(s_ent_sd + B.lds_ents);
  uint8_t* const s_flags = reinterpret_cast<uint8_t*>(s_ent_ch + B.lds_ents);                                       // ... and node flags
  const int t = threadIdx.x;
  uint8_t* const arena = B.arena + size_t(blockIdx.x) * size_t(B.arena_bytes);
  const int cap = B.cap;
  const size_t stride = size_t(B.cost_stride);
  int64_t* const e_first = reinterpret_cast<int64_t*>(arena);
  int64_t* const e_last = e_first + cap;
  int32_t* const e_len = reinterpret_cast<int32_t*>(e_last + cap);
  // add_mapping's "is this range already pending?" (server.c:1558-1620; the reference keeps a hash on (first,last),
  // queue_map.c): chained hashing over the stack -- heads[h] = the newest pending entry of bucket h, e_next[] the one before
  // it.  The stack is LIFO, so the entry being popped is always the head of its chain: removal is one store, no tombstones.
  // (Until round 4 every pop scanned the whole stack for every child: sp x children / 64 steps, minutes for a pattern like
  // `.*` with approximate matching.)
  int32_t* const e_next = e_len + cap;
  int32_t* const heads = e_next + cap;
  const uint32_t hmask = uint32_t(B.hash_size) - 1u;
  uint8_t* const e_cost = reinterpret_cast<uint8_t*>(heads + B.hash_size);
  auto hash_of = [&](int64_t f, int64_t l) -> uint32_t {
    uint64_t x = uint64_t(f) * 0x9e3779b97f4a7c15ull ^ (uint64_t(l) + 0x7f4a7c15ull) * 0xff51afd7ed558ccdull;
    x ^= x >> 29;
    return uint32_t(x) & hmask;
  };
  // The characters of the text, once per workgroup: stepping a range with a character the text lacks gives the empty range
  // (Occ == 0), which add_mapping ignores (server.c:1565) -- such characters never become children, so an APPROX search on DNA
  // fans out over 4 characters per pop, not over the alphabet's 256 (five rounds of the fan-out and a 256-iteration child
  // loop per pop until round 5).  The ORDER of the children that remain is untouched.
  if (t < 9) s_text[t] = 0;
  __syncthreads();
  #pragma unroll 1
  for (int c = t; c < kAlphaSize; c += 64)
    if (P::code_of(ix, uint32_t(c)) != 0xffffu) atomicOr(&s_text[c >> 5], 1u << (c & 31));
  __syncthreads();
  for (;;) {
    if (t == 0) s_q = atomicAdd(B.next, 1);
    __syncthreads();
    const int qi = uni(s_q);
    __syncthreads();
    if (qi >= B.nq) break;
    const int q = B.order ? B.order[qi] : qi;
    const NfaQueryDev Q = B.queries[q];
    const int N = Q.num_nodes, T = Q.num_ents, bound = Q.cost_bound;
    const bool approx = bound > 1;
    const uint8_t* const g_flags = B.node_flags + Q.node_off;
    const uint32_t* const g_ent_sd = B.ent_sd + Q.ent_off;
    const uint16_t* const g_ent_ch = B.ent_ch + Q.ent_off;
    if constexpr (kLds) {
      #pragma unroll 1
      for (int e = t; e < T; e += 64) {
        s_ent_sd[e] = g_ent_sd[e];
        s_ent_ch[e] = g_ent_ch[e];
      }
      #pragma unroll 1
      for (int i = t; i < N; i += 64) s_flags[i] = g_flags[i];
    }
    auto ENT_SD = [&](int e) -> uint32_t { if constexpr (kLds) return s_ent_sd[e]; else return g_ent_sd[e]; };
    auto ENT_CH = [&](int e) -> uint32_t { if constexpr (kLds) return s_ent_ch[e]; else return g_ent_ch[e]; };
    auto FLAGS = [&](int i) -> uint32_t { if constexpr (kLds) return s_flags[i]; else return g_flags[i]; };
    #pragma unroll 1
    for (int i = t; i < 262; i += 64) s_bychar[i] = B.bychar[Q.bychar_off + i];
    // the initial mapping: the whole index -> the start states (server.c:1786-1812)
    #pragma unroll 1
    for (int i = t; i < N; i += 64) e_cost[i] = (g_flags[i] & 1u) ? 0 : kNfaDead;
    #pragma unroll 1
    for (int i = t; i < B.hash_size; i += 64) heads[i] = -1;
    #pragma unroll 1
    for (int i = t; i < kNfaPend / 4; i += 64) reinterpret_cast<uint32_t*>(s_pend)[i] = 0;
    __syncthreads();
    if (t == 0) {
      e_first[0] = 0;
      e_last[0] = ix.total_length - 1;
      e_len[0] = 0;
      e_next[0] = -1;
      const uint32_t h0 = hash_of(0, ix.total_length - 1);
      heads[h0] = 0;
      s_pend[h0 & (kNfaPend - 1)] = 1;
    }
    int sp = 1, status = 0;
    int64_t iters = 0;
    // "top": the entry pushed LAST by a pop is the one the next pop takes.  Its range, match length and costs stay in
    // registers / LDS (s_top) and it never reaches the arena, the hash or the counters: nothing can look it up or merge into
    // it before it is popped (lookups happen in the NEXT pop's fan-out, after it is gone).  A pop of it costs no memory round
    // trip; a search that goes down a path pops such entries most of the time.
    // "pend": s_pend[b] counts the pending (arena) entries whose hash bucket is b mod kNfaPend (saturating at 255, then never
    // decremented).  Zero means the bucket's chain is empty: the fan-out then knows "not pending, head = -1" without
    // touching the arena -- the lookup that was one dependent round trip per pop.
    int top_slot = -1, top_len = 0;
    int64_t top_f = 0, top_l = 0;
    __syncthreads();
    for (;;) {
      if (iters > B.max_iterations) { status = kNfaStatusOverworked; break; }   // server.c:1821
      if (sp == 0) break;
      sp--;
      const bool from_top = sp == top_slot;
      top_slot = -1;
      int64_t first, last;
      int len;
      if (from_top) {
        first = top_f;
        last = top_l;
        len = top_len;
      } else {
        first = uni64(e_first[sp]);
        last = uni64(e_last[sp]);
        len = uni(e_len[sp]);
        if (t == 0) {      // the top of the stack is the head of its chain
          const uint32_t h = hash_of(first, last);
          heads[h] = e_next[sp];
          const uint8_t pc = s_pend[h & (kNfaPend - 1)];
          if (pc != 255) s_pend[h & (kNfaPend - 1)] = uint8_t(pc - 1);
        }
      }
      // ---- a final state alive: a result, not extended (approx_is_final_state: the first such node's cost)
      int fin = INT_MAX;
      #pragma unroll 1
      for (int i = t; i < N; i += 64) {
        const uint8_t c = from_top ? s_top[i] : e_cost[size_t(sp) * stride + i];
        s_cur[i] = c;
        if (int(c) < bound && (FLAGS(i) & 2u) && i < fin) fin = i;
      }
      fin = uni(wave_min_i32(fin));
      __syncthreads();
      if (fin != INT_MAX) {
        if (t == 0) {
          const unsigned long long slot = atomicAdd(B.result_count, 1ull);
          if (int64_t(slot) < B.result_cap) B.results[slot] = NfaResultDev{first, last, q, len, int(s_cur[fin]), B.pass};
        }
        __syncthreads();
        continue;
      }
      const int base5 = s_bychar[kNfaOffset];    // entries of characters >= CHARACTER_OFFSET start here
      // ---- deletions: states after reading ANY character at delete_cost, merged in (server.c:1854-1863)
      if (approx) {
        #pragma unroll 1
        for (int i = t; i < N; i += 64) s_tmp[i] = kNfaDead;
        __syncthreads();
        #pragma unroll 1
        for (int e = base5 + t; e < T; e += 64) {
          const uint32_t sd = ENT_SD(e);
          const int c = int(s_cur[sd & 0xffffu]) + Q.del;
          if (c < bound) atomicMin(&s_tmp[sd >> 16], uint32_t(c));
        }
        __syncthreads();
        #pragma unroll 1
        for (int i = t; i < N; i += 64) {
          const uint32_t v = s_tmp[i];
          if (v < uint32_t(s_cur[i])) s_cur[i] = uint8_t(v);
        }
        __syncthreads();
      }
      // ---- may any character be an error from here?  (nfa_errcnt_t arithmetic: one byte, as the reference computes it)
      int m = bound;
      #pragma unroll 1
      for (int i = t; i < N; i += 64) m = int(s_cur[i]) < m ? int(s_cur[i]) : m;
      m = uni(wave_min_i32(m));
      const int ms = m + Q.subst, mi = m + Q.ins;
      const int min_err = (ms < mi ? ms : mi) & 0xff;
      const bool allchars = min_err < bound && iters > 0;
      // ---- r_c: the characters an alive state can read
      if (t < 9) s_rc[t] = 0;
      __syncthreads();
      #pragma unroll 1
      for (int e = t; e < T; e += 64)
        if (int(s_cur[ENT_SD(e) & 0xffffu]) < bound) {
          const uint32_t ch = ENT_CH(e);
          atomicOr(&s_rc[ch >> 5], 1u << (ch & 31u));
        }
      __syncthreads();
      if (t < 9) {
        uint32_t w = s_rc[t];
        if (allchars) {               // CHARACTER_OFFSET .. ALPHA_SIZE - 1
          uint32_t mask = ~0u;
          if (t == 0) mask = ~0u << kNfaOffset;
          if (t == 8) mask = (1u << (kAlphaSize - 256)) - 1u;
          w |= mask;
        }
        s_rc[t] = w & s_text[t];      // ... that the text holds
      }
      __syncthreads();
      // ---- the children in push order: characters >= CHARACTER_OFFSET ascending (the substitution loop creates their
      // entries first, server.c:2114-2120), then the characters below it (regular loop, :2123-2130)
      int n_all = 0, n_low;
      {
        const uint32_t w = t < 9 ? s_rc[t] : 0u;
        n_all = uni(wave_sum_i32(__popc(w)));
        n_low = uni(__popc(s_rc[0] & ((1u << kNfaOffset) - 1u)));
      }
      const int nchild = n_all;       // <= characters of the text <= lds_children
      #pragma unroll 1
      for (int c = t; c < kAlphaSize; c += 64) {
        if (!((s_rc[c >> 5] >> (c & 31)) & 1u)) continue;
        int rank = __popc(s_rc[c >> 5] & ((1u << (c & 31)) - 1u));
        for (int w = 0; w < (c >> 5); w++) rank += __popc(s_rc[w]);
        const int pos = c >= kNfaOffset ? rank - n_low : (n_all - n_low) + rank;
        s_child_ch[pos] = uint16_t(c);
      }
      __syncthreads();
      // ---- the fan-out: lane k steps the range with the k-th character (server.c:1954-2060); then add_mapping's lookup: a
      // pending entry with the child's range?  (children of one pop have disjoint ranges); the children with rows are listed
      // in order (s_live): the merge loop below visits only those
      int nlive = 0;
      for (int k0 = 0; k0 < nchild; k0 += 64) {
        const int k = k0 + t;
        bool live = false;
        if (k < nchild) {
          const uint32_t ch = s_child_ch[k];
          int64_t f = first, l = last;
          P::search_step(ix, 1, P::code_of(ix, ch), f, l);
          s_child_f[k] = f;
          s_child_l[k] = l;
          int found = -1, head = -1;
          live = l >= f;                               // add_mapping ignores empty ranges (server.c:1565)
          if (live) {
            const uint32_t h = hash_of(f, l);
            s_child_h[k] = h;
            if (s_pend[h & (kNfaPend - 1)] != 0) {     // ("pend": an empty bucket needs no look at the arena)
              head = heads[h];
              for (int s2 = head; s2 >= 0; s2 = e_next[s2])
                if (e_first[s2] == f && e_last[s2] == l) { found = s2; break; }
            }
            s_child_head[k] = head;
            s_child_slot[k] = -1;
          }
          s_child_found[k] = found;
        }
        const unsigned long long mask = __ballot(live);
        if (live) s_live[nlive + __popcll(mask & ((1ull << t) - 1ull))] = k;
        nlive += __popcll(mask);
      }
      // the last NEW child in push order becomes the "top" (it is the next entry popped)
      int last_new = -1;
      {
        __syncthreads();
        for (int j0 = 0; j0 < nlive; j0 += 64) {
          const int j = j0 + t;
          const bool isnew = j < nlive && s_child_found[s_live[j]] < 0;
          const unsigned long long nm = __ballot(isnew);
          if (nm) last_new = j0 + 63 - __builtin_clzll(nm);
        }
      }
      // ---- substitutions: states after reading any character at subst_cost (server.c:2107-2110)
      if (approx) {
        #pragma unroll 1
        for (int i = t; i < N; i += 64) s_tmp[i] = kNfaDead;
        __syncthreads();
        #pragma unroll 1
        for (int e = base5 + t; e < T; e += 64) {
          const uint32_t sd = ENT_SD(e);
          const int c = int(s_cur[sd & 0xffffu]) + Q.subst;
          if (c < bound) atomicMin(&s_tmp[sd >> 16], uint32_t(c));
        }
        __syncthreads();
        #pragma unroll 1
        for (int i = t; i < N; i += 64) s_sub[i] = uint8_t(s_tmp[i]);
      }
      __syncthreads();
      for (int kk = 0; kk < nlive; kk++) {
        const int k = uni(s_live[kk]);
        const int64_t cf = uni64(s_child_f[k]), cl = uni64(s_child_l[k]);
        const int ch = uni(s_child_ch[k]);
        #pragma unroll 1
        for (int i = t; i < N; i += 64) s_tmp[i] = kNfaDead;
        __syncthreads();
        #pragma unroll 1
        for (int e = s_bychar[ch] + t; e < s_bychar[ch + 1]; e += 64) {    // approx_get_reachable_states(ch)
          const uint32_t sd = ENT_SD(e);
          const uint32_t c = s_cur[sd & 0xffffu];
          if (int(c) < bound) atomicMin(&s_tmp[sd >> 16], c);
        }
        __syncthreads();
        const int found = uni(s_child_found[k]);
        if (found < 0 && sp >= cap) { status = kNfaStatusFull; break; }
        const int slot = found < 0 ? sp : found;
        const bool to_top = kk == last_new;                                     // ("top": registers and s_top instead of the arena)
        uint8_t* const dst = e_cost + size_t(slot) * stride;
        #pragma unroll 1
        for (int i = t; i < N; i += 64) {
          uint32_t v = s_tmp[i];
          if (approx) {
            const uint32_t ins = uint32_t(s_cur[i]) + uint32_t(Q.ins);          // approx_add_error_allchars (nfa.c:305)
            v = ins < v ? ins : v;
            if (ch >= kNfaOffset && uint32_t(s_sub[i]) < v) v = s_sub[i];
          }
          if (int(v) >= bound) v = kNfaDead;                                    // beyond the bound is dead, whatever the number
          if (found >= 0) {
            const uint32_t old = dst[i];
            v = old < v ? old : v;                                              // nfa_states_union
          }
          if (to_top) s_top[i] = uint8_t(v);
          else dst[i] = uint8_t(v);
        }
        if (to_top) {
          top_slot = slot;
          top_f = cf;
          top_l = cl;
          top_len = len + 1;
        } else if (found < 0) {
          // the bucket's head NOW: an earlier new child of this pop in the same bucket, else what the fan-out saw
          const uint32_t h = uni(int(s_child_h[k]));
          int cur_head = uni(s_child_head[k]);
          for (int j0 = 0; j0 < kk; j0 += 64) {
            const int j = j0 + t;
            int sl = -1;
            if (j < kk) {
              const int k2 = s_live[j];
              if (s_child_h[k2] == h) sl = s_child_slot[k2];
            }
            const unsigned long long hm = __ballot(sl >= 0);
            if (hm) cur_head = __shfl(sl, 63 - __builtin_clzll(hm), 64);
          }
          if (t == 0) {
            e_first[slot] = cf;
            e_last[slot] = cl;
            e_len[slot] = len + 1;
            e_next[slot] = cur_head;
            heads[h] = slot;
            s_child_slot[k] = slot;
            const uint8_t pc = s_pend[h & (kNfaPend - 1)];
            if (pc != 255) s_pend[h & (kNfaPend - 1)] = uint8_t(pc + 1);
          }
        } else if (t == 0 && len + 1 > e_len[slot]) {
          e_len[slot] = len + 1;                                                // the longer match is kept (server.c:1611-1619)
        }
        if (found < 0) sp++;
        __syncthreads();
      }
      if (status) break;
      iters++;
    }
    if (t == 0) {
      B.status[q] = status;
      if (B.iters_out) B.iters_out[q] = int32_t(iters < INT_MAX ? iters : INT_MAX);
    }
    __syncthreads();
  }
}

}  // namespace femto_amd

namespace {

struct NfaHostResult { int64_t first, last; int32_t len, cost; int64_t seq; };

int validate_nfa(const femto_amd_nfa_t& a, int64_t qi) {
  auto bad = [&](const char* what) { return set_err(FEMTO_AMD_ERR_PARAM, "automaton " + std::to_string(qi) + ": " + what); };
  if (a.num_nodes < 1 || a.num_nodes > kNfaMaxNodes) return bad("1 <= num_nodes <= 2048");
  if (a.num_transitions < 0 || int64_t(a.num_transitions) > kNfaMaxTransitions) return bad("too many transitions");
  if (!a.trans_start || !a.is_start || !a.is_final || (a.num_transitions && (!a.trans_char || !a.trans_dest))) return bad("null array");
  // regexp_settings_t as compile_regexp_from_ast accepts them (src/main/compile_regexp.c:673-685); errors are counted in one byte
  if (a.cost_bound < 1 || a.cost_bound > kNfaDead) return bad("1 <= cost_bound <= 255");
  if (a.subst_cost < 1 || a.subst_cost > kNfaDead || a.delete_cost < 1 || a.delete_cost > kNfaDead || a.insert_cost < 1 || a.insert_cost > kNfaDead)
    return bad("1 <= subst_cost, delete_cost, insert_cost <= 255");
  if (a.trans_start[0] != 0 || a.trans_start[a.num_nodes] != a.num_transitions) return bad("trans_start does not span the transitions");
  for (int i = 0; i < a.num_nodes; i++)
    if (a.trans_start[i + 1] < a.trans_start[i]) return bad("trans_start is not monotone");
  for (int e = 0; e < a.num_transitions; e++) {
    if (a.trans_char[e] < 0 || a.trans_char[e] >= kAlphaSize) return bad("transition character >= ALPHA_SIZE (261)");
    if (a.trans_dest[e] < 0 || a.trans_dest[e] >= a.num_nodes) return bad("transition destination out of range");
  }
  return 0;
}

template <class P>
void launch_nfa(const DevIndex& d, const NfaBatchDev& B, int blocks, size_t lds, hipStream_t st) {
  if (B.lds_ents) hipLaunchKernelGGL((nfa_search_kernel<P, true>), dim3(uint32_t(blocks)), dim3(64), lds, st, d, B);
  else hipLaunchKernelGGL((nfa_search_kernel<P, false>), dim3(uint32_t(blocks)), dim3(64), lds, st, d, B);
}
// workgroups of 64 lanes a CU holds at once with `lds` bytes of dynamic LDS each (the grid is sized to fill the chip once:
// the workgroups take automata from a counter)
template <class P>
int nfa_blocks_per_cu(size_t lds, bool lds_ents) {
  int n = 0;
  const hipError_t e = lds_ents ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, nfa_search_kernel<P, true>, 64, lds)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, nfa_search_kernel<P, false>, 64, lds);
  if (e != hipSuccess || n < 1) n = 8;
  return n;
}

}  // namespace

// regexp_result_list_sort (src/main/server.c:1528-1573): sort by first ascending, last descending; drop equal ranges (the one
// appended first stays: glibc's qsort is a stable merge sort) and ranges inside the last kept one
static void sort_results(std::vector<NfaHostResult>& r) {
  std::stable_sort(r.begin(), r.end(), [](const NfaHostResult& a, const NfaHostResult& b) {
    if (a.first != b.first) return a.first < b.first;
    if (a.last != b.last) return a.last > b.last;
    return a.seq < b.seq;
  });
  size_t n = 0;
  for (size_t k = 0; k < r.size(); k++) {
    if (n && r[k].first == r[n - 1].first && r[k].last == r[n - 1].last) continue;
    r[n++] = r[k];
  }
  r.resize(n);
  if (r.empty()) return;
  int64_t first = r[0].first, last = r[0].last;
  size_t i = 1;
  for (size_t k = 1; k < r.size(); k++) {
    if (r[k].first >= first && r[k].last <= last) continue;
    first = r[k].first;
    last = r[k].last;
    r[i++] = r[k];
  }
  r.resize(i);
}

int femto_amd_nfa_search_batch(femto_amd_index_t* ix, int64_t nq, const femto_amd_nfa_t* nfas, int64_t max_results,
                               int64_t* result_start, int64_t* first_out, int64_t* last_out, int32_t* len_out, int32_t* cost_out,
                               int32_t* status_out, int64_t* n_out) {
  API_BEGIN
  if (!ix || nq < 0 || nq > INT_MAX / 2 || (nq && (!nfas || !result_start)) || max_results < 0 || !n_out ||
      (max_results && (!first_out || !last_out)))
    return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (!ix->children.empty())
    return femto_amd_nfa_search_batch(ix->children[0], nq, nfas, max_results, result_start, first_out, last_out, len_out, cost_out, status_out, n_out);
  *n_out = 0;
  int rc;
  for (int64_t qi = 0; qi < nq; qi++)       // malformed automata are refused before anything touches the device
    if ((rc = validate_nfa(nfas[qi], qi))) return rc;
  if ((rc = ensure_device(ix))) return rc;
  if (nq == 0) return FEMTO_AMD_OK;
  const int mode = ix->mode;
  if (mode != 3 && mode != 4 && !ix->host.dir_regular)
    return set_err(FEMTO_AMD_ERR_INVALID, "regular-expression search needs the derived segment lines");
  // ---- the automata, flat: transitions sorted by character (reading ch touches only its own entries)
  std::vector<NfaQueryDev> hq(static_cast<size_t>(nq));
  std::vector<uint8_t> h_flags;
  std::vector<uint32_t> h_sd;
  std::vector<uint16_t> h_ch;
  std::vector<int32_t> h_bychar(size_t(nq) * 262);
  int max_nodes = 1;
  for (int64_t qi = 0; qi < nq; qi++) {
    const femto_amd_nfa_t& a = nfas[qi];
    NfaQueryDev& Q = hq[size_t(qi)];
    Q.node_off = int64_t(h_flags.size());
    Q.ent_off = int64_t(h_sd.size());
    Q.bychar_off = qi * 262;
    Q.num_nodes = a.num_nodes;
    Q.num_ents = a.num_transitions;
    Q.cost_bound = a.cost_bound;
    Q.subst = a.subst_cost;
    Q.del = a.delete_cost;
    Q.ins = a.insert_cost;
    max_nodes = std::max(max_nodes, int(a.num_nodes));
    for (int i = 0; i < a.num_nodes; i++) h_flags.push_back(uint8_t((a.is_start[i] ? 1 : 0) | (a.is_final[i] ? 2 : 0)));
    int32_t* bc = h_bychar.data() + qi * 262;
    std::fill(bc, bc + 262, 0);
    for (int e = 0; e < a.num_transitions; e++) bc[a.trans_char[e] + 1]++;
    for (int c = 0; c < 261; c++) bc[c + 1] += bc[c];
    std::vector<int32_t> fill(bc, bc + 261);
    const size_t base = h_sd.size();
    h_sd.resize(base + size_t(a.num_transitions));
    h_ch.resize(base + size_t(a.num_transitions));
    for (int i = 0; i < a.num_nodes; i++)
      for (int e = a.trans_start[i]; e < a.trans_start[i + 1]; e++) {
        const int c = a.trans_char[e];
        const size_t at = base + size_t(fill[size_t(c)]++);
        h_sd[at] = uint32_t(i) | (uint32_t(a.trans_dest[e]) << 16);
        h_ch[at] = uint16_t(c);
      }
  }
  Lease L(ix);
  if (!L.s) return L.rc;
  hipStream_t st = L.s->stream;
  DeviceBuffer d_q, d_flags, d_sd, d_ch, d_bychar, d_arena, d_results, d_misc, d_order;
  struct Free {
    std::vector<DeviceBuffer*> b;
    ~Free() { for (DeviceBuffer* x : b) x->release(); }
  } guard{{&d_q, &d_flags, &d_sd, &d_ch, &d_bychar, &d_arena, &d_results, &d_misc, &d_order}};
  // Raw results (before the per-automaton sort drops ranges inside other results, and including the attempts of searches
  // that are run again in a larger arena) land in a buffer of the library's own, which grows and the batch runs again
  // when it is too small: max_results bounds what the CALLER's arrays receive, nothing else (max_results == 0: count only).
  // (the first size is a guess, not a promise: a generous max_results does not reserve memory up front)
  int64_t result_cap = std::max<int64_t>(std::min<int64_t>(2 * max_results, int64_t(1) << 22), 1 << 12);
  if ((rc = d_q.reserve(hq.size() * sizeof(NfaQueryDev))) || (rc = d_flags.reserve(h_flags.size() + 16)) ||
      (rc = d_sd.reserve(h_sd.size() * 4 + 16)) || (rc = d_ch.reserve(h_ch.size() * 2 + 16)) || (rc = d_bychar.reserve(h_bychar.size() * 4)) ||
      (rc = d_misc.reserve(64 + size_t(nq) * 8)) ||
      (rc = d_order.reserve(size_t(nq) * 4)))
    return rc;
  HIP_TRY(hipMemcpyAsync(d_q.p, hq.data(), hq.size() * sizeof(NfaQueryDev), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_flags.p, h_flags.data(), h_flags.size(), hipMemcpyHostToDevice, st));
  if (!h_sd.empty()) {
    HIP_TRY(hipMemcpyAsync(d_sd.p, h_sd.data(), h_sd.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_ch.p, h_ch.data(), h_ch.size() * 2, hipMemcpyHostToDevice, st));
  }
  HIP_TRY(hipMemcpyAsync(d_bychar.p, h_bychar.data(), h_bychar.size() * 4, hipMemcpyHostToDevice, st));
  // d_misc: [0] work counter (i32), [8] result count (u64), [64...] status per query
  int32_t* d_next = d_misc.as<int32_t>();
  unsigned long long* d_count = reinterpret_cast<unsigned long long*>(static_cast<char*>(d_misc.p) + 8);
  int32_t* d_status = reinterpret_cast<int32_t*>(static_cast<char*>(d_misc.p) + 64);
  NfaBatchDev B{};
  B.queries = d_q.as<NfaQueryDev>();
  B.node_flags = d_flags.as<uint8_t>();
  B.ent_sd = d_sd.as<uint32_t>();
  B.ent_ch = d_ch.as<uint16_t>();
  B.bychar = d_bychar.as<int32_t>();
  B.next = d_next;
  B.result_count = d_count;
  B.status = d_status;
  const bool want_stats = getenv("FEMTO_AMD_NFA_STATS") != nullptr;
  B.iters_out = want_stats ? d_status + nq : nullptr;
  B.max_iterations = ix->regexp_max_iterations;
  B.cost_stride = (max_nodes + 3) & ~3;
  B.lds_nodes = (max_nodes + 7) & ~7;
  {
    int nchars = 0;       // characters the text holds: the most children a pop can have
    for (int c = 0; c < kAlphaSize && size_t(c) + 1 < ix->host.C.size(); c++)
      if (ix->host.C[size_t(c) + 1] > ix->host.C[size_t(c)]) nchars++;
    B.lds_children = (std::max(nchars, 4) + 3) & ~3;
  }
  {
    size_t max_ents = 0;
    for (const NfaQueryDev& Q : hq) max_ents = std::max(max_ents, size_t(Q.num_ents));
    B.lds_ents = max_ents <= size_t(kNfaLdsEnts) ? int32_t((std::max<size_t>(max_ents, 2) + 1) & ~size_t(1)) : 0;
  }
  const size_t lds = nfa_lds_bytes(B.lds_nodes, B.lds_children, B.lds_ents);
  int per_cu = 8;
  if (mode == 3 && ix->dev.ru && ix->dev.ru_marks) per_cu = nfa_blocks_per_cu<RumPolicy>(lds, B.lds_ents != 0);
  else if (mode == 3 && ix->dev.ru) per_cu = nfa_blocks_per_cu<RuPolicy>(lds, B.lds_ents != 0);
  else if (mode == 3) per_cu = nfa_blocks_per_cu<PackPolicy>(lds, B.lds_ents != 0);
  else if (mode == 4 && ix->dev.ind) per_cu = nfa_blocks_per_cu<IndPolicy>(lds, B.lds_ents != 0);
  else if (mode == 4) per_cu = nfa_blocks_per_cu<Pack2Policy>(lds, B.lds_ents != 0);
  else per_cu = nfa_blocks_per_cu<WavePolicy>(lds, B.lds_ents != 0);
  per_cu = std::min(per_cu, 32);
  // Stack capacity: most searches keep a few dozen pending entries; the ones that run out (status FULL) are run again
  // with a larger arena and fewer workgroups, up to regexp_stack_cap entries.
  const size_t entry_bytes = 24 + size_t(B.cost_stride);      // first, last, match length, hash link, costs
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t budget = std::min<size_t>(free_b / 2, size_t(16) << 30);
  std::vector<int32_t> todo, status(static_cast<size_t>(nq), 0), last_pass(static_cast<size_t>(nq), 0);
  unsigned long long count = 0;
  for (int attempt = 0;; attempt++) {
  if ((rc = d_results.reserve(size_t(result_cap) * sizeof(NfaResultDev)))) return rc;
  B.results = d_results.as<NfaResultDev>();
  B.result_cap = result_cap;
  HIP_TRY(hipMemsetAsync(d_misc.p, 0, 64 + size_t(nq) * 8, st));
  todo.resize(static_cast<size_t>(nq));
  for (int64_t i = 0; i < nq; i++) todo[size_t(i)] = int32_t(i);
  int64_t cap = std::min<int64_t>(1024, ix->regexp_stack_cap);
  for (int pass = 0;; pass++) {
    int blocks = int(std::min<int64_t>(int64_t(todo.size()), int64_t(ix->num_cus) * per_cu));
    int64_t hsize = 64;
    while (hsize < 2 * cap) hsize <<= 1;
    const size_t per_block = (size_t(cap) * entry_bytes + size_t(hsize) * 4 + 255) & ~size_t(255);
    B.hash_size = int32_t(hsize);
    if (size_t(blocks) * per_block > budget) blocks = int(std::max<size_t>(1, budget / per_block));
    if ((rc = d_arena.reserve(size_t(blocks) * per_block))) return rc;
    B.arena = d_arena.as<uint8_t>();
    B.arena_bytes = int64_t(per_block);
    B.cap = int32_t(cap);
    B.nq = int32_t(todo.size());
    B.order = d_order.as<int32_t>();
    B.pass = pass;
    for (int32_t q : todo) last_pass[size_t(q)] = pass;
    HIP_TRY(hipMemcpyAsync(d_order.p, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_next, 0, 4, st));
    hipEvent_t te0 = nullptr, te1 = nullptr;
    const bool timed = timer_begin(ix, ix->t_regexp, st, &te0, &te1);
    if (mode == 3 && ix->dev.ru && ix->dev.ru_marks) launch_nfa<RumPolicy>(ix->dev, B, blocks, lds, st);
    else if (mode == 3 && ix->dev.ru) launch_nfa<RuPolicy>(ix->dev, B, blocks, lds, st);
    else if (mode == 3) launch_nfa<PackPolicy>(ix->dev, B, blocks, lds, st);
    else if (mode == 4 && ix->dev.ind) launch_nfa<IndPolicy>(ix->dev, B, blocks, lds, st);
    else if (mode == 4) launch_nfa<Pack2Policy>(ix->dev, B, blocks, lds, st);
    else launch_nfa<WavePolicy>(ix->dev, B, blocks, lds, st);
    if (timed) timer_end(ix, ix->t_regexp, st, te0, te1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(status.data(), d_status, size_t(nq) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (want_stats) {      // entries popped per automaton: the kernel ends with its longest search (profiles/r05_regexp_*)
      std::vector<int32_t> it(static_cast<size_t>(nq));
      HIP_TRY(hipMemcpy(it.data(), d_status + nq, size_t(nq) * 4, hipMemcpyDeviceToHost));
      std::vector<int32_t> v;
      for (int32_t q : todo) v.push_back(it[size_t(q)]);
      std::sort(v.begin(), v.end());
      double sum = 0;
      for (int32_t x : v) sum += x;
      auto pct = [&](double p) { return v.empty() ? 0 : v[std::min(v.size() - 1, size_t(p * double(v.size())))]; };
      fprintf(stderr, "[femto_amd] nfa pass %d: %zu automata, %d workgroups, pops: mean %.0f  p50 %d  p90 %d  p99 %d  p99.9 %d  max %d  (sum %.3g; max / (sum / workgroups) = %.2f)\n",
              pass, v.size(), blocks, v.empty() ? 0.0 : sum / double(v.size()), pct(0.5), pct(0.9), pct(0.99), pct(0.999), v.empty() ? 0 : v.back(), sum,
              sum > 0 ? double(v.empty() ? 0 : v.back()) / (sum / double(blocks)) : 0.0);
    }
    std::vector<int32_t> again;
    for (int32_t q : todo)
      if (status[size_t(q)] == kNfaStatusFull) again.push_back(q);
    if (again.empty() || cap >= ix->regexp_stack_cap) break;
    cap = std::min<int64_t>(cap * 64, ix->regexp_stack_cap);
    todo.swap(again);
  }
  HIP_TRY(hipMemcpy(&count, d_count, 8, hipMemcpyDeviceToHost));
  if (int64_t(count) <= result_cap) break;
  // the raw buffer was too small: once more from the start with room for what this attempt produced (a search's raw
  // count does not depend on the buffer, only on the arena sizes its passes ran with, which repeat)
  result_cap = int64_t(count) + int64_t(count) / 8 + 1024;
  if (attempt >= 3 || size_t(result_cap) * sizeof(NfaResultDev) > budget) {
    *n_out = int64_t(count);
    return set_err(FEMTO_AMD_ERR_FULL, "more raw result ranges than the device buffer may hold (" + std::to_string(count) + ")");
  }
  }
  std::vector<NfaResultDev> raw(static_cast<size_t>(count));
  if (count) HIP_TRY(hipMemcpy(raw.data(), d_results.p, size_t(count) * sizeof(NfaResultDev), hipMemcpyDeviceToHost));
  // a search that was run again appended its earlier attempts' results too: only the last attempt's count
  std::vector<std::vector<NfaHostResult>> per(static_cast<size_t>(nq));
  for (size_t k = 0; k < raw.size(); k++)
    if (raw[k].pass == last_pass[size_t(raw[k].query)])
      per[size_t(raw[k].query)].push_back({raw[k].first, raw[k].last, raw[k].len, raw[k].cost, int64_t(k)});
  int64_t n = 0;
  for (int64_t qi = 0; qi < nq; qi++) {
    std::vector<NfaHostResult>& r = per[size_t(qi)];
    if (status[size_t(qi)] != 0) r.clear();        // the reference returns an error and no results (RETURN_ERROR)
    sort_results(r);
    result_start[qi] = n;
    n += int64_t(r.size());
  }
  result_start[nq] = n;
  *n_out = n;
  if (status_out) std::memcpy(status_out, status.data(), size_t(nq) * 4);
  if (max_results == 0) return FEMTO_AMD_OK;        // count only: *n_out and result_start[] are what a second call needs
  if (n > max_results) return set_err(FEMTO_AMD_ERR_FULL, "more results than max_results: *n_out holds the number to call again with");
  for (int64_t qi = 0; qi < nq; qi++) {
    const std::vector<NfaHostResult>& r = per[size_t(qi)];
    int64_t at = result_start[qi];
    for (const NfaHostResult& x : r) {
      first_out[at] = x.first;
      last_out[at] = x.last;
      if (len_out) len_out[at] = x.len;
      if (cost_out) cost_out[at] = x.cost;
      at++;
    }
  }
  return FEMTO_AMD_OK;
  API_END
}

// ---- regular expressions: pattern text -> automaton (query_parser.hpp, regexp_nfa.hpp) ----------------------------------------
struct femto_amd_regexp {
  NfaDesc desc;
  femto_amd_nfa_t view;
  bool literal = false;                 // simplify_query: the query is one string
  std::vector<uint16_t> literal_syms;
  std::string echo;                     // ast_to_string(ast, 0, 1)
};

static int settings_check(int max_cost, int subst_cost, int delete_cost, int insert_cost) {
  // compile_regexp_from_ast (src/main/compile_regexp.c:673-685): cost_bound = max_cost + 1 (approx_node_new, ast.c:168-191)
  const int bound = max_cost + 1;
  if (max_cost < 0 || bound > kNfaDead || subst_cost < 1 || delete_cost < 1 || insert_cost < 1)
    return set_err(FEMTO_AMD_ERR_PARAM, "approximate search: 0 <= max_cost <= 254, costs >= 1");
  if (3 * std::min(subst_cost, kNfaDead) < bound || 3 * std::min(insert_cost, kNfaDead) < bound)
    return set_err(FEMTO_AMD_ERR_PARAM, "approximate search: three substitutions or insertions are not allowed (3 * cost >= max_cost + 1)");
  return 0;
}

// tree -> automaton + the handle's views; the tree's own APPROX settings are already in q
static int finish_regexp(QRegexp& q, femto_amd_regexp_t** out) {
  int rc = settings_check(q.cost_bound - 1, q.subst_cost, q.delete_cost, q.insert_cost);
  if (rc) return rc;
  std::unique_ptr<femto_amd_regexp> r(new femto_amd_regexp());
  r->literal = q_simple(q, &r->literal_syms);
  if (!r->literal) r->literal_syms.clear();
  q_echo(q, r->echo, true);
  RegexNfa nfa;
  std::string perr;
  QueryCompiler comp(&nfa);
  if (!comp.compile(q, &perr)) return set_err(FEMTO_AMD_ERR_PARAM, "regular expression: " + perr);
  if (!build_reversed_nfa(nfa, &r->desc)) return set_err(FEMTO_AMD_ERR_PARAM, "regular expression: too many transitions");
  if (r->desc.num_nodes > kNfaMaxNodes) return set_err(FEMTO_AMD_ERR_PARAM, "regular expression too large");
  r->desc.cost_bound = q.cost_bound;
  r->desc.subst_cost = std::min(q.subst_cost, kNfaDead);
  r->desc.delete_cost = std::min(q.delete_cost, kNfaDead);
  r->desc.insert_cost = std::min(q.insert_cost, kNfaDead);
  femto_amd_nfa_t& v = r->view;
  v.num_nodes = r->desc.num_nodes;
  v.num_transitions = int32_t(r->desc.trans_char.size());
  v.trans_start = r->desc.trans_start.data();
  v.trans_char = r->desc.trans_char.data();
  v.trans_dest = r->desc.trans_dest.data();
  v.is_start = r->desc.is_start.data();
  v.is_final = r->desc.is_final.data();
  v.cost_bound = r->desc.cost_bound;
  v.subst_cost = r->desc.subst_cost;
  v.delete_cost = r->desc.delete_cost;
  v.insert_cost = r->desc.insert_cost;
  *out = r.release();
  return FEMTO_AMD_OK;
}

int femto_amd_regexp_compile(const uint8_t* regex, int64_t regex_len, int max_cost, int subst_cost, int delete_cost, int insert_cost,
                             femto_amd_regexp_t** out) {
  API_BEGIN
  if (!out || (regex_len && !regex) || regex_len < 0) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  *out = nullptr;
  int rc = settings_check(max_cost, subst_cost, delete_cost, insert_cost);
  if (rc) return rc;
  QRegexp q;
  std::string perr;
  if (!parse_query(regex, regex_len, &q, &perr)) return set_err(FEMTO_AMD_ERR_PARAM, "regular expression: " + perr);
  if (q.cost_bound == 1) {               // no APPROX in the text: the arguments decide
    q.cost_bound = max_cost + 1;
    q.subst_cost = subst_cost;
    q.delete_cost = delete_cost;
    q.insert_cost = insert_cost;
  }
  return finish_regexp(q, out);
  API_END
}

int femto_amd_query_compile(const uint8_t* query, int64_t query_len, int flags, femto_amd_regexp_t** out) {
  API_BEGIN
  if (!out || (query_len && !query) || query_len < 0) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  *out = nullptr;
  QRegexp q;
  std::string perr;
  if (!parse_query(query, query_len, &q, &perr)) return set_err(FEMTO_AMD_ERR_PARAM, "query: " + perr);
  if (!(flags & FEMTO_AMD_QUERY_NO_STREAMLINE)) q_streamline(q);
  q_simplify(q);
  if (flags & FEMTO_AMD_QUERY_ICASE) {
    // search_tool.cc:732-751: the string is extracted (simplify_query) BEFORE icase_ast widens it
    q_icase(q);
  }
  return finish_regexp(q, out);
  API_END
}

int femto_amd_regexp_literal(const femto_amd_regexp_t* r, const uint16_t** syms, int64_t* n) {
  if (!r) return 0;
  if (syms) *syms = r->literal ? r->literal_syms.data() : nullptr;
  if (n) *n = r->literal ? int64_t(r->literal_syms.size()) : 0;
  return r->literal ? 1 : 0;
}

const char* femto_amd_regexp_echo(const femto_amd_regexp_t* r) { return r ? r->echo.c_str() : ""; }

/* test hook (src/main/query_planning_test.c; tests/golden/make_query_golden.py): parse, then -- flags bit 0 streamline_query,
 * bit 1 simplify_query, bit 2 icase_ast, in femto_search's order -- print the tree back as ast_to_string does; usequotes
 * == 2: the PARSED tree in the text form oracle/ref_tool.c `ast` reads (query_parser.hpp q_dump) */
int femto_amd_query_echo(const uint8_t* query, int64_t query_len, int flags, int usequotes, char* out, int64_t cap) {
  try {
    if ((query_len && !query) || query_len < 0 || !out || cap < 1) return -1;
    QRegexp q;
    std::string perr;
    if (!parse_query(query, query_len, &q, &perr)) { set_err(FEMTO_AMD_ERR_PARAM, "query: " + perr); return -1; }
    std::string o;
    if (usequotes == 2) {
      q_dump(q, o);
    } else {
      if (flags & 1) q_streamline(q);
      if (flags & 2) q_simplify(q);
      if (flags & 4) q_icase(q);
      q_echo(q, o, usequotes != 0);
    }
    if (int64_t(o.size()) + 1 > cap) return -1;
    std::memcpy(out, o.c_str(), o.size() + 1);
    return int(o.size());
  } catch (...) {
    return -1;
  }
}

const femto_amd_nfa_t* femto_amd_regexp_nfa(const femto_amd_regexp_t* r) { return r ? &r->view : nullptr; }

void femto_amd_regexp_free(femto_amd_regexp_t* r) { delete r; }

int femto_amd_regexp_search_batch(femto_amd_index_t* ix, int64_t nq, const uint8_t* const* regex, const int64_t* regex_len, int max_cost,
                                  int subst_cost, int delete_cost, int insert_cost, int64_t max_results, int64_t* result_start,
                                  int64_t* first_out, int64_t* last_out, int32_t* len_out, int32_t* cost_out, int32_t* status_out,
                                  int64_t* n_out) {
  API_BEGIN
  if (nq < 0 || (nq && (!regex || !regex_len))) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  struct Owner {
    std::vector<femto_amd_regexp_t*> v;
    ~Owner() { for (auto* r : v) femto_amd_regexp_free(r); }
  } own;
  std::vector<femto_amd_nfa_t> views;
  for (int64_t qi = 0; qi < nq; qi++) {
    femto_amd_regexp_t* r = nullptr;
    int rc = femto_amd_regexp_compile(regex[qi], regex_len[qi], max_cost, subst_cost, delete_cost, insert_cost, &r);
    if (rc) return rc;
    own.v.push_back(r);
    views.push_back(r->view);
  }
  return femto_amd_nfa_search_batch(ix, nq, views.data(), max_results, result_start, first_out, last_out, len_out, cost_out, status_out, n_out);
  API_END
}

int femto_amd_regexp_search_approx(femto_amd_index_t* ix, const uint8_t* regex, int64_t regex_len, int max_cost, int subst_cost,
                                   int delete_cost, int insert_cost, int64_t max_results, int64_t* first_out, int64_t* last_out,
                                   int32_t* len_out, int32_t* cost_out, int64_t* n_out) {
  int64_t start[2] = {0, 0};
  int32_t status = 0;
  const int rc = femto_amd_regexp_search_batch(ix, 1, &regex, &regex_len, max_cost, subst_cost, delete_cost, insert_cost, max_results, start,
                                               first_out, last_out, len_out, cost_out, &status, n_out);
  if (rc) return rc;
  if (status == kNfaStatusOverworked) return set_err(FEMTO_AMD_ERR_OVERWORKED, "regular expression: too much work (more than MAX_REGEXP_ITERATIONS steps)");
  if (status == kNfaStatusFull) return set_err(FEMTO_AMD_ERR_FULL, "regular expression: more pending ranges than the search stack holds");
  return FEMTO_AMD_OK;
}

int femto_amd_regexp_search(femto_amd_index_t* ix, const uint8_t* regex, int64_t regex_len, int64_t max_results, int64_t* first_out,
                            int64_t* last_out, int32_t* len_out, int64_t* n_out) {
  return femto_amd_regexp_search_approx(ix, regex, regex_len, 0, 1, 1, 1, max_results, first_out, last_out, len_out, nullptr, n_out);
}

/* test hook: does the automaton built from `regex` accept exactly the byte string s?  1 yes, 0 no, -1 syntax error */
int femto_amd_regexp_match(const uint8_t* regex, int64_t regex_len, const uint8_t* s, int64_t len) {
  try {
    if ((regex_len && !regex) || (len && !s) || regex_len < 0 || len < 0) return -1;
    QRegexp q;
    std::string perr;
    if (!parse_query(regex, regex_len, &q, &perr)) { set_err(FEMTO_AMD_ERR_PARAM, "regular expression: " + perr); return -1; }
    RegexNfa nfa;
    QueryCompiler comp(&nfa);
    if (!comp.compile(q, &perr)) { set_err(FEMTO_AMD_ERR_PARAM, "regular expression: " + perr); return -1; }
    return nfa_full_match(nfa, s, len) ? 1 : 0;
  } catch (...) {
    return -1;
  }
}
