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
#include <chrono>
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
  int32_t* iters_out;        // NULL, or per query: entries popped | shader cycles / 1024 | start | duration, the last two on the device's
                             // WALL clock in units of 16 ticks (wall_clock64: one constant-rate counter for the whole device -- the shader
                             // clocks of the eight XCDs are not synchronised with each other) (femto_amd_nfa_stats; FEMTO_AMD_NFA_STATS=1 prints)
  const int32_t* active;     // NULL, or a word in pinned host memory: automaton batches in flight on this handle (see "FAIR SHARE")
  int32_t slots;             // workgroups the GPU holds of this kernel at once (what one batch alone is launched with)
  int32_t nq_all;
  int32_t warm;              // 0: do not request the next pop's rank lines ahead (FEMTO_AMD_NFA_WARM=0: experiments)
  int64_t max_iterations;    // MAX_REGEXP_ITERATIONS (src/main/server.c:40)
  int32_t pass;
  int32_t lds_nodes, lds_children;   // sizes of the workgroup's dynamic LDS arrays (nfa_lds_bytes)
  int32_t lds_group;                 // children merged together per round of the child phase (s_tmpg holds their tmp_states)
  int32_t lds_tc;                    // > 0: tmp_states of EVERY character of the text fit s_tmpg (this many characters): the by-character pass
  int32_t lds_ents;                  // > 0: every automaton's transitions and node flags are copied to LDS (room for this many entries)
};

// mode 1: femto's own wavelet tree through the derived segment lines (alphabets of more than 256 characters, range-split
// indexes); "code" is the alpha code itself
struct WavePolicy : NoSpec {
  static constexpr int kNfaWaves = 3;
  static __device__ __forceinline__ void search_step(const DevIndex& ix, int, uint32_t ch, int64_t& f, int64_t& l) {
    const int64_t nf = f == 0 ? ix.C[ch] : c_plus_occ_lane(ix, ch, f - 1);
    const int64_t nl = c_plus_occ_lane(ix, ch, l) - 1;
    f = nf;
    l = nl;
  }
  static __device__ __forceinline__ uint32_t code_of(const DevIndex& ix, uint32_t ch) { return ix.C[ch + 1] > ix.C[ch] ? ch : 0xffffu; }
  static __device__ __forceinline__ uint32_t touch(const DevIndex&, uint32_t, int64_t) { return 0; }
};

// min over the wavefront's 64 lanes by DPP (four steps inside each row of 16 lanes, two row broadcasts; the result lands in
// lane 63): six VALU instructions.  The xor-shuffle form it replaces was six ds_bpermute round trips through the LDS pipe, each
// waited for -- and this kernel's time IS such chains (one wavefront per automaton, every phase of a pop waits for the one before).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_min_step(int v) {
  const int o = __builtin_amdgcn_update_dpp(v, v, kCtrl, kRowMask, 0xf, false);
  return o < v ? o : v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = dpp_min_step<0xb1, 0xf>(v);     // quad_perm [1,0,3,2]
  v = dpp_min_step<0x4e, 0xf>(v);     // quad_perm [2,3,0,1]
  v = dpp_min_step<0x114, 0xf>(v);    // row_shr:4
  v = dpp_min_step<0x118, 0xf>(v);    // row_shr:8   (lane 15 of every row: the row's minimum)
  v = dpp_min_step<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
  v = dpp_min_step<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_readlane(v, 63);
}
// a value every lane of the wavefront holds (read from one address, or the result of a wavefront reduction), moved to a
// scalar register: the search loop below is one wavefront working on ONE automaton, and most of what it carries -- the
// popped range, counts, slots -- is uniform; left in vector registers it cost the kernel its occupancy (105 VGPRs)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
  const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uint64_t(v))))), hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uint64_t(v) >> 32))));
  return int64_t((uint64_t(hi) << 32) | lo);
}
// Dynamic LDS of one workgroup (sized per launch from the LARGEST automaton of the batch and the number of characters the
// text holds, so that typical automata -- a few dozen nodes on a small alphabet -- leave room for a CU's full complement of
// wavefronts; until round 5 the arrays were static for 2048 nodes and 264 children: 19 KB, 8 workgroups per CU):
//   s_tmp u32[nn] | s_child_f i64[cc] | s_child_l i64[cc] | s_child_found i32[cc] | s_lv_ch u16[cc] (+ room) | s_child_h u32[cc] |
//   s_child_head i32[cc] | s_child_slot i32[cc] | s_child_ch u16[cc] | s_cur u8[nn] | s_sub u8[nn] | s_top u8[nn]
//                                           nn = lds_nodes (multiple of 8), cc = lds_children (multiple of 4)
//   kLds (the batch's largest automaton has at most kNfaLdsEnts transitions): + s_ent_sd u32[ne] | s_ent_ch u16[ne] | s_flags u8[nn] --
//   the automaton itself.  A pop reads the transition list five or six times (deletions, reachable characters,
//   substitutions, one slice per live child) and the node flags once; from global memory each of those is a dependent
//   round trip of a kernel that waits two thirds of its cycles (profiles/r05_regexp_stats.txt: SQ_WAIT_ANY 68 % of the
//   wave cycles at 4 waves per SIMD); the copy is made once per automaton.
constexpr int kNfaLdsEnts = 2048;
constexpr int kNfaPend = 256;             // counters of pending entries by hash bucket (mod this), in LDS
__host__ __device__ inline size_t nfa_lds_bytes(int nn, int cc, int ne, int group) {
  return size_t(nn) * 4 + size_t(cc) * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 2) + size_t(nn) * 3 + 16 + size_t(ne) * 6 + (ne ? size_t(nn) + 16 : 0) +
         (group ? size_t(group) * size_t(nn) * 4 + 16 : 0);
}

template <class P, bool kLds>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(P::kNfaWaves, P::kNfaWaves))) void nfa_search_kernel(const DevIndex ix, const NfaBatchDev B) {
  extern __shared__ __align__(16) uint8_t s_dyn[];
  __shared__ int32_t s_bychar[264];
  __shared__ uint32_t s_text[9];                // characters that occur in the text (a range stepped with any other is empty)
  __shared__ uint16_t s_code[264];              // ... and their codes in the index's rank layout (P::code_of, read from memory once)
  __shared__ uint16_t s_textch[264];            // the text's characters as a list, ascending
  __shared__ uint8_t s_alive[264];              // per character: an alive state can read it (set and cleared within a pop)
  __shared__ uint8_t s_textpos[264];            // per character: its place in s_textch, or 255 (the by-character pass: B.lds_tc)
  __shared__ int32_t s_q;
  const int nn = B.lds_nodes, cc = B.lds_children;
  uint32_t* const s_tmp = reinterpret_cast<uint32_t*>(s_dyn);                    // tmp_states being accumulated (atomicMin)
  int64_t* const s_child_f = reinterpret_cast<int64_t*>(s_tmp + nn);
  int64_t* const s_child_l = s_child_f + cc;
  int32_t* const s_child_found = reinterpret_cast<int32_t*>(s_child_l + cc);
  uint16_t* const s_lv_ch = reinterpret_cast<uint16_t*>(s_child_found + cc);     // (i32[cc] of room: the live children's characters, push order)
  uint32_t* const s_child_h = reinterpret_cast<uint32_t*>(s_child_found + 2 * cc);  // the child's hash bucket,
  int32_t* const s_child_head = reinterpret_cast<int32_t*>(s_child_h + cc);      // that bucket's head as the fan-out saw it,
  int32_t* const s_child_slot = s_child_head + cc;                               // and the slot the child was pushed into (-1: merged / not yet)
  uint16_t* const s_child_ch = reinterpret_cast<uint16_t*>(s_child_slot + cc);
  uint8_t* const s_cur = reinterpret_cast<uint8_t*>(s_child_ch + cc);            // nfa_states: the popped entry's costs, deletions merged in
  uint8_t* const s_sub = s_cur + nn;                                             // states after one substitution error (any character)
  uint8_t* const s_top = s_sub + nn;                                             // the costs of the entry on top of the stack (see "top")
  __shared__ uint32_t s_pend[kNfaPend];                                          // pending entries per hash bucket mod kNfaPend (see "pend")
  __shared__ int16_t s_char_child[264];                                          // character -> its child's place in the group being merged, or -1
  uint32_t* const s_tmpg = reinterpret_cast<uint32_t*>(s_dyn + ((nfa_lds_bytes(nn, cc, B.lds_ents, 0) + 3) & ~size_t(3)));   // [lds_group][N]: tmp_states of a GROUP of children
  uint32_t* const s_ent_sd = reinterpret_cast<uint32_t*>(s_dyn + ((nfa_lds_bytes(nn, cc, 0, 0) + 3) & ~size_t(3)));   // kLds: the automaton's transitions ...
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
  #pragma unroll 1
  for (int c = t; c < 264; c += 64) s_char_child[c] = -1;
  __syncthreads();
  #pragma unroll 1
  for (int c = t; c < kAlphaSize; c += 64) {
    const uint32_t code = P::code_of(ix, uint32_t(c));
    s_code[c] = uint16_t(code);
    if (code != 0xffffu) atomicOr(&s_text[c >> 5], 1u << (c & 31));
  }
  __syncthreads();
  #pragma unroll 1
  for (int c = t; c < 264; c += 64) s_textpos[c] = 255;
  __syncthreads();
  int ntext = 0;
  {
    uint32_t tw[9];
#pragma unroll
    for (int w = 0; w < 9; w++) tw[w] = s_text[w];
#pragma unroll
    for (int w = 0; w < 9; w++) ntext += __popc(tw[w]);
    ntext = uni(ntext);
    #pragma unroll 1
    for (int c = t; c < kAlphaSize; c += 64) {
      const int cw = c >> 5;
      uint32_t mine = 0;
      int rank = 0;
#pragma unroll
      for (int w = 0; w < 9; w++) {
        if (w == cw) mine = tw[w];
        rank += w < cw ? __popc(tw[w]) : 0;
      }
      if (!((mine >> (c & 31)) & 1u)) continue;
      rank += __popc(mine & ((1u << (c & 31)) - 1u));
      s_textch[rank] = uint16_t(c);
      if (rank < 255) s_textpos[c] = uint8_t(rank);
    }
    #pragma unroll 1
    for (int c = t; c < 264; c += 64) s_alive[c] = 0;
  }
  const bool tcmode = B.lds_tc > 0 && ntext <= B.lds_tc;      // (the host counted the same characters: ntext == B.lds_tc <= 64)
  // tcmode: lane j IS the text's j-th character for the whole kernel (its character and code in registers)
  __syncthreads();
  const uint32_t my_ch = (tcmode && t < ntext) ? uint32_t(s_textch[t]) : 0u;
  const uint32_t my_code = (tcmode && t < ntext) ? uint32_t(s_code[my_ch]) : 0u;
  const int ntext_all = ntext;      // <= 261
  int n_lowtext = 0;                // ... of them below CHARACTER_OFFSET (the first entries of the list)
  for (int c = 0; c < kNfaOffset; c++) n_lowtext += int((s_text[0] >> c) & 1u);
  n_lowtext = uni(n_lowtext);
  if (ntext > 64) ntext = 0;        // ("warm": byte alphabets of more than 64 characters: no guessing ahead)
  __syncthreads();
  for (;;) {
    // FAIR SHARE.  A batch's kernel is launched with a workgroup for every slot of the GPU, and a workgroup lives until the
    // batch's counter runs out -- which, with one search 25 x longer than the mean, is most of the kernel's time: a second
    // caller's kernel would find no slot until then (measured, four callers: each batch started when its predecessor's short
    // searches were done, 2.1 s for what the longest search needs 1.35 s for).  So before it takes its next automaton a
    // workgroup looks at the number of batches in flight on the handle (one word of pinned host memory, a few hundred
    // nanoseconds per automaton of milliseconds): with `a` of them, the workgroups beyond the first 1/a of the grid leave.
    if (t == 0) {
      int act = 1;
      if (B.active) act = __builtin_nontemporal_load(B.active);
      const bool stay = act <= 1 || int(blockIdx.x) < (B.slots + act - 1) / act;
      s_q = stay ? atomicAdd(B.next, 1) : INT_MAX;
    }
    __syncthreads();
    const int qi = uni(s_q);
    __syncthreads();
    if (qi >= B.nq) break;
    const int q = B.order ? B.order[qi] : qi;
    const long long t_begin = B.iters_out ? clock64() : 0;
    const long long w_begin = B.iters_out ? wall_clock64() : 0;
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
    for (int i = t; i < kNfaPend; i += 64) s_pend[i] = 0;
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
    if (tcmode) {      // tmp_states of every character of the text, and the substitution states: "dead" whenever a pop begins
      #pragma unroll 1
      for (int x = t; x < B.lds_tc * N; x += 64) s_tmpg[x] = kNfaDead;
      #pragma unroll 1
      for (int i = t; i < N; i += 64) s_tmp[i] = kNfaDead;
    }
    int sp = 1, status = 0;
    int64_t iters = 0;
    // "top": the entry pushed LAST by a pop is the one the next pop takes.  Its range, match length and costs stay in
    // registers / LDS (s_top) and it never reaches the arena, the hash or the counters: nothing can look it up or merge into
    // it before it is popped (lookups happen in the NEXT pop's fan-out, after it is gone).  A pop of it costs no memory round
    // trip; a search that goes down a path pops such entries most of the time.
    // "pend": s_pend[b] counts the pending (arena) entries whose hash bucket is b mod kNfaPend.  Zero means the bucket's chain is empty: the fan-out then knows "not pending, head = -1" without
    // touching the arena -- the lookup that was one dependent round trip per pop.
    // "warm": the moment a pop knows its top child's range it asks for the lines that child's own fan-out will read -- one rank
    // line per character of the text and range end -- and lets them arrive while the rest of this pop and the start of the next
    // run.  A search is ONE chain of dependent pops, each with one rank request in the middle (4.3 us per pop measured, half of
    // it that request's latency); the batch ends with its longest search (profiles/r05_regexp_stats.txt), so the latency of a
    // single chain is what the kernel's time is made of.
    uint32_t warm = 0, warm2 = 0;
    // "spec" (tcmode on a policy whose step is one load per range end): not just a word of the top child's lines but the UNITS
    // its fan-out will need are loaded when its range is known -- lane j for the text's j-th character -- and stay in registers:
    // the next pop's fan-out is arithmetic, the rank request's latency is off the chain altogether.
    typename P::Spec spec;
    bool spec_ok = false;
    // built with -DFEMTO_AMD_NFA_PROF (tools/ab_bench.sh): shader-clock cycles per phase of a pop, summed over the automaton (phase k =
    // from PROF(k) to the next), printed by FEMTO_AMD_NFA_STATS=1.  Not in the product build: the accumulators cost registers.
#ifdef FEMTO_AMD_NFA_PROF
    long long prof_t = 0, prof_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int prof_k = 8;
#define PROF(k)                                             \
  if (B.iters_out) {                                        \
    const long long now_ = clock64();                       \
    prof_acc[prof_k] += now_ - prof_t;                      \
    prof_t = now_;                                          \
    prof_k = (k);                                           \
  }
    if (B.iters_out) prof_t = clock64();
#else
#define PROF(k)
#endif
    int top_slot = -1, top_len = 0;
    int64_t top_f = 0, top_l = 0;
    __syncthreads();
    for (;;) {
      if (iters > B.max_iterations) { status = kNfaStatusOverworked; break; }   // server.c:1821
      if (sp == 0) break;
      sp--;
      const bool from_top = sp == top_slot;
      top_slot = -1;
      const bool use_spec = from_top && spec_ok;
      spec_ok = false;
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
          s_pend[h & (kNfaPend - 1)] -= 1u;
        }
      }
      PROF(0);
      // ---- a final state alive: a result, not extended (approx_is_final_state: the first such node's cost)
      int fin = INT_MAX;
      #pragma unroll 1
      for (int i0 = 0; i0 < N; i0 += 64) {        // (the first alive final node: the first lane of the first round that has one)
        const int i = i0 + t;
        bool is_fin = false;
        if (i < N) {
          const uint8_t c = from_top ? s_top[i] : e_cost[size_t(sp) * stride + i];
          s_cur[i] = c;
          is_fin = int(c) < bound && (FLAGS(i) & 2u);
        }
        const unsigned long long fm = __ballot(is_fin);
        if (fm && fin == INT_MAX) fin = i0 + __ffsll(static_cast<long long>(fm)) - 1;
      }
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
      PROF(1);
      // ---- deletions: states after reading ANY character at delete_cost, merged in (server.c:1854-1863)
      if (approx) {
        if (!tcmode) {
          #pragma unroll 1
          for (int i = t; i < N; i += 64) s_tmp[i] = kNfaDead;
          __syncthreads();
        }
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
          if (tcmode) s_tmp[i] = kNfaDead;      // (dead again for the substitution states of the pass below)
        }
        __syncthreads();
      }
      PROF(2);
      // ---- may any character be an error from here?  (nfa_errcnt_t arithmetic: one byte, as the reference computes it)
      int m = bound;
      #pragma unroll 1
      for (int i = t; i < N; i += 64) m = int(s_cur[i]) < m ? int(s_cur[i]) : m;
      m = uni(wave_min_i32(m));
      const int ms = m + Q.subst, mi = m + Q.ins;
      const int min_err = (ms < mi ? ms : mi) & 0xff;
      const bool allchars = min_err < bound && iters > 0;
      // ---- r_c: the characters an alive state can read, as one flag per character (cleared again as they are read below)
      // By-character pass (B.lds_tc: small alphabets and automata): the SAME pass over the transitions also accumulates, for
      // every character of the text, the states reached by reading it (approx_get_reachable_states -- the children's tmp_states,
      // whichever of them turn out to have rows) and the states after a substitution error (server.c:2107-2110): one pass and
      // one barrier instead of three passes of three dependent LDS phases each.
      #pragma unroll 1
      for (int e = t; e < T; e += 64) {
        const uint32_t sd = ENT_SD(e);
        const uint32_t c = s_cur[sd & 0xffffu];
        if (int(c) >= bound) continue;
        const uint32_t ch = ENT_CH(e);
        s_alive[ch] = 1;
        if (tcmode) {
          const uint32_t tp = s_textpos[ch];
          if (tp != 255u) atomicMin(&s_tmpg[tp * uint32_t(N) + (sd >> 16)], c);
          if (approx && e >= base5 && int(c) + Q.subst < bound) atomicMin(&s_tmp[sd >> 16], c + uint32_t(Q.subst));
        }
      }
      __syncthreads();
      int nlive = 0, n_new = 0, last_new = -1;
      if (tcmode) {
        // ---- children and fan-out, one lane per character of the text (small alphabets): lane j's character is a child when an
        // alive state reads it (or any character may be an error); the lane steps the range with it -- from the units loaded
        // ahead when the popped entry was the top ("spec") -- looks the child's range up among the pending entries, and the
        // places in push order (characters >= CHARACTER_OFFSET ascending, then those below it: server.c:2114-2130) are counts
        // of ballot bits.  No list of children in LDS, no second round trip for the character and its code.
        bool isch = false;
        if (t < ntext) {
          isch = s_alive[my_ch] != 0 || (allchars && my_ch >= uint32_t(kNfaOffset));
          s_alive[my_ch] = 0;
        }
        PROF(3);
        asm volatile("" ::"v"(warm), "v"(warm2));
        bool live = false;
        uint32_t h = 0;
        int64_t f = first, l = last;
        int found = -1, head = -1;
        if (isch) {
          if constexpr (P::kNfaSpec) {
            if (use_spec) P::spec_step(ix, my_code, f, l, spec);
            else P::search_step(ix, 1, my_code, f, l);
          } else {
            P::search_step(ix, 1, my_code, f, l);
          }
          live = l >= f;                               // add_mapping ignores empty ranges (server.c:1565)
          if (live) {
            h = hash_of(f, l);
            if (s_pend[h & (kNfaPend - 1)] != 0) {     // ("pend": an empty bucket needs no look at the arena)
              head = heads[h];
              for (int s2 = head; s2 >= 0; s2 = e_next[s2])
                if (e_first[s2] == f && e_last[s2] == l) { found = s2; break; }
            }
          }
        }
        const bool isnew = live && found < 0;
        const unsigned long long lm = __ballot(live), nm = __ballot(isnew);
        const unsigned long long lowm = (1ull << n_lowtext) - 1ull, below = (1ull << t) - 1ull;
        const bool lo = t < n_lowtext;      // a character below CHARACTER_OFFSET: pushed after all the others
        if (live) {
          const int j = lo ? int(__popcll(lm & ~lowm)) + int(__popcll(lm & lowm & below)) : int(__popcll(lm & ~lowm & below));
          const int nb = lo ? int(__popcll(nm & ~lowm)) + int(__popcll(nm & lowm & below)) : int(__popcll(nm & ~lowm & below));
          s_lv_ch[j] = uint16_t(my_ch);
          s_child_f[j] = f;
          s_child_l[j] = l;
          s_child_found[j] = found;
          s_child_slot[j] = isnew ? sp + nb : found;
          s_child_h[j] = h;
          s_child_head[j] = head;
        }
        nlive = int(__popcll(lm));
        n_new = int(__popcll(nm));
        if (nm) {
          const unsigned long long pick = (nm & lowm) ? (nm & lowm) : nm;      // the new child pushed last
          const int lane = 63 - __builtin_clzll(pick);
          last_new = (nm & lowm) ? int(__popcll(lm & ~lowm)) + int(__popcll(lm & lowm & ((1ull << lane) - 1ull))) : int(__popcll(lm & ~lowm & ((1ull << lane) - 1ull)));
          top_f = int64_t((uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(uint64_t(f) >> 32)), lane))) << 32) |
                          uint32_t(__builtin_amdgcn_readlane(int(uint32_t(uint64_t(f))), lane)));
          top_l = int64_t((uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(uint64_t(l) >> 32)), lane))) << 32) |
                          uint32_t(__builtin_amdgcn_readlane(int(uint32_t(uint64_t(l))), lane)));
          top_slot = sp + n_new - 1;
          top_len = len + 1;
        }
      } else {
        // ---- the children in push order: characters >= CHARACTER_OFFSET ascending (the substitution loop creates their
        // entries first, server.c:2114-2120), then the characters below it (regular loop, :2123-2130).  Only characters the text
        // holds can have rows, so the lanes go over the TEXT's characters (ascending: those below CHARACTER_OFFSET come first in
        // the list), 64 at a time, and a child's place is a count of ballot bits.  (Until round 5 this went over all 261 characters
        // with nine-word bit arithmetic per lane: 4 300 of a pop's 12 000 cycles, measured with FEMTO_AMD_NFA_STATS.)
        unsigned long long cm[5] = {0, 0, 0, 0, 0};
        int nchild = 0;
#pragma unroll
        for (int w = 0; w < 5; w++) {
          if (64 * w >= ntext_all) break;
          const int j = 64 * w + t;
          bool isch = false;
          if (j < ntext_all) {
            const int c = s_textch[j];
            isch = s_alive[c] != 0 || (allchars && c >= kNfaOffset);
            s_alive[c] = 0;
          }
          cm[w] = __ballot(isch);
          nchild += int(__popcll(cm[w]));
        }
        const int n_lo = int(__popcll(cm[0] & ((1ull << n_lowtext) - 1ull)));      // children below CHARACTER_OFFSET: pushed last
        {
          int before = 0;
#pragma unroll
          for (int w = 0; w < 5; w++) {
            if (64 * w >= ntext_all) break;
            const int j = 64 * w + t;
            if ((cm[w] >> t) & 1ull) {
              const int rank = before + int(__popcll(cm[w] & ((1ull << t) - 1ull)));      // children before this one in character order
              s_child_ch[j < n_lowtext ? (nchild - n_lo) + rank : rank - n_lo] = s_textch[j];
            }
            before += int(__popcll(cm[w]));
          }
        }
        __syncthreads();
        PROF(3);
        asm volatile("" ::"v"(warm), "v"(warm2));      // ("warm": the lines asked for by the previous pop have arrived by now, or are waited for here)
        // ---- the fan-out: lane k steps the range with the k-th character (server.c:1954-2060); then add_mapping's lookup: a
        // pending entry with the child's range?  (children of one pop have disjoint ranges.)  The children WITH rows are written
        // out in push order (index j: s_lv_ch / s_child_*), each with its slot -- a child whose range is pending keeps that entry's
        // slot, the new ones take sp, sp + 1, ... -- all from ballots over the lanes' own registers; the LAST new one becomes the
        // "top" (the next entry popped) and its range goes straight into scalar registers.
        for (int k0 = 0; k0 < nchild; k0 += 64) {
          const int k = k0 + t;
          bool live = false;
          uint32_t ch = 0, h = 0;
          int64_t f = first, l = last;
          int found = -1, head = -1;
          if (k < nchild) {
            ch = s_child_ch[k];
            P::search_step(ix, 1, s_code[ch], f, l);
            live = l >= f;                               // add_mapping ignores empty ranges (server.c:1565)
            if (live) {
              h = hash_of(f, l);
              if (s_pend[h & (kNfaPend - 1)] != 0) {     // ("pend": an empty bucket needs no look at the arena)
                head = heads[h];
                for (int s2 = head; s2 >= 0; s2 = e_next[s2])
                  if (e_first[s2] == f && e_last[s2] == l) { found = s2; break; }
              }
            }
          }
          const bool isnew = live && found < 0;
          const unsigned long long lm = __ballot(live), nm = __ballot(isnew);
          const unsigned long long below = (1ull << t) - 1ull;
          if (live) {
            const int j = nlive + int(__popcll(lm & below));
            s_lv_ch[j] = uint16_t(ch);
            s_child_f[j] = f;
            s_child_l[j] = l;
            s_child_found[j] = found;
            s_child_slot[j] = isnew ? sp + n_new + int(__popcll(nm & below)) : found;
            s_child_h[j] = h;
            s_child_head[j] = head;
          }
          if (nm) {
            const int lane = 63 - __builtin_clzll(nm);
            last_new = nlive + int(__popcll(lm & ((1ull << lane) - 1ull)));
            top_f = int64_t((uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(uint64_t(f) >> 32)), lane))) << 32) |
                            uint32_t(__builtin_amdgcn_readlane(int(uint32_t(uint64_t(f))), lane)));
            top_l = int64_t((uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(uint64_t(l) >> 32)), lane))) << 32) |
                            uint32_t(__builtin_amdgcn_readlane(int(uint32_t(uint64_t(l))), lane)));
            top_slot = sp + n_new + int(__popcll(nm)) - 1;
            top_len = len + 1;
          }
          nlive += int(__popcll(lm));
          n_new += int(__popcll(nm));
        }
      }
      PROF(4);
      if (sp + n_new > cap) { status = kNfaStatusFull; break; }     // (some new child would find the stack full)
      if (P::kNfaSpec && tcmode && B.warm) {
        if (last_new >= 0) {      // "spec" (uniform condition; every lane loads, unconditionally)
          P::spec_load(ix, my_code, top_f, top_l, spec);
          spec_ok = true;
        }
      } else if (last_new >= 0 && ntext > 0 && B.warm) {      // "warm" (uniform condition; every lane loads, unconditionally: a load inside a divergent
                                             // block is waited for where the block ends)
        const uint32_t code = s_code[s_textch[t < ntext ? t : 0]];
        warm = P::touch(ix, code, top_l);
        warm2 = P::touch(ix, code, top_f > 0 ? top_f - 1 : 0);      // (two registers, nothing computed from them here: a use is where the wait goes)
      }
      PROF(5);
      // ---- substitutions: states after reading any character at subst_cost (server.c:2107-2110)
      if (approx && !tcmode) {
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
      PROF(6);
      // ---- the children's states, a GROUP of children at a time (until round 5 one child at a time, five dependent LDS phases
      // each): tmp_states of every child of the group are accumulated in ONE pass over the transitions of the group's
      // characters (approx_get_reachable_states, nfa.c), then every (child, node) pair is finished and stored by its own lane
      // (approx_add_error_allchars nfa.c:305, the substitution states, nfa_states_union with a pending entry's)
      if (tcmode) {      // every (live child, node) pair by its own lane, from the rows the by-character pass filled
        #pragma unroll 1
        for (int x = t; x < nlive * N; x += 64) {
          const int k = x / N, i = x - k * N;
          const int ch = s_lv_ch[k], found = s_child_found[k], slot = s_child_slot[k];
          uint32_t v = s_tmpg[uint32_t(s_textpos[ch]) * uint32_t(N) + uint32_t(i)];
          if (approx) {
            const uint32_t ins = uint32_t(s_cur[i]) + uint32_t(Q.ins);          // approx_add_error_allchars (nfa.c:305)
            v = ins < v ? ins : v;
            const uint32_t sub = s_tmp[i];
            if (ch >= kNfaOffset && sub < v) v = sub;
          }
          if (int(v) >= bound) v = kNfaDead;                                    // beyond the bound is dead, whatever the number
          if (k == last_new) {
            s_top[i] = uint8_t(v);                                              // ("top": not the arena)
          } else {
            uint8_t* const dst = e_cost + size_t(slot) * stride + i;
            if (found >= 0) {
              const uint32_t old = *dst;
              v = old < v ? old : v;                                            // nfa_states_union
            }
            *dst = uint8_t(v);
          }
        }
        __syncthreads();
        #pragma unroll 1
        for (int x = t; x < B.lds_tc * N; x += 64) s_tmpg[x] = kNfaDead;     // "dead" again for the next pop
        if (approx) {
          #pragma unroll 1
          for (int i = t; i < N; i += 64) s_tmp[i] = kNfaDead;
        }
      }
      const int G = B.lds_group;
      for (int g0 = 0; g0 < nlive && !tcmode; g0 += G) {
        const int gn = nlive - g0 < G ? nlive - g0 : G;
        #pragma unroll 1
        for (int x = t; x < gn * N; x += 64) s_tmpg[x] = kNfaDead;
        int elo = INT_MAX, ehi = 0, my_ch = 0;
        if (t < gn) {
          my_ch = s_lv_ch[g0 + t];
          s_char_child[my_ch] = int16_t(t);
          elo = s_bychar[my_ch];
          ehi = s_bychar[my_ch + 1];
        }
        elo = wave_min_i32(elo);
        ehi = -wave_min_i32(-ehi);
        __syncthreads();
        #pragma unroll 1
        for (int e = elo + t; e < ehi; e += 64) {
          const int ci = s_char_child[ENT_CH(e)];
          if (ci < 0) continue;
          const uint32_t sd = ENT_SD(e);
          const uint32_t c = s_cur[sd & 0xffffu];
          if (int(c) < bound) atomicMin(&s_tmpg[ci * N + int(sd >> 16)], c);
        }
        __syncthreads();
        #pragma unroll 1
        for (int x = t; x < gn * N; x += 64) {
          const int ci = x / N, i = x - ci * N;
          const int k = g0 + ci;
          const int ch = s_lv_ch[k], found = s_child_found[k], slot = s_child_slot[k];
          uint32_t v = s_tmpg[x];
          if (approx) {
            const uint32_t ins = uint32_t(s_cur[i]) + uint32_t(Q.ins);          // approx_add_error_allchars (nfa.c:305)
            v = ins < v ? ins : v;
            if (ch >= kNfaOffset && uint32_t(s_sub[i]) < v) v = s_sub[i];
          }
          if (int(v) >= bound) v = kNfaDead;                                    // beyond the bound is dead, whatever the number
          if (g0 + ci == last_new) {
            s_top[i] = uint8_t(v);                                              // ("top": not the arena)
          } else {
            uint8_t* const dst = e_cost + size_t(slot) * stride + i;
            if (found >= 0) {
              const uint32_t old = *dst;
              v = old < v ? old : v;                                            // nfa_states_union
            }
            *dst = uint8_t(v);
          }
        }
        if (t < gn) s_char_child[my_ch] = -1;
        __syncthreads();
      }
      PROF(7);
      // ---- the entries themselves, one lane per child: a pending entry keeps the longer match (server.c:1611-1619); a new one
      // is linked into its bucket's chain behind the new children of this pop that precede it there (push order), and the last
      // of a bucket becomes its head
      for (int j0 = 0; j0 < nlive; j0 += 64) {
        const int j = j0 + t;
        if (j >= nlive) continue;
        const int k = j;
        const int found = s_child_found[k], slot = s_child_slot[k];
        if (found >= 0) {
          if (len + 1 > e_len[found]) e_len[found] = len + 1;
        } else if (j != last_new) {
          const uint32_t h = s_child_h[k];
          int prev = s_child_head[k];
          bool later = false;
          #pragma unroll 1
          for (int j2 = 0; j2 < nlive; j2++) {
            if (j2 == j || j2 == last_new) continue;
            const int k2 = j2;
            if (s_child_found[k2] >= 0 || s_child_h[k2] != h) continue;
            if (j2 < j) prev = s_child_slot[k2];
            else later = true;
          }
          e_first[slot] = s_child_f[k];
          e_last[slot] = s_child_l[k];
          e_len[slot] = len + 1;
          e_next[slot] = prev;
          if (!later) heads[h] = slot;
          atomicAdd(&s_pend[h & (kNfaPend - 1)], 1u);
        }
      }
      sp += n_new;
      __syncthreads();
      PROF(8);
      if (status) break;
      iters++;
    }
    if (t == 0) {
      B.status[q] = status;
      if (B.iters_out) {      // pops, shader-clock cycles / 1024 spent, and when it started
        B.iters_out[q] = int32_t(iters < INT_MAX ? iters : INT_MAX);
        const long long now = clock64();
        B.iters_out[B.nq_all + q] = int32_t((now - t_begin) >> 10);
        B.iters_out[2 * B.nq_all + q] = int32_t(uint32_t(uint64_t(w_begin) >> 4));
        B.iters_out[3 * B.nq_all + q] = int32_t(uint32_t(uint64_t(wall_clock64() - w_begin) >> 4));
#ifdef FEMTO_AMD_NFA_PROF
        prof_acc[prof_k] += now - prof_t;
        for (int k = 0; k < 9; k++) atomicAdd(reinterpret_cast<unsigned long long*>(B.iters_out + 4 * B.nq_all) + k, static_cast<unsigned long long>(prof_acc[k]));
#endif
      }
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
// regexp_result_list_sort (server.c:1528-1573) of one automaton's raw results, in place; returns how many are kept
static size_t sort_results(NfaHostResult* r, size_t count) {
  std::stable_sort(r, r + count, [](const NfaHostResult& a, const NfaHostResult& b) {
    if (a.first != b.first) return a.first < b.first;
    if (a.last != b.last) return a.last > b.last;
    return a.seq < b.seq;
  });
  size_t n = 0;
  for (size_t k = 0; k < count; k++) {
    if (n && r[k].first == r[n - 1].first && r[k].last == r[n - 1].last) continue;
    r[n++] = r[k];
  }
  if (n == 0) return 0;
  int64_t first = r[0].first, last = r[0].last;
  size_t i = 1;
  for (size_t k = 1; k < n; k++) {
    if (r[k].first >= first && r[k].last <= last) continue;
    first = r[k].first;
    last = r[k].last;
    r[i++] = r[k];
  }
  return i;
}

namespace {
thread_local double g_nfa_stats_thread[8] = {0, 0, 0, 0, 0, 0, 0, 0};
}

int femto_amd_nfa_stats(femto_amd_index_t* ix, double* out8) {
  if (!out8) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (!ix) {      // the calling thread's own last batch (concurrent callers on one handle)
    for (int k = 0; k < 8; k++) out8[k] = g_nfa_stats_thread[k];
    return FEMTO_AMD_OK;
  }
  if (!ix->children.empty()) return femto_amd_nfa_stats(ix->children[0], out8);
  std::lock_guard<std::mutex> lk(ix->mu);
  for (int k = 0; k < 8; k++) out8[k] = ix->nfa_stats[k];
  return FEMTO_AMD_OK;
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
  const auto t_call = std::chrono::steady_clock::now();      // FEMTO_AMD_NFA_STATS: where the call's time goes on the host
  auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  double t_flat = 0, t_search = 0, t_copy = 0, t_sort = 0;
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
  t_flat = ms_since(t_call);
  {      // the handle's count of batches in flight: one word of pinned host memory the kernels read ("FAIR SHARE")
    std::lock_guard<std::mutex> lk(ix->mu);
    if (!ix->nfa_active) {
      HIP_TRY(hipSetDevice(ix->device));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ix->nfa_active), 64, hipHostMallocMapped));
      *ix->nfa_active = 0;
      void* dp = nullptr;
      HIP_TRY(hipHostGetDevicePointer(&dp, ix->nfa_active, 0));
      ix->nfa_active_dev = static_cast<int32_t*>(dp);
    }
  }
  struct InFlight {
    int32_t* w;
    explicit InFlight(int32_t* p) : w(p) { __atomic_add_fetch(w, 1, __ATOMIC_RELAXED); }
    ~InFlight() { __atomic_sub_fetch(w, 1, __ATOMIC_RELAXED); }
  } in_flight{ix->nfa_active};
  double wall_hz = 1e8;
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ix->device) == hipSuccess && khz > 0) wall_hz = double(khz) * 1e3;
    else (void)hipGetLastError();
  }
  Lease L(ix);
  if (!L.s) return L.rc;
  hipStream_t st = L.s->stream;
  Scratch& S = *L.s;
  DeviceBuffer &d_q = S.nfa_q, &d_flags = S.nfa_flags, &d_sd = S.nfa_sd, &d_ch = S.nfa_ch, &d_bychar = S.nfa_bychar, &d_arena = S.nfa_arena,
               &d_results = S.nfa_results, &d_misc = S.nfa_misc, &d_order = S.nfa_order;
  struct Trim {      // (the buffers stay with the scratch; an arena or a result buffer that a rare retry blew up does not)
    Scratch& S;
    ~Trim() {
      if (S.nfa_arena.cap > (size_t(2) << 30)) S.nfa_arena.release();
      if (S.nfa_results.cap > (size_t(1) << 30)) S.nfa_results.release();
    }
  } trim{S};
  // Raw results (before the per-automaton sort drops ranges inside other results, and including the attempts of searches
  // that are run again in a larger arena) land in a buffer of the library's own, which grows and the batch runs again
  // when it is too small: max_results bounds what the CALLER's arrays receive, nothing else (max_results == 0: count only).
  // (the first size is a guess, not a promise: a generous max_results does not reserve memory up front)
  int64_t result_cap = std::max<int64_t>(std::min<int64_t>(2 * max_results, int64_t(1) << 22), 1 << 12);
  if ((rc = d_q.reserve(hq.size() * sizeof(NfaQueryDev))) || (rc = d_flags.reserve(h_flags.size() + 16)) ||
      (rc = d_sd.reserve(h_sd.size() * 4 + 16)) || (rc = d_ch.reserve(h_ch.size() * 2 + 16)) || (rc = d_bychar.reserve(h_bychar.size() * 4)) ||
      (rc = d_misc.reserve(64 + size_t(nq) * 20 + 128)) ||
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
  B.iters_out = d_status + nq;      // pops, cycles, start and duration of every search: femto_amd_nfa_stats (four words per automaton, written once)
  B.nq_all = int32_t(nq);
  B.active = knob(-1, "FEMTO_AMD_NFA_FAIR", 1) != 0 ? ix->nfa_active_dev : nullptr;
  B.warm = knob(-1, "FEMTO_AMD_NFA_WARM", 1) != 0;
  B.max_iterations = ix->regexp_max_iterations;
  B.cost_stride = (max_nodes + 3) & ~3;
  B.lds_nodes = (max_nodes + 7) & ~7;
  int text_chars = 0;
  {
    int nchars = 0;       // characters the text holds: the most children a pop can have
    for (int c = 0; c < kAlphaSize && size_t(c) + 1 < ix->host.C.size(); c++)
      if (ix->host.C[size_t(c) + 1] > ix->host.C[size_t(c)]) nchars++;
    B.lds_children = (std::max(nchars, 4) + 3) & ~3;
    text_chars = nchars;
  }
  {
    size_t max_ents = 0;
    for (const NfaQueryDev& Q : hq) max_ents = std::max(max_ents, size_t(Q.num_ents));
    B.lds_ents = max_ents <= size_t(kNfaLdsEnts) ? int32_t((std::max<size_t>(max_ents, 2) + 1) & ~size_t(1)) : 0;
  }
  B.lds_group = std::max(1, std::min({int(B.lds_children), 64, 16384 / (int(B.lds_nodes) * 4)}));
  // small alphabets and automata (DNA motifs: 5 characters x ~24 nodes): one row of tmp_states per character of the TEXT
  B.lds_tc = (text_chars <= 64 && text_chars <= B.lds_group && size_t(text_chars) * size_t(B.lds_nodes) * 4 <= 4096) ? text_chars : 0;
  const size_t lds = nfa_lds_bytes(B.lds_nodes, B.lds_children, B.lds_ents, B.lds_group);
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
  double st_pops = 0, st_max = 0, st_busy = 0, st_span = 0, st_blocks = 0, st_late = 0;      // femto_amd_nfa_stats (of the last attempt)
  for (int attempt = 0;; attempt++) {
  if ((rc = d_results.reserve(size_t(result_cap) * sizeof(NfaResultDev)))) return rc;
  B.results = d_results.as<NfaResultDev>();
  B.result_cap = result_cap;
  HIP_TRY(hipMemsetAsync(d_misc.p, 0, 64 + size_t(nq) * 20 + 128, st));
  todo.resize(static_cast<size_t>(nq));
  for (int64_t i = 0; i < nq; i++) todo[size_t(i)] = int32_t(i);
  // LONGEST-PREDICTED-FIRST.  The workgroups take automata off one counter, and the kernel lasts as long as its longest search
  // (profiles/r05_regexp_final.txt: 156 k pops of a mean of 6.3 k) -- plus however long that search waited to be taken.  The
  // order of the result lists is per automaton, not per batch, so the batch is handed out by falling predicted work: errors
  // allowed x transitions (an automaton that accepts more strings branches more) x nodes.  A search taken at t = 0 ends the
  // kernel at its own length; concurrent callers' kernels fill the workgroups the short searches leave.
  if (knob(-1, "FEMTO_AMD_NFA_LPT", 1) != 0) {
    auto weight = [&](int32_t q) { const NfaQueryDev& Q = hq[size_t(q)]; return int64_t(Q.cost_bound) * int64_t(Q.num_ents) * int64_t(Q.num_nodes); };
    std::stable_sort(todo.begin(), todo.end(), [&](int32_t a, int32_t b) { return weight(a) > weight(b); });
  }
  st_pops = st_max = st_busy = st_span = st_blocks = st_late = 0;
  int64_t cap = std::min<int64_t>(1024, ix->regexp_stack_cap);
  for (int pass = 0;; pass++) {
    // (a fair share of the GPU's workgroup slots when other batches are in flight on the handle: see "FAIR SHARE" in the kernel)
    const int active_now = std::max(1, int(__atomic_load_n(ix->nfa_active, __ATOMIC_RELAXED)));
    B.slots = int32_t(std::min<int64_t>(INT32_MAX, int64_t(ix->num_cus) * per_cu));
    int blocks = int(std::min<int64_t>(int64_t(todo.size()), (int64_t(B.slots) + active_now - 1) / active_now));
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
    std::vector<int32_t> it(static_cast<size_t>(nq) * 4);
    HIP_TRY(hipMemcpyAsync(it.data(), d_status + nq, size_t(nq) * 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    {      // what femto_amd_nfa_stats reports: pops, the span of the pass and how busy its workgroups were, on the device's wall clock
      // (starts are 32-bit counts of 16 ticks: differences to one of them, as signed numbers, survive the wrap)
      const uint32_t ref = todo.empty() ? 0u : uint32_t(it[size_t(2 * nq + todo[0])]);
      int64_t t0 = INT64_MAX, tend = INT64_MIN, late = 0;
      double busy = 0, pops = 0, pmax = -1;
      for (int32_t q : todo) t0 = std::min<int64_t>(t0, int32_t(uint32_t(it[size_t(2 * nq + q)]) - ref));
      for (int32_t q : todo) {
        const int64_t s0 = int64_t(int32_t(uint32_t(it[size_t(2 * nq + q)]) - ref)) - t0, run = int64_t(uint32_t(it[size_t(3 * nq + q)]));
        busy += double(run);
        pops += double(it[size_t(q)]);
        if (double(it[size_t(q)]) > pmax) { pmax = double(it[size_t(q)]); late = s0; }
        tend = std::max(tend, s0 + run);
      }
      const double tick_s = 16.0 / wall_hz;
      st_pops += pops;
      st_max = std::max(st_max, pmax);
      st_busy += busy * tick_s;
      st_span += todo.empty() ? 0.0 : double(tend) * tick_s;
      st_blocks = std::max(st_blocks, double(blocks));
      if (pass == 0) st_late = double(late) * tick_s;
    }
    if (want_stats) {      // entries popped per automaton: the kernel ends with its longest search (profiles/r05_regexp_*)
      std::vector<int32_t> v;
      for (int32_t q : todo) v.push_back(it[size_t(q)]);
      std::sort(v.begin(), v.end());
      double sum = 0;
      for (int32_t x : v) sum += x;
      auto pct = [&](double p) { return v.empty() ? 0 : v[std::min(v.size() - 1, size_t(p * double(v.size())))]; };
      fprintf(stderr, "[femto_amd] nfa pass %d: %zu automata, %d workgroups, pops: mean %.0f  p50 %d  p90 %d  p99 %d  p99.9 %d  max %d  (sum %.3g; max / (sum / workgroups) = %.2f)\n",
              pass, v.size(), blocks, v.empty() ? 0.0 : sum / double(v.size()), pct(0.5), pct(0.9), pct(0.99), pct(0.999), v.empty() ? 0 : v.back(), sum,
              sum > 0 ? double(v.empty() ? 0 : v.back()) / (sum / double(blocks)) : 0.0);
      {      // the clock: which search ended last, when it started, what a pop of it cost; how busy the workgroups were
        const uint32_t ref = todo.empty() ? 0u : uint32_t(it[size_t(2 * nq + todo[0])]);
        auto start_of = [&](int32_t q) { return int64_t(int32_t(uint32_t(it[size_t(2 * nq + q)]) - ref)); };
        int64_t t0 = INT64_MAX, tend = INT64_MIN;
        for (int32_t q : todo) t0 = std::min(t0, start_of(q));
        int32_t qlast = todo.empty() ? 0 : todo[0], qmax = qlast;
        double busy = 0;
        for (int32_t q : todo) {
          const int64_t en = start_of(q) - t0 + int64_t(uint32_t(it[size_t(3 * nq + q)]));
          busy += double(uint32_t(it[size_t(3 * nq + q)]));
          if (en > tend) { tend = en; qlast = q; }
          if (it[size_t(q)] > it[size_t(qmax)]) qmax = q;
        }
        const double tick_ms = 16e3 / wall_hz;
        auto line = [&](const char* what, int32_t q) {
          fprintf(stderr, "[femto_amd]   %s: automaton %d, %d pops, started at %.2f ms, ran %.2f ms = %.0f shader cycles per pop\n", what, q, it[size_t(q)],
                  double(start_of(q) - t0) * tick_ms, double(uint32_t(it[size_t(3 * nq + q)])) * tick_ms,
                  it[size_t(q)] ? double(uint32_t(it[size_t(nq + q)])) * 1024.0 / double(it[size_t(q)]) : 0.0);
        };
        line("ended last", qlast);
        line("most pops ", qmax);
        {
          unsigned long long ph[9];
          HIP_TRY(hipMemcpy(ph, d_status + 5 * nq, sizeof ph, hipMemcpyDeviceToHost));
          static const char* const names[9] = {"pop + final test", "deletions", "min cost + reachable characters + child list", "fan-out: rank step + pending lookup", "slots + warm",
                                               "substitutions", "children's states (groups)", "entries: link / top", "between pops (results, loop)"};
          double tot = 0;
          for (int k = 0; k < 9; k++) tot += double(ph[k]);
          for (int k = 0; k < 9 && tot > 0; k++)
            fprintf(stderr, "[femto_amd]     phase %d  %-46s %6.0f cycles per pop  %5.1f %%\n", k, names[k], sum > 0 ? double(ph[k]) / sum : 0.0, tot > 0 ? 100.0 * double(ph[k]) / tot : 0.0);
          HIP_TRY(hipMemset(d_status + 5 * nq, 0, sizeof ph));
        }
        fprintf(stderr, "[femto_amd]   all searches: %.2f ms on the device's wall clock; busy workgroup-time %.1f ms = %.2f of workgroups x span; mean time per pop %.2f us\n",
                double(tend) * tick_ms, busy * tick_ms, tend > 0 ? busy / (double(tend) * double(blocks)) : 0.0, sum > 0 ? busy * tick_ms * 1e3 / sum : 0.0);
      }
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
  t_search = ms_since(t_call) - t_flat;
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    const double v[8] = {double(nq), st_blocks, st_pops, st_max, st_busy, st_span, (st_span > 0 && st_blocks > 0) ? st_busy / (st_span * st_blocks) : 0.0, st_late};
    for (int k = 0; k < 8; k++) ix->nfa_stats[k] = g_nfa_stats_thread[k] = v[k];
  }
  std::vector<NfaResultDev> raw(static_cast<size_t>(count));
  if (count) HIP_TRY(hipMemcpy(raw.data(), d_results.p, size_t(count) * sizeof(NfaResultDev), hipMemcpyDeviceToHost));
  t_copy = ms_since(t_call) - t_flat - t_search;
  // a search that was run again appended its earlier attempts' results too: only the last attempt's count.  The raw results are
  // grouped by automaton in one flat array (count, prefix, scatter) and every automaton's list is sorted by the handle's host
  // threads (regexp_result_list_sort, server.c:1528-1573: 2.98 M raw ranges of 20 000 APPROX 1 motifs took one thread 167 ms --
  // a quarter of the call, next to 490 ms of search)
  std::vector<int64_t> seg(static_cast<size_t>(nq) + 1, 0);
  for (size_t k = 0; k < raw.size(); k++)
    if (raw[k].pass == last_pass[size_t(raw[k].query)] && status[size_t(raw[k].query)] == 0) seg[size_t(raw[k].query) + 1]++;   // (an error: the reference returns no results, RETURN_ERROR)
  for (int64_t qi = 0; qi < nq; qi++) seg[size_t(qi) + 1] += seg[size_t(qi)];
  std::vector<NfaHostResult> flat(static_cast<size_t>(seg[size_t(nq)]));
  {
    std::vector<int64_t> at(seg.begin(), seg.end() - 1);
    for (size_t k = 0; k < raw.size(); k++)
      if (raw[k].pass == last_pass[size_t(raw[k].query)] && status[size_t(raw[k].query)] == 0)
        flat[size_t(at[size_t(raw[k].query)]++)] = {raw[k].first, raw[k].last, raw[k].len, raw[k].cost, int64_t(k)};
  }
  std::vector<int64_t> kept(static_cast<size_t>(nq), 0);
  auto sort_some = [&](int t, int nt) {
    for (int64_t qi = t; qi < nq; qi += nt) kept[size_t(qi)] = int64_t(sort_results(flat.data() + seg[size_t(qi)], size_t(seg[size_t(qi) + 1] - seg[size_t(qi)])));
  };
  if (flat.size() >= (size_t(1) << 16)) {
    ensure_workers(ix);
    std::lock_guard<std::mutex> wl(ix->workers_mu);
    ix->workers->run(sort_some);
  } else {
    sort_some(0, 1);
  }
  int64_t n = 0;
  for (int64_t qi = 0; qi < nq; qi++) {
    result_start[qi] = n;
    n += kept[size_t(qi)];
  }
  result_start[nq] = n;
  *n_out = n;
  t_sort = ms_since(t_call) - t_flat - t_search - t_copy;
  if (want_stats)
    fprintf(stderr, "[femto_amd] nfa call, host side: flatten the automata %.1f ms, upload + search passes %.1f, results to the host %.1f (%llu raw), per-automaton sort %.1f\n",
            t_flat, t_search, t_copy, count, t_sort);
  if (status_out) std::memcpy(status_out, status.data(), size_t(nq) * 4);
  if (max_results == 0) return FEMTO_AMD_OK;        // count only: *n_out and result_start[] are what a second call needs
  if (n > max_results) return set_err(FEMTO_AMD_ERR_FULL, "more results than max_results: *n_out holds the number to call again with");
  for (int64_t qi = 0; qi < nq; qi++) {
    const NfaHostResult* r = flat.data() + seg[size_t(qi)];
    int64_t at = result_start[qi];
    for (int64_t k = 0; k < kept[size_t(qi)]; k++, at++) {
      first_out[at] = r[k].first;
      last_out[at] = r[k].last;
      if (len_out) len_out[at] = r[k].len;
      if (cost_out) cost_out[at] = r[k].cost;
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
  if (!q.has_approx) {                   // no APPROX in the text: the arguments decide (an explicit "APPROX 0 ..." keeps exact matching)
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
