// api_internal.hpp -- what the translation units of the C ABI share: the handle (femto_amd_index), the per-call scratch and its
// lease, device buffers, error reporting.  The C ABI is split by concern: femto_amd_api.hip (open / derive / batch pipeline / multi-device),
// regexp_search.hip (NFA search).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is loaded on first use (femto_amd_comm_*), never at link time

#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/femto_amd.h"
#include "device_tables.h"
#include "host_index.hpp"
#include "host_pipeline.hpp"

namespace femto_amd {

int set_err(int code, const std::string& msg);   // thread-local last error (femto_amd_last_error)

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return set_err(e_ == hipErrorOutOfMemory ? FEMTO_AMD_ERR_MEM : FEMTO_AMD_ERR_INVALID,        \
                     std::string(#expr) + ": " + hipGetErrorString(e_));                           \
  } while (0)

constexpr int kGroupW = 32;          // lanes per rank group; a count query uses one wavefront
constexpr int kBlockThreads = 256;

struct DeviceBuffer {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return set_err(FEMTO_AMD_ERR_MEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

// two timing events destroyed on every exit path (the derivations at open return early on any HIP error)
struct EventPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};

struct KernelTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // recorded, not yet read
  std::vector<hipEvent_t> free_list;                       // created once, reused
  double total_ms = 0;
  int64_t launches = 0;
  bool take(hipEvent_t* e0, hipEvent_t* e1) {
    while (free_list.size() < 2) {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) != hipSuccess) return false;
      free_list.push_back(e);
    }
    *e0 = free_list.back(); free_list.pop_back();
    *e1 = free_list.back(); free_list.pop_back();
    return true;
  }
  void give(hipEvent_t e0, hipEvent_t e1) { free_list.push_back(e0); free_list.push_back(e1); }
  void drain() {
    for (auto& pr : events) {
      float ms = 0;
      if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        total_ms += ms;
        launches++;
      }
      give(pr.first, pr.second);
    }
    events.clear();
  }
  void destroy() {
    drain();
    for (hipEvent_t e : free_list) (void)hipEventDestroy(e);
    free_list.clear();
  }
};

// ---- per-call scratch ----------------------------------------------------------------------------------------------
// The reference accepts blocking calls from many threads at once (each request has its own mutex and condition variable,
// src/main/server.c:3732-3793).  Here every call leases a Scratch -- all the device buffers, flags and (for host-pointer
// batches) the stream and pinned staging a call writes -- from a small pool owned by the handle, so concurrent calls
// on one handle never share mutable device state and overlap on the GPU.  An enqueue-only (device-pointer) call
// returns its lease with an event recorded on the caller's stream; the scratch is reused once that event is done.
constexpr int kPipeDepth = 3;            // chunks in flight: one being packed, one on the wire / in the kernel, one coming back
struct HostPipe {
  void* h_in[kPipeDepth] = {};           // [plen i32 x chunk | starts i64 x chunk | symbols u16 x sym_cap]
  void* h_out[kPipeDepth] = {};          // [first i64 x chunk | last i64 x chunk]
  void* d_in[kPipeDepth] = {};
  void* d_out[kPipeDepth] = {};
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;
  hipEvent_t in_done[kPipeDepth] = {}, k_done[kPipeDepth] = {}, out_done[kPipeDepth] = {};
  bool ready = false;
};

struct Scratch {
  DeviceBuffer plen, pats, starts, first, last, noccs, noccs64, out_starts, offsets, scan[3];
  DeviceBuffer rows, ch, occ, off;
  DeviceBuffer keys, keys2, idx, idx2, sorttmp, pairs, tail, bsums;
  // automaton batches (regexp_search.hip): the flattened automata, the workgroups' arenas, the raw results -- kept with the scratch,
  // so that a caller issuing batch after batch (or several callers at once, one scratch each) allocates and frees nothing per call
  // (hipFree waits for the whole device: a caller's return would wait for every other caller's search kernel)
  DeviceBuffer nfa_q, nfa_flags, nfa_sd, nfa_ch, nfa_bychar, nfa_arena, nfa_results, nfa_misc, nfa_order;
  int* d_flags = nullptr;       // [0] error flag, [1] "long ranges" flag of the row expansion, [2] tail item count, [3] see err
  int* err = nullptr;           // where kernels raise "symbol >= ALPHA_SIZE": d_flags (host-pointer calls check and clear it) or,
                                // for enqueue-only calls, d_flags + 3 (nobody reads it: such a pattern just has the empty range)
  int64_t* d_total = nullptr;   // [0] rows to locate, [1] 1 = more rows than the caller's buffer holds
  int64_t* total_user = nullptr;   // the caller's copy of d_total (enqueue-only locate): written by plan_rows_kernel
  int64_t bsums_nblocks = -1;      // the batch size `bsums` is laid out for (PlanSums, direct_kernels.hip.hpp)
  int bsums_parity = 0;            // which set of group sums the current launch accumulates into
  bool bsums_clean = false;        // the other set has been cleared by plan_rows_kernel
  hipStream_t stream = nullptr; // host-pointer calls launch here (non-blocking stream: calls of different threads overlap)
  hipEvent_t done = nullptr;
  bool busy = false, in_flight = false;
  hipStream_t flight_stream = nullptr;   // the stream of the enqueue-only call that used this scratch last
  HostPipe pipe;

  int init() {
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_flags), 8 * sizeof(int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_total), 4 * sizeof(int64_t)));
    // cleared ON the scratch's stream and waited for: a null-stream hipMemset is not ordered with a non-blocking stream,
    // and the memory may be a closed handle's flag word that was still set (seen once as a spurious "character code >=
    // ALPHA_SIZE" on a fresh handle)
    HIP_TRY(hipMemsetAsync(d_flags, 0, 8 * sizeof(int), stream));
    HIP_TRY(hipMemsetAsync(d_total, 0, 4 * sizeof(int64_t), stream));
    HIP_TRY(hipStreamSynchronize(stream));
    err = d_flags;
    return 0;
  }
  void release() {
    for (DeviceBuffer* b : {&plen, &pats, &starts, &first, &last, &noccs, &noccs64, &out_starts, &offsets, &scan[0], &scan[1], &scan[2],
                            &rows, &ch, &occ, &off, &keys, &keys2, &idx, &idx2, &sorttmp, &pairs, &tail, &bsums,
                            &nfa_q, &nfa_flags, &nfa_sd, &nfa_ch, &nfa_bychar, &nfa_arena, &nfa_results, &nfa_misc, &nfa_order})
      b->release();
    for (int b = 0; b < kPipeDepth; b++) {
      if (pipe.h_in[b]) (void)hipHostFree(pipe.h_in[b]);
      if (pipe.h_out[b]) (void)hipHostFree(pipe.h_out[b]);
      if (pipe.d_in[b]) (void)hipFree(pipe.d_in[b]);
      if (pipe.d_out[b]) (void)hipFree(pipe.d_out[b]);
      if (pipe.in_done[b]) (void)hipEventDestroy(pipe.in_done[b]);
      if (pipe.k_done[b]) (void)hipEventDestroy(pipe.k_done[b]);
      if (pipe.out_done[b]) (void)hipEventDestroy(pipe.out_done[b]);
    }
    if (pipe.s_h2d) (void)hipStreamDestroy(pipe.s_h2d);
    if (pipe.s_d2h) (void)hipStreamDestroy(pipe.s_d2h);
    if (d_flags) (void)hipFree(d_flags);
    if (d_total) (void)hipFree(d_total);
    if (stream) (void)hipStreamDestroy(stream);
    if (done) (void)hipEventDestroy(done);
  }
};

}  // namespace femto_amd

using namespace femto_amd;   // internal header: only the C ABI's own translation units include it

inline femto_amd_options_t femto_amd_auto_options() {
  femto_amd_options_t o;
  memset(&o, 0xff, sizeof o);      // every field -1: auto
  o.struct_size = uint32_t(sizeof o);
  return o;
}

struct femto_amd_index {
  HostIndex host;
  femto_amd_options_t opt = femto_amd_auto_options();   // the caller's options (femto_amd_open_opts); -1 = auto
  int64_t hbm_free_at_open = -1; // free HBM when this handle started allocating
  bool budget_is_default = false; // opt.hbm_budget_bytes was left on auto: the default bound is in force
  bool segs_released = false;    // d_segs was dropped after the derivations (release_wavelet_lines); the host copy is the source
  bool segs_drop_ok = false;     // this handle's open released them (a handle with a budget): calls that bring them back give them up
                                 // again when they leave the handle over its budget (WaveletLinesUse)
  int segs_users = 0;            // calls in flight that read them (under mu)
  int64_t hbm_held = 0;          // bytes of the handle's PERSISTENT device allocations (big arrays + uploaded tables): what
                                 // hbm_budget_bytes is counted against -- the scratch of the derivations at open comes and goes
  std::vector<std::pair<void*, size_t>> big_allocs;   // big_malloc()ed arrays and their sizes
  int device = -1;
  std::mutex mu;   // mode switches, timers
  // device-resident index
  uint8_t* d_image = nullptr;
  DevNode* d_nodes = nullptr;
  DevBucket* d_buckets = nullptr;
  DevSeq* d_seqs = nullptr;
  int64_t* d_occ_base = nullptr;
  uint32_t* d_leaf_code = nullptr;
  int64_t* d_C = nullptr;
  uint64_t* d_segs = nullptr;
  CumEntry* d_cum = nullptr;
  uint32_t* d_hint = nullptr;
  BlockDir* d_bdir = nullptr;
  LaneNode* d_lnodes = nullptr;
  LaneSeq* d_lseqs = nullptr;
  OccEntry* d_occ = nullptr;
  int mode = 1;  // 3: packed small-alphabet lines (default when the index has <= 8 characters); 4: two-level lines (<= 256
                 // characters); 1: lane-per-query kernels on femto's wavelet tree (default otherwise); 0: wavefront-cooperative
                 // walk of femto's raw A/S/D tables
  uint32_t* d_pack = nullptr;
  int64_t* d_pack_sa = nullptr;
  uint8_t* d_pack_code = nullptr;
  int64_t* d_pack_c = nullptr;
  int64_t* d_ktab2 = nullptr;
  int64_t* d_ktab2_deep = nullptr;
  int64_t ktab2_bytes = 0;
  uint64_t* d_ctx = nullptr;     // context tables of byte alphabets (ctx_kernels.hip.hpp)
  uint64_t* d_ctx2 = nullptr;
  uint64_t* d_ctxm = nullptr;
  int64_t ctx_bytes = 0, ctx_entries = 0, ctx2_bytes = 0, ctxm_bytes = 0;
  double ctx_build_ms = 0;
  uint8_t* d_txt = nullptr;
  int64_t* d_isa8 = nullptr;
  int64_t* d_sa_full = nullptr;
  uint32_t* d_ind = nullptr;
  int64_t ind_bytes = 0;
  int64_t text_bytes = 0;
  int64_t n_marks = 0;         // entries of pack_sa
  int64_t p2_lines1 = 0, p2_lines2 = 0;
  uint32_t *d_p2_l1 = nullptr, *d_p2_l2 = nullptr;
  int64_t *d_p2_base = nullptr, *d_p2_c = nullptr;
  uint16_t *d_p2_code = nullptr, *d_p2_alpha = nullptr;
  int64_t pack2_bytes = 0;
  double pack2_build_ms = 0;
  int64_t pack_bytes = 0;
  double pack_build_ms = 0;
  int64_t* d_ru_stop = nullptr;  // rows of the stop characters (ru_stop_step)
  uint64_t* d_ru = nullptr;         // rank units of small alphabets (ru_kernels.hip.hpp)
  int64_t ru_bytes = 0;
  int64_t last_trace_reads[2][16] = {};   // femto_amd_trace_lines: line READS of the count / the locate phase, per region
  int64_t marks_bytes = 0;       // pack_sa
  int mark_every_used = 0;       // distance between derived marks (0: femto's own marks)
  int num_cus = 256;
  DevIndex dev{};
  int64_t table_bytes = 0;
  DeviceBuffer open_scan[3];   // scan scratch of the derivations at open
  // scratch pool (see Scratch)
  std::mutex pool_mu;
  std::condition_variable pool_cv;
  std::vector<std::unique_ptr<Scratch>> pool;
  int pool_max = 8;
  std::unique_ptr<WorkerPool> workers;   // staging threads of host-pointer batches, created on first use
  std::mutex workers_mu;                 // one staged batch at a time uses the worker pool
  bool sort_queries = true;    // FEMTO_AMD_SORT=0 disables the suffix-order batch sort of the paths that use one
  uint8_t* d_dense = nullptr;  // alpha code -> dense sort digit (characters present in the text)
  std::vector<uint8_t> h_dense; // the same table on the host (key staging of host-pointer batches)
  std::vector<uint8_t> h_dense16; // ... indexed by any 16-bit symbol value (built on first use, under workers_mu)
  int dense_bits = 8;
  double dense_sigma = 256;    // distinct characters of the indexed text
  int64_t sort_min = 4096;
  bool trace_row_free = false;   // femto_amd_trace_lines follows the row-free form of the chain (option "trace_row_free")
  int64_t regexp_max_iterations = 1000000;   // MAX_REGEXP_ITERATIONS (src/main/server.c:40); option "regexp_max_iterations"
  int64_t regexp_stack_cap = int64_t(1) << 18; // pending ranges one search may hold (option "regexp_stack_cap", <= 2^22); the reference
                                               // has no bound: it runs on to ERR_OVERWORKED
  bool timing = false;
  KernelTimer t_count, t_locate, t_resolve, t_regexp;
  int64_t* d_doc_ends = nullptr;   // the header block's doc_ends[] on the device (resolve.hip; uploaded on first use)
  double nfa_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // the last automaton batch (femto_amd_nfa_stats)
  int32_t* nfa_active = nullptr;       // pinned host word: automaton batches in flight on this handle (regexp_search.hip "FAIR SHARE") ...
  int32_t* nfa_active_dev = nullptr;   // ... as the kernels address it
  double pipe_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the last staged host-pointer call (femto_amd_host_pipeline_stats)
  // range-split index (femto_amd_open_split): this handle holds the segment lines and the block images of
  // data blocks [split_blo[part], split_blo[part+1]); the other parts' slices are mapped from their owners
  int split_parts = 0, split_part = 0;
  bool split_ready = false;
  std::vector<int64_t> split_blo;        // nparts + 1 block boundaries
  std::vector<void*> peer_segs, peer_image;  // per part: base of that part's slices as seen from this process
  std::vector<char> peer_ipc;            // per part: 1 if opened with hipIpcOpenMemHandle (close on release)
  int64_t split_seg_bytes = 0, split_image_bytes = 0;
  // multi-device handle (femto_amd_open_multi): no device of its own, one replica per GPU; host-pointer batches are
  // sharded over the replicas by host threads
  std::vector<femto_amd_index*> children;
  // striped index (femto_amd_open_multi_striped): the big arrays are ONE address range each whose pages live in the HBM
  // of all the listed GPUs (HIP virtual memory management); the small tables are copied to every GPU
  std::vector<int> stripe_devices;           // non-empty while the builder handle allocates
  struct Striped { void* va; size_t size, chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
  std::vector<Striped> striped;
  std::vector<std::pair<void*, size_t>> small_tables;   // every upload()ed table: what a view on another GPU copies
  bool borrowed = false;                     // a view of another handle's arrays on a second GPU: owns only `owned_small`
  bool imported = false;                     // ... of another PROCESS's arrays: also owns its mappings of them (`striped`)
  std::vector<void*> owned_small;
  // RCCL communicator of the multi-process form (femto_amd_comm_init)
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_size = 0;
};

namespace femto_amd {

Scratch* scratch_acquire(femto_amd_index* ix, int* rc, bool enqueue_only = false, hipStream_t same_stream = nullptr);
void scratch_release(femto_amd_index* ix, Scratch* s, bool async, hipStream_t stream);

struct Lease {
  femto_amd_index* ix;
  Scratch* s = nullptr;
  bool async = false;           // enqueue-only call: the work is still running when the lease ends
  hipStream_t stream = nullptr;
  int rc = 0;
  explicit Lease(femto_amd_index* i) : ix(i) { s = scratch_acquire(ix, &rc); }
  Lease(femto_amd_index* i, hipStream_t st) : ix(i) {     // enqueue-only call on the caller's stream `st`
    s = scratch_acquire(ix, &rc, true, st);
    if (s) enqueue_only(st);
  }
  ~Lease() {
    if (s) s->err = s->d_flags;
    scratch_release(ix, s, async, stream);
  }
  // enqueue-only call on the caller's stream: nothing of it is checked on the host, so its kernels raise the error flag
  // in a word of their own (a later host-pointer call on this scratch must not inherit it)
  void enqueue_only(hipStream_t st) {
    async = true;
    stream = st;
    s->err = s->d_flags + 3;
  }
  Lease(const Lease&) = delete;
  Lease& operator=(const Lease&) = delete;
};
// `slack` zero bytes follow the data: a damaged index (counts that disagree with the bits they summarise) can make a
// kernel index a little past the end of the table it is walking -- at most one bucket's worth -- and must read
// zeros there, not fault.  (Results for such an index are garbage either way, as they are in the reference.)
extern thread_local std::vector<std::pair<void*, size_t>>* g_small_registry;   // set while a handle is being opened

template <class T>
int upload(T** dst, const std::vector<T>& src, int64_t* bytes, size_t slack = 0) {
  size_t n = src.size() * sizeof(T);
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), n + slack ? n + slack : 16));
  if (n) HIP_TRY(hipMemcpy(*dst, src.data(), n, hipMemcpyHostToDevice));
  if (slack) HIP_TRY(hipMemset(reinterpret_cast<char*>(*dst) + n, 0, slack));
  if (bytes) *bytes += int64_t(n);
  if (g_small_registry) g_small_registry->emplace_back(static_cast<void*>(*dst), (n + slack) ? n + slack : size_t(16));
  return 0;
}


// A knob's value: the caller's option when it is set, else the FEMTO_AMD_* test override, else the built-in rule (dflt).
int64_t knob(int64_t opt_value, const char* env_name, int64_t dflt);
// free HBM as far as this handle may use it: what the device has free, less whatever hbm_budget_bytes forbids
size_t hbm_free(const femto_amd_index* ix);
void ensure_workers(femto_amd_index* ix);         // creates ix->workers (api_host.hip) if it does not exist yet
int release_wavelet_lines(femto_amd_index* ix);   // a handle with a budget drops femto's segment lines once the derived layouts stand ...
int ensure_wavelet_lines(femto_amd_index* ix);    // ... and calls that need them (modes 0/1, forward steps) bring them back (caller holds ix->mu or is the only user)
int64_t hbm_held_all(const femto_amd_index* ix);  // big arrays + uploaded small tables: what hbm_budget_bytes is counted against
// A call that reads femto's own segment lines on a handle in modes 3 / 4 (LOCATION leaf requests, forward steps): brings them back
// under ix->mu for the duration of the call and -- when the upload left a bounded handle OVER its budget ("bytes this handle may
// HOLD in all", include/femto_amd.h) -- gives them up again once the last such call has ended.  Handles whose rank mode is 0 / 1
// keep them (their kernels read nothing else); femto_amd_set_rank_mode drops them on the way back to modes 3 / 4 by the same rule.
struct WaveletLinesUse {
  femto_amd_index* ix;
  bool held = false;
  explicit WaveletLinesUse(femto_amd_index* i) : ix(i) {}
  int acquire();
  ~WaveletLinesUse();
  WaveletLinesUse(const WaveletLinesUse&) = delete;
  WaveletLinesUse& operator=(const WaveletLinesUse&) = delete;
};
hipError_t big_malloc(femto_amd_index* ix, void** out, size_t bytes);
void big_free(femto_amd_index* ix, void* p);
hipError_t big_memset(femto_amd_index* ix, void* p, int v, size_t bytes);
hipError_t big_h2d(femto_amd_index* ix, void* p, const void* src, size_t bytes);
int ensure_device(femto_amd_index* ix);
void comm_destroy(femto_amd_index* ix);     // api_multi.hip: the RCCL communicator of femto_amd_comm_init, if any
// femto_amd_open and its variants: part / nparts: a range-split part; stripe: the big arrays over these GPUs' HBM
int open_impl(const char* index_path, int device, int part, int nparts, femto_amd_index_t** out, const std::vector<int>* stripe = nullptr,
              const femto_amd_options_t* opts = nullptr);
// ---- the launch layer (femto_amd_api.hip, the translation unit that holds the query kernels) as the host-pointer paths of
// api_host.hip see it
// The locate plan that can ride along with a count: do_locate_query's clamp (src/main/server.c:4405-4415) and the
// exclusive prefix sum of the row counts.  `done` is set when the count path produced noccs[], the block offsets in
// S.bsums and S.d_total (the direct pipeline); otherwise the caller runs clamp_kernel + device_scan.
struct Plan {
  int max_occs;
  int32_t* noccs;        // device, npats
  int64_t* out_starts;   // device, npats + 1
  int64_t capacity;      // rows the caller's offsets buffer holds (INT64_MAX when it is sized afterwards)
  bool done;
  int64_t* total_user = nullptr;   // device, 2 words: the caller's copy of S.d_total, written by plan_rows_kernel itself
  const int64_t* sa_known = nullptr;   // set by the count: text positions of one-row patterns it already knows (count_direct_kernel's sa_out)
  bool row_free = false;               // the caller takes no rows (parallel_locate's contract): DevIndex::row_free for this launch
};
// modes 3 / 4: the caller-order pipeline of direct_kernels.hip.hpp (with or without a level table)
inline bool use_direct(const femto_amd_index* ix) { return ix->mode == 3 || ix->mode == 4; }
int launch_count(femto_amd_index* ix, Scratch& S, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats, const int64_t* d_starts,
                 int64_t* d_first, int64_t* d_last, hipStream_t stream, Plan* plan = nullptr);
int launch_count_keys(femto_amd_index* ix, int64_t n, const uint64_t* d_keys, int2* out32, int64_t* d_first, int64_t* d_last, hipStream_t stream,
                      Scratch* S = nullptr, Plan* plan = nullptr);
int launch_count_plan(femto_amd_index* ix, Scratch& S, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats, const int64_t* d_starts,
                      int64_t* d_first, int64_t* d_last, Plan* plan, hipStream_t stream);
int launch_plan_rows(femto_amd_index* ix, Scratch& S, int64_t npats, const int32_t* d_noccs, const int64_t* d_first, int64_t* d_out_starts,
                     int64_t* d_offsets, int64_t capacity, hipStream_t stream, const int2* d_first32 = nullptr, bool fuse_walk = false,
                     const int64_t* d_sa_known = nullptr);
int launch_walk_device_total(femto_amd_index* ix, Scratch& S, int64_t* d_offsets, int64_t capacity, hipStream_t stream);
int launch_locate(femto_amd_index* ix, Scratch& S, int64_t npats, const int64_t* d_first, const int64_t* d_out_starts, int64_t total,
                  int64_t* d_offsets, hipStream_t stream);
int launch_clamp(int64_t npats, const int64_t* d_first, const int64_t* d_last, int max_occs, int32_t* d_noccs, int64_t* d_noccs64, hipStream_t stream);
// exclusive prefix sum of n 64-bit counts on the device (out has n + 1 entries); femto_amd_api.hip
int device_scan(DeviceBuffer* scan, int64_t n, const int64_t* in, int64_t* out, int level, hipStream_t stream);
void block_image_range(const HostIndex& h, int64_t b0, int64_t b1, uint64_t* lo, uint64_t* hi);   // image bytes of data blocks [b0, b1)
int check_err_flag(Scratch& S, hipStream_t stream);
bool timer_begin(femto_amd_index* ix, KernelTimer& t, hipStream_t stream, hipEvent_t* e0, hipEvent_t* e1);
void timer_end(femto_amd_index* ix, KernelTimer& t, hipStream_t stream, hipEvent_t e0, hipEvent_t e1);

}  // namespace femto_amd

// no exception crosses the C boundary
#define API_BEGIN try {
#define API_END                                                                            \
  } catch (const std::bad_alloc&) {                                                        \
    return set_err(FEMTO_AMD_ERR_MEM, "out of memory");                                    \
  } catch (const std::exception& ex) {                                                     \
    return set_err(FEMTO_AMD_ERR_INVALID, std::string("internal error: ") + ex.what());    \
  } catch (...) {                                                                          \
    return set_err(FEMTO_AMD_ERR_INVALID, "internal error");                               \
  }
