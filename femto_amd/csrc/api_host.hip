// api_host.hip -- HOST-pointer batches: the reference's own calling convention (src/main/femto.c:275-386: pageable arrays in,
// results in the caller's arrays, offsets[i] malloc()ed per pattern) staged onto the GPU.  Split off femto_amd_api.hip in round 4:
// this file owns the staging pipeline (worker pool packing patterns into pinned key / symbol chunks while the previous
// chunks travel and are searched, host_pipeline.hpp), the staged return of large results and the sharding of a batch
// over the replicas of a multi-device handle; it launches no kernel itself -- the launch layer (launch_count ...,
// api_internal.hpp) lives with the kernels in femto_amd_api.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "api_internal.hpp"
#include "host_pack.hpp"
#include "../../include/femto_amd.h"
#include "host_pipeline.hpp"

using namespace femto_amd;

namespace {

int validate_patterns(int64_t npats, const int32_t* plen, const int64_t* starts) {
  if (npats < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative pattern count");
  if (npats && (!plen || !starts)) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern arrays");
  for (int64_t i = 0; i < npats; i++)
    if (plen[i] < 0 || starts[i] < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative pattern length/start");
  return 0;
}

// copies a flat host pattern set to device scratch
int stage_patterns(Scratch& S, int64_t npats, const int32_t* plen, const uint16_t* pats, const int64_t* starts) {
  int rc = validate_patterns(npats, plen, starts);
  if (rc) return rc;
  int64_t total = 0;
  for (int64_t i = 0; i < npats; i++) total = std::max<int64_t>(total, starts[i] + plen[i]);
  if (total && !pats) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern symbols");
  if ((rc = S.plen.reserve(size_t(npats + 1) * 4))) return rc;
  if ((rc = S.starts.reserve(size_t(npats + 1) * 8))) return rc;
  if ((rc = S.pats.reserve(size_t(total + 4) * 2))) return rc;
  if (npats) {
    HIP_TRY(hipMemcpyAsync(S.plen.p, plen, size_t(npats) * 4, hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipMemcpyAsync(S.starts.p, starts, size_t(npats) * 8, hipMemcpyHostToDevice, S.stream));
  }
  if (total) HIP_TRY(hipMemcpyAsync(S.pats.p, pats, size_t(total) * 2, hipMemcpyHostToDevice, S.stream));
  return 0;
}

// ---- host-pointer batches, pipelined ----------------------------------------------------------------------------
constexpr int64_t kPipeChunk = int64_t(1) << 21;      // patterns per chunk
constexpr int64_t kPipeSymCap = int64_t(1) << 26;     // symbols per chunk (128 MB)
constexpr int64_t kPipeMin = int64_t(1) << 18;        // smaller batches take the plain path

// CPUs the process may use on average: cgroup v2 cpu.max ("<quota> <period>" or "max ..."), v1 cpu.cfs_quota_us / _period_us; 0: none
double cgroup_cpu_quota() {
  double q = 0, per = 0;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char buf[64] = {0};
    const int n = fscanf(f, "%63s %lf", buf, &per);
    fclose(f);
    if (n == 2 && strcmp(buf, "max") != 0 && per > 0) return atof(buf) / per;
    return 0;
  }
  FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
  FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
  if (fq && fp && fscanf(fq, "%lf", &q) == 1 && fscanf(fp, "%lf", &per) == 1 && q > 0 && per > 0) q = q / per;
  else q = 0;
  if (fq) fclose(fq);
  if (fp) fclose(fp);
  return q;
}

size_t pipe_in_bytes() { return size_t(kPipeChunk) * 12 + size_t(kPipeSymCap) * 2 + 64; }

}  // namespace
// the handle's host worker pool (staging threads of host-pointer batches; the automaton batch's result sort), created on first use
void femto_amd::ensure_workers(femto_amd_index* ix) {
  {
    std::lock_guard<std::mutex> lk(ix->workers_mu);
    if (!ix->workers) {
      // half of the host's hardware threads, at least 4, at most 128: packing 10 M patterns into keys is ~200 M table
      // look-ups, and the staging threads -- not PCIe, not the GPU -- bound this path (measured on the GPU box's
      // 256-thread host, 10 M 20-mers: 16 threads 12.7 ms, 32 9.2 ms, 64 6-9 ms, 128 5.6 ms)
      int nthreads = std::max(4, int(std::thread::hardware_concurrency()) / 2);
      // ... unless the process runs under a CPU quota (cgroup cpu.max): threads beyond about twice the quota only burn it, and a
      // process that has spent its quota stands still until the period ends.  Measured on the GPU box (256 hardware threads,
      // quota 16 CPUs, 20 calls of 10 M 20-mers back to back): 128 threads 4.1 ms best / 38 ms mean (calls of 50-100 ms),
      // 64 threads 4.7 / 16.4, 32 threads 4.0 / 6.6, 16 threads 6.9 / 7.4 -- packing is memory-bound, 32 threads pack as fast as 128.
      if (const double quota = cgroup_cpu_quota(); quota > 0) nthreads = std::min(nthreads, std::max(8, int(2.0 * quota + 0.5)));
      nthreads = int(knob(ix->opt.host_threads, "FEMTO_AMD_HOST_THREADS", nthreads));
      nthreads = std::max(1, std::min(nthreads, 128));
      ix->workers.reset(new WorkerPool(nthreads));
    }
  }
}
namespace {

int pipe_init(femto_amd_index* ix, Scratch& S) {
  ensure_workers(ix);
  auto& P = S.pipe;
  if (P.ready) return 0;
  // The staging buffers (kPipeDepth x ~140 MB pinned host + as much device memory per leased scratch) live until the handle is
  // closed; they belong to a CALL's scratch, like the pattern and result buffers of the plain path, and are not part of what
  // hbm_budget_bytes / femto_amd_structures[13] count (the index's own structures).  When any of them cannot be had, what
  // was allocated is released again and the caller takes the unpipelined path (-1), which needs none of it.
  bool ok = true;
  for (int b = 0; b < kPipeDepth && ok; b++) {
    ok = hipHostMalloc(&P.h_in[b], pipe_in_bytes(), hipHostMallocDefault) == hipSuccess &&
         hipHostMalloc(&P.h_out[b], size_t(kPipeChunk) * 16, hipHostMallocDefault) == hipSuccess &&
         hipMalloc(&P.d_in[b], pipe_in_bytes()) == hipSuccess && hipMalloc(&P.d_out[b], size_t(kPipeChunk) * 16) == hipSuccess &&
         hipEventCreateWithFlags(&P.in_done[b], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&P.k_done[b], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&P.out_done[b], hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipStreamCreateWithFlags(&P.s_h2d, hipStreamNonBlocking) == hipSuccess &&
       hipStreamCreateWithFlags(&P.s_d2h, hipStreamNonBlocking) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    for (int b = 0; b < kPipeDepth; b++) {
      if (P.h_in[b]) (void)hipHostFree(P.h_in[b]);
      if (P.h_out[b]) (void)hipHostFree(P.h_out[b]);
      if (P.d_in[b]) (void)hipFree(P.d_in[b]);
      if (P.d_out[b]) (void)hipFree(P.d_out[b]);
      if (P.in_done[b]) (void)hipEventDestroy(P.in_done[b]);
      if (P.k_done[b]) (void)hipEventDestroy(P.k_done[b]);
      if (P.out_done[b]) (void)hipEventDestroy(P.out_done[b]);
      P.h_in[b] = P.h_out[b] = P.d_in[b] = P.d_out[b] = nullptr;
      P.in_done[b] = P.k_done[b] = P.out_done[b] = nullptr;
    }
    if (P.s_h2d) (void)hipStreamDestroy(P.s_h2d);
    if (P.s_d2h) (void)hipStreamDestroy(P.s_d2h);
    P.s_h2d = P.s_d2h = nullptr;
    return -1;
  }
  P.ready = true;
  return 0;
}

// One chunk's patterns, in either calling convention, copied into a pinned buffer.  Returns the number of symbols
// staged, -1 when the chunk does not fit the buffer (the caller falls back to the unpipelined path), or -2 - code
// on invalid input.
struct HostBatch {
  int64_t npats = 0;
  const int32_t* plen = nullptr;
  const uint16_t* flat = nullptr;          // flat form: pattern i = flat[starts[i] .. +plen[i])
  const int64_t* starts = nullptr;
  const uint16_t* const* ptrs = nullptr;   // pointer-array form (parallel_count's alpha_t**)
};

int64_t pipe_stage(femto_amd_index* ix, const HostBatch& hb, int64_t a, int64_t b, void* h_in) {
  WorkerPool& pool = *ix->workers;
  std::lock_guard<std::mutex> wl(ix->workers_mu);   // the pool runs one job at a time; concurrent batches take turns per chunk
  const int64_t n = b - a;
  int32_t* o_plen = static_cast<int32_t*>(h_in);
  int64_t* o_starts = reinterpret_cast<int64_t*>(static_cast<char*>(h_in) + size_t(kPipeChunk) * 4);
  uint16_t* o_sym = reinterpret_cast<uint16_t*>(static_cast<char*>(h_in) + size_t(kPipeChunk) * 12);
  const int T = pool.size();
  std::vector<int64_t> part(size_t(T) + 1, 0), lo_t(size_t(T), INT64_MAX), hi_t(size_t(T), 0);
  std::vector<int> bad(size_t(T), 0);
  // pass 1: validate; flat form: symbol range of the chunk; pointer form: symbols per thread slice
  pool.run([&](int t, int nt) {
    const int64_t i0 = a + n * t / nt, i1 = a + n * (t + 1) / nt;
    int64_t sum = 0, lo = INT64_MAX, hi = 0;
    for (int64_t i = i0; i < i1; i++) {
      const int64_t l = hb.plen[i];
      if (l < 0) { bad[size_t(t)] = 1; return; }
      if (hb.ptrs) {
        if (l && !hb.ptrs[i]) { bad[size_t(t)] = 1; return; }
        sum += l;
      } else {
        const int64_t s0 = hb.starts[i];
        if (s0 < 0) { bad[size_t(t)] = 1; return; }
        lo = std::min(lo, s0);
        hi = std::max(hi, s0 + l);
      }
    }
    part[size_t(t) + 1] = sum;
    lo_t[size_t(t)] = lo;
    hi_t[size_t(t)] = hi;
  });
  for (int t = 0; t < T; t++) if (bad[size_t(t)]) return -2 - FEMTO_AMD_ERR_PARAM;
  int64_t nsym, lo = 0;
  if (hb.ptrs) {
    for (int t = 0; t < T; t++) part[size_t(t) + 1] += part[size_t(t)];
    nsym = part[size_t(T)];
  } else {
    lo = INT64_MAX;
    int64_t hi = 0;
    for (int t = 0; t < T; t++) { lo = std::min(lo, lo_t[size_t(t)]); hi = std::max(hi, hi_t[size_t(t)]); }
    if (lo == INT64_MAX) lo = 0;
    nsym = std::max<int64_t>(0, hi - lo);
  }
  if (nsym > kPipeSymCap) return -1;
  // pass 2: copy
  pool.run([&](int t, int nt) {
    const int64_t i0 = a + n * t / nt, i1 = a + n * (t + 1) / nt;
    if (hb.ptrs) {
      int64_t at = part[size_t(t)];
      for (int64_t i = i0; i < i1; i++) {  // patterns are copied, as setup_string_query does (src/main/server.c:691-695)
        const int64_t l = hb.plen[i];
        o_plen[i - a] = int32_t(l);
        o_starts[i - a] = at;
        if (l) memcpy(o_sym + at, hb.ptrs[i], size_t(l) * 2);
        at += l;
      }
    } else {
      memcpy(o_plen + (i0 - a), hb.plen + i0, size_t(i1 - i0) * 4);
      for (int64_t i = i0; i < i1; i++) o_starts[i - a] = hb.starts[i] - lo;
      const int64_t s0 = nsym * t / nt, s1 = nsym * (t + 1) / nt;
      if (s1 > s0) memcpy(o_sym + s0, hb.flat + lo + s0, size_t(s1 - s0) * 2);
    }
  });
  return nsym;
}

// Key staging (count_keys_kernel): every pattern of the chunk packed into 8 bytes.  Returns 1 when all of them are
// described completely by their keys (written to h_in as u64[n]), 0 when some pattern is not (the chunk then travels as
// symbols; malformed input is reported by that path).
int pipe_stage_keys(femto_amd_index* ix, const HostBatch& hb, int64_t a, int64_t b, void* h_in) {
  WorkerPool& pool = *ix->workers;
  std::lock_guard<std::mutex> wl(ix->workers_mu);
  const int64_t n = b - a;
  uint64_t* o_key = static_cast<uint64_t*>(h_in);
  const int bits = ix->dense_bits, nsym = 63 / bits;
  // field of every 16-bit symbol value (0: not a character of the text, or >= ALPHA_SIZE): no bounds test in the loop,
  // no early exit -- a bad symbol is remembered and the chunk given up afterwards (packing is what bounds this path)
  if (ix->h_dense16.empty()) {
    ix->h_dense16.assign(65536, 0);
    for (size_t c = 0; c < ix->h_dense.size() && c < size_t(kAlphaSize); c++) ix->h_dense16[c] = ix->h_dense[c];
  }
  const uint8_t* dense = ix->h_dense16.data();
  const int T = pool.size();
  std::vector<int> partial(size_t(T), 0);
  const PackSource src{hb.plen, hb.ptrs, hb.flat, hb.starts};
  static const int force_scalar = [] { const char* e = getenv("FEMTO_AMD_PACK_SCALAR"); return (e && atoi(e) != 0) ? 1 : 0; }();
  pool.run([&](int t, int nt) {
    const int64_t i0 = a + n * t / nt, i1 = a + n * (t + 1) / nt;
    if (pack_keys_span(src, dense, bits, nsym, i0, i1, o_key + (i0 - a), force_scalar)) partial[size_t(t)] = 1;
  });
  for (int t = 0; t < T; t++) if (partial[size_t(t)]) return 0;
  return 1;
}

// returns 0, an error code, or -1: "not applicable, use the plain path"
// With dev_first != nullptr the ranges stay on the device (whole-batch arrays dev_first / dev_last, the locate plan's
// input) and nothing is copied back.
int count_host_pipelined(femto_amd_index* ix, Scratch& S, const HostBatch& hb, int64_t* first, int64_t* last, int64_t* dev_first = nullptr,
                         int64_t* dev_last = nullptr) {
  if (hb.npats < kPipeMin) return -1;
  if (knob(ix->opt.host_pipeline, "FEMTO_AMD_HOST_PIPELINE", 1) == 0) return -1;
  if (!hb.plen) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern lengths");
  if (!hb.ptrs && (!hb.flat || !hb.starts)) {
    // the flat form without its arrays: only a batch of empty patterns needs neither (the plain path validates the rest)
    for (int64_t i = 0; i < hb.npats; i++)
      if (hb.plen[i] != 0) return set_err(FEMTO_AMD_ERR_PARAM, hb.starts ? "null pattern symbols" : "null pattern starts");
    return -1;
  }
  int rc = pipe_init(ix, S);
  if (rc) return rc;
  auto& P = S.pipe;
  hipStream_t s_k = S.stream;
  // patterns per pipeline stage (the buffers are laid out for kPipeChunk).  Default 2^20: the first chunk's staging and
  // the last chunk's return trip are not overlapped with anything, so smaller stages shorten the call until the per-stage
  // costs take over (10 M 20-mers, 128 staging threads: 2^21 5.5 ms, 2^20 4.0 ms, 2^19 4.6 ms)
  int64_t chunk = kPipeChunk / 2;
  if (const int64_t lg = knob(ix->opt.host_pipe_chunk_log2, "FEMTO_AMD_PIPE_CHUNK_LOG2", -1); lg >= 0)
    chunk = std::min<int64_t>(kPipeChunk, int64_t(1) << std::max<int64_t>(12, std::min<int64_t>(30, lg)));
  const int64_t nchunks = (hb.npats + chunk - 1) / chunk;
  bool keys_ok = use_direct(ix) && !ix->h_dense.empty();
  keys_ok = keys_ok && knob(ix->opt.host_keys, "FEMTO_AMD_HOST_KEYS", 1) != 0;
  const bool rows32 = ix->host.total_length < (int64_t(1) << 31) - 1;    // rows (and last + 1, -1) fit 32 bits
  int kind[kPipeDepth] = {1, 1, 1};   // what h_out[b] holds: 1 int64 arrays, 2 int32 (first,last) pairs, 3 int64 arrays of a key chunk (both present)
  // chunk c - kLag is handed back while chunk c is packed: with kLag = 2 its results have had a whole packing stage more to
  // arrive (kLag = 1, two buffers in use: 0.7-0.9 ms of a 3.7 ms call waited for them).  FEMTO_AMD_PIPE_LAG=1 for A/B runs.
  int kLag = kPipeDepth - 1;
  if (const char* e = getenv("FEMTO_AMD_PIPE_LAG")) kLag = std::max(1, std::min(kPipeDepth - 1, atoi(e)));
  const int depth = kLag + 1;
  // (streaming stores into the caller's arrays measured SLOWER on the box's host: 1.07 instead of 0.80 ms handing 10 M ranges
  // back; kept behind FEMTO_AMD_NT_STORES=1 for other hosts)
  static const bool nt_stores = [] { const char* e = getenv("FEMTO_AMD_NT_STORES"); return e && atoi(e) != 0; }();
  // every exit leaves nothing in flight on the pinned buffers
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(P.s_h2d);
    (void)hipStreamSynchronize(s_k);
    (void)hipStreamSynchronize(P.s_d2h);
    return code;
  };
#define PIPE_TRY(expr)                                                                                              \
  do {                                                                                                              \
    hipError_t e_ = (expr);                                                                                         \
    if (e_ != hipSuccess)                                                                                           \
      return fail(set_err(e_ == hipErrorOutOfMemory ? FEMTO_AMD_ERR_MEM : FEMTO_AMD_ERR_INVALID,                    \
                          std::string(#expr) + ": " + hipGetErrorString(e_)));                                      \
  } while (0)
  // where the call's wall time goes (femto_amd_host_pipeline_stats): [0] staging threads packing the caller's patterns,
  // [1] waiting for a pinned input buffer (its previous chunk's kernel), [2] enqueueing copies / kernels / events,
  // [3] waiting for a chunk's results to arrive, [4] staging threads moving results into the caller's arrays, [5] whole call.
  // Measured on the GPU box (256 hardware threads, 10 M random 20-mers, 3.7 ms per call): packing 1.9, waiting for results
  // 0.7, handing back 0.7, enqueueing 0.3 -- the HOST's packing bounds this path, not PCIe (160 MB both ways: 1.6 ms) and
  // not the kernels (0.4 ms).  A second pool handing results back WHILE the first packs the next chunk made the call
  // slower (4.2 ms: the two compete for the host's memory system, packing rose to 3.3 ms) and was removed.
  using clk = std::chrono::steady_clock;
  auto since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
  double st[6] = {0, 0, 0, 0, 0, 0};
  const clk::time_point t_call = clk::now();
  for (int64_t c = 0; c < nchunks + kLag; c++) {
    if (c < nchunks) {
      const int b = int(c % depth);
      const int64_t a = c * chunk, e = std::min(hb.npats, a + chunk), n = e - a;
      clk::time_point t0 = clk::now();
      if (c >= depth) PIPE_TRY(hipEventSynchronize(P.k_done[b]));  // chunk c-3 no longer reads d_in[b] (and h_in[b] was uploaded)
      st[1] += since(t0);
      t0 = clk::now();
      char* din = static_cast<char*>(P.d_in[b]);
      const char* hin = static_cast<const char*>(P.h_in[b]);
      // keys when every pattern of the chunk fits one (8 B per pattern over PCIe), symbols otherwise
      const bool as_keys = keys_ok && pipe_stage_keys(ix, hb, a, e, P.h_in[b]) == 1;
      const bool out32 = as_keys && !dev_first && rows32;
      kind[b] = out32 ? 2 : (as_keys ? 3 : 1);
      int64_t nsym = 0;
      if (!as_keys) {
        nsym = pipe_stage(ix, hb, a, e, P.h_in[b]);
        if (nsym == -1) return fail(-1);
        if (nsym < -1) return fail(set_err(int(-2 - nsym), "negative pattern length/start or null pattern"));
      }
      st[0] += since(t0);
      t0 = clk::now();
      if (as_keys) {
        PIPE_TRY(hipMemcpyAsync(din, hin, size_t(n) * 8, hipMemcpyHostToDevice, P.s_h2d));
      } else {
        PIPE_TRY(hipMemcpyAsync(din, hin, size_t(n) * 4, hipMemcpyHostToDevice, P.s_h2d));
        PIPE_TRY(hipMemcpyAsync(din + size_t(kPipeChunk) * 4, hin + size_t(kPipeChunk) * 4, size_t(n) * 8, hipMemcpyHostToDevice, P.s_h2d));
        if (nsym)
          PIPE_TRY(hipMemcpyAsync(din + size_t(kPipeChunk) * 12, hin + size_t(kPipeChunk) * 12, size_t(nsym) * 2, hipMemcpyHostToDevice, P.s_h2d));
      }
      PIPE_TRY(hipEventRecord(P.in_done[b], P.s_h2d));
      PIPE_TRY(hipStreamWaitEvent(s_k, P.in_done[b], 0));
      if (c >= depth) PIPE_TRY(hipStreamWaitEvent(s_k, P.out_done[b], 0));  // results of chunk c-3 have left d_out[b]
      int64_t* d_first = static_cast<int64_t*>(P.d_out[b]);
      int64_t* d_last = (last || as_keys) ? d_first + kPipeChunk : nullptr;
      if (dev_first) {
        d_first = dev_first + a;
        d_last = dev_last + a;
      }
      if (as_keys) rc = launch_count_keys(ix, n, reinterpret_cast<const uint64_t*>(din), out32 ? static_cast<int2*>(P.d_out[b]) : nullptr, d_first, d_last, s_k);
      else rc = launch_count(ix, S, n, reinterpret_cast<const int32_t*>(din), reinterpret_cast<const uint16_t*>(din + size_t(kPipeChunk) * 12),
                             reinterpret_cast<const int64_t*>(din + size_t(kPipeChunk) * 4), d_first, d_last, s_k);
      if (rc) return fail(rc);
      PIPE_TRY(hipEventRecord(P.k_done[b], s_k));
      if (dev_first) { st[2] += since(t0); continue; }
      PIPE_TRY(hipStreamWaitEvent(P.s_d2h, P.k_done[b], 0));
      char* hout = static_cast<char*>(P.h_out[b]);
      if (out32) {
        PIPE_TRY(hipMemcpyAsync(hout, P.d_out[b], size_t(n) * 8, hipMemcpyDeviceToHost, P.s_d2h));
      } else {
        PIPE_TRY(hipMemcpyAsync(hout, d_first, size_t(n) * 8, hipMemcpyDeviceToHost, P.s_d2h));
        if (last || as_keys) PIPE_TRY(hipMemcpyAsync(hout + size_t(kPipeChunk) * 8, d_last, size_t(n) * 8, hipMemcpyDeviceToHost, P.s_d2h));
      }
      PIPE_TRY(hipEventRecord(P.out_done[b], P.s_d2h));
      st[2] += since(t0);
    }
    if (c >= kLag && !dev_first) {  // hand chunk c-2 back while chunks c-1 and c are on their way
      const int b = int((c - kLag) % depth);
      const int64_t a = (c - kLag) * chunk, e = std::min(hb.npats, a + chunk), n = e - a;
      clk::time_point t0 = clk::now();
      PIPE_TRY(hipEventSynchronize(P.out_done[b]));
      st[3] += since(t0);
      t0 = clk::now();
      const char* hout = static_cast<const char*>(P.h_out[b]);
      const int k = kind[b];      // still chunk c-2's: chunks c-1 and c went into the other buffers
      std::lock_guard<std::mutex> wl(ix->workers_mu);
      ix->workers->run([&](int t, int nt_all) {
        // a copy: sixteen threads saturate the host's memory (measured: 1.06 ms per 10 M ranges with 32 threads, 1.10 with 16);
        // the others go back to waiting instead of spending a container's CPU quota
        const int nt = nt_all < 16 ? nt_all : 16;
        if (t >= nt) return;
        const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
        if (k == 2) {          // 32-bit (first,last) pairs: widened into the caller's arrays (or the counts, femto.c:313-318)
          const int32_t* pr = reinterpret_cast<const int32_t*>(hout);
          if (nt_stores) {
            for (int64_t i = i0; i < i1; i++) {
              const int64_t f = pr[2 * i], l = pr[2 * i + 1];
              if (last) { __builtin_nontemporal_store(f, first + a + i); __builtin_nontemporal_store(l, last + a + i); }
              else __builtin_nontemporal_store(l - f + 1, first + a + i);
            }
          } else {
            widen_pairs_span(pr + 2 * i0, i1 - i0, first + a + i0, last ? last + a + i0 : nullptr);
          }
        } else if (!last && k == 3) {   // key chunk with 64-bit rows and no `last` array: counts from both
          const int64_t* pf = reinterpret_cast<const int64_t*>(hout);
          const int64_t* pl = reinterpret_cast<const int64_t*>(hout + size_t(kPipeChunk) * 8);
          for (int64_t i = i0; i < i1; i++) first[a + i] = pl[i] - pf[i] + 1;
        } else {
          memcpy(first + a + i0, hout + size_t(i0) * 8, size_t(i1 - i0) * 8);
          if (last) memcpy(last + a + i0, hout + size_t(kPipeChunk) * 8 + size_t(i0) * 8, size_t(i1 - i0) * 8);
        }
      });
      st[4] += since(t0);
    }
  }
  st[5] = since(t_call);
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    for (int k = 0; k < 6; k++) ix->pipe_stats[k] = st[k];
    ix->pipe_stats[6] = double(nchunks);
    ix->pipe_stats[7] = double(ix->workers->size());
  }
  if (getenv("FEMTO_AMD_PIPE_TRACE"))      // a caller that cannot reach femto_amd_host_pipeline_stats (the reference's tools through the shim)
    fprintf(stderr, "[femto_amd] host batch of %lld (%s): pack %.2f  wait-in %.2f  enqueue %.2f  wait-out %.2f  hand back %.2f  call %.2f ms, %lld chunks, %d threads\n",
            (long long)hb.npats, hb.ptrs ? "pointer per pattern" : "flat", st[0], st[1], st[2], st[3], st[4], st[5], (long long)nchunks, ix->workers->size());
#undef PIPE_TRY
  if (dev_first) {
    HIP_TRY(hipStreamSynchronize(s_k));
    return 0;
  }
  return check_err_flag(S, s_k);
}

// count (pipelined staging when the batch is large) + clamp + scan: fills S.first/S.last/S.noccs/S.out_starts and
// S.d_total; with the direct pipeline the rows are expanded into S.offsets as well (*rows_done)
int plan_host(femto_amd_index* ix, Scratch& S, const HostBatch& hb, int max_occs_each, int64_t* total, bool* direct_plan) {
  const int64_t npats = hb.npats;
  hipStream_t st = S.stream;
  int rc;
  *direct_plan = false;
  if ((rc = S.first.reserve(size_t(npats + 1) * 8))) return rc;
  if ((rc = S.last.reserve(size_t(npats + 1) * 8))) return rc;
  if ((rc = S.noccs.reserve(size_t(npats + 1) * 4))) return rc;
  if ((rc = S.out_starts.reserve(size_t(npats + 2) * 8))) return rc;
  Plan plan{max_occs_each, S.noccs.as<int32_t>(), S.out_starts.as<int64_t>(), INT64_MAX, false};
  rc = count_host_pipelined(ix, S, hb, nullptr, S.last.as<int64_t>(), S.first.as<int64_t>(), S.last.as<int64_t>());
  if (rc == -1) {
    if (hb.ptrs) return -1;
    if ((rc = stage_patterns(S, npats, hb.plen, hb.flat, hb.starts))) return rc;
    rc = launch_count_plan(ix, S, npats, S.plen.as<int32_t>(), S.pats.as<uint16_t>(), S.starts.as<int64_t>(),
                           S.first.as<int64_t>(), S.last.as<int64_t>(), &plan, st);
    if (rc) return rc;
  } else {
    if (rc) return rc;
    // the chunks were counted without a plan: clamp + scan over the whole batch
    if ((rc = S.noccs64.reserve(size_t(npats + 1) * 8))) return rc;
    if ((rc = launch_clamp(npats, S.first.as<int64_t>(), S.last.as<int64_t>(), max_occs_each, S.noccs.as<int32_t>(), S.noccs64.as<int64_t>(), st))) return rc;
    if ((rc = device_scan(S.scan, npats, S.noccs64.as<int64_t>(), S.out_starts.as<int64_t>(), 0, st))) return rc;
  }
  if (plan.done) {
    *direct_plan = true;
    if ((rc = launch_plan_rows(ix, S, npats, S.noccs.as<int32_t>(), S.first.as<int64_t>(), S.out_starts.as<int64_t>(), nullptr, INT64_MAX, st))) return rc;   // (rows only: no offsets buffer yet)
  }
  if ((rc = check_err_flag(S, st))) return rc;
  if (max_occs_each == 0 && npats) {
    // The reference fails here: a pattern with more than one match is clamped to an empty locate range and
    // setup_locate_range rejects it (src/main/server.c:4411-4421 -> ERR_PARAM); one match is returned whole.
    std::vector<int64_t> f((size_t(npats))), l((size_t(npats)));
    HIP_TRY(hipMemcpy(f.data(), S.first.p, size_t(npats) * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(l.data(), S.last.p, size_t(npats) * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < npats; i++)
      if (l[size_t(i)] - f[size_t(i)] > 0) return set_err(FEMTO_AMD_ERR_PARAM, "max_occs_each == 0 with a multi-match pattern: Error during query processing");
  }
  HIP_TRY(hipMemcpy(total, S.out_starts.as<int64_t>() + npats, 8, hipMemcpyDeviceToHost));
  return 0;
}

// Results of a large host-pointer batch back into the caller's PAGEABLE memory: a plain hipMemcpy stages through the
// runtime's own bounce buffer on one thread (~10 GB/s, and a freshly malloc()ed destination faults its pages in on that
// thread); here the copy runs in 32 MB pieces into the call's two pinned buffers while the staging threads move the
// previous piece into place (parallel_locate's offsets of 10 M located rows: 80 MB).  Everything enqueued on S.stream so
// far is waited for; blocking.
int d2h_staged(femto_amd_index* ix, Scratch& S, void* dst, const void* d_src, size_t bytes) {
  auto& P = S.pipe;
  const size_t piece = size_t(kPipeChunk) * 16;
  bool staged = P.ready && ix->workers && bytes >= (size_t(4) << 20);
  staged = staged && knob(ix->opt.host_d2h_staged, "FEMTO_AMD_D2H_STAGED", 1) != 0;
  if (!staged) {
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    return 0;
  }
  HIP_TRY(hipEventRecord(P.k_done[0], S.stream));
  HIP_TRY(hipStreamWaitEvent(P.s_d2h, P.k_done[0], 0));
  const size_t npieces = (bytes + piece - 1) / piece;
  auto fail = [&](hipError_t e) {
    (void)hipStreamSynchronize(P.s_d2h);
    return set_err(FEMTO_AMD_ERR_INVALID, std::string("staged copy to the host: ") + hipGetErrorString(e));
  };
  for (size_t c = 0; c <= npieces; c++) {
    if (c < npieces) {
      const int b = int(c & 1);
      const size_t off = c * piece, len = std::min(piece, bytes - off);
      hipError_t e = hipMemcpyAsync(P.h_out[b], static_cast<const char*>(d_src) + off, len, hipMemcpyDeviceToHost, P.s_d2h);
      if (e == hipSuccess) e = hipEventRecord(P.out_done[b], P.s_d2h);
      if (e != hipSuccess) return fail(e);
    }
    if (c >= 1) {      // piece c-1 into the caller's memory while piece c is on its way
      const int b = int((c - 1) & 1);
      const size_t off = (c - 1) * piece, len = std::min(piece, bytes - off);
      const hipError_t e = hipEventSynchronize(P.out_done[b]);
      if (e != hipSuccess) return fail(e);
      const char* src = static_cast<const char*>(P.h_out[b]);
      char* out = static_cast<char*>(dst) + off;
      std::lock_guard<std::mutex> wl(ix->workers_mu);
      ix->workers->run([&](int t, int nt) {
        const size_t i0 = (len * size_t(t) / size_t(nt)) & ~size_t(63), i1 = t + 1 == nt ? len : (len * size_t(t + 1) / size_t(nt)) & ~size_t(63);
        if (i1 > i0) memcpy(out + i0, src + i0, i1 - i0);
      });
    }
  }
  return 0;
}

// plan, walk, offsets copied to `dst` (host, room for the total) -- shared by the flat and the malloc forms
int walk_to_host(femto_amd_index* ix, Scratch& S, int64_t npats, int64_t total, int64_t* dst) {
  int rc;
  if ((rc = S.offsets.reserve(size_t(total) * 8))) return rc;
  if ((rc = launch_locate(ix, S, npats, S.first.as<int64_t>(), S.out_starts.as<int64_t>(), total, S.offsets.as<int64_t>(), S.stream))) return rc;
  return d2h_staged(ix, S, dst, S.offsets.p, size_t(total) * 8);
}

// one pass: plan, walk, offsets returned in one malloc()ed array (caller frees); noccs / out_starts optional
int locate_host(femto_amd_index* ix, Scratch& S, const HostBatch& hb, int max_occs_each, int32_t* noccs, int64_t* out_starts, int64_t** offsets_out,
                int64_t* total_out) {
  int64_t total = 0;
  bool direct_plan = false;
  int rc = plan_host(ix, S, hb, max_occs_each, &total, &direct_plan);
  if (rc) return rc;
  const int64_t npats = hb.npats;
  if (total_out) *total_out = total;
  if (noccs && npats && (rc = d2h_staged(ix, S, noccs, S.noccs.p, size_t(npats) * 4))) return rc;
  if (out_starts && (rc = d2h_staged(ix, S, out_starts, S.out_starts.p, size_t(npats + 1) * 8))) return rc;
  *offsets_out = nullptr;
  if (total == 0) return 0;
  int64_t* buf = static_cast<int64_t*>(malloc(size_t(total) * 8));
  if (!buf) return set_err(FEMTO_AMD_ERR_MEM, "malloc failed");
  if ((rc = walk_to_host(ix, S, npats, total, buf))) {
    free(buf);
    return rc;
  }
  *offsets_out = buf;
  return 0;
}

}  // namespace
namespace {

// ---- multi-device handle: contiguous shards of a host-pointer batch, one host thread per replica --------------------
// fn(child, lo, hi) runs the ordinary single-device call on patterns [lo, hi); the first failure is reported.
template <class Fn>
int multi_run(femto_amd_index* ix, int64_t npats, Fn fn) {
  const int N = int(ix->children.size());
  std::vector<int> rcs(size_t(N), 0);
  std::vector<std::string> msgs((size_t(N)));
  std::vector<std::thread> th;
  for (int i = 0; i < N; i++)
    th.emplace_back([&, i] {
      const int64_t lo = npats * i / N, hi = npats * (i + 1) / N;
      try {
        rcs[size_t(i)] = fn(ix->children[size_t(i)], i, lo, hi);
      } catch (...) {
        rcs[size_t(i)] = FEMTO_AMD_ERR_INVALID;
      }
      if (rcs[size_t(i)]) msgs[size_t(i)] = femto_amd_last_error();   // thread-local message of the failing call
    });
  for (auto& t : th) t.join();
  for (int i = 0; i < N; i++)
    if (rcs[size_t(i)]) return set_err(rcs[size_t(i)], "device shard " + std::to_string(i) + ": " + msgs[size_t(i)]);
  return 0;
}


}  // namespace

extern "C" {

int femto_amd_count_flat(femto_amd_index_t* ix, int64_t npats, const int32_t* plen, const uint16_t* pats,
                         const int64_t* starts, int64_t* first, int64_t* last) {
  API_BEGIN
  if (!ix || (npats && !first)) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (!ix->children.empty()) {   // multi-device handle: contiguous shards, one host thread per GPU
    if (npats < 0 || (npats && (!plen || !starts))) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern arrays");
    return multi_run(ix, npats, [&](femto_amd_index* c, int, int64_t lo, int64_t hi) {
      return femto_amd_count_flat(c, hi - lo, plen + lo, pats, starts + lo, first + lo, last ? last + lo : nullptr);
    });
  }
  int rc = ensure_device(ix);
  if (rc) return rc;
  Lease L(ix);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  if (npats && plen && starts && pats) {
    HostBatch hb;
    hb.npats = npats;
    hb.plen = plen;
    hb.flat = pats;
    hb.starts = starts;
    rc = count_host_pipelined(ix, S, hb, first, last);
    if (rc != -1) return rc;
  }
  if ((rc = stage_patterns(S, npats, plen, pats, starts))) return rc;
  if ((rc = S.first.reserve(size_t(npats + 1) * 8))) return rc;
  if ((rc = S.last.reserve(size_t(npats + 1) * 8))) return rc;
  rc = launch_count(ix, S, npats, S.plen.as<int32_t>(), S.pats.as<uint16_t>(), S.starts.as<int64_t>(),
                    S.first.as<int64_t>(), last ? S.last.as<int64_t>() : nullptr, S.stream);
  if (rc) return rc;
  if ((rc = check_err_flag(S, S.stream))) return rc;
  if (npats) {
    HIP_TRY(hipMemcpyAsync(first, S.first.p, size_t(npats) * 8, hipMemcpyDeviceToHost, S.stream));
    if (last) HIP_TRY(hipMemcpyAsync(last, S.last.p, size_t(npats) * 8, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
  }
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_count_bytes(femto_amd_index_t* ix, int64_t npats, const int32_t* plen, const uint8_t* bytes,
                          const int64_t* starts, int64_t* first, int64_t* last) {
  API_BEGIN
  int rc = validate_patterns(npats, plen, starts);
  if (rc) return rc;
  int64_t total = 0;
  for (int64_t i = 0; i < npats; i++) total = std::max<int64_t>(total, starts[i] + plen[i]);
  if (total && !bytes) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern bytes");
  std::vector<uint16_t> codes(size_t(total) + 1);
  for (int64_t i = 0; i < total; i++) codes[size_t(i)] = uint16_t(bytes[i]) + FEMTO_AMD_CHARACTER_OFFSET;
  return femto_amd_count_flat(ix, npats, plen, codes.data(), starts, first, last);
  API_END
}

int femto_amd_parallel_count(femto_amd_index_t* ix, int npats, const int* plen, const uint16_t* const* pats,
                             int64_t* first, int64_t* last) {
  API_BEGIN
  if (npats < 0 || (npats && (!plen || !pats))) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (ix && !ix->children.empty()) {
    if (npats && !first) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
    return multi_run(ix, npats, [&](femto_amd_index* c, int, int64_t lo, int64_t hi) {
      return femto_amd_parallel_count(c, int(hi - lo), plen + lo, pats + lo, first + lo, last ? last + lo : nullptr);
    });
  }
  if (ix && first && npats >= kPipeMin && ix->device >= 0) {  // large batches: gathered chunk by chunk into pinned memory
    int rc = ensure_device(ix);
    if (rc) return rc;
    Lease L(ix);
    if (!L.s) return L.rc;
    HostBatch hb;
    hb.npats = npats;
    hb.plen = plen;
    hb.ptrs = pats;
    rc = count_host_pipelined(ix, *L.s, hb, first, last);
    if (rc != -1) return rc;
  }
  std::vector<int64_t> starts(size_t(npats) + 1, 0);
  for (int i = 0; i < npats; i++) {
    if (plen[i] < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative pattern length");
    if (plen[i] && !pats[i]) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern");
    starts[size_t(i) + 1] = starts[size_t(i)] + plen[i];
  }
  std::vector<uint16_t> flat(size_t(starts[size_t(npats)]) + 1);
  for (int i = 0; i < npats; i++)  // patterns are copied, as setup_string_query does (src/main/server.c:691-695)
    if (plen[i]) memcpy(flat.data() + starts[size_t(i)], pats[i], size_t(plen[i]) * 2);
  return femto_amd_count_flat(ix, npats, plen, flat.data(), starts.data(), first, last);
  API_END
}

int femto_amd_locate_flat(femto_amd_index_t* ix, int64_t npats, const int32_t* plen, const uint16_t* pats,
                          const int64_t* starts, int max_occs_each, int32_t* noccs, int64_t* out_starts,
                          int64_t* offsets, int64_t offsets_capacity, int64_t* total_out) {
  API_BEGIN
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (max_occs_each < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative max_occs_each");
  if (!ix->children.empty()) {   // through the one-pass form, then into the caller's buffer
    int64_t* all = nullptr;
    int64_t tot = 0;
    int rc = femto_amd_locate_flat_alloc(ix, npats, plen, pats, starts, max_occs_each, noccs, out_starts, &all, &tot);
    if (rc) return rc;
    if (total_out) *total_out = tot;
    if (offsets && offsets_capacity < tot) { free(all); return set_err(FEMTO_AMD_ERR_PARAM, "offsets buffer too small"); }
    if (offsets && tot) memcpy(offsets, all, size_t(tot) * 8);
    free(all);
    return FEMTO_AMD_OK;
  }
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (npats < 0 || (npats && (!plen || !starts))) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern arrays");
  int64_t total = 0;
  Lease L(ix);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  HostBatch hb;
  hb.npats = npats;
  hb.plen = plen;
  hb.flat = pats;
  hb.starts = starts;
  bool direct_plan = false;
  if ((rc = plan_host(ix, S, hb, max_occs_each, &total, &direct_plan))) return rc;
  if (total_out) *total_out = total;
  if (noccs && npats) HIP_TRY(hipMemcpy(noccs, S.noccs.p, size_t(npats) * 4, hipMemcpyDeviceToHost));
  if (out_starts) HIP_TRY(hipMemcpy(out_starts, S.out_starts.p, size_t(npats + 1) * 8, hipMemcpyDeviceToHost));
  if (!offsets) return FEMTO_AMD_OK;
  if (offsets_capacity < total) return set_err(FEMTO_AMD_ERR_PARAM, "offsets buffer too small");
  if (total == 0) return FEMTO_AMD_OK;
  return walk_to_host(ix, S, npats, total, offsets);
  API_END
}

int femto_amd_locate_flat_alloc(femto_amd_index_t* ix, int64_t npats, const int32_t* plen, const uint16_t* pats,
                                const int64_t* starts, int max_occs_each, int32_t* noccs, int64_t* out_starts,
                                int64_t** offsets_out, int64_t* total_out) {
  API_BEGIN
  if (!ix || !offsets_out) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (max_occs_each < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative max_occs_each");
  if (!ix->children.empty()) {   // shards locate independently; their offsets are concatenated in batch order
    if (npats < 0 || (npats && (!plen || !starts))) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern arrays");
    const int N = int(ix->children.size());
    std::vector<int64_t*> part(size_t(N), nullptr);
    std::vector<int64_t> ptotal(size_t(N), 0);
    std::vector<std::vector<int64_t>> pstarts((size_t(N)));
    std::vector<int32_t> tmp_noccs;
    if (!noccs) { tmp_noccs.resize(size_t(npats) + 1); noccs = tmp_noccs.data(); }
    int rc = multi_run(ix, npats, [&](femto_amd_index* c, int i, int64_t lo, int64_t hi) {
      pstarts[size_t(i)].assign(size_t(hi - lo) + 1, 0);
      return femto_amd_locate_flat_alloc(c, hi - lo, plen + lo, pats, starts + lo, max_occs_each, noccs + lo, pstarts[size_t(i)].data(),
                                         &part[size_t(i)], &ptotal[size_t(i)]);
    });
    int64_t total = 0;
    for (int i = 0; i < N; i++) total += ptotal[size_t(i)];
    int64_t* all = nullptr;
    if (!rc && total) {
      all = static_cast<int64_t*>(malloc(size_t(total) * 8));
      if (!all) rc = set_err(FEMTO_AMD_ERR_MEM, "malloc failed");
    }
    int64_t at = 0;
    for (int i = 0; i < N; i++) {
      const int64_t lo = npats * i / N, hi = npats * (i + 1) / N;
      if (!rc) {
        if (ptotal[size_t(i)]) memcpy(all + at, part[size_t(i)], size_t(ptotal[size_t(i)]) * 8);
        if (out_starts) for (int64_t k = lo; k < hi; k++) out_starts[k] = at + pstarts[size_t(i)][size_t(k - lo)];
      }
      at += ptotal[size_t(i)];
      free(part[size_t(i)]);
    }
    if (rc) return rc;
    if (out_starts) out_starts[npats] = total;
    *offsets_out = all;
    if (total_out) *total_out = total;
    return FEMTO_AMD_OK;
  }
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (npats < 0 || (npats && (!plen || !starts))) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern arrays");
  Lease L(ix);
  if (!L.s) return L.rc;
  HostBatch hb;
  hb.npats = npats;
  hb.plen = plen;
  hb.flat = pats;
  hb.starts = starts;
  return locate_host(ix, *L.s, hb, max_occs_each, noccs, out_starts, offsets_out, total_out);
  API_END
}

int femto_amd_parallel_locate(femto_amd_index_t* ix, int npats, const int* plen, const uint16_t* const* pats,
                              int max_occs_each, int* noccs, int64_t** offsets) {
  API_BEGIN
  if (npats < 0 || (npats && (!plen || !pats || !noccs || !offsets))) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (max_occs_each < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative max_occs_each");
  if (!ix->children.empty())     // every pattern's offsets are its own malloc(): shards are independent
    return multi_run(ix, npats, [&](femto_amd_index* c, int, int64_t lo, int64_t hi) {
      return femto_amd_parallel_locate(c, int(hi - lo), plen + lo, pats + lo, max_occs_each, noccs + lo, offsets + lo);
    });
  int rc = ensure_device(ix);
  if (rc) return rc;
  int64_t* all = nullptr;
  int64_t total = 0;
  {
    Lease L(ix);
    if (!L.s) return L.rc;
    HostBatch hb;
    hb.npats = npats;
    hb.plen = plen;
    hb.ptrs = pats;
    // (no out_starts array comes back: pattern i's offsets start at the sum of the noccs before it, which the hand-out below
    // accumulates itself -- 80 MB less to allocate, clear and copy per 10 M patterns)
    rc = locate_host(ix, *L.s, hb, max_occs_each, noccs, nullptr, &all, &total);
    if (rc == -1) {  // small batch: flatten here (patterns are copied, as setup_string_query does, server.c:691-695)
      std::vector<int64_t> starts(size_t(npats) + 1, 0);
      for (int i = 0; i < npats; i++) {
        if (plen[i] < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative pattern length");
        if (plen[i] && !pats[i]) return set_err(FEMTO_AMD_ERR_PARAM, "null pattern");
        starts[size_t(i) + 1] = starts[size_t(i)] + plen[i];
      }
      std::vector<uint16_t> flat(size_t(starts[size_t(npats)]) + 1);
      for (int i = 0; i < npats; i++)
        if (plen[i]) memcpy(flat.data() + starts[size_t(i)], pats[i], size_t(plen[i]) * 2);
      hb.ptrs = nullptr;
      hb.flat = flat.data();
      hb.starts = starts.data();
      rc = locate_host(ix, *L.s, hb, max_occs_each, noccs, nullptr, &all, &total);
    }
    if (rc) return rc;
  }
  // femto.c:372-386: offsets[i] is the callee's malloc() per matching pattern, NULL otherwise.  Large batches are handed out by
  // the staging threads, each a contiguous slice (its first offset = the rows of the slices before it): one thread spent
  // ~20 ms of a 10 M-pattern call here, writing 80 MB of pointers.
  std::atomic<int> failed{0};
  auto hand_out = [&](int64_t i0, int64_t i1, int64_t at) {
    for (int64_t i = i0; i < i1; i++) {
      offsets[i] = nullptr;
      const int64_t n = noccs[i];
      if (n > 0) {
        int64_t* o = failed.load(std::memory_order_relaxed) ? nullptr : static_cast<int64_t*>(malloc(sizeof(int64_t) * size_t(n)));
        if (!o) failed.store(1);
        else memcpy(o, all + at, sizeof(int64_t) * size_t(n));
        offsets[i] = o;
        at += n;
      }
    }
  };
  if (npats >= (1 << 18) && ix->workers) {
    WorkerPool& pool = *ix->workers;
    std::lock_guard<std::mutex> wl(ix->workers_mu);
    const int T = pool.size();
    std::vector<int64_t> rows(size_t(T) + 1, 0);
    pool.run([&](int t, int nt) {
      int64_t sum = 0;
      for (int64_t i = int64_t(npats) * t / nt, e = int64_t(npats) * (t + 1) / nt; i < e; i++) sum += noccs[i] > 0 ? noccs[i] : 0;
      rows[size_t(t) + 1] = sum;
    });
    for (int t = 0; t < T; t++) rows[size_t(t) + 1] += rows[size_t(t)];
    pool.run([&](int t, int nt) { hand_out(int64_t(npats) * t / nt, int64_t(npats) * (t + 1) / nt, rows[size_t(t)]); });
  } else {
    hand_out(0, npats, 0);
  }
  if (failed.load()) {
    for (int i = 0; i < npats; i++) { free(offsets[i]); offsets[i] = nullptr; }
    free(all);
    return set_err(FEMTO_AMD_ERR_MEM, "malloc failed");
  }
  free(all);
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_host_pipeline_stats(femto_amd_index_t* ix, double* out8) {
  if (!ix || !out8) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (!ix->children.empty()) return femto_amd_host_pipeline_stats(ix->children[0], out8);
  std::lock_guard<std::mutex> lk(ix->mu);
  for (int k = 0; k < 8; k++) out8[k] = ix->pipe_stats[k];
  return FEMTO_AMD_OK;
}

int femto_amd_parallel_locate_range(femto_amd_index_t* ix, int64_t first, int64_t last, int64_t* offsets) {
  API_BEGIN
  if (!ix || !offsets) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (first < 0 || last < first || last >= ix->host.total_length)
    return set_err(FEMTO_AMD_ERR_PARAM, "row range outside the index");   // the reference has no query to set up (server.c:4061)
  if (!ix->children.empty())
    return multi_run(ix, last - first + 1, [&](femto_amd_index* c, int, int64_t lo, int64_t hi) {
      return hi > lo ? femto_amd_parallel_locate_range(c, first + lo, first + hi - 1, offsets + lo) : 0;
    });
  int rc = ensure_device(ix);
  if (rc) return rc;
  Lease L(ix);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  const int64_t max_chunk = int64_t(1) << 26;   // rows per launch (bounded scratch; far below the 2^32 work-item limit)
  if ((rc = S.first.reserve(16))) return rc;
  if ((rc = S.out_starts.reserve(32))) return rc;
  for (int64_t at = first; at <= last; at += max_chunk) {
    const int64_t cnt = std::min<int64_t>(max_chunk, last - at + 1);
    const int64_t os[2] = {0, cnt};
    HIP_TRY(hipMemcpyAsync(S.first.p, &at, 8, hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipMemcpyAsync(S.out_starts.p, os, 16, hipMemcpyHostToDevice, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    if ((rc = walk_to_host(ix, S, 1, cnt, offsets + (at - first)))) return rc;
  }
  return FEMTO_AMD_OK;
  API_END
}

}  // extern "C"
