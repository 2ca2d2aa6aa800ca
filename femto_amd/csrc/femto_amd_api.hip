// femto_amd_api.hip -- the C ABI (include/femto_amd.h) over the HIP kernels.  C++ host code that
// owns device memory, streams and launch configuration; no compute happens on the host.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is loaded on first use (femto_amd_comm_*), never at link time
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <poll.h>
#include <sys/un.h>
#include <unistd.h>
#include <cerrno>
#include <chrono>

#include <algorithm>
#include <climits>
#include <condition_variable>
#include <new>
#include <stdexcept>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "api_internal.hpp"
#include "../../include/femto_amd.h"
#include "host_index.hpp"
#include "host_pipeline.hpp"
#include "index_builder.hpp"
#include "kernels.hip.hpp"
#include "pack_kernels.hip.hpp"
#include "ru_kernels.hip.hpp"
#include "pack2_kernels.hip.hpp"
#include "ind_kernels.hip.hpp"
#include "text_kernels.hip.hpp"
#include "ctx_kernels.hip.hpp"
#include "direct_kernels.hip.hpp"
#include "trace_api.hpp"

using namespace femto_amd;

namespace femto_amd {
size_t query_sort_temp_bytes(int64_t npats, int bits, int sort_syms);
hipError_t query_sort(int64_t npats, const int32_t* d_plen, const uint16_t* d_pats, const int64_t* d_starts,
                      const uint8_t* d_dense, int bits, int sort_syms, uint64_t* keys, uint64_t* keys2, uint32_t* idx,
                      uint32_t* idx2, void* tmp, size_t tmp_bytes, hipStream_t stream);
}

namespace {
thread_local std::string g_last_error;
}  // namespace

namespace femto_amd {

int set_err(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

thread_local std::vector<std::pair<void*, size_t>>* g_small_registry = nullptr;   // set while a handle is being opened

// lease of one Scratch for the duration of a call
// `same_stream`: the call only enqueues work on that stream.  A scratch whose previous use was enqueued on the SAME stream
// can be taken at once -- stream order keeps the two uses apart -- so a caller issuing step after step on one stream keeps
// ONE warm scratch instead of cycling through the pool (creating a scratch allocates its buffers, which synchronises the
// device: 0.2 ms per step of a 10-step run, measured).
Scratch* scratch_acquire(femto_amd_index* ix, int* rc, bool enqueue_only, hipStream_t same_stream) {
  std::unique_lock<std::mutex> lk(ix->pool_mu);
  for (;;) {
    Scratch* waiting = nullptr;
    if (enqueue_only)
      for (auto& s : ix->pool)
        if (!s->busy && s->in_flight && s->flight_stream == same_stream) {
          s->busy = true;      // still in flight: the event is re-recorded when this lease ends
          return s.get();
        }
    for (auto& s : ix->pool) {
      if (s->busy) continue;
      if (s->in_flight) {
        if (hipEventQuery(s->done) != hipSuccess) {
          (void)hipGetLastError();
          waiting = s.get();
          continue;
        }
        s->in_flight = false;
      }
      s->busy = true;
      return s.get();
    }
    if (int(ix->pool.size()) < ix->pool_max) {
      std::unique_ptr<Scratch> s(new (std::nothrow) Scratch());
      if (!s) { *rc = set_err(FEMTO_AMD_ERR_MEM, "out of memory"); return nullptr; }
      if ((*rc = s->init())) { s->release(); return nullptr; }
      s->busy = true;
      ix->pool.push_back(std::move(s));
      return ix->pool.back().get();
    }
    if (waiting) {   // every scratch is leased or still in flight on some stream: wait for the device
      waiting->busy = true;
      lk.unlock();
      (void)hipEventSynchronize(waiting->done);
      lk.lock();
      waiting->in_flight = false;
      return waiting;
    }
    ix->pool_cv.wait(lk);
  }
}

void scratch_release(femto_amd_index* ix, Scratch* s, bool async, hipStream_t stream) {
  if (!s) return;
  bool flying = false;
  if (async) flying = hipEventRecord(s->done, stream) == hipSuccess;
  {
    std::lock_guard<std::mutex> lk(ix->pool_mu);
    s->busy = false;
    s->in_flight = flying;
    s->flight_stream = stream;
  }
  ix->pool_cv.notify_one();
}

int64_t knob(int64_t opt_value, const char* env_name, int64_t dflt) {
  if (opt_value != -1) return opt_value;
  if (const char* e = getenv(env_name)) return atoll(e);
  return dflt;
}

size_t hbm_free(const femto_amd_index* ix) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
  if (ix->opt.hbm_budget_bytes >= 0) {
    int64_t held = ix->hbm_held;
    for (const auto& t : ix->small_tables) held += int64_t(t.second);
    const int64_t left = ix->opt.hbm_budget_bytes - held;
    if (left <= 0) return 0;
    if (size_t(left) < free_b) free_b = size_t(left);
  }
  return free_b;
}

// ---- big arrays: plain hipMalloc, or -- striped index -- one address range backed by the HBM of several GPUs ---------
static hipError_t big_malloc_raw(femto_amd_index* ix, void** out, size_t bytes);
hipError_t big_malloc(femto_amd_index* ix, void** out, size_t bytes) {
  const hipError_t e = big_malloc_raw(ix, out, bytes);
  if (e == hipSuccess) {
    ix->hbm_held += int64_t(bytes);
    ix->big_allocs.emplace_back(*out, bytes);
  }
  return e;
}
static hipError_t big_malloc_raw(femto_amd_index* ix, void** out, size_t bytes) {
  if (ix->stripe_devices.empty()) return hipMalloc(out, bytes);
  const int N = int(ix->stripe_devices.size());
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = ix->stripe_devices[0];
  prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;    // shareable with other processes (femto_amd_striped_serve)
  size_t gran = 0;
  hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  if (e != hipSuccess) return e;
  if (gran == 0) gran = size_t(2) << 20;
  femto_amd_index::Striped st;
  st.chunk = ((bytes + size_t(N) - 1) / size_t(N) + gran - 1) / gran * gran;
  st.size = st.chunk * size_t(N);
  st.va = nullptr;
  if ((e = hipMemAddressReserve(&st.va, st.size, gran, nullptr, 0)) != hipSuccess) return e;
  for (int i = 0; i < N; i++) {
    prop.location.id = ix->stripe_devices[size_t(i)];
    hipMemGenericAllocationHandle_t h;
    if ((e = hipMemCreate(&h, st.chunk, &prop, 0)) != hipSuccess) break;
    st.handles.push_back(h);
    if ((e = hipMemMap(static_cast<char*>(st.va) + size_t(i) * st.chunk, st.chunk, 0, h, 0)) != hipSuccess) break;
  }
  if (e == hipSuccess) {
    std::vector<hipMemAccessDesc> acc;
    std::vector<int> seen;
    for (int d : ix->stripe_devices) {
      if (std::find(seen.begin(), seen.end(), d) != seen.end()) continue;
      seen.push_back(d);
      hipMemAccessDesc a{};
      a.location.type = hipMemLocationTypeDevice;
      a.location.id = d;
      a.flags = hipMemAccessFlagsProtReadWrite;
      acc.push_back(a);
    }
    e = hipMemSetAccess(st.va, st.size, acc.data(), acc.size());
  }
  if (e != hipSuccess) {
    for (size_t i = 0; i < st.handles.size(); i++) {
      (void)hipMemUnmap(static_cast<char*>(st.va) + i * st.chunk, st.chunk);
      (void)hipMemRelease(st.handles[i]);
    }
    (void)hipMemAddressFree(st.va, st.size);
    return e;
  }
  ix->striped.push_back(st);
  *out = st.va;
  return hipSuccess;
}

void big_free(femto_amd_index* ix, void* p) {
  if (!p) return;
  for (size_t k = 0; k < ix->big_allocs.size(); k++)
    if (ix->big_allocs[k].first == p) {
      ix->hbm_held -= int64_t(ix->big_allocs[k].second);
      ix->big_allocs.erase(ix->big_allocs.begin() + long(k));
      break;
    }
  for (size_t k = 0; k < ix->striped.size(); k++)
    if (ix->striped[k].va == p) {
      auto& st = ix->striped[k];
      for (size_t i = 0; i < st.handles.size(); i++) {
        (void)hipMemUnmap(static_cast<char*>(st.va) + i * st.chunk, st.chunk);
        (void)hipMemRelease(st.handles[i]);
      }
      (void)hipMemAddressFree(st.va, st.size);
      ix->striped.erase(ix->striped.begin() + long(k));
      return;
    }
  (void)hipFree(p);
}

// memset / host-to-device copy that never crosses a stripe boundary in one call
template <class Fn>
hipError_t big_pieces(femto_amd_index* ix, void* p, size_t bytes, Fn fn) {
  char* c = static_cast<char*>(p);
  for (auto& st : ix->striped) {
    char* va = static_cast<char*>(st.va);
    if (c >= va && c < va + st.size) {
      size_t done = 0;
      while (done < bytes) {
        const size_t off = size_t(c + done - va);
        const size_t piece = std::min(bytes - done, st.chunk - off % st.chunk);
        hipError_t e = fn(c + done, done, piece);
        if (e != hipSuccess) return e;
        done += piece;
      }
      return hipSuccess;
    }
  }
  return fn(c, 0, bytes);
}
hipError_t big_memset(femto_amd_index* ix, void* p, int v, size_t bytes) {
  return big_pieces(ix, p, bytes, [&](char* dst, size_t, size_t n) { return hipMemset(dst, v, n); });
}
hipError_t big_h2d(femto_amd_index* ix, void* p, const void* src, size_t bytes) {
  return big_pieces(ix, p, bytes, [&](char* dst, size_t off, size_t n) { return hipMemcpy(dst, static_cast<const char*>(src) + off, n, hipMemcpyHostToDevice); });
}

int ensure_device(femto_amd_index* ix) {
  if (ix->device < 0) return set_err(FEMTO_AMD_ERR_INVALID, "index was opened without a device (parse-only handle)");
  if (ix->split_parts > 0 && !ix->split_ready)
    return set_err(FEMTO_AMD_ERR_INVALID, "range-split index: attach every part and call femto_amd_split_commit first");
  HIP_TRY(hipSetDevice(ix->device));
  return 0;
}

// image byte range [lo, hi) of data blocks [b0, b1) (block starts are 256-byte aligned inside HostIndex::image)
void block_image_range(const HostIndex& h, int64_t b0, int64_t b1, uint64_t* lo, uint64_t* hi) {
  if (b0 >= b1) { *lo = *hi = 0; return; }
  *lo = h.block_off[size_t(b0)];
  *hi = h.block_off[size_t(b1) - 1] + h.block_len[size_t(b1) - 1];
}

int check_err_flag(Scratch& S, hipStream_t stream) {
  int flag = 0;
  HIP_TRY(hipMemcpyAsync(&flag, S.d_flags, sizeof(int), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (flag) {
    HIP_TRY(hipMemsetAsync(S.d_flags, 0, sizeof(int), stream));
    return set_err(FEMTO_AMD_ERR_PARAM, "pattern contains a character code >= ALPHA_SIZE (261)");
  }
  return 0;
}

// timing events are created once and reused (the timer keeps a free list)
bool timer_begin(femto_amd_index* ix, KernelTimer& t, hipStream_t stream, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (!ix->timing) return false;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (!t.take(e0, e1)) return false;
  if (hipEventRecord(*e0, stream) != hipSuccess) { t.give(*e0, *e1); *e0 = *e1 = nullptr; return false; }
  return true;
}
void timer_end(femto_amd_index* ix, KernelTimer& t, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
  if (!e0) return;
  (void)hipEventRecord(e1, stream);
  std::lock_guard<std::mutex> lk(ix->mu);
  t.events.emplace_back(e0, e1);
}


}  // namespace femto_amd

namespace {

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
};

// modes 3 / 4: the caller-order pipeline of direct_kernels.hip.hpp (with or without a level table)
bool use_direct(const femto_amd_index* ix) { return ix->mode == 3 || ix->mode == 4; }

int tail_setup(femto_amd_index* ix, Scratch& S, DevIndex& d, int64_t npats, hipStream_t stream) {
  int rc = S.tail.reserve(size_t(npats) * sizeof(TailItem));
  if (rc) return rc;
  d.tail_items = S.tail.p;
  d.tail_min = ix->mode == 3 ? 12 : 10;   // about where the walk + compare + ISA lookup beats stepping (1 / 2 lines a step)
  d.tail_min = std::max(2, int(knob(ix->opt.tail_min, "FEMTO_AMD_TAIL_MIN", d.tail_min)));
  d.tail_count = S.d_flags + 2;
  HIP_TRY(hipMemsetAsync(d.tail_count, 0, sizeof(int), stream));
  return 0;
}

// count_tail_kernel of the handle's layout; with the full suffix array resident the row's position is one read
void launch_tail(femto_amd_index* ix, const DevIndex& d, dim3 grid, hipStream_t stream, const TailItem* items, const int32_t* d_plen,
                 const uint16_t* d_pats, const int64_t* d_starts, const uint32_t* perm, const uint64_t* keys, int bits, int nsym,
                 const TailOut& out, int* err_flag) {
  const dim3 block{uint32_t(kBlockThreads)};
  const int* n_items = d.tail_count;
  if (ix->mode == 3 && d.ru) {
    if (d.sa_full) hipLaunchKernelGGL((count_tail_kernel<RuPolicy, true>), grid, block, 0, stream, d, items, n_items, d_plen, d_pats, d_starts, perm, keys, bits, nsym, out, err_flag);
    else hipLaunchKernelGGL((count_tail_kernel<RuPolicy, false>), grid, block, 0, stream, d, items, n_items, d_plen, d_pats, d_starts, perm, keys, bits, nsym, out, err_flag);
  } else if (ix->mode == 3) {
    if (d.sa_full) hipLaunchKernelGGL((count_tail_kernel<PackPolicy, true>), grid, block, 0, stream, d, items, n_items, d_plen, d_pats, d_starts, perm, keys, bits, nsym, out, err_flag);
    else hipLaunchKernelGGL((count_tail_kernel<PackPolicy, false>), grid, block, 0, stream, d, items, n_items, d_plen, d_pats, d_starts, perm, keys, bits, nsym, out, err_flag);
  } else {
    if (d.sa_full) hipLaunchKernelGGL((count_tail_kernel<Pack2Policy, true>), grid, block, 0, stream, d, items, n_items, d_plen, d_pats, d_starts, perm, keys, bits, nsym, out, err_flag);
    else hipLaunchKernelGGL((count_tail_kernel<Pack2Policy, false>), grid, block, 0, stream, d, items, n_items, d_plen, d_pats, d_starts, perm, keys, bits, nsym, out, err_flag);
  }
}

// S.bsums holds a PlanSums (direct_kernels.hip.hpp): block sums + two alternating sets of group sums.  The kernels keep
// the set the next launch uses cleared; the host only clears when the buffer is new or the batch size (= the layout) changes,
// or when the group sums are computed by plan_super_kernel (which accumulates the second level).
int reserve_plan_sums(Scratch& S, int64_t nblocks, bool fold, hipStream_t stream) {
  const void* before = S.bsums.p;
  int rc = S.bsums.reserve(plan_sums_bytes(nblocks));
  if (rc) return rc;
  S.bsums_parity ^= 1;
  if (S.bsums.p != before || S.bsums_nblocks != nblocks || !fold || !S.bsums_clean) {
    int64_t* sets = S.bsums.as<int64_t>() + ((nblocks + 63) & ~int64_t(63));
    HIP_TRY(hipMemsetAsync(sets, 0, size_t(2 * plan_set_words(nblocks)) * 8, stream));
  }
  S.bsums_nblocks = nblocks;
  S.bsums_clean = false;       // true again once launch_plan_rows has run for this launch (it clears the other set)
  return 0;
}

// Thresholds of the inline text tail (count_direct_kernel<.., kDense = true>): SA read + text compare + ISA read are
// three dependent lines, so it pays from four symbols to go (measured: cfg 3 5.22 -> 5.08 ms against the hand-over
// thresholds 12 / 10).  Packed lines (<= 8 characters): only once the row has survived two steps -- a random pattern's
// last row usually dies on the next step, one line, and 10 M random DNA 20-mers ran 0.86 instead of 0.65 ms when they
// jumped at once; a pattern that occurs pays two lines more.
void inline_tail_setup(const femto_amd_index* ix, DevIndex& d) {
  d.tail_min = 4;
  d.tail_ones = ix->mode == 3 ? 2 : 0;
  d.tail_min = std::max(2, int(knob(ix->opt.tail_min, "FEMTO_AMD_TAIL_MIN", d.tail_min)));
  d.tail_ones = std::max(0, int(knob(ix->opt.tail_ones, "FEMTO_AMD_TAIL_ONES", d.tail_ones)));
  // Ranges of 2-4 rows can take the tail too (each row compared, the survivors' rows from the inverse suffix array), but on
  // the sigma~96 workload that loses: the lanes of a wavefront then serialise up to 3 x rows dependent reads while the
  // others wait (10 M sampled patterns, round 2: rows = 1 3.48 ms, 2 3.63 ms, 4 3.87-4.08 ms; round 3, with eight symbols
  // compared per load: 2.08 / 2.14 / 2.20 ms -- a wavefront lives as long as its slowest lane's chain of dependent reads, and
  // rows x (SA, text, ISA) is no shorter a chain than the steps it replaces).  Default 1; FEMTO_AMD_TAIL_ROWS <= 4.
  d.tail_rows = 1;
  d.tail_row_cost = 8;
  d.tail_rows = std::max(1, int(knob(ix->opt.tail_rows, "FEMTO_AMD_TAIL_ROWS", d.tail_rows)));
  d.tail_row_cost = std::max(0, int(knob(ix->opt.tail_row_cost, "FEMTO_AMD_TAIL_ROW_COST", d.tail_row_cost)));
}

// modes 3/4, caller order, no sort (direct_kernels.hip.hpp)
int launch_count_direct(femto_amd_index* ix, Scratch& S, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                        const int64_t* d_starts, int64_t* d_first, int64_t* d_last, hipStream_t stream, Plan* plan) {
  const int64_t nblocks = (npats + kBlockThreads - 1) / kBlockThreads;
  DevIndex d = ix->dev;
  int rc;
  // full suffix array + inverse suffix array resident: the text tail is compared inline, after the wavefront's stepping
  // loop (direct_kernels.hip.hpp); with the sampled arrays the pattern is handed over to count_tail_kernel instead
  const bool inline_tail = d.txt && d.sa_full && d.isa8 && d.isa_shift == 0;
  const bool tail = d.txt != nullptr && !inline_tail;
  if (tail && (rc = tail_setup(ix, S, d, npats, stream))) return rc;
  if (d.txt && !tail) inline_tail_setup(ix, d);
  PlanSums ps{nullptr, nullptr, nullptr, nullptr, 0, nblocks, 0};
  int* big_flag = nullptr;
  if (plan) {
    if ((rc = reserve_plan_sums(S, nblocks, !tail, stream))) return rc;
    ps = plan_sums_at(S.bsums.p, nblocks, /*fold=*/!tail, S.bsums_parity);     // (count_tail_kernel still adds to the sums: see plan_super_kernel)
    big_flag = S.d_flags + 1;
  }
  int64_t* bsums = ps.sums;
  hipEvent_t e0, e1;
  timer_begin(ix, ix->t_count, stream, &e0, &e1);
  const dim3 grid{uint32_t(nblocks)}, block{uint32_t(kBlockThreads)};
  const int mo = plan ? plan->max_occs : 0;
  int32_t* noccs = plan ? plan->noccs : nullptr;
  const bool dense = inline_tail;
#define LAUNCH_COUNT_DIRECT(POLICY)                                                                                                        \
  do {                                                                                                                                     \
    if (plan && dense) hipLaunchKernelGGL((count_direct_kernel<POLICY, true, true>), grid, block, 0, stream, d, npats, d_plen, d_pats, d_starts, d_first, d_last, S.err, mo, noccs, ps, big_flag); \
    else if (plan) hipLaunchKernelGGL((count_direct_kernel<POLICY, true, false>), grid, block, 0, stream, d, npats, d_plen, d_pats, d_starts, d_first, d_last, S.err, mo, noccs, ps, big_flag);       \
    else if (dense) hipLaunchKernelGGL((count_direct_kernel<POLICY, false, true>), grid, block, 0, stream, d, npats, d_plen, d_pats, d_starts, d_first, d_last, S.err, mo, noccs, ps, big_flag);     \
    else hipLaunchKernelGGL((count_direct_kernel<POLICY, false, false>), grid, block, 0, stream, d, npats, d_plen, d_pats, d_starts, d_first, d_last, S.err, mo, noccs, ps, big_flag);              \
  } while (0)
  if (ix->mode == 3 && d.ru) LAUNCH_COUNT_DIRECT(RuPolicy);     // rank units: one 16-byte load per range end and step
  else if (ix->mode == 3) LAUNCH_COUNT_DIRECT(PackPolicy);
  else if (d.ind) LAUNCH_COUNT_DIRECT(IndPolicy);     // per-character rank lines: one line per range end and step
  else LAUNCH_COUNT_DIRECT(Pack2Policy);
#undef LAUNCH_COUNT_DIRECT
  HIP_TRY(hipGetLastError());
  if (tail) {   // persistent grid: the number of handed-over patterns is only known on the device
    const TailOut out{nullptr, d_first, d_last, noccs, bsums, mo};
    const dim3 tgrid{uint32_t(std::min<int64_t>(nblocks, int64_t(ix->num_cus) * 8))};
    launch_tail(ix, d, tgrid, stream, static_cast<const TailItem*>(S.tail.p), d_plen, d_pats, d_starts, nullptr, nullptr, 1, 0, out, S.err);
    HIP_TRY(hipGetLastError());
  }
  timer_end(ix, ix->t_count, stream, e0, e1);
  if (plan) {
    if (tail) {     // the sums became final in count_tail_kernel: the group sums in a launch of their own
      hipLaunchKernelGGL(plan_super_kernel, dim3(uint32_t(((nblocks + 63) / 64 + 3) / 4)), dim3(256), 0, stream, ps);
      HIP_TRY(hipGetLastError());
    }
    S.total_user = plan->total_user;
    plan->done = true;
  }
  return 0;
}

int launch_count_chunk(femto_amd_index* ix, Scratch& S, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                       const int64_t* d_starts, int64_t* d_first, int64_t* d_last, hipStream_t stream, Plan* plan = nullptr) {
  if (npats <= 0) return 0;
  if (use_direct(ix)) return launch_count_direct(ix, S, npats, d_plen, d_pats, d_starts, d_first, d_last, stream, plan);
  constexpr int lanes_per_query = 2 * kGroupW;
  const int64_t threads = npats * lanes_per_query;
  const int64_t blocks = (threads + kBlockThreads - 1) / kBlockThreads;
  if (blocks > 0x7fffffffLL) return set_err(FEMTO_AMD_ERR_PARAM, "batch too large for one launch");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const DevIndex& d = ix->dev;
  if (ix->mode == 1) {
    // femto's own wavelet tree, one lane per pattern (alphabets of more than 256 characters, range-split indexes)
    const int64_t lblocks = (npats + kBlockThreads - 1) / kBlockThreads;
    const uint32_t* perm = nullptr;
    if (ix->sort_queries && npats >= ix->sort_min && npats < (int64_t(1) << 32)) {
      // order the batch by pattern suffix (query_sort.hip: the reference's "sort requests by block and row",
      // server.h:930-971); results still land at the caller's indexes
      int rc;
      if ((rc = S.keys.reserve(size_t(npats) * 8))) return rc;
      if ((rc = S.keys2.reserve(size_t(npats) * 8))) return rc;
      if ((rc = S.idx.reserve(size_t(npats) * 4))) return rc;
      if ((rc = S.idx2.reserve(size_t(npats) * 4))) return rc;
      // symbols that matter for the order: sigma^s >= 4 * npats
      int sort_syms = 1;
      for (double reach = ix->dense_sigma; reach < 4.0 * double(npats) && sort_syms < 64; reach *= ix->dense_sigma) sort_syms++;
      const size_t tb = query_sort_temp_bytes(npats, ix->dense_bits, sort_syms);
      if ((rc = S.sorttmp.reserve(tb ? tb : 16))) return rc;
      HIP_TRY(query_sort(npats, d_plen, d_pats, d_starts, ix->d_dense, ix->dense_bits, sort_syms, S.keys.as<uint64_t>(),
                         S.keys2.as<uint64_t>(), S.idx.as<uint32_t>(), S.idx2.as<uint32_t>(), S.sorttmp.p, tb, stream));
      perm = S.idx2.as<uint32_t>();
    }
    timer_begin(ix, ix->t_count, stream, &e0, &e1);   // events bracket the search kernel itself (the sort is separate)
    hipLaunchKernelGGL(count_kernel_lane, dim3(uint32_t(lblocks)), dim3(kBlockThreads), 0, stream, d, npats, d_plen,
                       d_pats, d_starts, d_first, d_last, S.err, perm);
  } else {
    // mode 0: the wavefront-cooperative walk of femto's raw tables (north_star's sketch; the documented reference kernel)
    timer_begin(ix, ix->t_count, stream, &e0, &e1);
    hipLaunchKernelGGL((count_kernel<kGroupW>), dim3(uint32_t(blocks)), dim3(kBlockThreads), 0, stream, d, npats,
                       d_plen, d_pats, d_starts, d_first, d_last, S.err);
  }
  HIP_TRY(hipGetLastError());
  timer_end(ix, ix->t_count, stream, e0, e1);
  return 0;
}

// An AQL dispatch carries at most 2^32 - 1 work-items per dimension: larger batches go out in chunks
// (pattern starts are absolute, so only the per-pattern arrays are offset).
int launch_count(femto_amd_index* ix, Scratch& S, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                 const int64_t* d_starts, int64_t* d_first, int64_t* d_last, hipStream_t stream, Plan* plan = nullptr) {
  const int64_t max_chunk = ix->mode == 0 ? (int64_t(1) << 25) : (int64_t(1) << 31);
  if (plan && npats <= max_chunk) return launch_count_chunk(ix, S, npats, d_plen, d_pats, d_starts, d_first, d_last, stream, plan);
  for (int64_t off = 0; off < npats; off += max_chunk) {
    const int64_t cnt = std::min<int64_t>(max_chunk, npats - off);
    int rc = launch_count_chunk(ix, S, cnt, d_plen + off, d_pats, d_starts + off, d_first + off, d_last ? d_last + off : nullptr, stream);
    if (rc) return rc;
  }
  return 0;
}

}  // namespace

// (shared with api_open.hip, whose derivations prefix-sum their per-line counts)
namespace femto_amd {
int device_scan(DeviceBuffer* scan, int64_t n, const int64_t* in, int64_t* out /* n+1 */, int level, hipStream_t stream) {
  if (n <= 0) {
    HIP_TRY(hipMemsetAsync(out, 0, sizeof(int64_t), stream));
    return 0;
  }
  const int64_t tiles = (n + kScanTile - 1) / kScanTile;
  if (level >= 3) return set_err(FEMTO_AMD_ERR_PARAM, "scan too deep");
  int rc = scan[level].reserve(size_t(2 * (tiles + 1)) * sizeof(int64_t));
  if (rc) return rc;
  int64_t* tile_sums = scan[level].as<int64_t>();
  int64_t* tile_offs = tile_sums + tiles + 1;
  hipLaunchKernelGGL(scan_tile_kernel, dim3(uint32_t(tiles)), dim3(kScanBlock), 0, stream, n, in, out, tile_sums);
  if (tiles > 1) {
    rc = device_scan(scan, tiles, tile_sums, tile_offs, level + 1, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(scan_add_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, stream, n, out, tile_offs);
  }
  hipLaunchKernelGGL(set_total_kernel, dim3(1), dim3(64), 0, stream, n, out, in, out + n);
  HIP_TRY(hipGetLastError());
  return 0;
}
}  // namespace femto_amd

namespace {

// key batches (count_keys_kernel): host-pointer key chunks, and -- with a plan -- femto_amd_locate_keys_device
int launch_count_keys(femto_amd_index* ix, int64_t n, const uint64_t* d_keys, int2* out32, int64_t* d_first, int64_t* d_last, hipStream_t stream,
                      Scratch* S = nullptr, Plan* plan = nullptr) {
  if (n <= 0) return 0;
  const int64_t nblocks = (n + kBlockThreads - 1) / kBlockThreads;
  const dim3 grid{uint32_t(nblocks)}, block{uint32_t(kBlockThreads)};
  const int bits = ix->dense_bits, nsym = 63 / bits;
  PlanSums ps{nullptr, nullptr, nullptr, nullptr, 0, nblocks, 0};
  int* big_flag = nullptr;
  int rc;
  if (plan) {
    if ((rc = reserve_plan_sums(*S, nblocks, true, stream))) return rc;
    ps = plan_sums_at(S->bsums.p, nblocks, true, S->bsums_parity);
    big_flag = S->d_flags + 1;
  }
  const int mo = plan ? plan->max_occs : 0;
  int32_t* noccs = plan ? plan->noccs : nullptr;
  hipEvent_t e0, e1;
  timer_begin(ix, ix->t_count, stream, &e0, &e1);
#define LAUNCH_KEYS(POLICY)                                                                                                                                  \
  do {                                                                                                                                                       \
    if (plan) hipLaunchKernelGGL((count_keys_kernel<POLICY, true>), grid, block, 0, stream, ix->dev, n, d_keys, bits, nsym, out32, d_first, d_last, mo, noccs, ps, big_flag); \
    else hipLaunchKernelGGL((count_keys_kernel<POLICY, false>), grid, block, 0, stream, ix->dev, n, d_keys, bits, nsym, out32, d_first, d_last, mo, noccs, ps, big_flag);     \
  } while (0)
  if (ix->mode == 3 && ix->dev.ru) LAUNCH_KEYS(RuPolicy);
  else if (ix->mode == 3) LAUNCH_KEYS(PackPolicy);
  else if (ix->dev.ind) LAUNCH_KEYS(IndPolicy);
  else LAUNCH_KEYS(Pack2Policy);
#undef LAUNCH_KEYS
  HIP_TRY(hipGetLastError());
  timer_end(ix, ix->t_count, stream, e0, e1);
  if (plan) {
    S->total_user = plan->total_user;
    plan->done = true;
  }
  return 0;
}

// count + locate plan: on return (stream order) d_noccs[npats], d_out_starts[npats+1] and S.d_total are valid; with
// the direct pipeline d_out_starts[0..npats) is filled by launch_plan_rows (which also expands the rows)
int launch_count_plan(femto_amd_index* ix, Scratch& S, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                      const int64_t* d_starts, int64_t* d_first, int64_t* d_last, Plan* plan, hipStream_t stream) {
  plan->done = false;
  int rc;
  if (npats <= 0) {
    HIP_TRY(hipMemsetAsync(plan->out_starts, 0, sizeof(int64_t), stream));
    HIP_TRY(hipMemsetAsync(S.d_total, 0, 2 * sizeof(int64_t), stream));
    return 0;
  }
  if ((rc = launch_count(ix, S, npats, d_plen, d_pats, d_starts, d_first, d_last, stream, plan))) return rc;
  if (plan->done) return 0;
  if ((rc = S.noccs64.reserve(size_t(npats + 1) * 8))) return rc;
  hipLaunchKernelGGL(clamp_kernel, dim3(uint32_t((npats + 255) / 256)), dim3(256), 0, stream, npats, d_first, d_last,
                     plan->max_occs, plan->noccs, S.noccs64.as<int64_t>());
  HIP_TRY(hipGetLastError());
  if ((rc = device_scan(S.scan, npats, S.noccs64.as<int64_t>(), plan->out_starts, 0, stream))) return rc;
  hipLaunchKernelGGL(copy_total_kernel, dim3(1), dim3(64), 0, stream, plan->out_starts + npats, S.d_total, plan->capacity);
  HIP_TRY(hipGetLastError());
  return 0;
}

// direct pipeline, after launch_count_plan: out_starts[] and -- when d_offsets is given -- the rows to locate
int launch_plan_rows(femto_amd_index* ix, Scratch& S, int64_t npats, const int32_t* d_noccs, const int64_t* d_first,
                     int64_t* d_out_starts, int64_t* d_offsets, int64_t capacity, hipStream_t stream, const int2* d_first32 = nullptr,
                     bool fuse_walk = false) {
  if (npats <= 0) return 0;
  int* big_flag = S.d_flags + 1;      // cleared by the count kernel
  const int64_t nblocks = (npats + kBlockThreads - 1) / kBlockThreads;
  const dim3 grid{uint32_t(nblocks)}, bgrid{uint32_t(std::min<int64_t>(nblocks, int64_t(ix->num_cus) * 8))}, block{uint32_t(kBlockThreads)};
  const PlanSums ps = plan_sums_at(S.bsums.p, nblocks, false, S.bsums_parity);
  S.bsums_clean = true;
  // what lands in d_offsets: the offsets themselves from the resident suffix array (no walk afterwards); the offsets
  // themselves by a walk per row inside the expansion (fuse_walk: sampled marks, one-call chains); or the rows
  const int mode = (d_offsets && ix->dev.sa_full) ? kRowsSa : ((d_offsets && fuse_walk) ? kRowsWalk : kRowsOnly);
  const bool timed = mode != kRowsOnly;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (timed) timer_begin(ix, ix->t_locate, stream, &e0, &e1);
  const int64_t* starts_c = d_out_starts;
  const int64_t* total_c = S.d_total;
  const int* big_c = big_flag;
#define LAUNCH_PLAN(MODE, POLICY)                                                                                                         \
  do {                                                                                                                                    \
    hipLaunchKernelGGL((plan_rows_kernel<MODE, POLICY>), grid, block, 0, stream, npats, d_noccs, d_first, d_first32, ps, d_out_starts,  \
                       d_offsets, capacity, big_flag, ix->dev, S.d_total, S.total_user);                                                  \
    if (d_offsets)                                                                                                                        \
      hipLaunchKernelGGL((plan_big_rows_kernel<MODE, POLICY>), bgrid, block, 0, stream, npats, d_first, d_first32, starts_c, total_c,    \
                         capacity, d_offsets, big_c, ix->dev);                                                                            \
  } while (0)
  if (mode == kRowsSa) LAUNCH_PLAN(kRowsSa, PackPolicy);            // (the policy only matters to the walk)
  else if (mode == kRowsOnly) LAUNCH_PLAN(kRowsOnly, PackPolicy);
  else if (ix->mode == 3) LAUNCH_PLAN(kRowsWalk, PackPolicy);
  else LAUNCH_PLAN(kRowsWalk, Pack2Policy);
#undef LAUNCH_PLAN
  HIP_TRY(hipGetLastError());
  if (timed) timer_end(ix, ix->t_locate, stream, e0, e1);
  return 0;
}

// direct pipeline: the walk of the rows plan_rows_kernel wrote; the row count is read from the device
int launch_walk_device_total(femto_amd_index* ix, Scratch& S, int64_t* d_offsets, int64_t capacity, hipStream_t stream) {
  hipEvent_t e0, e1;
  timer_begin(ix, ix->t_locate, stream, &e0, &e1);
  int64_t want = (capacity + kBlockThreads - 1) / kBlockThreads;
  const dim3 grid{uint32_t(std::max<int64_t>(1, std::min<int64_t>(want, int64_t(ix->num_cus) * 8)))};
  if (ix->mode == 3)
    hipLaunchKernelGGL(locate_walk_kernel<PackPolicy>, grid, dim3(kBlockThreads), 0, stream, ix->dev, static_cast<const int64_t*>(S.d_total), capacity, d_offsets);
  else
    hipLaunchKernelGGL(locate_walk_kernel<Pack2Policy>, grid, dim3(kBlockThreads), 0, stream, ix->dev, static_cast<const int64_t*>(S.d_total), capacity, d_offsets);
  HIP_TRY(hipGetLastError());
  timer_end(ix, ix->t_locate, stream, e0, e1);
  return 0;
}

int launch_locate(femto_amd_index* ix, Scratch& S, int64_t npats, const int64_t* d_first, const int64_t* d_out_starts,
                  int64_t total, int64_t* d_offsets, hipStream_t stream) {
  if (total <= 0) return 0;
  const int64_t threads = ix->mode == 0 ? total * kGroupW : total;   // mode 0 walks with a 32-lane group per row
  const int64_t blocks = (total * kGroupW + kBlockThreads - 1) / kBlockThreads;
  if (threads >= (int64_t(1) << 32)) return set_err(FEMTO_AMD_ERR_PARAM, "too many rows to locate in one call (2^32 work-items per launch): lower max_occs_each or split the batch");
  if (ix->mode == 3 || ix->mode == 4) {  // rows first (one thread per pattern), then the walk -- no per-row search for the owning pattern
    int* big_flag = S.d_flags + 1;
    HIP_TRY(hipMemsetAsync(big_flag, 0, sizeof(int), stream));
    hipLaunchKernelGGL(expand_rows_kernel, dim3(uint32_t((npats + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, stream,
                       npats, d_first, d_out_starts, d_offsets, big_flag);
    hipLaunchKernelGGL(expand_big_rows_kernel, dim3(uint32_t((total + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, stream,
                       npats, d_first, d_out_starts, total, d_offsets, big_flag);
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  timer_begin(ix, ix->t_locate, stream, &e0, &e1);
  if ((ix->mode == 3 || ix->mode == 4) && ix->dev.sa_full) {
    const int64_t lblocks = (total + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(gather_sa_kernel, dim3(uint32_t(lblocks)), dim3(kBlockThreads), 0, stream, ix->dev, total, d_offsets);
  } else if (ix->mode == 4) {
    const int64_t lblocks = (total + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(locate_kernel_pack2, dim3(uint32_t(lblocks)), dim3(kBlockThreads), 0, stream, ix->dev, total, d_offsets);
  } else if (ix->mode == 3) {
    const int64_t lblocks = (total + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(locate_kernel_pack, dim3(uint32_t(lblocks)), dim3(kBlockThreads), 0, stream, ix->dev, total, d_offsets);
  } else if (ix->mode == 1) {
    const int64_t lblocks = (total + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(locate_kernel_lane, dim3(uint32_t(lblocks)), dim3(kBlockThreads), 0, stream, ix->dev, npats, d_first,
                       d_out_starts, total, d_offsets);
  } else {
    hipLaunchKernelGGL((locate_kernel<kGroupW>), dim3(uint32_t(blocks)), dim3(kBlockThreads), 0, stream, ix->dev, npats,
                       d_first, d_out_starts, total, d_offsets);
  }
  HIP_TRY(hipGetLastError());
  timer_end(ix, ix->t_locate, stream, e0, e1);
  return 0;
}

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

size_t pipe_in_bytes() { return size_t(kPipeChunk) * 12 + size_t(kPipeSymCap) * 2 + 64; }

int pipe_init(femto_amd_index* ix, Scratch& S) {
  {
    std::lock_guard<std::mutex> lk(ix->workers_mu);
    if (!ix->workers) {
      // half of the host's hardware threads, at least 4, at most 128: packing 10 M patterns into keys is ~200 M table
      // look-ups, and the staging threads -- not PCIe, not the GPU -- bound this path (measured on the GPU box's
      // 256-thread host, 10 M 20-mers: 16 threads 12.7 ms, 32 9.2 ms, 64 6-9 ms, 128 5.6 ms)
      int nthreads = std::max(4, int(std::thread::hardware_concurrency()) / 2);
      nthreads = int(knob(ix->opt.host_threads, "FEMTO_AMD_HOST_THREADS", nthreads));
      nthreads = std::max(1, std::min(nthreads, 128));
      ix->workers.reset(new WorkerPool(nthreads));
    }
  }
  auto& P = S.pipe;
  if (P.ready) return 0;
  for (int b = 0; b < kPipeDepth; b++) {
    HIP_TRY(hipHostMalloc(&P.h_in[b], pipe_in_bytes(), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&P.h_out[b], size_t(kPipeChunk) * 16, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&P.d_in[b], pipe_in_bytes()));
    HIP_TRY(hipMalloc(&P.d_out[b], size_t(kPipeChunk) * 16));
    HIP_TRY(hipEventCreateWithFlags(&P.in_done[b], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&P.k_done[b], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&P.out_done[b], hipEventDisableTiming));
  }
  HIP_TRY(hipStreamCreateWithFlags(&P.s_h2d, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&P.s_d2h, hipStreamNonBlocking));
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
  pool.run([&](int t, int nt) {
    const int64_t i0 = a + n * t / nt, i1 = a + n * (t + 1) / nt;
    uint32_t bad = 0;
    for (int64_t i = i0; i < i1; i++) {
      const int64_t l = hb.plen[i];
      const uint16_t* pat = hb.ptrs ? hb.ptrs[i] : (hb.starts[i] >= 0 ? hb.flat + hb.starts[i] : nullptr);
      if (l < 0 || l > nsym || (l && !pat)) { partial[size_t(t)] = 1; return; }
      uint64_t key = 0;
      for (int64_t s = l - 1; s >= 0; s--) {   // last symbol first: it lands in the top field
        const uint32_t c = dense[pat[s]];
        bad |= uint32_t(c == 0);
        key = (key << bits) | c;
      }
      o_key[i - a] = l ? key << (64 - int(l) * bits) : 0;   // field j (from the top) = j-th symbol from the end; 0 = end
    }
    if (bad) partial[size_t(t)] = 1;
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
  static const bool nt_stores = [] { const char* e = getenv("FEMTO_AMD_NT_STORES"); return !e || atoi(e) != 0; }();
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
      ix->workers->run([&](int t, int nt) {
        const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
        if (k == 2) {          // 32-bit (first,last) pairs: widened into the caller's arrays (or the counts, femto.c:313-318)
          const int32_t* pr = reinterpret_cast<const int32_t*>(hout);
          // (streaming stores: the caller's arrays are written once and not read here -- no read-for-ownership of 160 MB)
          if (nt_stores) {
            for (int64_t i = i0; i < i1; i++) {
              const int64_t f = pr[2 * i], l = pr[2 * i + 1];
              if (last) { __builtin_nontemporal_store(f, first + a + i); __builtin_nontemporal_store(l, last + a + i); }
              else __builtin_nontemporal_store(l - f + 1, first + a + i);
            }
          } else {
            for (int64_t i = i0; i < i1; i++) {
              const int64_t f = pr[2 * i], l = pr[2 * i + 1];
              if (last) { first[a + i] = f; last[a + i] = l; }
              else first[a + i] = l - f + 1;
            }
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
    if (npats) {
      hipLaunchKernelGGL(clamp_kernel, dim3(uint32_t((npats + 255) / 256)), dim3(256), 0, st, npats, S.first.as<int64_t>(),
                         S.last.as<int64_t>(), max_occs_each, S.noccs.as<int32_t>(), S.noccs64.as<int64_t>());
      HIP_TRY(hipGetLastError());
    }
    if ((rc = device_scan(S.scan, npats, S.noccs64.as<int64_t>(), S.out_starts.as<int64_t>(), 0, st))) return rc;
  }
  if (plan.done) {
    *direct_plan = true;
    if ((rc = launch_plan_rows(ix, S, npats, S.noccs.as<int32_t>(), S.first.as<int64_t>(), S.out_starts.as<int64_t>(), nullptr, INT64_MAX, st))) return rc;
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
      if (rcs[size_t(i)]) msgs[size_t(i)] = g_last_error;   // thread-local message of the failing call
    });
  for (auto& t : th) t.join();
  for (int i = 0; i < N; i++)
    if (rcs[size_t(i)]) return set_err(rcs[size_t(i)], "device shard " + std::to_string(i) + ": " + msgs[size_t(i)]);
  return 0;
}


}  // namespace


const char* femto_amd_last_error(void) { return g_last_error.c_str(); }

extern "C" {

void femto_amd_options_init(femto_amd_options_t* o) {
  if (!o) return;
  memset(o, 0xff, sizeof *o);             // every field -1: auto
  o->struct_size = uint32_t(sizeof *o);
}

int femto_amd_open_opts(const char* index_path, int device, const femto_amd_options_t* opts, femto_amd_index_t** out) {
  API_BEGIN
  return open_impl(index_path, device, 0, 0, out, nullptr, opts);
  API_END
}

int femto_amd_open(const char* index_path, int device, femto_amd_index_t** out) {
  return open_impl(index_path, device, 0, 0, out);
}

int femto_amd_open_split(const char* index_path, int device, int part, int nparts, femto_amd_index_t** out) {
  if (nparts < 1 || nparts > 64 || part < 0 || part >= nparts) return set_err(FEMTO_AMD_ERR_PARAM, "bad part / nparts");
  if (device < 0) return set_err(FEMTO_AMD_ERR_PARAM, "a range-split index needs a device");
  return open_impl(index_path, device, part, nparts, out);
}

int femto_amd_split_export(femto_amd_index_t* ix, void* handles /* 128 bytes */) {
  if (!ix || !handles) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (ix->split_parts <= 0) return set_err(FEMTO_AMD_ERR_INVALID, "not a range-split index");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle blob layout");
  HIP_TRY(hipSetDevice(ix->device));
  hipIpcMemHandle_t hs[2];
  HIP_TRY(hipIpcGetMemHandle(&hs[0], ix->d_segs));
  HIP_TRY(hipIpcGetMemHandle(&hs[1], ix->d_image));
  memcpy(handles, hs, sizeof hs);
  return FEMTO_AMD_OK;
}

int femto_amd_split_attach(femto_amd_index_t* ix, int part, const void* handles) {
  if (!ix || !handles) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (ix->split_parts <= 0) return set_err(FEMTO_AMD_ERR_INVALID, "not a range-split index");
  if (part < 0 || part >= ix->split_parts) return set_err(FEMTO_AMD_ERR_PARAM, "bad part");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->split_ready) return set_err(FEMTO_AMD_ERR_INVALID, "already committed");
  if (part == ix->split_part || ix->peer_segs[size_t(part)]) return FEMTO_AMD_OK;  // own slices / already mapped
  HIP_TRY(hipSetDevice(ix->device));
  hipIpcMemHandle_t hs[2];
  memcpy(hs, handles, sizeof hs);
  void *ps = nullptr, *pi = nullptr;
  HIP_TRY(hipIpcOpenMemHandle(&ps, hs[0], hipIpcMemLazyEnablePeerAccess));
  hipError_t e2 = hipIpcOpenMemHandle(&pi, hs[1], hipIpcMemLazyEnablePeerAccess);
  if (e2 != hipSuccess) {
    (void)hipIpcCloseMemHandle(ps);
    return set_err(FEMTO_AMD_ERR_INVALID, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e2));
  }
  ix->peer_segs[size_t(part)] = ps;
  ix->peer_image[size_t(part)] = pi;
  ix->peer_ipc[size_t(part)] = 1;
  return FEMTO_AMD_OK;
}

int femto_amd_split_attach_local(femto_amd_index_t* ix, femto_amd_index_t* owner) {
  if (!ix || !owner) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (ix->split_parts <= 0 || owner->split_parts != ix->split_parts)
    return set_err(FEMTO_AMD_ERR_INVALID, "both handles must be parts of the same range-split");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->split_ready) return set_err(FEMTO_AMD_ERR_INVALID, "already committed");
  const int part = owner->split_part;
  if (part == ix->split_part || ix->peer_segs[size_t(part)]) return FEMTO_AMD_OK;
  if (owner->device != ix->device) {  // same process, another GPU: direct peer loads over xGMI
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, ix->device, owner->device));
    if (!can) return set_err(FEMTO_AMD_ERR_INVALID, "no peer access between the two devices");
    HIP_TRY(hipSetDevice(ix->device));
    hipError_t pe = hipDeviceEnablePeerAccess(owner->device, 0);
    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
      return set_err(FEMTO_AMD_ERR_INVALID, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
    (void)hipGetLastError();
  }
  ix->peer_segs[size_t(part)] = owner->d_segs;
  ix->peer_image[size_t(part)] = owner->d_image;
  return FEMTO_AMD_OK;
}

int femto_amd_split_commit(femto_amd_index_t* ix) {
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (ix->split_parts <= 0) return set_err(FEMTO_AMD_ERR_INVALID, "not a range-split index");
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->split_ready) return FEMTO_AMD_OK;
  for (int p = 0; p < ix->split_parts; p++)
    if (!ix->peer_segs[size_t(p)] || !ix->peer_image[size_t(p)])
      return set_err(FEMTO_AMD_ERR_INVALID, "part " + std::to_string(p) + " is not attached");
  HIP_TRY(hipSetDevice(ix->device));
  HostIndex& h = ix->host;
  // Rebase every sequence onto its owner's slice: offsets are relative to THIS part's bases and wrap
  // modulo 2^64 (kernels add them with integer arithmetic, wrap_ptr in kernels.hip.hpp).
  std::vector<LaneNode> ln = h.lnodes;
  std::vector<LaneSeq> ls = h.lseqs;
  const uint64_t my_segs = uint64_t(reinterpret_cast<uintptr_t>(ix->d_segs));
  const uint64_t my_img = uint64_t(reinterpret_cast<uintptr_t>(ix->d_image));
  for (int p = 0; p < ix->split_parts; p++) {
    const int64_t b0 = ix->split_blo[size_t(p)], b1 = ix->split_blo[size_t(p) + 1];
    if (b0 >= b1) continue;
    const uint64_t dseg = uint64_t(reinterpret_cast<uintptr_t>(ix->peer_segs[size_t(p)])) - my_segs;
    const uint64_t dimg = uint64_t(reinterpret_cast<uintptr_t>(ix->peer_image[size_t(p)])) - my_img;
    if (dseg & 63) return set_err(FEMTO_AMD_ERR_INVALID, "peer mapping is not 64-byte aligned");
    const uint64_t slot_delta = (dseg >> 6) - h.block_slot_start[size_t(b0)];   // in 64-byte slots, mod 2^58
    uint64_t i0, i1;
    block_image_range(h, b0, b1, &i0, &i1);
    const uint64_t img_delta = dimg - i0;
    for (uint64_t i = h.block_lnode_start[size_t(b0)]; i < h.block_lnode_start[size_t(b1)]; i++)
      ln[size_t(i)].bs.seg_base += slot_delta;
    for (uint64_t i = h.block_lseq_start[size_t(b0)]; i < h.block_lseq_start[size_t(b1)]; i++) {
      ls[size_t(i)].mark_table.seg_base += slot_delta;
      ls[size_t(i)].mark_array += img_delta;
    }
  }
  int r;
  if ((r = upload(&ix->d_lnodes, ln, &ix->table_bytes))) return r;
  if ((r = upload(&ix->d_lseqs, ls, &ix->table_bytes))) return r;
  ix->dev.lnodes = ix->d_lnodes;
  ix->dev.lseqs = ix->d_lseqs;
  ix->split_ready = true;
  // the host copies of the big tables are no longer needed
  std::vector<uint64_t>().swap(h.segs);
  return FEMTO_AMD_OK;
}

int femto_amd_split_info(const femto_amd_index_t* ix, int* part, int* nparts, int64_t* seg_bytes, int64_t* image_bytes) {
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (part) *part = ix->split_part;
  if (nparts) *nparts = ix->split_parts;
  if (seg_bytes) *seg_bytes = ix->split_seg_bytes;
  if (image_bytes) *image_bytes = ix->split_image_bytes;
  return FEMTO_AMD_OK;
}

void femto_amd_close(femto_amd_index_t* ix) {
  if (!ix) return;
  for (size_t c = ix->children.size(); c-- > 0;) femto_amd_close(ix->children[c]);   // views before the builder of a striped index
  ix->children.clear();
  if (ix->comm) {
    comm_destroy(ix);
    ix->comm = nullptr;
  }
  if (ix->device >= 0) {
    (void)hipSetDevice(ix->device);
    ix->t_count.destroy();
    ix->t_locate.destroy();
    (void)hipDeviceSynchronize();   // enqueue-only calls may still be running on the caller's streams
    for (auto& s : ix->pool) s->release();
    ix->pool.clear();
    ix->workers.reset();
    for (size_t p = 0; p < ix->peer_ipc.size(); p++)
      if (ix->peer_ipc[p]) {
        (void)hipIpcCloseMemHandle(ix->peer_segs[p]);
        (void)hipIpcCloseMemHandle(ix->peer_image[p]);
      }
    if (ix->borrowed) {      // a view on a second GPU: the arrays belong to the builder handle
      for (void* q : ix->owned_small) (void)hipFree(q);
      if (ix->imported)      // ... of another process: this process's mappings of them
        while (!ix->striped.empty()) big_free(ix, ix->striped.back().va);
    } else {
      for (void* q : {static_cast<void*>(ix->d_image), static_cast<void*>(ix->d_segs), static_cast<void*>(ix->d_pack), static_cast<void*>(ix->d_pack_sa),
                      static_cast<void*>(ix->d_txt), static_cast<void*>(ix->d_isa8), static_cast<void*>(ix->d_p2_l1), static_cast<void*>(ix->d_p2_l2),
                      static_cast<void*>(ix->d_ktab2), static_cast<void*>(ix->d_ktab2_deep), static_cast<void*>(ix->d_sa_full), static_cast<void*>(ix->d_ind),
                      static_cast<void*>(ix->d_ctx), static_cast<void*>(ix->d_ctx2), static_cast<void*>(ix->d_ru), static_cast<void*>(ix->d_ru_stop)})
        big_free(ix, q);
      for (void* q : {static_cast<void*>(ix->d_nodes), static_cast<void*>(ix->d_buckets), static_cast<void*>(ix->d_seqs), static_cast<void*>(ix->d_occ_base),
                      static_cast<void*>(ix->d_leaf_code), static_cast<void*>(ix->d_C), static_cast<void*>(ix->d_cum), static_cast<void*>(ix->d_hint),
                      static_cast<void*>(ix->d_bdir), static_cast<void*>(ix->d_lnodes), static_cast<void*>(ix->d_lseqs), static_cast<void*>(ix->d_occ),
                      static_cast<void*>(ix->d_dense), static_cast<void*>(ix->d_pack_code), static_cast<void*>(ix->d_pack_c),
                      static_cast<void*>(ix->d_p2_base), static_cast<void*>(ix->d_p2_c), static_cast<void*>(ix->d_p2_code), static_cast<void*>(ix->d_p2_alpha)})
        (void)hipFree(q);
    }
    for (DeviceBuffer& b : ix->open_scan) b.release();
  }
  delete ix;
}

int femto_amd_info(const femto_amd_index_t* ix, femto_amd_info_t* out) {
  if (!ix || !out) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  const HostIndex& h = ix->host;
  out->total_length = h.total_length;
  out->number_of_blocks = h.number_of_blocks;
  out->number_of_documents = h.number_of_documents;
  out->block_size = h.block_size;
  out->bucket_size = h.b_size;
  out->mark_period = h.mark_period;
  out->chunk_size = h.chunk_size;
  out->text_size_bits = h.text_size_bits;
  out->total_buckets = h.total_buckets;
  out->image_bytes = int64_t(h.image.size());
  out->table_bytes = int64_t(h.nodes.size() * sizeof(DevNode) + h.buckets.size() * sizeof(DevBucket) +
                             h.seqs.size() * sizeof(DevSeq) + h.occ_base.size() * 8 + h.leaf_code.size() * 4 +
                             h.C.size() * 8 + h.segs.size() * 8 + h.cum.size() * sizeof(CumEntry) + h.hint.size() * 4 + h.bdir.size() * sizeof(BlockDir) +
                             h.lnodes.size() * sizeof(LaneNode) + h.lseqs.size() * sizeof(LaneSeq) +
                             h.occ.size() * sizeof(OccEntry));
  if (ix->device >= 0) out->table_bytes = ix->table_bytes;   // what is actually resident, derived fast-path layouts included
  if (!ix->children.empty()) out->table_bytes = ix->children[0]->table_bytes;   // per GPU
  return FEMTO_AMD_OK;
}

int femto_amd_resolve_location(const femto_amd_index_t* ix, int64_t offset, int64_t* doc, int64_t* doc_offset) {
  if (!ix || !doc || !doc_offset) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  return ix->host.resolve_location(offset, doc, doc_offset);
}

int femto_amd_document_info(const femto_amd_index_t* ix, int64_t doc, const char** info, int64_t* len) {
  if (!ix || !info || !len) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  const uint8_t* p = nullptr;
  int rc = ix->host.document_info(doc, &p, len);
  if (rc) return set_err(rc, rc == FEMTO_AMD_ERR_PARAM ? "document number out of range" : "corrupt document info table");
  *info = reinterpret_cast<const char*>(p);
  return FEMTO_AMD_OK;
}

int femto_amd_count_device(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                           const int64_t* d_starts, int64_t* d_first, int64_t* d_last, void* stream) {
  API_BEGIN
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (reinterpret_cast<uintptr_t>(d_pats) & 1u) return set_err(FEMTO_AMD_ERR_PARAM, "d_pats must be 2-byte aligned (uint16 symbols; the kernels read them in aligned 16-byte pieces)");
  int rc = ensure_device(ix);
  if (rc) return rc;
  Lease L(ix, static_cast<hipStream_t>(stream));
  if (!L.s) return L.rc;
  return launch_count(ix, *L.s, npats, d_plen, d_pats, d_starts, d_first, d_last, L.stream);
  API_END
}

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

int femto_amd_locate_plan_device(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                                 const int64_t* d_starts, int max_occs_each, int64_t* d_first, int64_t* d_last,
                                 int32_t* d_noccs, int64_t* d_out_starts, void* stream_) {
  API_BEGIN
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (reinterpret_cast<uintptr_t>(d_pats) & 1u) return set_err(FEMTO_AMD_ERR_PARAM, "d_pats must be 2-byte aligned (uint16 symbols; the kernels read them in aligned 16-byte pieces)");
  if (max_occs_each < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative max_occs_each");
  int rc = ensure_device(ix);
  if (rc) return rc;
  Lease L(ix, static_cast<hipStream_t>(stream_));
  if (!L.s) return L.rc;
  Plan plan{max_occs_each, d_noccs, d_out_starts, INT64_MAX, false};
  if ((rc = launch_count_plan(ix, *L.s, npats, d_plen, d_pats, d_starts, d_first, d_last, &plan, L.stream))) return rc;
  if (plan.done) rc = launch_plan_rows(ix, *L.s, npats, d_noccs, d_first, d_out_starts, nullptr, INT64_MAX, L.stream);
  return rc;
  API_END
}

int femto_amd_locate_walk_device(femto_amd_index_t* ix, int64_t npats, const int64_t* d_first,
                                 const int64_t* d_out_starts, int64_t total, int64_t* d_offsets, void* stream) {
  API_BEGIN
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  int rc = ensure_device(ix);
  if (rc) return rc;
  Lease L(ix, static_cast<hipStream_t>(stream));
  if (!L.s) return L.rc;
  return launch_locate(ix, *L.s, npats, d_first, d_out_starts, total, d_offsets, L.stream);
  API_END
}

int femto_amd_locate_device(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                            const int64_t* d_starts, int max_occs_each, int64_t* d_first, int64_t* d_last,
                            int32_t* d_noccs, int64_t* d_out_starts, int64_t* d_offsets, int64_t offsets_capacity,
                            int64_t* d_total, void* stream_) {
  API_BEGIN
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (reinterpret_cast<uintptr_t>(d_pats) & 1u) return set_err(FEMTO_AMD_ERR_PARAM, "d_pats must be 2-byte aligned (uint16 symbols; the kernels read them in aligned 16-byte pieces)");
  if (max_occs_each < 0 || offsets_capacity < 0) return set_err(FEMTO_AMD_ERR_PARAM, "negative max_occs_each / capacity");
  if (npats && (!d_first || !d_last || !d_noccs || !d_out_starts || !d_total)) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  int rc = ensure_device(ix);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Lease L(ix, stream);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  Plan plan{max_occs_each, d_noccs, d_out_starts, offsets_capacity, false, d_total};
  if ((rc = launch_count_plan(ix, S, npats, d_plen, d_pats, d_starts, d_first, d_last, &plan, stream))) return rc;
  if (plan.done) {   // direct pipeline: one stream-ordered chain, nothing returns to the host (d_total: plan_rows_kernel's last block)
    if ((rc = launch_plan_rows(ix, S, npats, d_noccs, d_first, d_out_starts, d_offsets, offsets_capacity, stream, nullptr, /*fuse_walk=*/true))) return rc;
  } else {           // other kernel families size the walk on the host
    int64_t tot[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(tot, S.d_total, sizeof tot, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const int64_t walk = std::min(tot[0], offsets_capacity);
    if (walk == tot[0] && d_offsets && (rc = launch_locate(ix, S, npats, d_first, d_out_starts, walk, d_offsets, stream))) return rc;
    HIP_TRY(hipMemcpyAsync(d_total, S.d_total, 2 * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
  }
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_key_format(const femto_amd_index_t* ix, int* bits, int* max_syms, uint8_t* field_of_alpha /* [261] or NULL */) {
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (!ix->children.empty()) return femto_amd_key_format(ix->children[0], bits, max_syms, field_of_alpha);
  if (!use_direct(ix) || ix->h_dense.empty())
    return set_err(FEMTO_AMD_ERR_INVALID, "keys need the packed layouts (modes 3 / 4) and at most 255 distinct characters");
  if (bits) *bits = ix->dense_bits;
  if (max_syms) *max_syms = 63 / ix->dense_bits;
  if (field_of_alpha) memcpy(field_of_alpha, ix->h_dense.data(), std::min<size_t>(ix->h_dense.size(), size_t(kAlphaSize)));
  return FEMTO_AMD_OK;
}

int femto_amd_pack_keys_device(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats, const int64_t* d_starts,
                               uint64_t* d_keys, int64_t* d_bad, void* stream_) {
  API_BEGIN
  if (!ix || npats < 0 || !d_bad || (npats && (!d_plen || !d_starts || !d_keys))) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (!use_direct(ix) || ix->h_dense.empty())
    return set_err(FEMTO_AMD_ERR_INVALID, "keys need the packed layouts (modes 3 / 4) and at most 255 distinct characters");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(int64_t), stream));
  if (npats) {
    hipLaunchKernelGGL(pack_keys_kernel, dim3(uint32_t((npats + 255) / 256)), dim3(256), 0, stream, npats, d_plen, d_pats, d_starts,
                       static_cast<const uint8_t*>(ix->d_dense), ix->dense_bits, 63 / ix->dense_bits, d_keys,
                       reinterpret_cast<unsigned long long*>(d_bad));
    HIP_TRY(hipGetLastError());
  }
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_locate_keys_device(femto_amd_index_t* ix, int64_t npats, const uint64_t* d_keys, int max_occs_each, int32_t* d_ranges32,
                                 int64_t* d_first, int64_t* d_last, int32_t* d_noccs, int64_t* d_out_starts, int64_t* d_offsets,
                                 int64_t offsets_capacity, int64_t* d_total, void* stream_) {
  API_BEGIN
  if (!ix || npats < 0 || max_occs_each < 0 || offsets_capacity < 0) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (npats && (!d_keys || (!d_ranges32 && (!d_first || !d_last)))) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (d_noccs && (!d_out_starts || !d_total)) return set_err(FEMTO_AMD_ERR_PARAM, "a locate plan needs d_out_starts and d_total");
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (!use_direct(ix) || ix->h_dense.empty())
    return set_err(FEMTO_AMD_ERR_INVALID, "keys need the packed layouts (modes 3 / 4) and at most 255 distinct characters");
  if (d_ranges32 && ix->host.total_length >= int64_t(INT32_MAX)) return set_err(FEMTO_AMD_ERR_PARAM, "32-bit ranges need an index of fewer than 2^31 - 1 rows");
  if (npats > (int64_t(1) << 31)) return set_err(FEMTO_AMD_ERR_PARAM, "at most 2^31 keys per call");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Lease L(ix, stream);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  int2* r32 = reinterpret_cast<int2*>(d_ranges32);
  if (!d_noccs) return launch_count_keys(ix, npats, d_keys, r32, d_first, d_last, stream);
  if (npats == 0) {
    HIP_TRY(hipMemsetAsync(d_out_starts, 0, sizeof(int64_t), stream));
    HIP_TRY(hipMemsetAsync(d_total, 0, 2 * sizeof(int64_t), stream));
    return FEMTO_AMD_OK;
  }
  Plan plan{max_occs_each, d_noccs, d_out_starts, offsets_capacity, false, d_total};
  if ((rc = launch_count_keys(ix, npats, d_keys, r32, d_first, d_last, stream, &S, &plan))) return rc;
  if ((rc = launch_plan_rows(ix, S, npats, d_noccs, d_first, d_out_starts, d_offsets, offsets_capacity, stream, r32, /*fuse_walk=*/true))) return rc;
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_pack_counts_device(femto_amd_index_t* ix, int64_t npats, const int64_t* d_first, const int64_t* d_last, uint8_t* d_counts8,
                                 int64_t* d_big, int64_t big_capacity, int64_t* d_big_n, void* stream_) {
  API_BEGIN
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (npats < 0 || big_capacity < 0 || !d_big_n || (npats && (!d_first || !d_last || !d_counts8)) || (big_capacity && !d_big))
    return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  int rc = ensure_device(ix);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  HIP_TRY(hipMemsetAsync(d_big_n, 0, sizeof(int64_t), stream));
  if (npats) {
    hipLaunchKernelGGL(pack_counts_kernel, dim3(uint32_t((npats + 255) / 256)), dim3(256), 0, stream, npats, d_first, d_last, d_counts8, d_big,
                       big_capacity, reinterpret_cast<unsigned long long*>(d_big_n));
    HIP_TRY(hipGetLastError());
  }
  return FEMTO_AMD_OK;
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
  std::vector<int64_t> ostarts(size_t(npats) + 2);
  int64_t* all = nullptr;
  int64_t total = 0;
  {
    Lease L(ix);
    if (!L.s) return L.rc;
    HostBatch hb;
    hb.npats = npats;
    hb.plen = plen;
    hb.ptrs = pats;
    rc = locate_host(ix, *L.s, hb, max_occs_each, noccs, ostarts.data(), &all, &total);
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
      rc = locate_host(ix, *L.s, hb, max_occs_each, noccs, ostarts.data(), &all, &total);
    }
    if (rc) return rc;
  }
  for (int i = 0; i < npats; i++) {  // femto.c:372-386
    offsets[i] = nullptr;
    if (noccs[i] > 0) {
      offsets[i] = static_cast<int64_t*>(malloc(sizeof(int64_t) * size_t(noccs[i])));
      if (!offsets[i]) {
        for (int j = 0; j < i; j++) { free(offsets[j]); offsets[j] = nullptr; }
        free(all);
        return set_err(FEMTO_AMD_ERR_MEM, "malloc failed");
      }
      memcpy(offsets[i], all + ostarts[size_t(i)], sizeof(int64_t) * size_t(noccs[i]));
    }
  }
  free(all);
  return FEMTO_AMD_OK;
  API_END
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

int femto_amd_block_requests(femto_amd_index_t* ix, int64_t n, const int64_t* rows, const uint16_t* ch_in,
                             uint16_t* ch_out, int32_t* occ_out, int64_t* off_out) {
  API_BEGIN
  if (!ix || (n && !rows)) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (!ix->children.empty()) return femto_amd_block_requests(ix->children[0], n, rows, ch_in, ch_out, occ_out, off_out);
  int rc = ensure_device(ix);
  if (rc) return rc;
  const HostIndex& h = ix->host;
  for (int64_t i = 0; i < n; i++) {
    if (rows[i] < 0 || rows[i] >= h.total_length) return set_err(FEMTO_AMD_ERR_PARAM, "row out of range");
    if (ch_in && ch_in[i] >= kAlphaSize) return set_err(FEMTO_AMD_ERR_PARAM, "character out of range");
  }
  if (n == 0) return FEMTO_AMD_OK;
  Lease L(ix);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  hipStream_t st = S.stream;
  if ((rc = S.rows.reserve(size_t(n) * 8))) return rc;
  if ((rc = S.ch.reserve(size_t(n) * 4))) return rc;
  if ((rc = S.occ.reserve(size_t(n) * 8))) return rc;
  if ((rc = S.off.reserve(size_t(n) * 8))) return rc;
  HIP_TRY(hipMemcpyAsync(S.rows.p, rows, size_t(n) * 8, hipMemcpyHostToDevice, st));
  uint16_t* d_chin = nullptr;
  uint16_t* d_chout = S.ch.as<uint16_t>();
  if (ch_in) {
    d_chin = S.ch.as<uint16_t>() + n;
    HIP_TRY(hipMemcpyAsync(d_chin, ch_in, size_t(n) * 2, hipMemcpyHostToDevice, st));
  }
  const int64_t blocks = (n * kGroupW + kBlockThreads - 1) / kBlockThreads;
  if (ix->mode == 4)
    hipLaunchKernelGGL(block_request_kernel_pack2, dim3(uint32_t((n + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads),
                       0, st, ix->dev, n, S.rows.as<int64_t>(), d_chin, d_chout, S.occ.as<int64_t>(), S.off.as<int64_t>());
  else if (ix->mode == 3)
    hipLaunchKernelGGL(block_request_kernel_pack, dim3(uint32_t((n + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads),
                       0, st, ix->dev, n, S.rows.as<int64_t>(), d_chin, d_chout, S.occ.as<int64_t>(), S.off.as<int64_t>());
  else if (ix->mode >= 1)
    hipLaunchKernelGGL(block_request_kernel_lane, dim3(uint32_t((n + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads),
                       0, st, ix->dev, n, S.rows.as<int64_t>(), d_chin, d_chout, S.occ.as<int64_t>(), S.off.as<int64_t>());
  else
    hipLaunchKernelGGL((block_request_kernel<kGroupW>), dim3(uint32_t(blocks)), dim3(kBlockThreads), 0, st, ix->dev, n,
                       S.rows.as<int64_t>(), d_chin, d_chout, S.occ.as<int64_t>(), S.off.as<int64_t>());
  HIP_TRY(hipGetLastError());
  std::vector<uint16_t> chs((size_t(n)));
  std::vector<int64_t> occ((size_t(n)));
  HIP_TRY(hipMemcpyAsync(chs.data(), d_chout, size_t(n) * 2, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(occ.data(), S.occ.p, size_t(n) * 8, hipMemcpyDeviceToHost, st));
  if (off_out) HIP_TRY(hipMemcpyAsync(off_out, S.off.p, size_t(n) * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (ch_out) memcpy(ch_out, chs.data(), size_t(n) * 2);
  if (occ_out) {
    // occs_in_block = Occ - (C[ch] + block_occs[ch][block])   (HDR_BACK sum, src/main/index.c:1740-1746)
    const size_t bo_off = 88 + 8 * size_t(kAlphaSize);
    for (int64_t i = 0; i < n; i++) {
      const int ch = ch_in ? ch_in[i] : chs[size_t(i)];
      const int64_t blk = rows[i] / h.block_size;
      const uint8_t* p = h.header.data() + bo_off + 8 * (size_t(ch) * size_t(h.number_of_blocks) + size_t(blk));
      uint64_t v = 0;
      for (int k = 0; k < 8; k++) v = (v << 8) | p[k];
      occ_out[i] = int32_t(occ[size_t(i)] - h.C[size_t(ch)] - int64_t(v));
    }
  }
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_forward_steps(femto_amd_index_t* ix, int64_t n, const int64_t* rows, uint16_t* ch_out, int64_t* row_out,
                            int64_t* off_out) {
  API_BEGIN
  if (!ix || (n && (!rows || !ch_out || !row_out || !off_out))) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (!ix->children.empty()) return femto_amd_forward_steps(ix->children[0], n, rows, ch_out, row_out, off_out);
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (!ix->host.dir_regular) return set_err(FEMTO_AMD_ERR_INVALID, "forward steps need the derived mark-table directory");
  if (ix->split_parts > 0) return set_err(FEMTO_AMD_ERR_INVALID, "forward steps are not available on a range-split index");
  for (int64_t i = 0; i < n; i++)
    if (rows[i] < 0 || rows[i] >= ix->host.total_length) return set_err(FEMTO_AMD_ERR_PARAM, "row out of range");
  if (n == 0) return FEMTO_AMD_OK;
  Lease L(ix);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  hipStream_t st = S.stream;
  if ((rc = S.rows.reserve(size_t(n) * 8))) return rc;
  if ((rc = S.ch.reserve(size_t(n) * 4))) return rc;
  if ((rc = S.occ.reserve(size_t(n) * 8))) return rc;
  if ((rc = S.off.reserve(size_t(n) * 8))) return rc;
  HIP_TRY(hipMemcpyAsync(S.rows.p, rows, size_t(n) * 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(forward_kernel, dim3(uint32_t((n + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, st,
                     ix->dev, n, S.rows.as<int64_t>(), S.ch.as<uint16_t>(), S.occ.as<int64_t>(), S.off.as<int64_t>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ch_out, S.ch.p, size_t(n) * 2, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(row_out, S.occ.p, size_t(n) * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(off_out, S.off.p, size_t(n) * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return FEMTO_AMD_OK;
  API_END
}

// Compulsory traffic of a batch: runs count (+ clamp + scan) and then the row expansion + locate walk through the TRACED
// twins of the direct pipeline's kernels (trace_kernels.hip) and reports, per traced array, how many DISTINCT 128-byte
// lines each phase loaded.
int femto_amd_trace_lines(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                          const int64_t* d_starts, int max_occs_each, int64_t* count_lines /* [10] */,
                          int64_t* locate_lines /* [10] */, int64_t* rows_out) {
  API_BEGIN
  namespace ta = femto_amd_trace_api;
  if (!ix || !count_lines || !locate_lines) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (npats <= 0 || npats >= (int64_t(1) << 31)) return set_err(FEMTO_AMD_ERR_PARAM, "trace: 1 .. 2^31-1 patterns");
  if (!use_direct(ix)) return set_err(FEMTO_AMD_ERR_INVALID, "the line trace follows the direct pipeline (modes 3/4 with the level table)");
  if (ta::traced_dev_index_bytes() != sizeof(DevIndex)) return set_err(FEMTO_AMD_ERR_INVALID, "traced kernels were built from different tables");
  Lease L(ix);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  hipStream_t st = S.stream;
  const int64_t n = ix->host.total_length;
  int64_t region_lines[kTraceRegions] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  region_lines[kTraceInd] = ix->ind_bytes / 128;
  region_lines[kTracePack] = ix->dev.pack ? (n + kPackRows - 1) / kPackRows : 0;
  region_lines[kTraceKtab] = ix->ktab2_bytes / 128 + 4;
  region_lines[kTraceSa] = (ix->dev.sa_full ? n : ix->n_marks) / 16 + 1;
  region_lines[kTraceL1] = ix->p2_lines1;
  region_lines[kTraceL2] = ix->p2_lines2;
  region_lines[kTraceTxt] = ix->dev.txt ? (n + 64) / 128 + 1 : 0;
  region_lines[kTraceIsa] = ix->dev.isa8 ? ((n >> ix->dev.isa_shift) + 2) / 16 + 1 : 0;
  region_lines[kTraceCtx] = ix->ctx_bytes / 128 + ix->ctx2_bytes / 128;
  region_lines[kTraceRu] = ix->ru_bytes / 128 + 1;
  int64_t off[kTraceRegions + 1];
  off[0] = 0;
  for (int r = 0; r < kTraceRegions; r++) off[r + 1] = (off[r] + region_lines[r] + 63) & ~int64_t(63);
  DeviceBuffer bitmap, counts;
  static_assert(kTraceRegions <= 16, "counts buffer");
  auto body = [&]() -> int {
    int r2;
    const size_t bm_bytes = size_t(off[kTraceRegions]) / 8 + 64;
    const int64_t nblocks = (npats + kBlockThreads - 1) / kBlockThreads;
    if ((r2 = bitmap.reserve(bm_bytes))) return r2;
    if ((r2 = counts.reserve(32 * 8))) return r2;
    if ((r2 = S.first.reserve(size_t(npats + 1) * 8))) return r2;
    if ((r2 = S.last.reserve(size_t(npats + 1) * 8))) return r2;
    if ((r2 = S.noccs.reserve(size_t(npats + 1) * 4))) return r2;
    if ((r2 = S.out_starts.reserve(size_t(npats + 2) * 8))) return r2;
    if ((r2 = reserve_plan_sums(S, nblocks, true, st))) return r2;
    DevIndex d = ix->dev;
    ta::TraceArgs a{};
    a.dev = &d;
    a.mode = ix->mode;
    a.num_cus = ix->num_cus;
    a.npats = npats;
    a.plen = d_plen;
    a.pats = d_pats;
    a.starts = d_starts;
    a.max_occs = max_occs_each;
    a.first = S.first.as<int64_t>();
    a.last = S.last.as<int64_t>();
    a.noccs = S.noccs.as<int32_t>();
    a.out_starts = S.out_starts.as<int64_t>();
    a.bsums = S.bsums.as<int64_t>();
    a.parity = S.bsums_parity;
    S.bsums_clean = false;    // (the traced twins run plan_rows twice: simply clear before the next real launch)
    a.tail_items = nullptr;
    inline_tail_setup(ix, d);     // as launch_count_direct does (the hand-over case takes tail_setup's below)
    a.tail_min = d.tail_min;
    if (d.txt && !(d.sa_full && d.isa8 && d.isa_shift == 0)) {
      if ((r2 = tail_setup(ix, S, d, npats, st))) return r2;
      a.tail_items = d.tail_items;
      a.tail_min = d.tail_min;
    }
    a.flags = S.d_flags;
    a.total = S.d_total;
    a.bitmap = static_cast<uint32_t*>(bitmap.p);
    a.reads = reinterpret_cast<unsigned long long*>(counts.p) + 16;
    a.trace_off = off;
    a.stream = st;
    auto collect = [&](int64_t* out, int phase) -> int {
      HIP_TRY(hipMemcpyAsync(ix->last_trace_reads[phase], a.reads, kTraceRegions * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemsetAsync(counts.p, 0, 32 * 8, st));
      for (int r = 0; r < kTraceRegions; r++)
        if (region_lines[r])
          HIP_TRY(ta::traced_popcount(static_cast<const uint32_t*>(bitmap.p), off[r] / 32, (off[r] + region_lines[r] + 31) / 32,
                                      reinterpret_cast<unsigned long long*>(counts.p) + r, st));
      HIP_TRY(hipMemcpyAsync(out, counts.p, kTraceRegions * 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      return 0;
    };
    HIP_TRY(hipMemsetAsync(bitmap.p, 0, bm_bytes, st));
    HIP_TRY(hipMemsetAsync(counts.p, 0, 32 * 8, st));
    HIP_TRY(ta::traced_count_plan(a));
    if ((r2 = collect(count_lines, 0))) return r2;
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, S.d_total, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemsetAsync(bitmap.p, 0, bm_bytes, st));
    if (total > 0) {
      if ((r2 = S.offsets.reserve(size_t(total) * 8))) return r2;
      HIP_TRY(ta::traced_walk(a, S.offsets.as<int64_t>(), total));
    }
    if ((r2 = collect(locate_lines, 1))) return r2;
    if (rows_out) *rows_out = total;
    HIP_TRY(hipMemsetAsync(S.d_flags, 0, 4 * sizeof(int), st));
    return 0;
  };
  rc = body();
  (void)hipStreamSynchronize(st);
  bitmap.release();
  counts.release();
  return rc;
  API_END
}

int femto_amd_trace_reads(const femto_amd_index_t* ix, int64_t* count_reads, int64_t* locate_reads) {
  if (!ix || !count_reads || !locate_reads) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  for (int r = 0; r < 10; r++) {
    count_reads[r] = ix->last_trace_reads[0][r];
    locate_reads[r] = ix->last_trace_reads[1][r];
  }
  return FEMTO_AMD_OK;
}

int femto_amd_set_rank_mode(femto_amd_index_t* ix, int mode) {
  if (!ix || mode < 0 || mode > 4 || mode == 2) return set_err(FEMTO_AMD_ERR_PARAM, "bad rank mode (0, 1, 3, 4)");
  if (!ix->children.empty()) {
    for (femto_amd_index* c : ix->children) { int rc = femto_amd_set_rank_mode(c, mode); if (rc) return rc; }
    return FEMTO_AMD_OK;
  }
  if (mode == 4 && !ix->dev.p2_l1)
    return set_err(FEMTO_AMD_ERR_INVALID, "two-level lines (mode 4) are built for indexes with 9..256 distinct characters (FEMTO_AMD_PACK2=1 forces them for fewer)");
  if (mode == 3 && !ix->dev.pack)
    return set_err(FEMTO_AMD_ERR_INVALID, "packed lines (mode 3) exist only for indexes with at most 8 distinct characters");
  if (mode >= 1 && !ix->host.dir_regular)
    return set_err(FEMTO_AMD_ERR_INVALID, "this index has a short non-final segment: only the raw walk (mode 0) applies");
  if (ix->split_parts > 0 && mode != 1) return set_err(FEMTO_AMD_ERR_INVALID, "a range-split index runs the lane kernels (mode 1) only");
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->mode = mode;
  return FEMTO_AMD_OK;
}

int femto_amd_get_rank_mode(const femto_amd_index_t* ix) {
  if (ix && !ix->children.empty()) return ix->children[0]->mode;
  return ix ? ix->mode : -1;
}

int femto_amd_set_option(femto_amd_index_t* ix, const char* name, int value) {
  if (!ix || !name) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  for (femto_amd_index* c : ix->children) { int rc = femto_amd_set_option(c, name, value); if (rc) return rc; }
  std::lock_guard<std::mutex> lk(ix->mu);
  if (!strcmp(name, "sort")) ix->sort_queries = value != 0;
  else if (!strcmp(name, "regexp_max_iterations")) ix->regexp_max_iterations = value;
  else if (!strcmp(name, "regexp_stack_cap")) ix->regexp_stack_cap = std::min(1 << 22, std::max(16, value));
  else return set_err(FEMTO_AMD_ERR_PARAM, "unknown option");
  return FEMTO_AMD_OK;
}

int femto_amd_host_pipeline_stats(femto_amd_index_t* ix, double* out8) {
  if (!ix || !out8) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (!ix->children.empty()) return femto_amd_host_pipeline_stats(ix->children[0], out8);
  std::lock_guard<std::mutex> lk(ix->mu);
  for (int k = 0; k < 8; k++) out8[k] = ix->pipe_stats[k];
  return FEMTO_AMD_OK;
}

int femto_amd_pack_info(const femto_amd_index_t* ix, int* available, int64_t* bytes, double* build_ms, int* ktab_syms) {
  if (!ix) return set_err(FEMTO_AMD_ERR_PARAM, "null index");
  if (!ix->children.empty()) return femto_amd_pack_info(ix->children[0], available, bytes, build_ms, ktab_syms);
  if (available) *available = ix->dev.pack != nullptr;
  if (ktab_syms) *ktab_syms = ix->dev.ktab2 ? ix->dev.kt2_syms : 0;
  if (available && ix->dev.p2_l1) *available |= 2;   // bit 1: the two-level lines (mode 4) exist
  if (available && ix->dev.ktab2) *available |= 4;   // bit 2: the level table of the direct pipeline exists
  if (available && ix->dev.ind) *available |= 32;     // bit 5: per-character rank lines (byte alphabets)
  if (available && ix->dev.ctx) *available |= 64 | (ix->dev.ctx_syms << 8);   // bit 6: context table; bits 8-11: its H
  if (available && ix->dev.ctx2) *available |= ix->dev.ctx2_syms << 12;        // bits 12-16: H2 of the wide context table
  if (available && ix->dev.ru) *available |= 1 << 20;   // bit 20: rank units (small alphabets)
  if (available && ix->dev.sa_full) *available |= 8;  // bit 3: the full suffix array is resident
  if (available && ix->dev.isa8 && ix->dev.isa_shift == 0) *available |= 16;   // bit 4: the full inverse suffix array
  if (bytes) *bytes = ix->pack_bytes + ix->pack2_bytes;
  if (build_ms) *build_ms = ix->pack_build_ms + ix->pack2_build_ms;
  return FEMTO_AMD_OK;
}

int femto_amd_structures(const femto_amd_index_t* ix, int64_t* out, int n) {
  if (!ix || !out || n < 0 || n > 16) return set_err(FEMTO_AMD_ERR_PARAM, "bad argument");
  if (!ix->children.empty()) return femto_amd_structures(ix->children[0], out, n);
  int64_t v[16] = {0};
  v[0] = int64_t(ix->host.image.size());
  v[1] = ix->pack_bytes - ix->marks_bytes;
  v[2] = ix->marks_bytes;
  v[3] = ix->ru_bytes;
  v[4] = ix->ktab2_bytes;
  v[5] = ix->ctx_bytes + ix->ctx2_bytes;
  v[6] = ix->ind_bytes;
  v[7] = ix->text_bytes;
  v[8] = (ix->p2_lines1 + ix->p2_lines2) * 128;
  v[9] = ix->table_bytes;
  v[10] = ix->mark_every_used;
  v[11] = ix->dev.ktab2 ? ix->dev.kt2_syms : 0;
  v[12] = ix->dev.pack_sa ? (ix->dev.pack_sa32 ? 4 : 8) : 0;
  v[13] = ix->hbm_held;
  for (const auto& t : ix->small_tables) v[13] += int64_t(t.second);
  for (int i = 0; i < n; i++) out[i] = v[i];
  return FEMTO_AMD_OK;
}

void femto_amd_kernel_time_enable(femto_amd_index_t* ix, int on) {
  if (ix) ix->timing = on != 0;
}

void femto_amd_kernel_time_reset(femto_amd_index_t* ix) {
  if (!ix) return;
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->t_count.drain();
  ix->t_locate.drain();
  ix->t_count.total_ms = ix->t_locate.total_ms = 0;
  ix->t_count.launches = ix->t_locate.launches = 0;
}

int femto_amd_kernel_time_ms(femto_amd_index_t* ix, const char* kernel, double* avg_ms, int64_t* n_launches) {
  if (!ix || !kernel) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  KernelTimer* t = nullptr;
  if (!strcmp(kernel, "count")) t = &ix->t_count;
  else if (!strcmp(kernel, "locate")) t = &ix->t_locate;
  else return set_err(FEMTO_AMD_ERR_PARAM, "unknown kernel name");
  t->drain();
  if (avg_ms) *avg_ms = t->launches ? t->total_ms / double(t->launches) : 0.0;
  if (n_launches) *n_launches = t->launches;
  return FEMTO_AMD_OK;
}

namespace {
int collect_docs(int ndocs, const uint8_t* const* docs, const int64_t* doc_lens, const char* const* doc_infos,
                 std::vector<Document>* out) {
  if (ndocs <= 0 || !docs || !doc_lens) return set_err(FEMTO_AMD_ERR_PARAM, "an index needs at least one document");
  for (int i = 0; i < ndocs; i++) {
    if (doc_lens[i] < 0 || (doc_lens[i] && !docs[i])) return set_err(FEMTO_AMD_ERR_PARAM, "bad document");
    out->push_back(Document{docs[i], doc_lens[i], (doc_infos && doc_infos[i]) ? std::string(doc_infos[i]) : std::string()});
  }
  return 0;
}
int host_threads() {
  unsigned n = std::thread::hardware_concurrency();
  return n ? int(n) : 4;
}
}  // namespace

int femto_amd_build_index_from_sa(const char* out_dir, int ndocs, const uint8_t* const* docs, const int64_t* doc_lens,
                                  const char* const* doc_infos, const char* params, const int64_t* sa) {
  if (!out_dir || !sa) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  std::vector<Document> d;
  int rc = collect_docs(ndocs, docs, doc_lens, doc_infos, &d);
  if (rc) return rc;
  BuildParams bp;
  Error e{0, ""};
  if ((rc = parse_build_params(params, &bp, &e))) return set_err(rc, e.msg);
  if ((rc = build_index_from_sa(out_dir, d, bp, sa, host_threads(), &e))) return set_err(rc, e.msg);
  return FEMTO_AMD_OK;
}

int femto_amd_build_index(const char* out_dir, int ndocs, const uint8_t* const* docs, const int64_t* doc_lens,
                          const char* const* doc_infos, const char* params, int device) {
  if (!out_dir) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  std::vector<Document> d;
  int rc = collect_docs(ndocs, docs, doc_lens, doc_infos, &d);
  if (rc) return rc;
  BuildParams bp;
  Error e{0, ""};
  if ((rc = parse_build_params(params, &bp, &e))) return set_err(rc, e.msg);
  std::vector<uint16_t> text;
  std::vector<int64_t> doc_ends, sa;
  prepare_text(d, &text, &doc_ends);
  if ((rc = gpu_suffix_sort(text, device, &sa, &e))) return set_err(rc, e.msg);
  text.clear();
  text.shrink_to_fit();
  if ((rc = build_index_from_sa(out_dir, d, bp, sa.data(), host_threads(), &e))) return set_err(rc, e.msg);
  return FEMTO_AMD_OK;
}

int femto_amd_flatten_index(const char* index_dir, const char* out_path) {
  if (!index_dir || !out_path) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  Error e{0, ""};
  int rc = flatten_index_dir(index_dir, out_path, &e);
  return rc ? set_err(rc, e.msg) : FEMTO_AMD_OK;
}

/* test hook: bseq_construct_forcetype-compatible encoder (src/main/wtree.c:365) */
int femto_amd_bseq_encode(const uint8_t* bits_msb_first, int64_t bitlen, int force_type, uint8_t* out, int64_t cap,
                          int64_t* out_len) {
  if (!bits_msb_first || bitlen <= 0 || !out_len) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  std::vector<uint8_t> z;
  bseq_encode(bits_msb_first, bitlen, force_type, &z);
  *out_len = int64_t(z.size());
  if (out) {
    if (cap < int64_t(z.size())) return set_err(FEMTO_AMD_ERR_PARAM, "output buffer too small");
    memcpy(out, z.data(), z.size());
  }
  return FEMTO_AMD_OK;
}

}  // extern "C"
