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

// Concurrent callers lease one scratch -- one stream -- each (api_internal.hpp), and streams that share a hardware queue run their
// kernels one after the other: with the runtime's default of four queues, four callers' automaton batches ran as two lanes of two
// (2.2 s for what the longest search needs 1.45 s for; with 16 queues 1.6 s, eight callers 91 k automata/s --
// profiles/r06_regexp_concurrent.txt).  The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises, so the library asks for 16
// when it is loaded -- unless the variable is set already, and with no effect if the process has initialised HIP before.
__attribute__((constructor)) void femto_amd_ask_for_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "16", 0); }
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
    const int64_t held = hbm_held_all(ix);
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

// femto's wavelet tree as segment lines (HostIndex::segs -> d_segs): the source of every derivation and the data of modes 0/1,
// not read by the derived layouts' kernels.  A handle with a budget releases them after open (0.76 GB of a 1 GiB DNA index)
// and uploads them again -- counted in hbm_held -- when a call needs them.
// (round 6: the block images go and come with them -- in modes 3 / 4 only the same callers read them: LOCATION leaf requests and
// forward steps through femto's own mark arrays; 0.55 GB of a 1 GiB DNA index, 0.6 GB of a sigma~96 one)
int release_wavelet_lines(femto_amd_index* ix) {
  if ((!ix->d_segs && !ix->d_image && !ix->d_cum) || ix->split_parts > 0 || !ix->stripe_devices.empty() || ix->borrowed) return 0;
  HIP_TRY(hipDeviceSynchronize());
  if (ix->d_segs) {
    big_free(ix, ix->d_segs);
    ix->table_bytes -= int64_t(ix->host.segs.size() * 8);
  }
  big_free(ix, ix->d_image);
  ix->d_segs = nullptr;
  ix->dev.segs = nullptr;
  ix->d_image = nullptr;
  ix->dev.image = nullptr;
  // ... and the per-segment / per-block directories of femto's sequences (the lane kernels' cum / hint / block directory: 0.4 GB of a
  // 1 GiB sigma~96 index), which only those same kernels read
  auto drop_small = [&](void* p, size_t counted) {
    if (!p) return;
    for (size_t k = 0; k < ix->small_tables.size(); k++)
      if (ix->small_tables[k].first == p) {
        ix->small_tables.erase(ix->small_tables.begin() + long(k));
        break;
      }
    (void)hipFree(p);
    ix->table_bytes -= int64_t(counted);
  };
  drop_small(ix->d_cum, ix->host.cum.size() * sizeof(CumEntry));
  drop_small(ix->d_hint, ix->host.hint.size() * 4);
  drop_small(ix->d_bdir, ix->host.bdir.size() * sizeof(BlockDir));
  ix->d_cum = nullptr;
  ix->d_hint = nullptr;
  ix->d_bdir = nullptr;
  ix->dev.cum = nullptr;
  ix->dev.hint = nullptr;
  ix->dev.bdir = nullptr;
  ix->segs_released = true;
  ix->segs_drop_ok = true;
  return 0;
}
int ensure_wavelet_lines(femto_amd_index* ix) {
  if (!ix->segs_released || (ix->d_segs && ix->d_image && ix->d_cum && ix->d_hint && ix->d_bdir)) return 0;
  HostIndex& h = ix->host;
  if (!ix->d_segs) {
    const size_t sb = h.segs.size() * 8, slack = (size_t(h.b_size) / 511 + 4) * 128;
    if (big_malloc(ix, reinterpret_cast<void**>(&ix->d_segs), sb + slack) != hipSuccess) {
      (void)hipGetLastError();
      ix->d_segs = nullptr;
      return set_err(FEMTO_AMD_ERR_MEM, "no HBM for femto's wavelet segment lines (modes 0/1)");
    }
    if (sb) HIP_TRY(big_h2d(ix, ix->d_segs, h.segs.data(), sb));
    HIP_TRY(big_memset(ix, reinterpret_cast<char*>(ix->d_segs) + sb, 0, slack));
    ix->table_bytes += int64_t(sb);
  }
  if (!ix->d_image) {
    const size_t image_slack = size_t(h.b_size) * size_t(h.text_size_bits) / 8 + 64;      // (as at open: a mark array read one bucket too far)
    if (big_malloc(ix, reinterpret_cast<void**>(&ix->d_image), h.image.size() + image_slack) != hipSuccess) {
      (void)hipGetLastError();
      ix->d_image = nullptr;
      return set_err(FEMTO_AMD_ERR_MEM, "no HBM for femto's block images (modes 0/1, LOCATION requests, forward steps)");
    }
    HIP_TRY(big_h2d(ix, ix->d_image, h.image.data(), h.image.size()));
    HIP_TRY(big_memset(ix, ix->d_image + h.image.size(), 0, image_slack));
  }
  {      // the sequences' directories, registered with the handle's small tables like their first upload (they count against the budget)
    struct Registry {
      std::vector<std::pair<void*, size_t>>* old;
      explicit Registry(std::vector<std::pair<void*, size_t>>* mine) : old(g_small_registry) { g_small_registry = mine; }
      ~Registry() { g_small_registry = old; }
    } reg(&ix->small_tables);
    int r;
    if (!ix->d_cum && (r = upload(&ix->d_cum, h.cum, &ix->table_bytes))) return r;
    if (!ix->d_hint && (r = upload(&ix->d_hint, h.hint, &ix->table_bytes))) return r;
    if (!ix->d_bdir && (r = upload(&ix->d_bdir, h.bdir, &ix->table_bytes, (size_t(h.b_size) / 512 + 4) * sizeof(BlockDir)))) return r;
  }
  ix->dev.segs = ix->d_segs;
  ix->dev.image = ix->d_image;
  ix->dev.cum = ix->d_cum;
  ix->dev.hint = ix->d_hint;
  ix->dev.bdir = ix->d_bdir;
  ix->segs_released = false;
  return 0;
}

int64_t hbm_held_all(const femto_amd_index* ix) {
  int64_t held = ix->hbm_held;
  for (const auto& t : ix->small_tables) held += int64_t(t.second);
  return held;
}

int WaveletLinesUse::acquire() {
  std::lock_guard<std::mutex> lk(ix->mu);
  if (int rc = ensure_wavelet_lines(ix)) return rc;
  ix->segs_users++;
  held = true;
  return 0;
}
WaveletLinesUse::~WaveletLinesUse() {
  if (!held) return;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (--ix->segs_users > 0) return;
  if (ix->segs_drop_ok && ix->mode >= 3 && (ix->d_segs || ix->d_image) && ix->opt.hbm_budget_bytes >= 0 && hbm_held_all(ix) > ix->opt.hbm_budget_bytes)
    (void)release_wavelet_lines(ix);      // (waits for the device: every kernel that read them has ended)
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

namespace femto_amd {



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
  if (ix->mode == 3 && d.ru && d.ru_marks) {
    if (d.sa_full) hipLaunchKernelGGL((count_tail_kernel<RumPolicy, true>), grid, block, 0, stream, d, items, n_items, d_plen, d_pats, d_starts, perm, keys, bits, nsym, out, err_flag);
    else hipLaunchKernelGGL((count_tail_kernel<RumPolicy, false>), grid, block, 0, stream, d, items, n_items, d_plen, d_pats, d_starts, perm, keys, bits, nsym, out, err_flag);
  } else if (ix->mode == 3 && d.ru) {
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
// jumped at once; a pattern that occurs pays two lines more.  With the rank units a step is ONE line, so three symbols to go
// cost the same three requests either way -- but a tail also hands the row's text position to the row expansion, which then
// skips its suffix-array read: from three symbols on small alphabets (cfg 5, whose 20-mers leave the K = 16 table with ~2
// rows and reach one row a step or two later: 1.53 -> 1.47 ms/step, rows 0.28 -> 0.18 ms; two symbols: 1.48; cfg 3: no change).
void inline_tail_setup(const femto_amd_index* ix, DevIndex& d) {
  d.tail_min = ix->mode == 3 ? 3 : 4;
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
  // (a row-free launch needs no way back from a position to a row: the suffix array and the text are enough, inline_tail_applies)
  const bool inline_tail = inline_tail_applies(d, plan && plan->row_free);
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
  // one-row patterns whose text position the search already knows (the text tail's, the wide context table's): handed to
  // plan_rows_kernel, which then skips their suffix-array read (S.noccs64 is free on this path: the block sums replace the scan)
  // ... or, on a handle WITHOUT the suffix array whose rank units carry the rows' mark bits (ru_kernels.hip.hpp), a marked
  // row the search stood on: plan_rows_kernel starts from it instead of walking from the final row ("mark spotting")
  const bool spot = plan && !dense && ix->mode == 3 && d.ru && d.ru_marks && d.pack && d.pack_sa;
  // ... or, on a row-free launch (Plan::row_free), the position a text tail found the pattern at (count_tail_kernel on handles with
  // the sampled arrays): plan_rows_kernel then has nothing to walk for that pattern
  d.row_free = (plan && plan->row_free) ? 1 : 0;
  int64_t* sa_out = nullptr;
  if (plan && (dense || spot || (d.row_free && tail))) {
    if ((rc = S.noccs64.reserve(size_t(npats + 1) * 8))) return rc;
    sa_out = S.noccs64.as<int64_t>();
    plan->sa_known = sa_out;
  }
#define LAUNCH_COUNT_DIRECT(POLICY)                                                                                                        \
  do {                                                                                                                                     \
    if (plan && dense) hipLaunchKernelGGL((count_direct_kernel<POLICY, true, true>), grid, block, 0, stream, d, npats, d_plen, d_pats, d_starts, d_first, d_last, S.err, mo, noccs, ps, big_flag, sa_out); \
    else if (plan) hipLaunchKernelGGL((count_direct_kernel<POLICY, true, false>), grid, block, 0, stream, d, npats, d_plen, d_pats, d_starts, d_first, d_last, S.err, mo, noccs, ps, big_flag, sa_out);       \
    else if (dense) hipLaunchKernelGGL((count_direct_kernel<POLICY, false, true>), grid, block, 0, stream, d, npats, d_plen, d_pats, d_starts, d_first, d_last, S.err, mo, noccs, ps, big_flag, sa_out);     \
    else hipLaunchKernelGGL((count_direct_kernel<POLICY, false, false>), grid, block, 0, stream, d, npats, d_plen, d_pats, d_starts, d_first, d_last, S.err, mo, noccs, ps, big_flag, sa_out);              \
  } while (0)
  if (ix->mode == 3 && d.ru && d.ru_marks) LAUNCH_COUNT_DIRECT(RumPolicy);     // ... of 64 rows with their mark bits
  else if (ix->mode == 3 && d.ru) LAUNCH_COUNT_DIRECT(RuPolicy);     // rank units: one 16-byte load per range end and step
  else if (ix->mode == 3) LAUNCH_COUNT_DIRECT(PackPolicy);
  else if (d.ind) LAUNCH_COUNT_DIRECT(IndPolicy);     // per-character rank lines: one line per range end and step
  else LAUNCH_COUNT_DIRECT(Pack2Policy);
#undef LAUNCH_COUNT_DIRECT
  HIP_TRY(hipGetLastError());
  if (tail) {   // persistent grid: the number of handed-over patterns is only known on the device
    const TailOut out{nullptr, d_first, d_last, noccs, bsums, mo, sa_out, d.row_free};
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
                 const int64_t* d_starts, int64_t* d_first, int64_t* d_last, hipStream_t stream, Plan* plan) {
  const int64_t max_chunk = ix->mode == 0 ? (int64_t(1) << 25) : (int64_t(1) << 31);
  if (plan && npats <= max_chunk) return launch_count_chunk(ix, S, npats, d_plen, d_pats, d_starts, d_first, d_last, stream, plan);
  for (int64_t off = 0; off < npats; off += max_chunk) {
    const int64_t cnt = std::min<int64_t>(max_chunk, npats - off);
    int rc = launch_count_chunk(ix, S, cnt, d_plen + off, d_pats, d_starts + off, d_first + off, d_last ? d_last + off : nullptr, stream);
    if (rc) return rc;
  }
  return 0;
}

// (shared with api_open.hip, whose derivations prefix-sum their per-line counts)
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
// key batches (count_keys_kernel): host-pointer key chunks, and -- with a plan -- femto_amd_locate_keys_device
int launch_count_keys(femto_amd_index* ix, int64_t n, const uint64_t* d_keys, int2* out32, int64_t* d_first, int64_t* d_last, hipStream_t stream,
                      Scratch* S, Plan* plan) {
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
  // a handle that walks to marks and whose rank units carry the mark bits: the search hands plan_rows_kernel a marked row it
  // stood on (launch_count_direct's "spot")
  int64_t* sa_out = nullptr;
  if (plan && S && ix->mode == 3 && ix->dev.ru && ix->dev.ru_marks && ix->dev.pack && ix->dev.pack_sa && !ix->dev.sa_full) {
    if ((rc = S->noccs64.reserve(size_t(n + 1) * 8))) return rc;
    sa_out = S->noccs64.as<int64_t>();
    plan->sa_known = sa_out;
  }
  hipEvent_t e0, e1;
  timer_begin(ix, ix->t_count, stream, &e0, &e1);
#define LAUNCH_KEYS(POLICY)                                                                                                                                  \
  do {                                                                                                                                                       \
    if (plan) hipLaunchKernelGGL((count_keys_kernel<POLICY, true>), grid, block, 0, stream, ix->dev, n, d_keys, bits, nsym, out32, d_first, d_last, mo, noccs, ps, big_flag, sa_out); \
    else hipLaunchKernelGGL((count_keys_kernel<POLICY, false>), grid, block, 0, stream, ix->dev, n, d_keys, bits, nsym, out32, d_first, d_last, mo, noccs, ps, big_flag, sa_out);     \
  } while (0)
  if (ix->mode == 3 && ix->dev.ru && ix->dev.ru_marks) LAUNCH_KEYS(RumPolicy);
  else if (ix->mode == 3 && ix->dev.ru) LAUNCH_KEYS(RuPolicy);
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
                     int64_t* d_out_starts, int64_t* d_offsets, int64_t capacity, hipStream_t stream, const int2* d_first32,
                     bool fuse_walk, const int64_t* d_sa_known) {
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
                       d_offsets, capacity, big_flag, ix->dev, S.d_total, S.total_user, d_sa_known);                                      \
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

// clamp of a whole batch counted without a plan (host-pointer chunks): do_locate_query's rule, then the caller scans
int launch_clamp(int64_t npats, const int64_t* d_first, const int64_t* d_last, int max_occs, int32_t* d_noccs, int64_t* d_noccs64, hipStream_t stream) {
  if (npats <= 0) return 0;
  hipLaunchKernelGGL(clamp_kernel, dim3(uint32_t((npats + 255) / 256)), dim3(256), 0, stream, npats, d_first, d_last, max_occs, d_noccs, d_noccs64);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace femto_amd

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
    ix->t_resolve.destroy();
    ix->t_regexp.destroy();
    if (ix->d_doc_ends) (void)hipFree(ix->d_doc_ends);
    if (ix->nfa_active) (void)hipHostFree(ix->nfa_active);
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
                      static_cast<void*>(ix->d_ctx), static_cast<void*>(ix->d_ctx2), static_cast<void*>(ix->d_ctxm), static_cast<void*>(ix->d_ru), static_cast<void*>(ix->d_ru_stop)})
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
  out->image_bytes = (ix->device >= 0 && !ix->d_image && ix->split_parts == 0 && ix->children.empty() && !ix->borrowed) ? 0 : int64_t(h.image.size());      // (released with the wavelet lines on a bounded handle)
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
  // d_first == d_last == NULL: the ROW-FREE form -- what parallel_locate itself returns (noccs and offsets, src/main/femto.c:331-400)
  const bool row_free = !d_first && !d_last;
  if (npats && ((!row_free && (!d_first || !d_last)) || !d_noccs || !d_out_starts || !d_total)) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  int rc = ensure_device(ix);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  Lease L(ix, stream);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  Plan plan{max_occs_each, d_noccs, d_out_starts, offsets_capacity, false, d_total};
  if (row_free && npats) {      // the rows the row expansion still needs live in the call's scratch
    if ((rc = S.first.reserve(size_t(npats + 1) * 8))) return rc;
    d_first = S.first.as<int64_t>();
    if (use_direct(ix)) {
      plan.row_free = true;
    } else {                    // femto's own wavelet tree (modes 0 / 1): the search is the same either way
      if ((rc = S.last.reserve(size_t(npats + 1) * 8))) return rc;
      d_last = S.last.as<int64_t>();
    }
  }
  if ((rc = launch_count_plan(ix, S, npats, d_plen, d_pats, d_starts, d_first, d_last, &plan, stream))) return rc;
  if (plan.done) {   // direct pipeline: one stream-ordered chain, nothing returns to the host (d_total: plan_rows_kernel's last block)
    if ((rc = launch_plan_rows(ix, S, npats, d_noccs, d_first, d_out_starts, d_offsets, offsets_capacity, stream, nullptr, /*fuse_walk=*/true, plan.sa_known))) return rc;
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

int femto_amd_lf_steps_device(femto_amd_index_t* ix, int64_t n, const int64_t* d_rows, int64_t* d_next, int64_t* d_off, void* stream_) {
  API_BEGIN
  if (!ix || n < 0 || (n && (!d_rows || !d_next || !d_off))) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (n >= (int64_t(1) << 31)) return set_err(FEMTO_AMD_ERR_PARAM, "at most 2^31 - 1 rows per call");
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (n == 0) return FEMTO_AMD_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const dim3 grid{uint32_t((n + kBlockThreads - 1) / kBlockThreads)}, block{uint32_t(kBlockThreads)};
  if (ix->mode == 3) {
    hipLaunchKernelGGL(lf_steps_kernel<PackPolicy>, grid, block, 0, stream, ix->dev, n, d_rows, d_next, d_off);
  } else if (ix->mode == 4) {
    hipLaunchKernelGGL(lf_steps_kernel<Pack2Policy>, grid, block, 0, stream, ix->dev, n, d_rows, d_next, d_off);
  } else {      // femto's own tables (and their own marks): the leaf kernel's answers, then LF from them
    if (!ix->host.dir_regular) return set_err(FEMTO_AMD_ERR_INVALID, "LF steps need the derived segment lines");
    Lease L(ix, stream);
    if (!L.s) return L.rc;
    Scratch& S = *L.s;
    if ((rc = S.ch.reserve(size_t(n) * 2))) return rc;
    if ((rc = S.occ.reserve(size_t(n) * 8))) return rc;
    hipLaunchKernelGGL(block_request_kernel_lane, grid, block, 0, stream, ix->dev, n, d_rows, static_cast<const uint16_t*>(nullptr), S.ch.as<uint16_t>(),
                       S.occ.as<int64_t>(), d_off);
    hipLaunchKernelGGL(lf_from_leaf_kernel, grid, block, 0, stream, n, static_cast<const uint16_t*>(S.ch.as<uint16_t>()), static_cast<const int64_t*>(S.occ.as<int64_t>()), d_next, d_off);
  }
  HIP_TRY(hipGetLastError());
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_key_table_id(const femto_amd_index_t* ix, uint64_t* id) {
  if (!ix || !id) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (!ix->children.empty()) return femto_amd_key_table_id(ix->children[0], id);
  if (!use_direct(ix) || ix->h_dense.empty())
    return set_err(FEMTO_AMD_ERR_INVALID, "keys need the packed layouts (modes 3 / 4) and at most 255 distinct characters");
  uint64_t h = 0xcbf29ce484222325ull ^ uint64_t(ix->dense_bits);      // FNV-1a over bits and the 261 fields
  for (size_t c = 0; c < size_t(kAlphaSize) && c < ix->h_dense.size(); c++) h = (h ^ ix->h_dense[c]) * 0x100000001b3ull;
  *id = h;
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
  if ((rc = launch_plan_rows(ix, S, npats, d_noccs, d_first, d_out_starts, d_offsets, offsets_capacity, stream, r32, /*fuse_walk=*/true, plan.sa_known))) return rc;
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
  // LOCATION requests are answered from femto's own mark tables, whatever the mode (lane_mark_offset); CHAR / OCCS requests in
  // modes 3 / 4 read the derived lines only and leave femto's segment lines where they are
  WaveletLinesUse wl(ix);
  if ((off_out || ix->mode <= 1) && (rc = wl.acquire())) return rc;
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
  int64_t* const d_off = off_out ? S.off.as<int64_t>() : nullptr;      // (the kernels look a row's offset up only when asked)
  if (ix->mode == 4)
    hipLaunchKernelGGL(block_request_kernel_pack2, dim3(uint32_t((n + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads),
                       0, st, ix->dev, n, S.rows.as<int64_t>(), d_chin, d_chout, S.occ.as<int64_t>(), d_off);
  else if (ix->mode == 3)
    hipLaunchKernelGGL(block_request_kernel_pack, dim3(uint32_t((n + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads),
                       0, st, ix->dev, n, S.rows.as<int64_t>(), d_chin, d_chout, S.occ.as<int64_t>(), d_off);
  else if (ix->mode >= 1)
    hipLaunchKernelGGL(block_request_kernel_lane, dim3(uint32_t((n + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads),
                       0, st, ix->dev, n, S.rows.as<int64_t>(), d_chin, d_chout, S.occ.as<int64_t>(), d_off);
  else
    hipLaunchKernelGGL((block_request_kernel<kGroupW>), dim3(uint32_t(blocks)), dim3(kBlockThreads), 0, st, ix->dev, n,
                       S.rows.as<int64_t>(), d_chin, d_chout, S.occ.as<int64_t>(), d_off);
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
  WaveletLinesUse wl(ix);      // femto's own tables answer this: bring the segment lines back if the handle released them
  if ((rc = wl.acquire())) return rc;
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
  region_lines[kTraceSa] = ix->dev.sa_full ? n / (ix->dev.sa32 ? 32 : 16) + 1 : ix->n_marks / 16 + 1;
  region_lines[kTraceL1] = ix->p2_lines1;
  region_lines[kTraceL2] = ix->p2_lines2;
  region_lines[kTraceTxt] = ix->dev.txt ? (n + 64) / 128 + 1 : 0;
  region_lines[kTraceIsa] = ix->dev.isa8 ? ((n >> ix->dev.isa_shift) + 2) / (ix->dev.sa32 ? 32 : 16) + 1 : 0;
  region_lines[kTraceCtx] = ix->ctx_bytes / 128 + ix->ctx2_bytes / 128 + ix->ctxm_bytes / 128;
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
    if ((r2 = S.noccs64.reserve(size_t(npats + 1) * 8))) return r2;
    a.sa_known = S.noccs64.as<int64_t>();      // (used where launch_count_direct would: dense handles)
    a.out_starts = S.out_starts.as<int64_t>();
    a.bsums = S.bsums.as<int64_t>();
    a.parity = S.bsums_parity;
    S.bsums_clean = false;    // (the traced twins run plan_rows twice: simply clear before the next real launch)
    a.tail_items = nullptr;
    a.row_free = ix->trace_row_free ? 1 : 0;
    inline_tail_setup(ix, d);     // as launch_count_direct does (the hand-over case takes tail_setup's below)
    a.tail_min = d.tail_min;
    if (d.txt && !inline_tail_applies(d, a.row_free != 0)) {
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
  if (!ix->children.empty()) return femto_amd_trace_reads(ix->children[0], count_reads, locate_reads);   // like femto_amd_structures
  std::lock_guard<std::mutex> lk(const_cast<femto_amd_index_t*>(ix)->mu);
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
  if (mode <= 1 && ix->device >= 0) {      // femto's own wavelet tree: its segment lines may have been released (a handle with a budget)
    HIP_TRY(hipSetDevice(ix->device));
    if (int rc = ensure_wavelet_lines(ix)) return rc;
  }
  ix->mode = mode;
  // ... and back on the derived layouts a bounded handle gives them up again when they put it over its budget
  if (mode >= 3 && ix->device >= 0 && ix->segs_drop_ok && (ix->d_segs || ix->d_image) && ix->segs_users == 0 && ix->opt.hbm_budget_bytes >= 0 &&
      hbm_held_all(ix) > ix->opt.hbm_budget_bytes) {
    HIP_TRY(hipSetDevice(ix->device));
    if (int rc = release_wavelet_lines(ix)) return rc;
  }
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
  else if (!strcmp(name, "trace_row_free")) ix->trace_row_free = value != 0;
  else if (!strcmp(name, "regexp_max_iterations")) ix->regexp_max_iterations = value;
  else if (!strcmp(name, "regexp_stack_cap")) ix->regexp_stack_cap = std::min(1 << 22, std::max(16, value));
  else return set_err(FEMTO_AMD_ERR_PARAM, "unknown option");
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
  if (available && ix->dev.ctxm) *available |= ix->dev.ctxm_syms << 24;        // bits 24-28: HM of the table in between
  if (available && ix->dev.ru) *available |= 1 << 20;   // bit 20: rank units (small alphabets)
  if (available && ix->dev.ru && ix->dev.ru_marks) *available |= 1 << 21;   // bit 21: ... the marked ones (64 rows + mark bits)
  if (available && (ix->dev.sa_full || ix->dev.isa8) && ix->dev.sa32) *available |= 1 << 22;   // bit 22: the suffix / inverse suffix arrays hold 4-byte entries
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
  v[0] = (ix->device >= 0 && !ix->d_image && ix->split_parts == 0 && !ix->borrowed) ? 0 : int64_t(ix->host.image.size());
  v[1] = ix->dev.pack ? ix->pack_bytes - ix->marks_bytes : 0;      // (byte alphabets: the marks belong to the two-level lines)
  v[2] = ix->marks_bytes;
  v[3] = ix->ru_bytes;
  v[4] = ix->ktab2_bytes;
  v[5] = ix->ctx_bytes + ix->ctx2_bytes + ix->ctxm_bytes;
  v[6] = ix->ind_bytes;
  v[7] = ix->text_bytes;
  v[8] = (ix->p2_lines1 + ix->p2_lines2) * 128;
  v[9] = ix->table_bytes;
  v[10] = ix->mark_every_used;
  v[11] = ix->dev.ktab2 ? ix->dev.kt2_syms : 0;
  v[12] = ix->dev.pack_sa ? (ix->dev.pack_sa32 ? 4 : 8) : 0;
  v[14] = ix->opt.hbm_budget_bytes;                    // the budget in force (-1: everything that is free)
  v[15] = ix->budget_is_default ? 1 : 0;
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
  for (KernelTimer* t : {&ix->t_count, &ix->t_locate, &ix->t_resolve, &ix->t_regexp}) {
    t->drain();
    t->total_ms = 0;
    t->launches = 0;
  }
}

int femto_amd_kernel_time_ms(femto_amd_index_t* ix, const char* kernel, double* avg_ms, int64_t* n_launches) {
  if (!ix || !kernel) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  KernelTimer* t = nullptr;
  if (!strcmp(kernel, "count")) t = &ix->t_count;
  else if (!strcmp(kernel, "locate")) t = &ix->t_locate;
  else if (!strcmp(kernel, "resolve")) t = &ix->t_resolve;
  else if (!strcmp(kernel, "regexp")) t = &ix->t_regexp;
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
