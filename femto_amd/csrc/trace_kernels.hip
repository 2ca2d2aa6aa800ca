// trace_kernels.hip -- the query kernels of the direct pipeline compiled a SECOND time with the line trace switched on.
//
// bench.py's roofline needs the COMPULSORY bytes of a launch: 128 B for every distinct line of a derived array the
// kernels load.  That is counted on the GPU by running the batch once through traced twins of the very same kernel
// sources: this translation unit includes the kernel headers with FEMTO_AMD_TRACE defined (trace_touch() then sets one
// bit per loaded line, pack_kernels.hip.hpp) and with the namespace renamed, so the twins do not collide with the
// production kernels -- which contain no trace code at all.
#define FEMTO_AMD_TRACE 1
#define femto_amd femto_amd_traced
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "kernels.hip.hpp"
#include "pack_kernels.hip.hpp"
#include "ru_kernels.hip.hpp"
#include "pack2_kernels.hip.hpp"
#include "ind_kernels.hip.hpp"
#include "text_kernels.hip.hpp"
#include "ctx_kernels.hip.hpp"
#include "direct_kernels.hip.hpp"
#undef femto_amd

#include "trace_api.hpp"

namespace femto_amd_trace_api {

using namespace femto_amd_traced;

size_t traced_dev_index_bytes() { return sizeof(DevIndex); }

static DevIndex make_dev(const TraceArgs& a) {
  DevIndex d;
  memcpy(&d, a.dev, sizeof d);
  d.trace = a.bitmap;
  d.trace_reads = a.reads;
  for (int r = 0; r < kTraceRegions; r++) d.trace_off[r] = a.trace_off[r];
  d.tail_items = a.tail_items;
  d.tail_count = a.flags + 2;
  d.tail_min = a.tail_min;
  d.row_free = a.row_free;
  if (!a.tail_items && !inline_tail_applies(d, a.row_free != 0)) d.txt = nullptr;
  return d;
}

// count_direct_kernel + count_tail_kernel (+ plan_super_kernel) + plan_rows_kernel (out_starts only)
hipError_t traced_count_plan(const TraceArgs& a) {
  const DevIndex d = make_dev(a);
  const int64_t nblocks = (a.npats + 255) / 256;
  const dim3 grid{uint32_t(nblocks)}, block{256};
  hipError_t e = hipMemsetAsync(a.flags, 0, 4 * sizeof(int), a.stream);
  if (e != hipSuccess) return e;
  const bool dense = inline_tail_applies(d, a.row_free != 0);   // as launch_count_direct decides
  const bool tail = d.txt && !dense;
  const PlanSums ps = plan_sums_at(a.bsums, nblocks, !tail, a.parity);
  int* big_flag = a.flags + 1;
#define LAUNCH_COUNT_DIRECT(POLICY)                                                                                                     \
  do {                                                                                                                                  \
    if (dense) hipLaunchKernelGGL((count_direct_kernel<POLICY, true, true>), grid, block, 0, a.stream, d, a.npats, a.plen, a.pats, a.starts, a.first, a.last, a.flags, a.max_occs, a.noccs, ps, big_flag, a.sa_known); \
    else hipLaunchKernelGGL((count_direct_kernel<POLICY, true, false>), grid, block, 0, a.stream, d, a.npats, a.plen, a.pats, a.starts, a.first, a.last, a.flags, a.max_occs, a.noccs, ps, big_flag, (spot || (a.row_free && tail)) ? a.sa_known : static_cast<int64_t*>(nullptr));     \
  } while (0)
  const bool spot = a.mode == 3 && d.ru && d.ru_marks && d.pack && d.pack_sa && !dense;    // as launch_count_direct decides
  if (a.mode == 3 && d.ru && d.ru_marks) LAUNCH_COUNT_DIRECT(RumPolicy);
  else if (a.mode == 3 && d.ru) LAUNCH_COUNT_DIRECT(RuPolicy);
  else if (a.mode == 3) LAUNCH_COUNT_DIRECT(PackPolicy);
  else if (d.ind) LAUNCH_COUNT_DIRECT(IndPolicy);
  else LAUNCH_COUNT_DIRECT(Pack2Policy);
#undef LAUNCH_COUNT_DIRECT
  if (tail) {
    const TailOut out{nullptr, a.first, a.last, a.noccs, a.bsums, a.max_occs, a.sa_known, a.row_free};
    const dim3 tgrid{uint32_t(std::min<int64_t>(nblocks, int64_t(a.num_cus) * 8))};
    const TailItem* items = static_cast<const TailItem*>(a.tail_items);
    const int* n_items = d.tail_count;
    const uint32_t* perm = nullptr;
    const uint64_t* keys = nullptr;
    if (a.mode == 3 && d.ru && d.ru_marks) {
      if (d.sa_full) hipLaunchKernelGGL((count_tail_kernel<RumPolicy, true>), tgrid, block, 0, a.stream, d, items, n_items, a.plen, a.pats, a.starts, perm, keys, 1, 0, out, a.flags);
      else hipLaunchKernelGGL((count_tail_kernel<RumPolicy, false>), tgrid, block, 0, a.stream, d, items, n_items, a.plen, a.pats, a.starts, perm, keys, 1, 0, out, a.flags);
    } else if (a.mode == 3 && d.ru) {
      if (d.sa_full) hipLaunchKernelGGL((count_tail_kernel<RuPolicy, true>), tgrid, block, 0, a.stream, d, items, n_items, a.plen, a.pats, a.starts, perm, keys, 1, 0, out, a.flags);
      else hipLaunchKernelGGL((count_tail_kernel<RuPolicy, false>), tgrid, block, 0, a.stream, d, items, n_items, a.plen, a.pats, a.starts, perm, keys, 1, 0, out, a.flags);
    } else if (a.mode == 3) {
      if (d.sa_full) hipLaunchKernelGGL((count_tail_kernel<PackPolicy, true>), tgrid, block, 0, a.stream, d, items, n_items, a.plen, a.pats, a.starts, perm, keys, 1, 0, out, a.flags);
      else hipLaunchKernelGGL((count_tail_kernel<PackPolicy, false>), tgrid, block, 0, a.stream, d, items, n_items, a.plen, a.pats, a.starts, perm, keys, 1, 0, out, a.flags);
    } else {
      if (d.sa_full) hipLaunchKernelGGL((count_tail_kernel<Pack2Policy, true>), tgrid, block, 0, a.stream, d, items, n_items, a.plen, a.pats, a.starts, perm, keys, 1, 0, out, a.flags);
      else hipLaunchKernelGGL((count_tail_kernel<Pack2Policy, false>), tgrid, block, 0, a.stream, d, items, n_items, a.plen, a.pats, a.starts, perm, keys, 1, 0, out, a.flags);
    }
  }
  if (tail) hipLaunchKernelGGL(plan_super_kernel, dim3(uint32_t(((nblocks + 63) / 64 + 3) / 4)), block, 0, a.stream, ps);
  hipLaunchKernelGGL((plan_rows_kernel<kRowsOnly, PackPolicy>), grid, block, 0, a.stream, a.npats, static_cast<const int32_t*>(a.noccs), static_cast<const int64_t*>(a.first), static_cast<const int2*>(nullptr),
                     ps, a.out_starts, static_cast<int64_t*>(nullptr), INT64_MAX, a.flags + 1, d, a.total, static_cast<int64_t*>(nullptr), static_cast<const int64_t*>(nullptr));
  return hipGetLastError();
}

// plan_rows_kernel (rows) + locate_walk_kernel
hipError_t traced_walk(const TraceArgs& a, int64_t* offsets, int64_t capacity) {
  const DevIndex d = make_dev(a);
  const int64_t nblocks = (a.npats + 255) / 256;
  const dim3 grid{uint32_t(nblocks)}, block{256};
  hipError_t e = hipMemsetAsync(a.flags + 1, 0, sizeof(int), a.stream);
  if (e != hipSuccess) return e;
  const dim3 wgrid{uint32_t(std::max<int64_t>(1, std::min<int64_t>((capacity + 255) / 256, int64_t(a.num_cus) * 8)))};
  const int32_t* noccs = a.noccs;
  const int64_t *first = a.first, *ostarts = a.out_starts, *total = a.total;
  const PlanSums boffs = plan_sums_at(a.bsums, nblocks, false, a.parity);
  const int* big = a.flags + 1;
  if (d.sa_full) {   // the offsets themselves, no walk
    hipLaunchKernelGGL((plan_rows_kernel<kRowsSa, PackPolicy>), grid, block, 0, a.stream, a.npats, noccs, first, static_cast<const int2*>(nullptr), boffs, a.out_starts, offsets, capacity, a.flags + 1, d, a.total, static_cast<int64_t*>(nullptr), static_cast<const int64_t*>(a.sa_known));
    hipLaunchKernelGGL((plan_big_rows_kernel<kRowsSa, PackPolicy>), wgrid, block, 0, a.stream, a.npats, first, static_cast<const int2*>(nullptr), ostarts, total, capacity, offsets, big, d);
    return hipGetLastError();
  }
  // sampled marks: the walk runs inside the row expansion, as femto_amd_locate_device launches it
  if (a.mode == 3) {
    const bool spot = (d.ru && d.ru_marks && d.pack && d.pack_sa) || (a.row_free && a.tail_items);      // the count twin left its hints in a.sa_known
    hipLaunchKernelGGL((plan_rows_kernel<kRowsWalk, PackPolicy>), grid, block, 0, a.stream, a.npats, noccs, first, static_cast<const int2*>(nullptr), boffs, a.out_starts, offsets, capacity, a.flags + 1, d, a.total, static_cast<int64_t*>(nullptr), spot ? static_cast<const int64_t*>(a.sa_known) : static_cast<const int64_t*>(nullptr));
    hipLaunchKernelGGL((plan_big_rows_kernel<kRowsWalk, PackPolicy>), wgrid, block, 0, a.stream, a.npats, first, static_cast<const int2*>(nullptr), ostarts, total, capacity, offsets, big, d);
  } else {
    hipLaunchKernelGGL((plan_rows_kernel<kRowsWalk, Pack2Policy>), grid, block, 0, a.stream, a.npats, noccs, first, static_cast<const int2*>(nullptr), boffs, a.out_starts, offsets, capacity, a.flags + 1, d, a.total, static_cast<int64_t*>(nullptr), (a.row_free && a.tail_items) ? static_cast<const int64_t*>(a.sa_known) : static_cast<const int64_t*>(nullptr));
    hipLaunchKernelGGL((plan_big_rows_kernel<kRowsWalk, Pack2Policy>), wgrid, block, 0, a.stream, a.npats, first, static_cast<const int2*>(nullptr), ostarts, total, capacity, offsets, big, d);
  }
  return hipGetLastError();
}

hipError_t traced_popcount(const uint32_t* bitmap, int64_t w0, int64_t w1, unsigned long long* out, hipStream_t stream) {
  hipLaunchKernelGGL(trace_popcount_kernel, dim3(1024), dim3(256), 0, stream, bitmap, w0, w1, out);
  return hipGetLastError();
}

}  // namespace femto_amd_trace_api
