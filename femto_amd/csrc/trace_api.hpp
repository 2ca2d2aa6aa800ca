// trace_api.hpp -- interface between the API (femto_amd_api.hip) and the traced kernel twins (trace_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace femto_amd_trace_api {

struct TraceArgs {
  const void* dev;          // the handle's DevIndex, as bytes (the traced twins use an identical struct in their own namespace)
  int mode, num_cus;
  int64_t npats;
  const int32_t* plen;
  const uint16_t* pats;
  const int64_t* starts;
  int max_occs;
  int64_t *first, *last;
  int32_t* noccs;
  int64_t* sa_known;        // npats: count_direct_kernel's sa_out on a dense handle (text positions the search already knows)
  int64_t *out_starts, *bsums;
  int parity;               // which set of group sums of bsums this launch uses (PlanSums)
  void* tail_items;         // NULL: no text tail
  int tail_min;
  int row_free;             // 1: the row-free form of the chain (DevIndex::row_free)
  int* flags;               // [0] error, [1] long ranges, [2] tail item count
  int64_t* total;
  uint32_t* bitmap;
  unsigned long long* reads; // [kTraceRegions] read counters (zeroed by the caller)
  const int64_t* trace_off; // [kTraceRegions]
  hipStream_t stream;
};

size_t traced_dev_index_bytes();
hipError_t traced_count_plan(const TraceArgs& a);
hipError_t traced_walk(const TraceArgs& a, int64_t* offsets, int64_t capacity);
hipError_t traced_popcount(const uint32_t* bitmap, int64_t w0, int64_t w1, unsigned long long* out, hipStream_t stream);

}  // namespace femto_amd_trace_api
