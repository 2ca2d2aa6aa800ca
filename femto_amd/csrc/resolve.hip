// resolve.hip -- text offset -> (document, offset in document) on the device: resolve_location
// (src/main/index.c:1587-1611) for a whole batch of located offsets, the step the reference runs right after every locate
// (do_range_to_results_query schedules one header_loc_query_t per located row, src/main/server.c:4800-4823).
//
// resolve_location finds, by binary search over the header block's doc_ends[] (exclusive end offsets, ascending), the last
// document end <= offset (bsearch_int64_ntoh_arr, src/utils/util.c:346): document = that index + 1, offset in document =
// offset - that end (document 0: the offset itself).  Here: one lane per offset.  doc_ends is uploaded once per handle; a
// workgroup keeps up to 2048 evenly spaced entries of it in LDS (16 KB) -- for an index of at most 2048 documents that IS
// the table and a lane never touches memory beyond its own offset and its two results; beyond that the LDS search picks a
// window of `stride` entries and the search finishes on the (L2-resident) table itself.  HBM-bound: 8 bytes in, 16 out per
// offset (4 + 8 with the int32 document form).
#include "api_internal.hpp"

namespace femto_amd {

constexpr int kResolveTop = 2048;

struct ResolveArgs {
  const int64_t* offsets;
  int64_t n;
  const int64_t* d_n;          // when not NULL: only min(n, *d_n) offsets are live (the row total of an enqueue-only locate)
  const int64_t* doc_ends;
  int64_t ndocs, stride;
  int32_t ntop;
  int64_t* doc_out;            // any of the three may be NULL
  int32_t* doc32_out;
  int64_t* off_out;
};

__global__ __launch_bounds__(256) void resolve_kernel(const ResolveArgs A) {
  __shared__ int64_t s_top[kResolveTop];
  for (int k = threadIdx.x; k < A.ntop; k += 256) s_top[k] = A.doc_ends[int64_t(k) * A.stride];
  __syncthreads();
  int64_t n = A.n;
  if (A.d_n) {
    const int64_t live = *A.d_n;
    n = live < n ? (live < 0 ? 0 : live) : n;
  }
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const int64_t t = A.offsets[i];
    // entries of s_top that are <= t: [0, lo)
    int lo = 0, hi = A.ntop;
    while (lo < hi) {
      const int m = (lo + hi) >> 1;
      if (s_top[m] <= t) lo = m + 1; else hi = m;
    }
    int64_t cnt;           // doc_ends entries <= t  ==  the document number
    int64_t prev_end = 0;
    if (lo == 0) {
      cnt = 0;
    } else if (A.stride == 1) {
      cnt = lo;
      prev_end = s_top[lo - 1];
    } else {
      // doc_ends[(lo - 1) * stride] <= t; the first entry > t lies in ((lo - 1) * stride, min(lo * stride, ndocs)]
      int64_t a = int64_t(lo - 1) * A.stride + 1, b = int64_t(lo) * A.stride;
      b = b < A.ndocs ? b : A.ndocs;
      prev_end = s_top[lo - 1];
      while (a < b) {
        const int64_t m = (a + b) >> 1;
        const int64_t v = A.doc_ends[m];
        if (v <= t) { a = m + 1; prev_end = v; } else b = m;
      }
      cnt = a;
    }
    if (A.doc_out) A.doc_out[i] = cnt;
    if (A.doc32_out) A.doc32_out[i] = int32_t(cnt);
    if (A.off_out) A.off_out[i] = t - prev_end;
  }
}

// the handle's device copy of doc_ends (made on first use)
static int ensure_doc_ends(femto_amd_index* ix) {
  // (the fast path takes no lock: femto_amd_resolve_device is an enqueue-only call issued every step; the pointer is published
  // with release order once the table is complete)
  if (__atomic_load_n(&ix->d_doc_ends, __ATOMIC_ACQUIRE)) return 0;
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->d_doc_ends) return 0;
  const size_t n = ix->host.doc_ends.size();
  int64_t* p = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * 8));
  if (n) {
    const hipError_t e = hipMemcpy(p, ix->host.doc_ends.data(), n * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(p);
      return set_err(FEMTO_AMD_ERR_INVALID, std::string("hipMemcpy(doc_ends): ") + hipGetErrorString(e));
    }
  }
  __atomic_store_n(&ix->d_doc_ends, p, __ATOMIC_RELEASE);
  ix->table_bytes += int64_t(n * 8);
  ix->hbm_held += int64_t(n * 8);
  return 0;
}

static int launch_resolve(femto_amd_index* ix, const int64_t* d_offsets, int64_t n, const int64_t* d_n, int64_t* d_doc, int32_t* d_doc32,
                          int64_t* d_doc_offset, hipStream_t st) {
  if (d_doc32 && ix->host.doc_ends.size() >= (size_t(1) << 31))
    return set_err(FEMTO_AMD_ERR_PARAM, "32-bit document numbers need an index of fewer than 2^31 documents: pass d_doc");
  int rc = ensure_doc_ends(ix);
  if (rc) return rc;
  ResolveArgs A{};
  A.offsets = d_offsets;
  A.n = n;
  A.d_n = d_n;
  A.doc_ends = ix->d_doc_ends;
  A.ndocs = int64_t(ix->host.doc_ends.size());
  A.stride = (A.ndocs + kResolveTop - 1) / kResolveTop;
  if (A.stride < 1) A.stride = 1;
  A.ntop = int32_t((A.ndocs + A.stride - 1) / A.stride);
  A.doc_out = d_doc;
  A.doc32_out = d_doc32;
  A.off_out = d_doc_offset;
  // every workgroup fills its LDS copy once and then strides over the batch: enough workgroups to fill the chip (8 per CU),
  // no more than the batch has tiles of four
  int64_t blocks = (n + 1023) / 1024;
  const int64_t cap = int64_t(ix->num_cus) * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = timer_begin(ix, ix->t_resolve, st, &e0, &e1);
  hipLaunchKernelGGL(resolve_kernel, dim3(uint32_t(blocks)), dim3(256), 0, st, A);
  if (timed) timer_end(ix, ix->t_resolve, st, e0, e1);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace femto_amd

int femto_amd_resolve_device(femto_amd_index_t* ix, const int64_t* d_offsets, int64_t n, const int64_t* d_n, int64_t* d_doc,
                             int32_t* d_doc32, int64_t* d_doc_offset, void* stream) {
  API_BEGIN
  if (!ix || n < 0 || (n && !d_offsets)) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (!ix->children.empty()) return set_err(FEMTO_AMD_ERR_INVALID, "device-pointer calls are not available on a multi-device handle");
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (n == 0) return FEMTO_AMD_OK;
  return launch_resolve(ix, d_offsets, n, d_n, d_doc, d_doc32, d_doc_offset, static_cast<hipStream_t>(stream));
  API_END
}

int femto_amd_resolve_batch(femto_amd_index_t* ix, int64_t n, const int64_t* offsets, int64_t* doc, int64_t* doc_offset) {
  API_BEGIN
  if (!ix || n < 0 || (n && (!offsets || (!doc && !doc_offset)))) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (!ix->children.empty()) return femto_amd_resolve_batch(ix->children[0], n, offsets, doc, doc_offset);
  int rc = ensure_device(ix);
  if (rc) return rc;
  if (n == 0) return FEMTO_AMD_OK;
  Lease L(ix);
  if (!L.s) return L.rc;
  Scratch& S = *L.s;
  hipStream_t st = S.stream;
  if ((rc = S.rows.reserve(size_t(n) * 8)) || (rc = S.off.reserve(size_t(n) * 8)) || (rc = S.offsets.reserve(size_t(n) * 8))) return rc;
  HIP_TRY(hipMemcpyAsync(S.rows.p, offsets, size_t(n) * 8, hipMemcpyHostToDevice, st));
  if ((rc = launch_resolve(ix, S.rows.as<int64_t>(), n, nullptr, doc ? S.off.as<int64_t>() : nullptr, nullptr,
                           doc_offset ? S.offsets.as<int64_t>() : nullptr, st)))
    return rc;
  if (doc) HIP_TRY(hipMemcpyAsync(doc, S.off.p, size_t(n) * 8, hipMemcpyDeviceToHost, st));
  if (doc_offset) HIP_TRY(hipMemcpyAsync(doc_offset, S.offsets.p, size_t(n) * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return FEMTO_AMD_OK;
  API_END
}
