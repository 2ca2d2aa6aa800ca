// host_pack.cpp -- the staging threads' inner loops of the host-pointer batches (api_host.hip), plain C++ (no device pass).
//
// A host-pointer batch (parallel_count / parallel_locate, src/main/femto.c:275,331) is bounded by the HOST: every pattern's
// symbols are turned into one 64-bit key (dense code of every symbol, `bits` bits each, last symbol in the top field;
// count_keys_kernel, direct_kernels.hip.hpp) before they cross PCIe, and the ranges are widened into the caller's arrays on
// the way back.  The scalar packing loop is a chain of (load symbol, load its code, shift, or) per symbol: 11 ns per 20-mer,
// 0.11 CPU-seconds per 10 M patterns -- under a container's CPU quota that, not PCIe and not the GPU, is the call.  With
// AVX-512 VBMI + BMI2 a pattern of up to 32 symbols is ONE masked load, a 256-entry byte table look-up in registers
// (two vpermi2b), and a pext per eight symbols.  Chosen at run time (__builtin_cpu_supports); same keys either way
// (tests/test_host_logic.py compares the two paths through femto_amd_host_pack_keys).
#include "host_pack.hpp"

#include <immintrin.h>

namespace femto_amd {

namespace {

// one pattern, the plain way; returns 1 when a symbol has no code (the chunk then travels as symbols)
inline uint32_t pack_one_scalar(const uint8_t* dense16, int bits, int64_t l, const uint16_t* pat, uint64_t* out) {
  uint32_t bad = 0;
  uint64_t key = 0;
  for (int64_t s = l - 1; s >= 0; s--) {   // last symbol first: it lands in the top field
    const uint32_t c = dense16[pat[s]];
    bad |= uint32_t(c == 0);
    key = (key << bits) | c;
  }
  *out = l ? key << (64 - int(l) * bits) : 0;   // field j (from the top) = j-th symbol from the end; 0 = end
  return bad;
}

inline const uint16_t* pattern_at(const PackSource& src, int64_t i) {
  return src.ptrs ? src.ptrs[i] : (src.starts[i] >= 0 ? src.flat + src.starts[i] : nullptr);
}

uint32_t pack_span_scalar(const PackSource& src, const uint8_t* dense16, int bits, int nsym, int64_t i0, int64_t i1, uint64_t* out) {
  uint32_t bad = 0;
  for (int64_t i = i0; i < i1; i++) {
    const int64_t l = src.plen[i];
    const uint16_t* pat = pattern_at(src, i);
    if (l < 0 || l > nsym || (l && !pat)) return 1;
    bad |= pack_one_scalar(dense16, bits, l, pat, out + (i - i0));
  }
  return bad;
}

// kHigh = false: no symbol >= 128 has a code (ASCII texts): one table look-up instead of two, anything above takes the scalar way
template <bool kHigh>
__attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi,bmi2")))
uint32_t pack_span_avx512(const PackSource& src, const uint8_t* dense16, int bits, int nsym, int64_t i0, int64_t i1, uint64_t* out) {
  // the codes of the byte-sized symbols 0..255 live in four registers (symbols 256..260 -- bytes 251..255 -- take the scalar way)
  const __m512i t0 = _mm512_loadu_si512(dense16), t1 = _mm512_loadu_si512(dense16 + 64);
  const __m512i t2 = _mm512_loadu_si512(dense16 + 128), t3 = _mm512_loadu_si512(dense16 + 192);
  const __m512i v255 = _mm512_set1_epi16(kHigh ? 255 : 127);
  const uint64_t fm = 0x0101010101010101ull * ((1ull << bits) - 1ull);   // a field per byte
  uint32_t bad = 0;
  for (int64_t i = i0; i < i1; i++) {
    const int64_t l = src.plen[i];
    const uint16_t* pat = pattern_at(src, i);
    if (l < 0 || l > nsym || (l && !pat)) return 1;
    if (l == 0) { out[i - i0] = 0; continue; }
    if (l > 32) { bad |= pack_one_scalar(dense16, bits, l, pat, out + (i - i0)); continue; }
    const __mmask32 m = l == 32 ? 0xffffffffu : ((1u << int(l)) - 1u);
    const __m512i v = _mm512_maskz_loadu_epi16(m, pat);                 // (lanes beyond the pattern are neither read nor faulted on)
    if (_mm512_cmpgt_epu16_mask(v, v255)) { bad |= pack_one_scalar(dense16, bits, l, pat, out + (i - i0)); continue; }
    const __m512i idx = _mm512_castsi256_si512(_mm512_cvtepi16_epi8(v));  // 32 byte-sized symbols (upper half unused)
    __m512i c512 = _mm512_permutex2var_epi8(t0, idx, t1);                // index bits 0..6: symbols 0..127
    if (kHigh) c512 = _mm512_mask_blend_epi8(_mm512_movepi8_mask(idx), c512, _mm512_permutex2var_epi8(t2, idx, t3));   // symbols 128..255
    const __m256i c = _mm256_maskz_mov_epi8(m, _mm512_castsi512_si256(c512));
    bad |= uint32_t(_mm256_mask_cmpeq_epi8_mask(m, c, _mm256_setzero_si256()) != 0);
    // field of symbol s sits at bit (64 - bits * l) + bits * s: the little-endian compaction of the code bytes, shifted up
    uint64_t k = _pext_u64(uint64_t(_mm256_extract_epi64(c, 0)), fm);
    if (l > 8) k |= _pext_u64(uint64_t(_mm256_extract_epi64(c, 1)), fm) << (8 * bits);     // (l > 8 => 9 * bits <= 63)
    if (l > 16) k |= _pext_u64(uint64_t(_mm256_extract_epi64(c, 2)), fm) << (16 * bits);
    if (l > 24) k |= _pext_u64(uint64_t(_mm256_extract_epi64(c, 3)), fm) << (24 * bits);
    out[i - i0] = k << (64 - int(l) * bits);
  }
  return bad;
}

void widen_span_scalar(const int32_t* pr, int64_t n, int64_t* first, int64_t* last) {
  for (int64_t i = 0; i < n; i++) {
    const int64_t f = pr[2 * i], l = pr[2 * i + 1];
    if (last) { first[i] = f; last[i] = l; }
    else first[i] = l - f + 1;        // femto.c:313-318
  }
}

__attribute__((target("avx512f")))
void widen_span_avx512(const int32_t* pr, int64_t n, int64_t* first, int64_t* last) {
  int64_t i = 0;
  const __m512i one = _mm512_set1_epi64(1);
  for (; i + 8 <= n; i += 8) {      // eight (first,last) pairs = eight 64-bit lanes: first in the low half, last in the high half
    const __m512i p = _mm512_loadu_si512(pr + 2 * i);
    const __m512i f = _mm512_srai_epi64(_mm512_slli_epi64(p, 32), 32), l = _mm512_srai_epi64(p, 32);
    if (last) {
      _mm512_storeu_si512(first + i, f);
      _mm512_storeu_si512(last + i, l);
    } else {
      _mm512_storeu_si512(first + i, _mm512_add_epi64(_mm512_sub_epi64(l, f), one));
    }
  }
  widen_span_scalar(pr + 2 * i, n - i, first + i, last ? last + i : nullptr);
}

int simd_level() {
  static const int level = [] {
    __builtin_cpu_init();
    return (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
            __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("bmi2"))
               ? 1
               : 0;
  }();
  return level;
}

}  // namespace

bool host_pack_simd() { return simd_level() == 1; }

uint32_t pack_keys_span(const PackSource& src, const uint8_t* dense16, int bits, int nsym, int64_t i0, int64_t i1, uint64_t* out, int force_scalar) {
  if (!force_scalar && simd_level() == 1 && bits >= 1 && bits <= 8) {
    bool high = false;
    for (int c = 128; c < 256; c++) high = high || dense16[c] != 0;
    return high ? pack_span_avx512<true>(src, dense16, bits, nsym, i0, i1, out) : pack_span_avx512<false>(src, dense16, bits, nsym, i0, i1, out);
  }
  return pack_span_scalar(src, dense16, bits, nsym, i0, i1, out);
}

void widen_pairs_span(const int32_t* pairs, int64_t n, int64_t* first, int64_t* last) {
  if (simd_level() == 1) widen_span_avx512(pairs, n, first, last);
  else widen_span_scalar(pairs, n, first, last);
}

}  // namespace femto_amd

// diagnostic entry point (include/femto_amd.h): the packing loop on its own, no handle and no device -- the CPU tests compare
// the AVX-512 path with the scalar one through it
extern "C" int femto_amd_host_pack_keys(const uint8_t* dense, int ndense, int bits, int64_t npats, const int32_t* plen, const uint16_t* flat,
                                        const int64_t* starts, int force_scalar, uint64_t* keys_out, int* simd_used) {
  if (!dense || ndense < 0 || bits < 1 || bits > 8 || npats < 0 || (npats && (!plen || !starts || !keys_out))) return -1;
  static thread_local uint8_t dense16[65536];
  for (int c = 0; c < 65536; c++) dense16[c] = c < ndense ? dense[c] : 0;
  if (simd_used) *simd_used = (!force_scalar && femto_amd::host_pack_simd()) ? 1 : 0;
  const femto_amd::PackSource src{plen, nullptr, flat, starts};
  return femto_amd::pack_keys_span(src, dense16, bits, 63 / bits, 0, npats, keys_out, force_scalar) ? 0 : 1;
}
