// host_pack.hpp -- the staging threads' inner loops (host_pack.cpp): patterns -> 64-bit keys, 32-bit range pairs -> the caller's arrays
#pragma once
#include <stdint.h>

namespace femto_amd {

// where a batch's patterns are: one array of pointers (parallel_count's alpha_t**) or a flat symbol array + starts
struct PackSource {
  const int32_t* plen;
  const uint16_t* const* ptrs;   // or NULL
  const uint16_t* flat;
  const int64_t* starts;
};

bool host_pack_simd();           // AVX-512 VBMI + BMI2 found on this host

// keys of patterns [i0, i1) -> out[0 .. i1 - i0).  dense16[sym] = the symbol's field (0: not a character of the text), 65536
// entries; nsym = 63 / bits.  Returns non-zero when some pattern cannot travel as a key (longer than nsym, negative length or
// start, a symbol without a code): the caller sends the chunk as symbols, and that path reports malformed input.
uint32_t pack_keys_span(const PackSource& src, const uint8_t* dense16, int bits, int nsym, int64_t i0, int64_t i1, uint64_t* out,
                        int force_scalar = 0);

// n (first,last) int32 pairs -> first[0..n), last[0..n); last == NULL: first[i] = last - first + 1 (femto.c:313-318)
void widen_pairs_span(const int32_t* pairs, int64_t n, int64_t* first, int64_t* last);

}  // namespace femto_amd
