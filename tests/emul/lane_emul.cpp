// tests/emul/lane_emul.cpp -- TEST-ONLY host emulation of the lane kernels' table walk
// (bseq_rank_lane / wt_rank_lane of femto_amd/csrc/kernels.hip.hpp) over the tables the loader
// builds (HostIndex).  It lets the CPU suite check the DERIVED TABLES (segment lines, cum, hint,
// RLE skip tables, child links) against the reference's golden vectors without a GPU, with
// bounds-checked accesses.  It is not part of the product library and is never shipped as a fallback.
//
// usage: lane_emul <index path> <out.bin>   -> per row: u16 L, i64 Occ(L[row],row)-global, i64 mark offset
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../femto_amd/csrc/host_index.hpp"

using namespace femto_amd;

static const HostIndex* H;

template <class T>
static const T& at(const std::vector<T>& v, uint64_t i, const char* what) {
  if (i >= v.size()) { fprintf(stderr, "OUT OF BOUNDS %s[%llu] size %zu\n", what, (unsigned long long)i, v.size()); abort(); }
  return v[i];
}

struct RR { uint32_t o0, o1, bit; };

static uint64_t sel8(const uint64_t* w, int i) { return (i >= 0 && i < 8) ? w[i] : 0; }

static RR rank_lane(const LaneBseq bs, uint32_t index1) {
  const uint32_t t = index1 - 1;
  uint32_t seg, o0 = 0, o1 = 0;
  const bool uniform = bs.hint_base == kNoHint;
  if (uniform) seg = t / 511u;
  else {
    const BlockDir d = at(H->bdir, uint64_t(bs.hint_base) + (t >> 9), "bdir");
    seg = d.seg; o0 = d.o0; o1 = d.o1;
    if (t >= d.n0 + d.n1) { o0 = d.n0; o1 = d.n1; seg++; }
    // the merged directory must agree with hint[] / cum[]
    if (seg != at(H->hint, uint64_t(bs.hint_base) + (t >> 9), "hint") && seg != at(H->hint, uint64_t(bs.hint_base) + (t >> 9), "hint") + 1) abort();
  }
  const uint64_t slot = bs.seg_base + 2ull * seg;
  uint64_t w[8];
  for (int k = 0; k < 8; k++) w[k] = at(H->segs, slot * 8 + k, "segs");
  if (uniform) { const uint64_t c = at(H->segs, slot * 8 + 8, "segs.cum"); o0 = uint32_t(c); o1 = uint32_t(c >> 32); }
  RR r;
  if (w[0] >> 63) {
    uint32_t bit = uint32_t(w[0] >> 62) & 1u;
    int p = 2;
    if (!uniform) {
      const uint32_t rel = t - o0 - o1;
      uint64_t e[8];
      for (int k = 0; k < 8; k++) e[k] = at(H->segs, (slot + 1) * 8 + k, "aux");
      int kk = 0; uint64_t best = 0;
      for (int j = 1; j < 8; j++) if (uint32_t(e[j]) <= rel) { best = e[j]; kk = j; }
      if (kk) {
        const uint32_t total = uint32_t(best), hi = uint32_t(best >> 32), ones = hi & 0x7fffffffu;
        o0 += total - ones; o1 += ones; bit = hi >> 31; p = int((e[0] >> (9 * (kk - 1))) & 0x1ff);
      }
    }
    uint64_t win = 0; int avail = 0;
    for (int it = 0; it < 512; it++) {
      int k = win ? __builtin_clzll(win) : 64;
      if (2 * k + 1 > avail) {
        const int wi = p >> 6, sh = p & 63;
        const uint64_t a = sel8(w, wi), c = sel8(w, wi + 1);
        win = (a << sh) | (sh ? (c >> (64 - sh)) : 0);
        avail = 64;
        k = win ? __builtin_clzll(win) : 64;
        if (k >= 32) break;
      }
      const int nb = 2 * k + 1;
      const uint32_t v = uint32_t(win >> (64 - nb));
      win = nb < 64 ? (win << nb) : 0;
      avail -= nb; p += nb;
      const uint32_t tot = o0 + o1;
      if (tot + v <= t) { if (bit) o1 += v; else o0 += v; bit ^= 1u; }
      else { const uint32_t rem = t + 1 - tot; if (bit) o1 += rem; else o0 += rem; break; }
    }
    r.bit = bit;
  } else {
    const uint32_t nb = 1 + t - o0 - o1;
    if (nb > 511) { fprintf(stderr, "literal position %u out of segment (t=%u o0=%u o1=%u seg=%u uniform=%d)\n", nb, t, o0, o1, seg, int(uniform)); abort(); }
    uint32_t ones = 0;
    for (int k = 0; k < 8; k++) {
      const uint32_t lo = 64u * k;
      uint64_t m = 0;
      if (nb >= lo) m = (nb - lo >= 63) ? ~0ull : (~0ull << (63 - (nb - lo)));
      ones += uint32_t(__builtin_popcountll(w[k] & m));
    }
    o1 += ones; o0 += nb - ones;
    r.bit = uint32_t(w[nb >> 6] >> (63 - (nb & 63))) & 1u;
  }
  r.o0 = o0; r.o1 = o1;
  return r;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  HostIndex h;
  Error e{0, ""};
  if (int rc = h.load(argv[1], &e)) { fprintf(stderr, "load failed %d %s\n", rc, e.msg.c_str()); return 1; }
  H = &h;
  FILE* out = fopen(argv[2], "wb");
  for (int64_t row = 0; row < h.total_length; row++) {
    const int64_t gb = row / h.b_size;
    uint32_t idx = uint32_t(row - gb * h.b_size) + 1;
    const DevBucket bk = at(h.buckets, gb, "buckets");
    int cur = 0, seq = -1;
    for (int depth = 0; depth < 32; depth++) {
      const LaneNode nd = at(h.lnodes, bk.node_base + uint32_t(cur), "lnodes");
      const RR r = rank_lane(nd.bs, idx);
      idx -= r.bit ? r.o0 : r.o1;
      const int c = r.bit ? nd.child[1] : nd.child[0];
      if (c < 0) { seq = -1 - c; break; }
      cur = c;
    }
    if (seq < 0 || uint32_t(seq) >= bk.n_in_use) { fprintf(stderr, "bad seq at row %lld\n", (long long)row); return 1; }
    const LaneSeq sq = at(h.lseqs, bk.seq_base + uint32_t(seq), "lseqs");
    const RR m = rank_lane(sq.mark_table, idx);
    int64_t off = -1;
    if (m.bit) {
      const uint64_t bitpos = sq.mark_array * 8 + (uint64_t(m.o1) - 1) * uint64_t(h.text_size_bits);
      uint64_t v = 0;
      for (int i = 0; i < h.text_size_bits; i++) {
        const uint64_t bp = bitpos + uint64_t(i);
        v = (v << 1) | ((at(h.image, bp >> 3, "image") >> (7 - (bp & 7))) & 1u);
      }
      off = int64_t(v);
    }
    const uint16_t ch = uint16_t(sq.ch);
    const int64_t occ = at(h.occ, uint64_t(gb) * kAlphaSize + ch, "occ").base + int64_t(idx);
    fwrite(&ch, 2, 1, out);
    fwrite(&occ, 8, 1, out);
    fwrite(&off, 8, 1, out);
  }
  fclose(out);
  return 0;
}
