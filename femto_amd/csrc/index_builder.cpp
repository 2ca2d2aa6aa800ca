// index_builder.cpp -- femto index writer (host C++; the suffix array comes from the GPU sorter
// in suffix_sort.hip or from the caller).  For the same documents/parameters the block files are
// byte-identical to the reference's index_documents(map=NULL) (src/main/construct.c:572):
//   bucket compressor      compress_bucket          src/main/index.c:309-738
//   Huffman code lengths   BZ2_hbMakeCodeLengths    src/main/huffman.c:63-148 (bzip2's algorithm)
//   canonical codes        BZ2_hbAssignCodes        src/main/huffman.c:152-167
//   wavelet tree           wtree_construct          src/main/wtree.c:907-1078
//   binary sequences       bseq_construct           src/main/wtree.c:359-603 (+ save_run/save_segment :130-357)
//   block / header files   begin/update/finish_*    src/main/index.c:817-1197, constructor_construct_header
//                                                   src/main/construct.c:293-569
#include "index_builder.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>

namespace femto_amd {
namespace {

enum { OK = 0, ERR_MEM = 1, ERR_IO = 2, ERR_PARAM = 3, ERR_FORMAT = 4, ERR_INVALID = 6 };

int fail(Error* e, int code, const std::string& m) {
  if (e) { e->code = code; e->msg = m; }
  return code;
}

int num_bits64(int64_t v) { return v > 0 ? 64 - __builtin_clzll(uint64_t(v)) : 1; }
int ilog2(uint32_t v) { return 31 - __builtin_clz(v); }

void put32(std::vector<uint8_t>& b, uint32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back(uint8_t(v >> s)); }
void put64(std::vector<uint8_t>& b, uint64_t v) { for (int s = 56; s >= 0; s -= 8) b.push_back(uint8_t(v >> s)); }
void set32(std::vector<uint8_t>& b, size_t at, uint32_t v) { for (int k = 0; k < 4; k++) b[at + size_t(k)] = uint8_t(v >> (24 - 8 * k)); }
void set64(std::vector<uint8_t>& b, size_t at, uint64_t v) { for (int k = 0; k < 8; k++) b[at + size_t(k)] = uint8_t(v >> (56 - 8 * k)); }
void align8(std::vector<uint8_t>& b) { while (b.size() & 7) b.push_back(0); }

// MSB-first bit stream (bsW24/bsW64 + bsFinishWrite, src/utils/buffer_funcs.h:100-134)
struct BitWriter {
  std::vector<uint8_t> bytes;
  uint8_t cur = 0;
  int nbits = 0;       // bits in cur
  int64_t total = 0;   // bits written
  void put(uint64_t v, int n) {
    for (int i = n - 1; i >= 0; i--) {
      cur = uint8_t((cur << 1) | ((v >> i) & 1u));
      if (++nbits == 8) { bytes.push_back(cur); cur = 0; nbits = 0; }
    }
    total += n;
  }
  void put1(unsigned bit) {
    cur = uint8_t((cur << 1) | (bit & 1u));
    if (++nbits == 8) { bytes.push_back(cur); cur = 0; nbits = 0; }
    total++;
  }
  void finish() {
    if (nbits) { bytes.push_back(uint8_t(cur << (8 - nbits))); cur = 0; nbits = 0; }
  }
};

// ---------------------------------------------------------------- binary sequence encoder

struct SegWriter {  // segs_writer, src/main/wtree_funcs.h:175-275
  uint64_t w[kSegmentWords];
  int used;
  int appends;
  void reset() { memset(w, 0, sizeof w); used = 0; appends = 0; }
  bool has_room(int64_t need) const { return need <= int64_t(kSegmentWords * 64 - used); }
  void append(uint64_t v, int n) {
    appends++;
    int pos = used;
    used += n;
    while (n > 0) {
      const int word = pos >> 6, off = pos & 63, room = 64 - off;
      const int take = n < room ? n : room;
      const uint64_t chunk = (take == 64) ? v : ((v >> (n - take)) & ((1ull << take) - 1));
      w[word] |= chunk << (room - take);
      pos += take;
      n -= take;
    }
  }
};

struct BseqEncoder {
  SegWriter rle, unc;
  int seg_rle = 0, initial = 0;
  uint32_t occ_rle[2] = {0, 0}, occ_unc[2] = {0, 0}, occs[2] = {0, 0};
  int seg_in_group = 0;
  uint32_t segment_ap = 0;
  unsigned lastbit = 0;
  int last_words_used = 0;
  std::vector<uint32_t> A0, A1, AP;
  std::vector<uint8_t> S;
  std::vector<uint64_t> D;

  void start_segment() {  // start_segment_bseq for both writers (wtree_funcs.h:244-258)
    rle.reset();
    rle.append(1, 1);
    rle.append(lastbit, 1);
    rle.appends = 0;
    unc.reset();
    unc.append(0, 1);
    unc.appends = 0;
  }

  void varbyte(uint32_t v) {  // encode_varbyte, wtree_funcs.h:437-454
    for (;;) {
      uint8_t b = uint8_t(v & 0x7f);
      v >>= 7;
      if (v == 0) { S.push_back(uint8_t(b | 0x80)); break; }
      S.push_back(b);
    }
  }

  void save_segment() {  // wtree.c:125-220
    const SegWriter& a = seg_rle >= 0 ? rle : unc;
    const uint32_t* so = seg_rle >= 0 ? occ_rle : occ_unc;
    last_words_used = (a.used + 63) / 64;
    for (int k = 0; k < kSegmentWords; k++) D.push_back(a.w[k]);
    varbyte(so[0]);
    varbyte(so[1]);
    if (seg_in_group == 0) {  // save_group, wtree.c:84-112: totals BEFORE this segment
      A0.push_back(occs[0]);
      A1.push_back(occs[1]);
      AP.push_back(segment_ap);
    }
    if (++seg_in_group >= kGroupSize) seg_in_group = 0;
    occs[0] += so[0];
    occs[1] += so[1];
    occ_rle[0] = occ_rle[1] = occ_unc[0] = occ_unc[1] = 0;
    seg_rle = initial;
    start_segment();
    segment_ap = uint32_t(S.size());
  }

  void save_run(unsigned bit, uint32_t run) {  // wtree.c:222-357
    int enc_bits = 1 + 2 * ilog2(run);
    for (;;) {
      const bool rle_room = rle.has_room(enc_bits);
      const bool unc_room = unc.has_room(run);
      if (!rle_room && seg_rle == 0) {
        const uint32_t stored = occ_rle[0] + occ_rle[1];
        seg_rle = stored < uint32_t(kSegmentWords * 64 - 1) ? -1 : 1;
      }
      if (!unc_room && seg_rle == 0) seg_rle = 1;
      if ((!rle_room && !unc_room) || (seg_rle == 1 && !rle_room) || (seg_rle == -1 && !unc_room)) {
        if (seg_rle == -1 && !unc_room) {  // top up the literal segment with part of the run
          while (run > 0 && unc.has_room(1)) {
            unc.append(bit, 1);
            occ_unc[bit]++;
            run--;
          }
          if (run != 0) enc_bits = 1 + 2 * ilog2(run);
        }
        save_segment();
        if (run == 0) return;
      } else {
        break;
      }
    }
    if (seg_rle >= 0) {
      if (!rle.has_room(enc_bits)) save_segment();
      rle.append(run, enc_bits);  // Elias gamma: floor(log2 run) zeros then run (encode_gamma, wtree_funcs.h:78-96)
      occ_rle[bit] += run;
    }
    if (seg_rle <= 0) {
      for (uint32_t i = 0; i < run; i++) {
        if (!unc.has_room(1)) save_segment();
        unc.append(bit, 1);
        occ_unc[bit]++;
      }
    }
  }

  void encode(const uint8_t* data, int64_t bitlen, int force_type, std::vector<uint8_t>* out) {
    initial = force_type;
    seg_rle = initial;
    lastbit = (data[0] >> 7) & 1u;
    start_segment();
    uint32_t run = 0;
    for (int64_t i = 0; i < bitlen; i++) {
      const unsigned bit = (data[i >> 3] >> (7 - (i & 7))) & 1u;
      if (bit == lastbit) run++;
      else {
        save_run(lastbit, run);
        run = 1;
        lastbit = bit;
      }
    }
    if (run) save_run(lastbit, run);
    if ((seg_rle >= 0 ? rle.appends : unc.appends) > 0) save_segment();

    // layout (wtree.c:510-591): {0, NUM_GROUPS, TOTAL_SEGMENT_WORDS, D_OFFSET} A0 A1 AP S pad8 D
    const size_t nseg = D.size() / kSegmentWords;
    const uint32_t segment_words = uint32_t(kSegmentWords * (nseg ? nseg - 1 : 0) + size_t(last_words_used));
    const size_t base = out->size();
    put32(*out, 0);
    put32(*out, uint32_t(A0.size()));
    put32(*out, segment_words);
    put32(*out, 0);
    for (uint32_t v : A0) put32(*out, v);
    for (uint32_t v : A1) put32(*out, v);
    for (uint32_t v : AP) put32(*out, v);
    out->insert(out->end(), S.begin(), S.end());
    while ((out->size() - base) & 7) out->push_back(0);
    set32(*out, base + 12, uint32_t(out->size() - base));
    for (uint32_t k = 0; k < segment_words; k++) put64(*out, D[k]);
  }
};

// ---------------------------------------------------------------- Huffman (bzip2's length-limited scheme)

// The algorithm below follows bzip2's BZ2_hbMakeCodeLengths, which the reference vendors (src/main/huffman.c:63-148):
//   bzip2/libbzip2 version 1.0.6 of 6 September 2010, Copyright (C) 1996-2010 Julian Seward <jseward@bzip.org>,
//   released under the terms of the bzip2 licence (a BSD-style licence: redistribution in source and binary forms
//   permitted provided the copyright notice, the conditions and the disclaimer are retained; "THIS SOFTWARE IS PROVIDED
//   BY THE AUTHOR ``AS IS'' AND ANY EXPRESS OR IMPLIED WARRANTIES ... ARE DISCLAIMED").
// Byte-identical index files depend on reproducing its tie-breaking exactly, hence the close correspondence.
// BZ2_hbMakeCodeLengths (src/main/huffman.c:63-148): repeated Huffman construction on
// weights = freq<<8 | depth with a 1-based binary min-heap; on overflow of maxLen the
// frequencies are halved (1 + f/2) and the construction is repeated.  The heap discipline
// (strict '<' comparisons, insertion order) decides ties, so it is reproduced exactly.
void make_code_lengths(uint8_t* len, const int32_t* freq, int alphaSize, int maxLen) {
  const int cap = 2 * alphaSize + 4;
  std::vector<int32_t> heap(size_t(alphaSize) + 2), weight((size_t(cap))), parent((size_t(cap)));
  for (int i = 0; i < alphaSize; i++) weight[size_t(i) + 1] = (freq[i] == 0 ? 1 : freq[i]) << 8;
  auto wsum = [](int32_t a, int32_t b) {
    const int32_t da = a & 0xff, db = b & 0xff;
    return int32_t((uint32_t(a) & 0xffffff00u) + (uint32_t(b) & 0xffffff00u)) | (1 + (da > db ? da : db));
  };
  for (;;) {
    int nNodes = alphaSize, nHeap = 0;
    heap[0] = 0;
    weight[0] = 0;
    parent[0] = -2;
    auto up = [&](int z) {
      int zz = z;
      const int32_t tmp = heap[size_t(zz)];
      while (weight[size_t(tmp)] < weight[size_t(heap[size_t(zz >> 1)])]) {
        heap[size_t(zz)] = heap[size_t(zz >> 1)];
        zz >>= 1;
      }
      heap[size_t(zz)] = tmp;
    };
    auto down = [&](int z) {
      int zz = z;
      const int32_t tmp = heap[size_t(zz)];
      for (;;) {
        int yy = zz << 1;
        if (yy > nHeap) break;
        if (yy < nHeap && weight[size_t(heap[size_t(yy) + 1])] < weight[size_t(heap[size_t(yy)])]) yy++;
        if (weight[size_t(tmp)] < weight[size_t(heap[size_t(yy)])]) break;
        heap[size_t(zz)] = heap[size_t(yy)];
        zz = yy;
      }
      heap[size_t(zz)] = tmp;
    };
    for (int i = 1; i <= alphaSize; i++) {
      parent[size_t(i)] = -1;
      heap[size_t(++nHeap)] = i;
      up(nHeap);
    }
    while (nHeap > 1) {
      const int n1 = heap[1];
      heap[1] = heap[size_t(nHeap--)];
      down(1);
      const int n2 = heap[1];
      heap[1] = heap[size_t(nHeap--)];
      down(1);
      nNodes++;
      parent[size_t(n1)] = parent[size_t(n2)] = nNodes;
      weight[size_t(nNodes)] = wsum(weight[size_t(n1)], weight[size_t(n2)]);
      parent[size_t(nNodes)] = -1;
      heap[size_t(++nHeap)] = nNodes;
      up(nHeap);
    }
    bool tooLong = false;
    for (int i = 1; i <= alphaSize; i++) {
      int j = 0, k = i;
      while (parent[size_t(k)] >= 0) { k = parent[size_t(k)]; j++; }
      len[i - 1] = uint8_t(j);
      if (j > maxLen) tooLong = true;
    }
    if (!tooLong) break;
    for (int i = 1; i <= alphaSize; i++) {
      int32_t j = weight[size_t(i)] >> 8;
      j = 1 + (j / 2);
      weight[size_t(i)] = j << 8;
    }
  }
}

// ---------------------------------------------------------------- bucket compressor

struct BucketResult {
  std::vector<uint8_t> z;
  int32_t occs[kAlphaSize];
};

// wtree_construct, src/main/wtree.c:907-1078
void wtree_encode(const std::vector<uint32_t>& leaf /* alphaSize leaf numbers */, const std::vector<uint16_t>& Lseq,
                  std::vector<uint8_t>* out) {
  std::vector<uint32_t> nodes;
  for (uint32_t lf : leaf)
    for (uint32_t t = lf >> 1; t >= 1; t >>= 1) nodes.push_back(t);
  std::sort(nodes.begin(), nodes.end());
  nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
  const size_t ni = nodes.size();
  // per symbol: the (node index, bit) pairs on its root path
  std::vector<std::vector<std::pair<uint32_t, uint8_t>>> path(leaf.size());
  for (size_t s = 0; s < leaf.size(); s++) {
    uint32_t t = leaf[s];
    while (t > 1) {
      const uint8_t bit = uint8_t(t & 1);
      t >>= 1;
      const size_t k = size_t(std::lower_bound(nodes.begin(), nodes.end(), t) - nodes.begin());
      path[s].push_back({uint32_t(k), bit});
    }
  }
  std::vector<BitWriter> bw(ni);
  for (uint16_t s : Lseq)
    for (const auto& pr : path[s]) bw[pr.first].put1(pr.second);

  const size_t base = out->size();
  put32(*out, uint32_t(ni));
  for (size_t k = 0; k < ni; k++) { put32(*out, nodes[k]); put32(*out, 0); }
  while ((out->size() - base) & 7) out->push_back(0);
  for (size_t k = 0; k < ni; k++) {
    if (bw[k].total == 0) continue;  // offset stays 0: "no data" (wtree.c:1048)
    const int64_t bits = bw[k].total;
    bw[k].finish();
    set32(*out, base + 4 + 8 * k + 4, uint32_t(out->size() - base));
    BseqEncoder enc;
    enc.encode(bw[k].bytes.data(), bits, 0, out);
    while ((out->size() - base) & 7) out->push_back(0);
  }
}

// compress_bucket, src/main/index.c:309-738 (chunk_size <= 0: no chunk directory)
void compress_bucket(const uint16_t* L, const int64_t* offsets, int32_t len, int64_t total_length, BucketResult* r) {
  const int offset_bits = num_bits64(total_length);
  memset(r->occs, 0, sizeof r->occs);
  std::vector<BitWriter> table(kAlphaSize), records(kAlphaSize);
  std::vector<int32_t> mark_count(kAlphaSize, 0);
  for (int32_t i = 0; i < len; i++) {
    const int ch = L[i];
    r->occs[ch]++;
    if (offsets[i] != -1) {
      mark_count[size_t(ch)]++;
      table[size_t(ch)].put1(1);
      records[size_t(ch)].put(uint64_t(offsets[i]), offset_bits);
    } else {
      table[size_t(ch)].put1(0);
    }
  }
  int nInUse = 0;
  uint16_t unseqToSeq[kAlphaSize];
  bool inUse[kAlphaSize];
  int32_t rfreq[kAlphaSize + 1];
  for (int k = 0; k < kAlphaSize; k++) {
    inUse[k] = r->occs[k] > 0;
    if (inUse[k]) {
      unseqToSeq[k] = uint16_t(nInUse);
      rfreq[nInUse] = r->occs[k];
      nInUse++;
    }
  }
  const int alphaSize = nInUse + 1;
  rfreq[nInUse] = 1;  // end-of-bucket symbol
  uint8_t hlen[kAlphaSize + 1];
  make_code_lengths(hlen, rfreq, alphaSize, 20);
  int minLen = 32, maxLen = 0;
  for (int k = 0; k < alphaSize; k++) { if (hlen[k] > maxLen) maxLen = hlen[k]; if (hlen[k] < minLen) minLen = hlen[k]; }
  std::vector<uint32_t> leaf(size_t(alphaSize), 0);
  {  // BZ2_hbAssignCodes + leading 1 (index.c:290-300)
    uint32_t vec = 0;
    for (int n = minLen; n <= maxLen; n++) {
      for (int i = 0; i < alphaSize; i++) if (hlen[i] == n) { leaf[size_t(i)] = vec | (1u << n); vec++; }
      vec <<= 1;
    }
  }
  std::vector<uint16_t> Lseq((size_t(len)));
  for (int32_t i = 0; i < len; i++) Lseq[size_t(i)] = unseqToSeq[L[i]];

  std::vector<uint8_t>& z = r->z;
  z.clear();
  z.reserve(size_t(len) / 2 + 4096);
  for (int i = 0; i < 6; i++) put32(z, 0);
  set32(z, 0, 0xb140bcc7u);   // BUCKET_START
  set32(z, 20, 0);            // number of chunks
  // mapping table + coding table
  set32(z, 4, uint32_t(z.size()));
  {
    BitWriter bw;
    bool inUse16[17];
    for (int i = 0; i < 17; i++) {
      inUse16[i] = false;
      for (int j = 0; j < 16; j++) if (i * 16 + j < kAlphaSize && inUse[i * 16 + j]) inUse16[i] = true;
    }
    for (int i = 0; i < 17; i++) bw.put1(inUse16[i] ? 1 : 0);
    for (int i = 0; i < 17; i++)
      if (inUse16[i])
        for (int j = 0; j < 16; j++) bw.put1((i * 16 + j < kAlphaSize && inUse[i * 16 + j]) ? 1 : 0);
    int curr = hlen[0];
    bw.put(uint64_t(curr), 5);
    for (int i = 0; i < alphaSize; i++) {
      while (curr < hlen[i]) { bw.put(2, 2); curr++; }
      while (curr > hlen[i]) { bw.put(3, 2); curr--; }
      bw.put1(0);
    }
    bw.finish();
    z.insert(z.end(), bw.bytes.begin(), bw.bytes.end());
    align8(z);
  }
  // wavelet tree
  set32(z, 8, uint32_t(z.size()));
  wtree_encode(leaf, Lseq, &z);
  align8(z);
  // mark tables
  set32(z, 12, uint32_t(z.size()));
  {
    const size_t base = z.size();
    z.resize(z.size() + 4 * size_t(nInUse), 0);
    align8(z);
    for (int ch = 0, k = 0; ch < kAlphaSize; ch++) {
      if (!inUse[ch]) continue;
      set32(z, base + 4 * size_t(k), uint32_t(z.size() - base));
      const int64_t bits = table[size_t(ch)].total;
      table[size_t(ch)].finish();
      BseqEncoder enc;
      enc.encode(table[size_t(ch)].bytes.data(), bits, 0, &z);
      align8(z);
      k++;
    }
    align8(z);
  }
  // mark arrays
  set32(z, 16, uint32_t(z.size()));
  {
    const size_t base = z.size();
    z.resize(z.size() + 4 * size_t(nInUse), 0);
    align8(z);
    for (int ch = 0, k = 0; ch < kAlphaSize; ch++) {
      if (!inUse[ch]) continue;
      set32(z, base + 4 * size_t(k), uint32_t(z.size() - base));
      records[size_t(ch)].finish();
      z.insert(z.end(), records[size_t(ch)].bytes.begin(), records[size_t(ch)].bytes.end());
      k++;
    }
  }
  align8(z);
}

void block_header(std::vector<uint8_t>& b, uint32_t magic, int64_t block_number, int64_t nblocks, int64_t total_length,
                  int64_t ndocs, int32_t num_buckets, int32_t size, const BuildParams& p) {
  // write_block_header, src/main/index.c:817-868
  put32(b, magic);
  put32(b, 6);
  put64(b, uint64_t(block_number));
  put64(b, uint64_t(nblocks));
  put64(b, uint64_t(total_length));
  put64(b, uint64_t(ndocs));
  put32(b, uint32_t(num_buckets));
  put32(b, uint32_t(size));
  put32(b, 0);                       // variable_block_size
  put32(b, uint32_t(p.block_size));
  put32(b, uint32_t(p.b_size));
  put32(b, uint32_t(p.mark_period));
  put32(b, 1);                       // mark_type
  put32(b, 0);                       // variable_chunk_size
  put32(b, uint32_t(-1));            // chunk_size: index_documents sets -1 when there is no document map (construct.c:604)
  put32(b, uint32_t(kGroupSize + 0x1000 * kSegmentWords));
  put32(b, uint32_t(kAlphaSize));
  put32(b, 0xe0ffff4du);
}

int write_file(const std::string& path, const std::vector<uint8_t>& data, Error* e) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return fail(e, ERR_IO, "Could not create " + path);
  if (!data.empty() && fwrite(data.data(), 1, data.size(), f) != data.size()) { fclose(f); return fail(e, ERR_IO, "short write " + path); }
  if (fclose(f)) return fail(e, ERR_IO, "close failed " + path);
  return OK;
}

}  // namespace

int parse_build_params(const char* s, BuildParams* p, Error* e) {
  // parse_param_set / parse_param, src/main/index.c:144-219: name=value separated by spaces or commas
  if (s) {
    std::string str(s);
    size_t i = 0;
    while (i < str.size()) {
      while (i < str.size() && (str[i] == ' ' || str[i] == ',')) i++;
      size_t j = i;
      while (j < str.size() && (isalnum((unsigned char)str[j]) || str[j] == '_' || str[j] == '-')) j++;
      std::string tok = str.substr(i, j - i), val;
      if (j < str.size() && str[j] == '=') {
        size_t k = ++j;
        while (k < str.size() && (isalnum((unsigned char)str[k]) || str[k] == '_' || str[k] == '-')) k++;
        val = str.substr(j, k - j);
        j = k;
      }
      if (tok.empty() && j == i) { if (i < str.size()) return fail(e, ERR_INVALID, "Invalid parameter string"); break; }
      i = j;
      int v = int(strtol(val.c_str(), nullptr, 0));
      if (tok == "block_size") p->block_size = v;
      else if (tok == "bucket_size") p->b_size = v;
      else if (tok == "chunk_size") p->chunk_size = v;
      else if (tok == "mark_period") p->mark_period = v;
      else return fail(e, ERR_INVALID, "Invalid parameter " + tok);
    }
  }
  // calculate_params, src/main/index.c:793-815
  if (p->b_size <= 0 || p->block_size <= 0) return fail(e, ERR_PARAM, "block_size and bucket_size must be positive");
  if (p->block_size % p->b_size != 0) return fail(e, ERR_PARAM, "block_size must be a multiple of bucket_size");
  if (p->chunk_size > 0 && p->b_size % p->chunk_size != 0) return fail(e, ERR_PARAM, "bucket_size must be a multiple of chunk_size");
  return OK;
}

void prepare_text(const std::vector<Document>& docs, std::vector<uint16_t>* text, std::vector<int64_t>* doc_ends) {
  int64_t total = 0;
  for (const Document& d : docs) total += d.len + 1;
  text->resize(size_t(total));
  doc_ends->clear();
  size_t at = 0;
  for (const Document& d : docs) {
    for (int64_t i = 0; i < d.len; i++) (*text)[at++] = uint16_t(d.bytes[i]) + 5;
    (*text)[at++] = kSEOF;
    doc_ends->push_back(int64_t(at));
  }
}

void bseq_encode(const uint8_t* bits, int64_t bitlen, int force_type, std::vector<uint8_t>* out) {
  BseqEncoder enc;
  enc.encode(bits, bitlen, force_type, out);
}

// flatten_index, src/main/index.c:2260-2365: {u32 0xb1497dea, u32 6, i64 nblocks+1}, (nblocks+2) i64
// start/end offsets, then the header block and every data block, each padded to the page size.
int flatten_index_dir(const std::string& index_dir, const std::string& out_path, Error* e) {
  auto slurp = [&](const std::string& path, std::vector<uint8_t>* out) -> int {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return fail(e, ERR_IO, "Could not open " + path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize(size_t(n));
    if (n && fread(out->data(), 1, size_t(n), f) != size_t(n)) { fclose(f); return fail(e, ERR_IO, "short read " + path); }
    fclose(f);
    return OK;
  };
  std::vector<uint8_t> hdr;
  int rc = slurp(index_dir + "/00", &hdr);
  if (rc) return rc;
  if (hdr.size() < 88 || hdr[0] != 0xb1 || hdr[1] != 0x17 || hdr[2] != 0x7d || hdr[3] != 0xea) return fail(e, ERR_FORMAT, "Invalid block start");
  uint64_t nblocks = 0;
  for (int k = 0; k < 8; k++) nblocks = (nblocks << 8) | hdr[16 + size_t(k)];
  const size_t page = 4096;  // get_page_size() on the platforms femto runs on (src/utils/page_utils.c)
  std::vector<uint8_t> out;
  put32(out, 0xb1497deau);
  put32(out, 6);
  put64(out, nblocks + 1);
  for (uint64_t b = 0; b < nblocks + 2; b++) put64(out, 0);
  while (out.size() % page) out.push_back(0);
  for (uint64_t b = 0; b < nblocks + 1; b++) {
    std::vector<uint8_t> blk;
    if (b == 0) blk = hdr;
    else {
      char name[64];
      snprintf(name, sizeof name, "/%02llx", (unsigned long long)b);
      if ((rc = slurp(index_dir + name, &blk))) return rc;
    }
    const uint64_t start = out.size();
    out.insert(out.end(), blk.begin(), blk.end());
    while (out.size() % page) out.push_back(0);
    set64(out, 16 + 8 * size_t(b), start);
    set64(out, 16 + 8 * (size_t(b) + 1), out.size());
  }
  return write_file(out_path, out, e);
}

int build_index_from_sa(const std::string& out_dir, const std::vector<Document>& docs, const BuildParams& params,
                        const int64_t* sa, int nthreads, Error* e) {
  if (docs.empty()) return fail(e, ERR_PARAM, "an index needs at least one document");
  if (params.mark_period <= 0) return fail(e, ERR_PARAM, "Final EOF character of each document must be marked (mark_period > 0)");
  std::vector<uint16_t> text;
  std::vector<int64_t> doc_ends;
  prepare_text(docs, &text, &doc_ends);
  const int64_t n = int64_t(text.size());
  const int64_t ndocs = int64_t(docs.size());
  const int64_t nblocks = (n + params.block_size - 1) / params.block_size;
  const int32_t buckets_per_block = params.block_size / params.b_size;
  if (nthreads < 1) nthreads = 1;
  mkdir(out_dir.c_str(), 0777);

  auto doc_of = [&](int64_t off) {  // bwt_document_info_reader_find_doc
    return int64_t(std::upper_bound(doc_ends.begin(), doc_ends.end(), off) - doc_ends.begin());
  };
  auto marked = [&](int64_t off) {  // should_mark, src/main/index_types.h:134-144
    const int64_t d = doc_of(off);
    const int64_t start = d ? doc_ends[size_t(d) - 1] : 0;
    const int64_t doff = off - start, dlen = doc_ends[size_t(d)] - start;
    return doff == 0 || doff == dlen - 1 || doff % params.mark_period == 0;
  };

  std::vector<int64_t> occs_total(kAlphaSize, 0);                       // running occurrences before the current block
  std::vector<int64_t> block_occs(size_t(kAlphaSize) * size_t(nblocks), 0);

  for (int64_t b = 0; b < nblocks; b++) {
    const int64_t row0 = b * int64_t(params.block_size);
    const int32_t rows = int32_t(std::min<int64_t>(params.block_size, n - row0));
    const int32_t nbk = (rows + params.b_size - 1) / params.b_size;
    for (int ch = 0; ch < kAlphaSize; ch++) block_occs[size_t(ch) * size_t(nblocks) + size_t(b)] = occs_total[size_t(ch)];

    std::vector<BucketResult> res((size_t(nbk)));
    std::atomic<int32_t> next(0);
    auto worker = [&]() {
      std::vector<uint16_t> L;
      std::vector<int64_t> off;
      for (;;) {
        const int32_t k = next.fetch_add(1);
        if (k >= nbk) break;
        const int64_t r0 = row0 + int64_t(k) * params.b_size;
        const int32_t len = int32_t(std::min<int64_t>(params.b_size, row0 + rows - r0));
        L.resize(size_t(len));
        off.resize(size_t(len));
        for (int32_t i = 0; i < len; i++) {
          const int64_t s = sa[r0 + i];
          L[size_t(i)] = s == 0 ? uint16_t(kSEOF) : text[size_t(s) - 1];  // get_L_char_from_offsets, bwt_qsufsort.c:62-83
          off[size_t(i)] = marked(s) ? s : -1;
        }
        compress_bucket(L.data(), off.data(), len, n, &res[size_t(k)]);
      }
    };
    {
      std::vector<std::thread> th;
      const int nt = std::min<int>(nthreads, nbk);
      for (int t = 1; t < nt; t++) th.emplace_back(worker);
      worker();
      for (auto& t : th) t.join();
    }

    // begin_data_block / update_data_block / finish_data_block (index.c:1037-1197)
    std::vector<uint8_t> blk;
    block_header(blk, 0xb1501deau, b, nblocks, n, ndocs, nbk, rows, params);
    const size_t dir_at = blk.size();
    for (int i = 0; i < buckets_per_block + 1; i++) put32(blk, 0);
    const size_t occ_at = blk.size();
    blk.resize(blk.size() + 4 * size_t(kAlphaSize) * size_t(nbk), 0);
    align8(blk);
    std::vector<int64_t> since_block(kAlphaSize, 0);
    for (int32_t k = 0; k < nbk; k++) {
      for (int ch = 0; ch < kAlphaSize; ch++) set32(blk, occ_at + 4 * (size_t(ch) * size_t(nbk) + size_t(k)), uint32_t(since_block[size_t(ch)]));
      align8(blk);
      set32(blk, dir_at + 4 * size_t(k), uint32_t(blk.size()));
      blk.insert(blk.end(), res[size_t(k)].z.begin(), res[size_t(k)].z.end());
      align8(blk);
      set32(blk, dir_at + 4 * (size_t(k) + 1), uint32_t(blk.size()));
      for (int ch = 0; ch < kAlphaSize; ch++) since_block[size_t(ch)] += res[size_t(k)].occs[ch];
      res[size_t(k)].z.clear();
      res[size_t(k)].z.shrink_to_fit();
    }
    for (int ch = 0; ch < kAlphaSize; ch++) occs_total[size_t(ch)] += since_block[size_t(ch)];
    char name[64];
    snprintf(name, sizeof name, "/%02llx", (unsigned long long)(b + 1));
    int rc = write_file(out_dir + name, blk, e);
    if (rc) return rc;
  }

  // header block (begin_header_block index.c:958-1035; constructor_construct_header construct.c:293-569)
  std::vector<uint8_t> hdr;
  BuildParams hp = params;
  block_header(hdr, 0xb1177deau, -1, nblocks, n, ndocs, 0, 0, hp);
  int64_t sum = 0;
  for (int ch = 0; ch < kAlphaSize; ch++) { put64(hdr, uint64_t(sum)); sum += occs_total[size_t(ch)]; }
  for (int ch = 0; ch < kAlphaSize; ch++)
    for (int64_t b = 0; b < nblocks; b++) put64(hdr, uint64_t(block_occs[size_t(ch) * size_t(nblocks) + size_t(b)]));
  for (int64_t d = 0; d < ndocs; d++) put64(hdr, uint64_t(doc_ends[size_t(d)]));
  const size_t eof_at = hdr.size();
  for (int64_t d = 0; d < ndocs; d++) put64(hdr, 0);
  const size_t info_at = hdr.size();
  for (int64_t d = 0; d < ndocs + 1; d++) put64(hdr, 0);
  // rows 0..ndocs-1 are the suffixes starting with SEOF; each is marked with its own offset
  for (int64_t row = 0; row < ndocs; row++) set64(hdr, eof_at + 8 * size_t(doc_of(sa[row])), uint64_t(row));
  for (int64_t d = 0; d < ndocs; d++) {
    set64(hdr, info_at + 8 * size_t(d), uint64_t(hdr.size()));
    hdr.insert(hdr.end(), docs[size_t(d)].info.begin(), docs[size_t(d)].info.end());
    set64(hdr, info_at + 8 * (size_t(d) + 1), uint64_t(hdr.size()));
  }
  int rc = write_file(out_dir + "/00", hdr, e);
  if (rc) return rc;
  {  // marker file (construct.c:531-543)
    const std::string tag = "This is a FEMTO index constructed by femto_amd (MI355X-native builder)\n";
    std::vector<uint8_t> t(tag.begin(), tag.end());
    rc = write_file(out_dir + "/_femto_index", t, e);
    if (rc) return rc;
  }
  return OK;
}

}  // namespace femto_amd
