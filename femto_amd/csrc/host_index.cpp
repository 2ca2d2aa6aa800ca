// host_index.cpp -- femto index file parser (host side of the loader).
//
// Format facts (all integers big-endian; reference: src/main/block_format.txt and the code cited):
//   88-byte block header                         src/main/index.c:817-868, :1348-1404
//   header block: C[261], block_occs[261][nb], doc_ends, doc_eof_rows, doc infos   index.c:870-898
//   data block: bucket directory (buckets_per_block+1 u32), bucket_occs[261][num_buckets] u32,
//               pad 8, buckets                   index.c:900-910, :1085-1107
//   bucket: 6 x u32 {magic, map, wtree, mark tables, mark arrays, nchunks}      index.c:44-52
//   map: 17 bits inUse16, 16 bits per set group; Huffman lengths (5-bit start, 10=+1, 11=-1, 0=next)
//        for nInUse+1 symbols (last = end-of-bucket)                             index.c:571-611, :1264-1314
//   canonical codes by (length, symbol)          src/main/huffman.c:152-167; leaf = 1<<len | code  index.c:290-300
//   wavelet tree: u32 num_internal, sorted {u32 node,u32 offset}                 src/main/wtree.c:918-1059
//   bseq: {0, NUM_GROUPS, TOTAL_SEGMENT_WORDS, D_OFFSET}                          src/main/wtree_funcs.h:294-358
//   mark tables / mark arrays: nInUse u32 offsets then bodies                     index.c:645-720
#include "host_index.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace femto_amd {
namespace {

constexpr uint32_t kHeaderBlockStart = 0xb1177deau;  // src/main/index.h:182-186
constexpr uint32_t kDataBlockStart = 0xb1501deau;
constexpr uint32_t kBlockVersion = 6;
constexpr uint32_t kEndOfHeader = 0xe0ffff4du;
constexpr uint32_t kBucketStart = 0xb140bcc7u;
constexpr uint32_t kFlattenedStart = 0xb1497deau;    // src/main/block_storage.h:119
constexpr uint32_t kWtreeSettings = kGroupSize + 0x1000 * kSegmentWords;  // src/main/wtree.c:50-53
constexpr size_t kBlockHeaderSize = 88;

enum { OK = 0, ERR_MEM = 1, ERR_IO = 2, ERR_PARAM = 3, ERR_FORMAT = 4, ERR_BZ_DATA = 5, ERR_INVALID = 6 };

inline uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
inline uint64_t be64(const uint8_t* p) { return (uint64_t(be32(p)) << 32) | be32(p + 4); }

int fail(Error* e, int code, const std::string& m) {
  if (e) { e->code = code; e->msg = m; }
  return code;
}

int read_file(const std::string& path, std::vector<uint8_t>* out, Error* e) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fail(e, ERR_IO, "Could not open " + path);
  if (fseek(f, 0, SEEK_END)) { fclose(f); return fail(e, ERR_IO, "Could not seek " + path); }
  long n = ftell(f);
  if (n < 0 || fseek(f, 0, SEEK_SET)) { fclose(f); return fail(e, ERR_IO, "Could not size " + path); }
  out->resize(size_t(n));
  if (n && fread(out->data(), 1, size_t(n), f) != size_t(n)) { fclose(f); return fail(e, ERR_IO, "short read " + path); }
  fclose(f);
  return OK;
}

struct BlockHeader {
  int64_t block_number, number_of_blocks, total_length, number_of_documents;
  int32_t num_buckets, size, block_size, b_size, mark_period, mark_type, chunk_size;
};

// read_block_header, src/main/index.c:1348-1404 (same checks, same order)
int parse_block_header(const uint8_t* d, size_t len, uint32_t magic, BlockHeader* h, Error* e) {
  if (len < kBlockHeaderSize) return fail(e, ERR_FORMAT, "block too short");
  if (be32(d) != magic) return fail(e, ERR_FORMAT, "Invalid block start");
  if (be32(d + 4) != kBlockVersion) return fail(e, ERR_FORMAT, "Wrong block version");
  h->block_number = int64_t(be64(d + 8));
  h->number_of_blocks = int64_t(be64(d + 16));
  h->total_length = int64_t(be64(d + 24));
  h->number_of_documents = int64_t(be64(d + 32));
  h->num_buckets = int32_t(be32(d + 40));
  h->size = int32_t(be32(d + 44));
  h->block_size = int32_t(be32(d + 52));
  h->b_size = int32_t(be32(d + 56));
  h->mark_period = int32_t(be32(d + 60));
  h->mark_type = int32_t(be32(d + 64));
  h->chunk_size = int32_t(be32(d + 72));
  // calculate_params, src/main/index.c:793-815
  if (h->b_size <= 0 || h->block_size <= 0) return fail(e, ERR_PARAM, "bad block/bucket size");
  if (h->block_size % h->b_size != 0) return fail(e, ERR_PARAM, "block_size not a multiple of bucket_size");
  if (be32(d + 76) != kWtreeSettings) return fail(e, ERR_FORMAT, "Wrong wavelet tree settings");
  if (be32(d + 80) != uint32_t(kAlphaSize)) return fail(e, ERR_FORMAT, "Wrong alphabet size");
  if (be32(d + 84) != kEndOfHeader) return fail(e, ERR_FORMAT, "Bad end of header");
  return OK;
}

int num_bits64(int64_t v) { return v > 0 ? 64 - __builtin_clzll(uint64_t(v)) : 1; }  // src/utils/bit_funcs.h:191

struct BitReader {  // MSB-first, like bsR24 (src/utils/buffer_funcs.h:136)
  const uint8_t* p; size_t len; size_t bit = 0; bool overrun = false;
  unsigned get(int n) {
    unsigned v = 0;
    for (int i = 0; i < n; i++, bit++) {
      if ((bit >> 3) >= len) { overrun = true; return 0; }
      v = (v << 1) | ((p[bit >> 3] >> (7 - (bit & 7))) & 1u);
    }
    return v;
  }
};

// Re-expresses one sequence's A0/A1/AP + varbyte S tables (src/main/wtree_funcs.h:294-358) as the lane
// kernels' tables (device_tables.h): 128-byte segment lines, per-segment cumulative (zeros, ones), the
// per-block segment hint and the RLE skip tables.
struct LaneTables {
  std::vector<uint64_t>* segs;
  std::vector<CumEntry>* cum;
  std::vector<uint32_t>* hint;
  std::vector<BlockDir>* bdir;
};

int build_lane_tables(const std::vector<uint8_t>& img, DevBseq* bs, bool* regular, Error* e, LaneTables* lt, LaneBseq* lane) {
  const size_t cum0 = lt ? lt->cum->size() : 0;
  const size_t hint0 = lt ? lt->hint->size() : 0;
  bool uniform = true;
  if (lt) {
    if (lt->cum->size() > 0xfff00000ull || lt->hint->size() > 0xfff00000ull) return fail(e, ERR_MEM, "lane tables too large");
    lane->seg_base = lt->segs->size() / kSegmentWords;
    lane->cum_base = uint32_t(cum0);
  }
  const uint8_t* z = img.data() + bs->off;
  const uint32_t G = bs->num_groups;
  const uint8_t* A0 = z + 16;
  const uint8_t* A1 = A0 + 4ull * G;
  const uint8_t* AP = A1 + 4ull * G;
  const uint8_t* S = AP + 4ull * G;
  const uint8_t* Send = z + bs->d_off;
  const uint64_t nseg = (uint64_t(bs->total_words) + kSegmentWords - 1) / kSegmentWords;
  uint64_t pos = 0, next_block = 0;
  uint64_t seg = 0;
  uint32_t tot0 = 0, tot1 = 0;
  for (uint32_t g = 0; g < G && seg < nseg; g++) {
    uint32_t o0 = be32(A0 + 4ull * g), o1 = be32(A1 + 4ull * g);
    if (uint64_t(o0) + o1 != pos) return fail(e, ERR_FORMAT, "bseq group table disagrees with segment sums");
    const uint8_t* sp = S + be32(AP + 4ull * g);
    for (int j = 0; j < kGroupSize && seg < nseg; j++, seg++) {
      uint32_t v[2];
      for (int k = 0; k < 2; k++) {
        uint32_t val = 0;
        int i = 0;
        for (;;) {
          if (sp >= Send || i > 4) return fail(e, ERR_FORMAT, "bseq segment sums run past the S section");
          const uint8_t w = *sp++;
          val |= uint32_t(w & 0x7f) << (7 * i++);
          if (w & 0x80) break;
        }
        v[k] = val;
      }
      const uint64_t len = uint64_t(v[0]) + v[1];
      if (len < 511 && seg + 1 < nseg) *regular = false;
      if (seg + 1 < nseg ? len != 511 : len > 511) uniform = false;
      if (lt) lt->cum->push_back(CumEntry{o0, o1});
      while (next_block * 512 < pos + len) {
        if (lt) lt->hint->push_back(uint32_t(seg));
        next_block++;
      }
      pos += len;
      o0 += v[0];
      o1 += v[1];
      tot0 = o0;
      tot1 = o1;
    }
  }
  if (seg != nseg) return fail(e, ERR_FORMAT, "bseq has fewer segment sums than segments");
  if (lt) {
    // terminal entry: totals after the last segment (the two-candidate test reads cum[seg+1])
    const uint32_t t0 = tot0, t1 = tot1;
    lt->cum->push_back(CumEntry{t0, t1});
    if (uniform) {
      lt->hint->resize(hint0);
      lane->hint_base = kNoHint;
    } else {
      lane->hint_base = uint32_t(hint0);
      // merged block directory, parallel to hint[]: both candidate segments' cumulative counts in one entry
      lt->bdir->resize(hint0);
      for (size_t hpos = hint0; hpos < lt->hint->size(); hpos++) {
        const uint32_t sg = (*lt->hint)[hpos];
        const CumEntry& ca = (*lt->cum)[cum0 + sg];
        const CumEntry& cb = (*lt->cum)[cum0 + sg + 1];
        lt->bdir->push_back(BlockDir{sg, ca.o0, ca.o1, cb.o0, cb.o1, {0, 0, 0}});
      }
    }
    // copy the D words into 64-byte aligned native-endian slots (bseq_segment's zero fill included);
    // non-uniform sequences get a second slot per segment holding the RLE skip table
    // every sequence uses two 64-byte slots (one 128-byte memory line) per segment: the line the
    // memory system fetches anyway (all L2 read requests are 128 B on gfx950, TCC_EA0_RDREQ_128B).
    // Uniform sequences keep the segment's cumulative (zeros, ones) in word 8 of the line, so a rank
    // costs exactly ONE line; sequences with RLE segments keep the skip table there.
    const int stride = 2;
    const uint8_t* Dsrc = img.data() + bs->off + bs->d_off;
    const size_t at = lt->segs->size();
    lt->segs->resize(at + size_t(nseg) * kSegmentWords * size_t(stride), 0);
    for (uint64_t sg = 0; sg < nseg; sg++) {
      uint64_t w[kSegmentWords];
      for (int k = 0; k < kSegmentWords; k++) {
        const uint64_t wi = sg * kSegmentWords + uint64_t(k);
        w[k] = wi < bs->total_words ? be64(Dsrc + 8ull * wi) : 0;
      }
      uint64_t* dst = lt->segs->data() + at + size_t(sg) * kSegmentWords * size_t(stride);
      memcpy(dst, w, sizeof w);
      if (uniform) {
        const CumEntry& ce = (*lt->cum)[cum0 + size_t(sg)];
        dst[kSegmentWords] = uint64_t(ce.o0) | (uint64_t(ce.o1) << 32);
      }
      if (!uniform && (w[0] >> 63)) {
        // decode the whole RLE segment once (wtree.c:690-712) and record the state at each 64-bit boundary
        uint64_t* aux = dst + kSegmentWords;
        for (int k = 1; k < kSegmentWords; k++) aux[k] = 0xffffffffull;
        auto window = [&](int p) -> uint64_t {
          const int wi = p >> 6, sh = p & 63;
          uint64_t v = wi < kSegmentWords ? w[wi] << sh : 0;
          if (sh && wi + 1 < kSegmentWords) v |= w[wi + 1] >> (64 - sh);
          return v;
        };
        uint32_t bit = uint32_t(w[0] >> 62) & 1u, total = 0, ones = 0;
        int p = 2, next_k = 1;
        uint64_t pk = 0;
        while (p < kSegmentWords * 64) {
          const uint64_t win = window(p);
          if (win == 0) break;
          while (next_k < kSegmentWords && p >= 64 * next_k) {
            aux[next_k] = uint64_t(total) | (uint64_t(ones | (bit << 31)) << 32);
            pk |= uint64_t(p) << (9 * (next_k - 1));
            next_k++;
          }
          const int kz = __builtin_clzll(win);
          if (kz >= 32) break;
          const int nb = 2 * kz + 1;
          const uint32_t v = uint32_t(win >> (64 - nb));
          total += v;
          if (bit) ones += v;
          bit ^= 1u;
          p += nb;
        }
        aux[0] = pk;
      }
    }
    if (uniform) lt->cum->resize(cum0);
  }
  return OK;
}

int parse_bseq(const std::vector<uint8_t>& img, uint64_t abs, uint64_t limit, DevBseq* out, Error* e) {
  if (abs + 16 > limit) return fail(e, ERR_FORMAT, "bseq header out of range");
  const uint8_t* z = img.data() + abs;
  out->off = abs;
  out->num_groups = be32(z + 4);
  out->total_words = be32(z + 8);
  out->d_off = be32(z + 12);
  out->pad = 0;
  if (be32(z) != 0) return fail(e, ERR_FORMAT, "bseq header word 0 not zero");
  uint64_t need_dir = 16 + 12ull * out->num_groups;
  if (out->d_off < need_dir || abs + out->d_off + 8ull * out->total_words > limit)
    return fail(e, ERR_FORMAT, "bseq sections out of range");
  if (out->d_off & 7) return fail(e, ERR_FORMAT, "bseq D section not 8-byte aligned");
  return OK;
}

}  // namespace

int HostIndex::resolve_location(int64_t offset, int64_t* doc, int64_t* doc_offset) const {
  // resolve_location, src/main/index.c:1587-1611
  int64_t prev = -1;
  int64_t a = 0, b = int64_t(doc_ends.size()) - 1;
  if (doc_ends.empty() || offset < doc_ends[0]) prev = -1;
  else if (doc_ends[size_t(b)] <= offset) prev = b;
  else {
    while (b - a > 1) { int64_t m = (a + b) / 2; if (offset < doc_ends[size_t(m)]) b = m; else a = m; }
    prev = a;
  }
  if (prev == -1) { *doc = 0; *doc_offset = offset; }
  else { *doc = prev + 1; *doc_offset = offset - doc_ends[size_t(prev)]; }
  return OK;
}

int HostIndex::document_info(int64_t doc, const uint8_t** info, int64_t* len) const {
  if (doc < 0 || doc >= number_of_documents) return ERR_PARAM;
  const uint64_t start = be64(header.data() + doc_info_off + 8 * size_t(doc));
  const uint64_t end = be64(header.data() + doc_info_off + 8 * size_t(doc) + 8);
  if (start > end || end > header.size()) return ERR_FORMAT;
  *info = header.data() + start;
  *len = int64_t(end - start);
  return OK;
}

int HostIndex::load(const std::string& path, Error* e) {
  struct stat st;
  if (stat(path.c_str(), &st)) return fail(e, ERR_IO, "Could not stat " + path);
  const bool flat = S_ISREG(st.st_mode);
  if (!flat && !S_ISDIR(st.st_mode)) return fail(e, ERR_IO, "index not file or directory: " + path);

  std::vector<uint8_t> flat_file;
  std::vector<uint64_t> flat_off;
  int rc;
  if (flat) {  // flattened container, src/main/index.c:2293-2357, block_storage.c:157-190
    rc = read_file(path, &flat_file, e);
    if (rc) return rc;
    if (flat_file.size() < 16 || be32(flat_file.data()) != kFlattenedStart || be32(flat_file.data() + 4) != kBlockVersion)
      return fail(e, ERR_FORMAT, "not a flattened femto index: " + path);
    int64_t nb = int64_t(be64(flat_file.data() + 8));
    // bound the count by what the file can hold BEFORE multiplying (a damaged count must not wrap the size check)
    if (nb <= 0 || uint64_t(nb) > (flat_file.size() - 16) / 8 || uint64_t(nb) + 1 > (flat_file.size() - 16) / 8)
      return fail(e, ERR_FORMAT, "bad flattened block count");
    for (int64_t i = 0; i <= nb; i++) flat_off.push_back(be64(flat_file.data() + 16 + 8 * size_t(i)));
    for (int64_t i = 0; i < nb; i++)
      if (flat_off[size_t(i)] > flat_off[size_t(i) + 1] || flat_off[size_t(i) + 1] > flat_file.size())
        return fail(e, ERR_FORMAT, "bad flattened offsets");
    header.assign(flat_file.begin() + long(flat_off[0]), flat_file.begin() + long(flat_off[1]));
  } else {
    rc = read_file(path + "/00", &header, e);  // "%s/%02x", src/main/block_storage.c:257-263
    if (rc) return rc;
  }

  BlockHeader hh;
  rc = parse_block_header(header.data(), header.size(), kHeaderBlockStart, &hh, e);
  if (rc) return rc;
  total_length = hh.total_length;
  number_of_blocks = hh.number_of_blocks;
  number_of_documents = hh.number_of_documents;
  block_size = hh.block_size;
  b_size = hh.b_size;
  mark_period = hh.mark_period;
  chunk_size = hh.chunk_size;
  text_size_bits = num_bits64(total_length);  // open_data_block, src/main/index.c:1441
  buckets_per_block = block_size / b_size;
  if (number_of_blocks < 0 || total_length < 0 || number_of_documents < 0) return fail(e, ERR_FORMAT, "negative sizes in header");
  if (block_size <= 0 || b_size <= 0 || b_size > block_size) return fail(e, ERR_FORMAT, "bad block / bucket size in header");
  // every table below is sized by these counts: bound them by what the header block can hold (by division, so that a
  // damaged count cannot overflow the offset arithmetic)
  if (uint64_t(number_of_blocks) > header.size() / (8 * uint64_t(kAlphaSize)) || uint64_t(number_of_documents) > header.size() / 24)
    return fail(e, ERR_FORMAT, "header block too short for its block / document counts");
  if (flat && int64_t(flat_off.size()) - 2 != number_of_blocks) return fail(e, ERR_FORMAT, "flattened block count mismatch");
  {
    int64_t expect = (total_length + block_size - 1) / block_size;
    if (expect != number_of_blocks) return fail(e, ERR_FORMAT, "number_of_blocks does not match total_length");
  }
  const size_t c_off = kBlockHeaderSize;
  const size_t bo_off = c_off + 8 * size_t(kAlphaSize);
  const size_t de_off = bo_off + 8 * size_t(kAlphaSize) * size_t(number_of_blocks);
  doc_info_off = de_off + 16 * size_t(number_of_documents);  // doc_ends, doc_eof_rows, then ndocs+1 info offsets
  if (doc_info_off + 8 * (size_t(number_of_documents) + 1) > header.size()) return fail(e, ERR_FORMAT, "header block too short");
  C.resize(kAlphaSize + 1);
  for (int ch = 0; ch < kAlphaSize; ch++) C[size_t(ch)] = int64_t(be64(header.data() + c_off + 8 * size_t(ch)));
  C[kAlphaSize] = total_length;  // get_C, src/main/index.c:1545
  doc_ends.resize(size_t(number_of_documents));
  for (int64_t d = 0; d < number_of_documents; d++) doc_ends[size_t(d)] = int64_t(be64(header.data() + de_off + 8 * size_t(d)));
  // The kernels trust these tables for every row number they form (LF steps, range ends): reject a header whose
  // cumulative counts are not monotone or leave [0, total_length] -- a damaged file must fail here, not fault a GPU.
  for (int ch = 0; ch < kAlphaSize; ch++)
    if (C[size_t(ch)] < 0 || C[size_t(ch)] > C[size_t(ch) + 1]) return fail(e, ERR_FORMAT, "header C table is not monotone");
  for (int64_t d = 0; d < number_of_documents; d++)
    if (doc_ends[size_t(d)] < 0 || doc_ends[size_t(d)] > total_length || (d && doc_ends[size_t(d)] < doc_ends[size_t(d) - 1]))
      return fail(e, ERR_FORMAT, "header document ends are not monotone");

  total_buckets = (total_length + b_size - 1) / b_size;
  occ_base.assign(size_t(total_buckets) * kAlphaSize, 0);
  leaf_code.assign(size_t(total_buckets) * kAlphaSize, 0);
  buckets.resize(size_t(total_buckets));
  block_off.resize(size_t(number_of_blocks));
  block_len.resize(size_t(number_of_blocks));

  // pass 1: read the block files into one image
  image.clear();
  for (int64_t b = 0; b < number_of_blocks; b++) {
    std::vector<uint8_t> blk;
    if (flat) {
      blk.assign(flat_file.begin() + long(flat_off[size_t(b) + 1]), flat_file.begin() + long(flat_off[size_t(b) + 2]));
    } else {
      char name[64];
      snprintf(name, sizeof name, "/%02llx", (unsigned long long)(b + 1));
      rc = read_file(path + name, &blk, e);
      if (rc) return rc;
    }
    size_t at = (image.size() + 255) & ~size_t(255);
    image.resize(at + blk.size());
    memcpy(image.data() + at, blk.data(), blk.size());
    block_off[size_t(b)] = at;
    block_len[size_t(b)] = blk.size();
  }
  image.resize(((image.size() + 255) & ~size_t(255)) + 512);  // tail pad: kernels over-read S sections by < 256 B

  // pass 2: parse every block and bucket
  int64_t gb = 0;
  for (int64_t b = 0; b < number_of_blocks; b++) {
    const uint64_t boff = block_off[size_t(b)];
    const uint64_t blimit = boff + block_len[size_t(b)];
    const uint8_t* d = image.data() + boff;
    block_slot_start.push_back(segs.size() / kSegmentWords);
    block_lnode_start.push_back(lnodes.size());
    block_lseq_start.push_back(lseqs.size());
    BlockHeader bh;
    rc = parse_block_header(d, block_len[size_t(b)], kDataBlockStart, &bh, e);
    if (rc) return rc;
    if (bh.block_number != b || bh.number_of_blocks != number_of_blocks || bh.total_length != total_length ||
        bh.block_size != block_size || bh.b_size != b_size)
      return fail(e, ERR_FORMAT, "data block header disagrees with header block");
    int64_t expect_rows = std::min<int64_t>(block_size, total_length - b * int64_t(block_size));
    if (bh.size != expect_rows) return fail(e, ERR_FORMAT, "data block row count mismatch");
    int64_t nbk = (expect_rows + b_size - 1) / b_size;
    if (bh.num_buckets != nbk) return fail(e, ERR_FORMAT, "data block bucket count mismatch");
    const size_t dir_off = kBlockHeaderSize;
    const size_t bocc_off = dir_off + 4 * (size_t(buckets_per_block) + 1);  // data_block_bucket_occs_offset, index.c:893
    if (bocc_off + 4 * size_t(kAlphaSize) * size_t(nbk) > block_len[size_t(b)]) return fail(e, ERR_FORMAT, "data block too short");

    for (int64_t k = 0; k < nbk; k++, gb++) {
      // Occ bases
      for (int ch = 0; ch < kAlphaSize; ch++) {
        int64_t block_occs = int64_t(be64(header.data() + bo_off + 8 * (size_t(ch) * size_t(number_of_blocks) + size_t(b))));
        uint32_t bucket_occs = be32(d + bocc_off + 4 * (size_t(ch) * size_t(nbk) + size_t(k)));
        occ_base[size_t(gb) * kAlphaSize + size_t(ch)] = C[size_t(ch)] + block_occs + int64_t(bucket_occs);
      }
      // bucket header (b_fault, src/main/index.c:1222-1262)
      uint32_t bko = be32(d + dir_off + 4 * size_t(k));
      if (bko & 7 || uint64_t(bko) + 24 > block_len[size_t(b)]) return fail(e, ERR_FORMAT, "bad bucket offset");
      const uint8_t* bk = d + bko;
      if (be32(bk) != kBucketStart) return fail(e, ERR_FORMAT, "bad bucket magic");
      uint32_t map_off = bko + be32(bk + 4), wt_off = bko + be32(bk + 8);
      uint32_t mt_off = bko + be32(bk + 12), ma_off = bko + be32(bk + 16);
      if ((wt_off & 7) || (mt_off & 7) || (ma_off & 7)) return fail(e, ERR_FORMAT, "misaligned bucket sections");
      if (uint64_t(map_off) >= block_len[size_t(b)] || uint64_t(wt_off) + 4 > block_len[size_t(b)] ||
          uint64_t(mt_off) > block_len[size_t(b)] || uint64_t(ma_off) > block_len[size_t(b)])
        return fail(e, ERR_FORMAT, "bucket sections out of range");

      // mapping table + Huffman code lengths (index.c:1264-1314)
      BitReader br{d + map_off, size_t(block_len[size_t(b)] - map_off)};
      bool inUse16[17];
      for (int i = 0; i < 17; i++) inUse16[i] = br.get(1) == 1;
      bool inUse[kAlphaSize] = {false};
      for (int i = 0; i < 17; i++)
        if (inUse16[i])
          for (int j = 0; j < 16; j++) {
            unsigned uc = br.get(1);
            if (uc == 1 && i * 16 + j < kAlphaSize) inUse[i * 16 + j] = true;
          }
      int nInUse = 0;
      uint16_t seqToUnseq[kAlphaSize + 1];
      for (int i = 0; i < kAlphaSize; i++) if (inUse[i]) seqToUnseq[nInUse++] = uint16_t(i);
      const int alphaSize = nInUse + 1;
      uint8_t len[kAlphaSize + 1];
      int curr = int(br.get(5));
      for (int i = 0; i < alphaSize; i++) {
        for (;;) {
          if (curr < 1 || curr > 20) return fail(e, ERR_BZ_DATA, "bad Huffman code length");
          if (br.get(1) == 0) break;
          if (br.get(1) == 0) curr++; else curr--;
          if (br.overrun) return fail(e, ERR_FORMAT, "coding table runs past block end");
        }
        len[i] = uint8_t(curr);
      }
      if (br.overrun) return fail(e, ERR_FORMAT, "mapping table runs past block end");
      int minLen = 32, maxLen = 0;
      for (int i = 0; i < alphaSize; i++) { if (len[i] > maxLen) maxLen = len[i]; if (len[i] < minLen) minLen = len[i]; }
      // BZ2_hbAssignCodes (src/main/huffman.c:152-167) + leading 1 (index.c:290-300)
      uint32_t leaf[kAlphaSize + 1];
      {
        uint32_t vec = 0;
        for (int n = minLen; n <= maxLen; n++) {
          for (int i = 0; i < alphaSize; i++) if (len[i] == n) { leaf[i] = vec | (1u << n); vec++; }
          vec <<= 1;
        }
      }
      for (int s = 0; s < nInUse; s++) leaf_code[size_t(gb) * kAlphaSize + seqToUnseq[s]] = leaf[s];

      // wavelet tree directory -> DevNodes with explicit children
      const uint8_t* wt = d + wt_off;
      uint32_t n_internal = be32(wt);
      if (uint64_t(wt_off) + 4 + 8ull * n_internal > block_len[size_t(b)]) return fail(e, ERR_FORMAT, "wavelet directory out of range");
      DevBucket& B = buckets[size_t(gb)];
      B.node_base = uint32_t(nodes.size());
      B.seq_base = uint32_t(seqs.size());
      B.n_internal = n_internal;
      B.n_in_use = uint32_t(nInUse);
      std::vector<uint32_t> node_num(n_internal);
      for (uint32_t i = 0; i < n_internal; i++) {
        node_num[i] = be32(wt + 4 + 8 * size_t(i));
        if (i && node_num[i] <= node_num[i - 1]) return fail(e, ERR_FORMAT, "wavelet directory not sorted");
      }
      if (n_internal && node_num[0] != 1) return fail(e, ERR_FORMAT, "wavelet tree has no root");
      auto find_internal = [&](uint32_t num) -> int32_t {
        size_t lo = 0, hi = node_num.size();
        while (lo < hi) { size_t m = (lo + hi) / 2; if (node_num[m] < num) lo = m + 1; else hi = m; }
        return (lo < node_num.size() && node_num[lo] == num) ? int32_t(lo) : -1;
      };
      for (uint32_t i = 0; i < n_internal; i++) {
        DevNode nd;
        memset(&nd, 0, sizeof nd);
        LaneNode ln;
        memset(&ln, 0, sizeof ln);
        uint32_t off = be32(wt + 4 + 8 * size_t(i) + 4);
        if (off == 0) {  // "0 indicates no data" (src/main/wtree.c:1048): never ranked
          nd.bs.off = boff + wt_off; nd.bs.num_groups = 0; nd.bs.d_off = 0; nd.bs.total_words = 0;
        } else {
          rc = parse_bseq(image, boff + wt_off + off, blimit, &nd.bs, e);
          if (rc) return rc;
          LaneTables lt{&segs, &cum, &hint, &bdir};
          rc = build_lane_tables(image, &nd.bs, &dir_regular, e, &lt, &ln.bs);
          if (rc) return rc;
        }
        for (int bit = 0; bit < 2; bit++) {
          uint32_t childnum = node_num[i] * 2 + uint32_t(bit);
          int32_t ci = find_internal(childnum);
          if (ci >= 0) nd.child[bit] = ci;
          else {
            int seq = -1;
            for (int s = 0; s < alphaSize; s++) if (leaf[s] == childnum) { seq = s; break; }
            if (seq < 0) return fail(e, ERR_BZ_DATA, "wavelet tree does not match Huffman code");
            nd.child[bit] = -1 - seq;
          }
        }
        ln.child[0] = nd.child[0];
        ln.child[1] = nd.child[1];
        nodes.push_back(nd);
        lnodes.push_back(ln);
      }

      // mark tables and arrays (index.c:645-720): nInUse u32 offsets each
      if (uint64_t(mt_off) + 4ull * uint32_t(nInUse) > block_len[size_t(b)] ||
          uint64_t(ma_off) + 4ull * uint32_t(nInUse) > block_len[size_t(b)])
        return fail(e, ERR_FORMAT, "mark directories out of range");
      for (int s = 0; s < nInUse; s++) {
        DevSeq sq;
        memset(&sq, 0, sizeof sq);
        uint32_t toff = be32(d + mt_off + 4 * size_t(s));
        if (toff & 7) return fail(e, ERR_FORMAT, "misaligned mark table");
        rc = parse_bseq(image, boff + mt_off + toff, blimit, &sq.mark_table, e);
        if (rc) return rc;
        LaneSeq lsq;
        memset(&lsq, 0, sizeof lsq);
        LaneTables lt{&segs, &cum, &hint, &bdir};
        rc = build_lane_tables(image, &sq.mark_table, &dir_regular, e, &lt, &lsq.mark_table);
        if (rc) return rc;
        uint32_t aoff = be32(d + ma_off + 4 * size_t(s));
        sq.mark_array = boff + ma_off + aoff;
        if (sq.mark_array > blimit) return fail(e, ERR_FORMAT, "mark array out of range");
        // the arrays of a bucket follow one another (index.c:700-720): a start before its predecessor's is damage
        if (s > 0 && sq.mark_array < seqs.back().mark_array) return fail(e, ERR_FORMAT, "mark arrays are not in order");
        sq.ch = seqToUnseq[s];
        seqs.push_back(sq);
        lsq.mark_array = sq.mark_array;
        lsq.ch = sq.ch;
        lseqs.push_back(lsq);
      }
      {  // end-of-bucket pseudo symbol (never occurs in L)
        DevSeq sq;
        memset(&sq, 0, sizeof sq);
        sq.ch = kAlphaSize;
        seqs.push_back(sq);
        LaneSeq lsq;
        memset(&lsq, 0, sizeof lsq);
        lsq.ch = kAlphaSize;
        lseqs.push_back(lsq);
      }
    }
  }
  if (gb != total_buckets) return fail(e, ERR_FORMAT, "bucket count mismatch");
  // per bucket and character: C[ch] <= Occ base <= C[ch+1], non-decreasing from bucket to bucket, and the counts of a
  // bucket add up to its rows (block_occs + bucket_occs, src/main/index.c:1538-1569, :1828-1843)
  for (int64_t g = 0; g < total_buckets; g++) {
    int64_t rows_here = 0;
    for (int ch = 0; ch < kAlphaSize; ch++) {
      const int64_t v = occ_base[size_t(g) * kAlphaSize + size_t(ch)];
      const int64_t nx = g + 1 < total_buckets ? occ_base[size_t(g + 1) * kAlphaSize + size_t(ch)] : C[size_t(ch) + 1];
      if (v < C[size_t(ch)] || v > nx || nx > C[size_t(ch) + 1]) return fail(e, ERR_FORMAT, "occurrence tables are not monotone");
      rows_here += nx - v;
    }
    const int64_t expect = std::min<int64_t>(b_size, total_length - g * int64_t(b_size));
    if (rows_here != expect) return fail(e, ERR_FORMAT, "occurrence tables do not add up to the bucket's rows");
  }
  block_slot_start.push_back(segs.size() / kSegmentWords);
  block_lnode_start.push_back(lnodes.size());
  block_lseq_start.push_back(lseqs.size());
  occ.resize(size_t(total_buckets) * kAlphaSize);
  for (int64_t g = 0; g < total_buckets; g++)
    for (int ch = 0; ch < kAlphaSize; ch++) {
      OccEntry& oe = occ[size_t(g) * kAlphaSize + size_t(ch)];
      oe.base = occ_base[size_t(g) * kAlphaSize + size_t(ch)];
      oe.code = leaf_code[size_t(g) * kAlphaSize + size_t(ch)];
      oe.node_base = buckets[size_t(g)].node_base;
    }
  return OK;
}

}  // namespace femto_amd
