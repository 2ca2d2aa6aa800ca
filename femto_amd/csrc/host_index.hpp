// host_index.hpp -- host-side loader of a femto index (directory or flattened file).
// Parses what the reference parses lazily in b_fault()/read_block_header()
// (src/main/index.c:1222-1404) once, up front, into the flat tables of device_tables.h.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "device_tables.h"

namespace femto_amd {

struct Error {
  int code;
  std::string msg;
};

struct HostIndex {
  // block header facts
  int64_t total_length = 0, number_of_blocks = 0, number_of_documents = 0;
  int32_t block_size = 0, b_size = 0, mark_period = 0, chunk_size = 0, text_size_bits = 0;
  int32_t buckets_per_block = 0;
  int64_t total_buckets = 0;

  // raw images
  std::vector<uint8_t> header;          // header block bytes
  std::vector<uint8_t> image;           // all data blocks, each start 256-byte aligned (+ tail pad)
  std::vector<uint64_t> block_off;      // offset of data block b inside image
  std::vector<uint64_t> block_len;
  // first segs slot / LaneNode / LaneSeq of every data block (+ one terminal entry): the derived tables are
  // laid out block after block, which is what a range-split index (femto_amd_open_split) partitions on
  std::vector<uint64_t> block_slot_start, block_lnode_start, block_lseq_start;

  // derived tables
  std::vector<DevNode> nodes;
  std::vector<DevBucket> buckets;
  std::vector<DevSeq> seqs;
  std::vector<uint64_t> segs;           // 64-byte aligned native-endian segment slots
  std::vector<CumEntry> cum;
  std::vector<uint32_t> hint;
  std::vector<BlockDir> bdir;           // parallel to hint
  std::vector<LaneNode> lnodes;
  std::vector<LaneSeq> lseqs;
  std::vector<OccEntry> occ;
  bool dir_regular = true;              // false: some non-final segment holds < 511 bits -> raw A/S walk only
  std::vector<int64_t> occ_base;        // [gb*261+ch]
  std::vector<uint32_t> leaf_code;      // [gb*261+ch]
  std::vector<int64_t> C;               // 262 entries
  std::vector<int64_t> doc_ends;

  // Returns 0 or an err_code_t value; fills err.msg.
  int load(const std::string& path, Error* err);
  int resolve_location(int64_t offset, int64_t* doc, int64_t* doc_offset) const;
  // document_info (src/main/index.c:1768): the info string stored for a document (points into `header`)
  int document_info(int64_t doc, const uint8_t** info, int64_t* len) const;
  size_t doc_info_off = 0;              // header_block_doc_info_offset, src/main/index.c:895
};

}  // namespace femto_amd
