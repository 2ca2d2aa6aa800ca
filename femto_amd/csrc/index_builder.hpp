// index_builder.hpp -- writer of femto index directories.
// Produces, for the same documents and parameters, block files byte-identical to what the
// reference's index_documents(map=NULL) writes (src/main/construct.c:572).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "host_index.hpp"

namespace femto_amd {

struct BuildParams {             // index_block_param_t, defaults of set_default_param (src/main/index.c:122-142)
  int32_t block_size = 1024 * 1024 * 128;
  int32_t b_size = 1024 * 1024;
  int32_t mark_period = 20;
  int32_t chunk_size = 2048;     // accepted and validated like the reference; no chunks are written (map == NULL)
};

// parse_param, src/main/index.c:185-219
int parse_build_params(const char* s, BuildParams* p, Error* e);

struct Document {
  const uint8_t* bytes;
  int64_t len;
  std::string info;
};

// prepared text: bytes+5 per document followed by SEOF (src/main/bwt_prepare.c:227-311)
void prepare_text(const std::vector<Document>& docs, std::vector<uint16_t>* text, std::vector<int64_t>* doc_ends);

// Encoders exposed for unit tests (byte-identical to bseq_construct / wtree_construct,
// src/main/wtree.c:359-603, :907-1078).
void bseq_encode(const uint8_t* bits_msb_first, int64_t bitlen, int force_type, std::vector<uint8_t>* out);

// sa[i] = start (in the prepared text) of the i-th smallest suffix; the ordering is the plain
// suffix order of the concatenated prepared text with a virtual smallest end marker
// (the reference's test sorter, src/main/bwt_qsufsort.c:176-240).
int build_index_from_sa(const std::string& out_dir, const std::vector<Document>& docs, const BuildParams& params,
                        const int64_t* sa, int nthreads, Error* e);

// flatten_index (src/main/index.c:2260): directory index -> single flattened file, byte-identical
int flatten_index_dir(const std::string& index_dir, const std::string& out_path, Error* e);

// GPU suffix sort of the prepared text (suffix_sort.hip); sa_out has text.size() entries.
int gpu_suffix_sort(const std::vector<uint16_t>& text, int device, std::vector<int64_t>* sa_out, Error* e);

}  // namespace femto_amd
