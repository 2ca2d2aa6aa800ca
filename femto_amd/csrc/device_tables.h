// device_tables.h -- layout of the HBM-resident index (shared by the host loader and the kernels).
//
// The femto block files are uploaded UNCHANGED (big-endian, 8-byte aligned) into one device
// buffer `image`; the loader adds small derived side tables so that no kernel has to parse a
// bucket's mapping/Huffman header or search a wavelet-tree directory:
//
//   image        : [data block 0 | data block 1 | ...], each block start 256-byte aligned
//   nodes[]      : one DevNode per internal wavelet-tree node, bucket after bucket, in the order
//                  of the bucket's own (sorted) node directory; local index 0 is the root
//   buckets[]    : one DevBucket per global bucket gb = row / b_size
//   occ_base[]   : int64 [gb][261]  = C[ch] + block_occs[ch][block] + bucket_occs[ch][bucket]
//                  (the three additive terms of Occ outside the wavelet tree; reference
//                  src/main/index.c:1538-1569, :1828-1843, combined as HDR_BACK does, :1740-1746)
//   leaf_code[]  : uint32 [gb][261] = Huffman leaf number (code with leading 1,
//                  src/main/index.c:290-300) of ch in that bucket, 0 if ch does not occur
//   seqs[]       : one DevSeq per in-use character per bucket: its alpha code, its mark table
//                  (a bseq) and its mark array
//   segs/cum/hint: the lane kernels' view of every binary sequence (see LaneBseq below)
#pragma once
#include <stdint.h>

namespace femto_amd {

constexpr int kAlphaSize = 261;        // ALPHA_SIZE, src/main/index_types.h:66
constexpr int kSEOF = 2;               // ESCAPE_CODE_SEOF
constexpr int kGroupSize = 31;         // GROUP_SIZE, src/main/wtree.c:48
constexpr int kSegmentWords = 8;       // SEGMENT_WORDS, src/main/wtree_funcs.h:34

struct DevBseq {            // one encoded binary sequence (src/main/wtree_funcs.h:294-358)
  uint64_t off;             // absolute byte offset of the bseq header inside `image`
  uint32_t num_groups;      // NUM_GROUPS
  uint32_t d_off;           // D_OFFSET (relative to off)
  uint32_t total_words;     // TOTAL_SEGMENT_WORDS
  uint32_t pad;
};

// ---- lane-kernel view of a binary sequence (derived at load time) --------------------------
// segs  : every sequence's D words copied into 128-byte-ALIGNED lines, one per segment: words 0..7
//         = the segment (native-endian u64, most significant bit first; zero padded), word 8 = the
//         zeros/ones before the segment (uniform sequences) or words 8..15 = the RLE skip table.
//         gfx950's L2 fetches 128-byte lines (all TCC_EA0_RDREQ are 128 B), so a rank on a uniform
//         sequence costs exactly ONE memory line;
// cum   : per segment the zeros/ones BEFORE it (the varbyte S sums and A0/A1 group tables,
//         prefix-summed), plus one terminal entry per sequence;
// hint  : per 512-bit block the segment holding the block's first bit.  Every segment but a
//         sequence's last stores >= 511 bits (save_run, src/main/wtree.c:240-290), so bit t lies in
//         segment hint[t>>9] or the next one.  Sequences whose segments all hold exactly 511 bits
//         (all-literal: random ACGT) need no hint: segment = t / 511.
// Sequences that need a hint (some segment is RLE) use TWO 64-byte slots per segment: the segment
// and, for RLE segments, a skip table so that a rank decodes at most the gamma codes of one 64-bit
// word instead of half the segment (a sigma~96 text averages 67 codes per RLE rank):
//   aux word 0      : p_1..p_7, 9 bits each (p_k in the low bits ... p_7 highest): bit position of the
//                     first gamma code that starts at or after bit 64*k of the segment
//   aux word k (1-7): low 32 bits = data bits covered by the codes before p_k (0xffffffff: no such
//                     code), high 32 bits = ones among them | (value of the run starting at p_k) << 31
struct CumEntry { uint32_t o0, o1; };
// One 32-byte entry per 512-bit block of a sequence that has RLE segments: the segment holding the
// block's first bit, the zeros/ones before it, and the zeros/ones before the NEXT segment (the only other
// segment a bit of this block can lie in).  One load replaces hint[] + two cum[] entries.
struct BlockDir {
  uint32_t seg;
  uint32_t o0, o1;      // before segment seg
  uint32_t n0, n1;      // before segment seg + 1
  uint32_t pad[3];
};
constexpr uint32_t kNoHint = 0xffffffffu;
struct LaneBseq {           // 16 bytes
  uint64_t seg_base;        // first 64-byte slot of this sequence in DevIndex::segs (two slots per segment)
  uint32_t cum_base;        // first CumEntry
  uint32_t hint_base;       // first hint entry, or kNoHint for uniform 511-bit segments
};
struct LaneNode {           // 32 bytes
  LaneBseq bs;
  int32_t child[2];
  uint32_t pad[2];
};
struct LaneSeq {            // 32 bytes
  LaneBseq mark_table;
  uint64_t mark_array;      // absolute byte offset in `image`
  uint32_t ch;
  uint32_t pad;
};
struct OccEntry {           // 16 bytes, [gb*261 + ch]
  int64_t base;             // C[ch] + block_occs + bucket_occs
  uint32_t code;            // Huffman leaf number (0: ch absent from the bucket)
  uint32_t node_base;       // first LaneNode of the bucket
};

struct DevNode {            // 32 bytes
  DevBseq bs;
  int32_t child[2];         // >= 0: local index of the internal child; < 0: leaf, seq = -1 - child
};

struct DevSeq {             // 40 bytes
  DevBseq mark_table;       // bseq over this character's occurrences in the bucket (1 = marked)
  uint64_t mark_array;      // absolute byte offset of this character's mark records
  uint32_t ch;              // alpha code (0..260); kAlphaSize for the end-of-bucket symbol
  uint32_t pad;
};

struct DevBucket {          // 16 bytes
  uint32_t node_base;       // first DevNode of this bucket
  uint32_t seq_base;        // first DevSeq of this bucket
  uint32_t n_internal;
  uint32_t n_in_use;
};

// regions of the compulsory-traffic trace (femto_amd_trace_lines): every 128-byte line a query kernel loads from one
// of these arrays sets one bit; the number of set bits x 128 B is what the launch MUST move from HBM at least once
enum { kTracePack = 0, kTraceKtab = 1, kTraceSa = 2, kTraceL1 = 3, kTraceL2 = 4, kTraceTxt = 5, kTraceIsa = 6, kTraceRu = 7, kTraceInd = 8,
       kTraceCtx = 9, kTraceRegions = 10 };

struct DevIndex {           // passed by value to kernels
  const uint8_t* image;
  const DevNode* nodes;
  const DevBucket* buckets;
  const DevSeq* seqs;
  const int64_t* occ_base;  // [gb*261 + ch]
  const uint32_t* leaf_code;// [gb*261 + ch]
  const int64_t* C;         // [262]; C[261] == total_length (get_C, src/main/index.c:1545)
  const uint64_t* segs;     // 64-byte aligned native-endian segment slots (8 words each)
  const CumEntry* cum;
  const uint32_t* hint;
  const BlockDir* bdir;     // indexed like hint[]
  const LaneNode* lnodes;   // parallel to nodes[]
  const LaneSeq* lseqs;     // parallel to seqs[]
  const OccEntry* occ;      // [gb*261 + ch]
  // small-alphabet packed lines (pack_kernels.hip.hpp); null when the index has more than 8 characters
  const uint32_t* pack;     // 32 dwords per 160 rows
  const int64_t* pack_sa;   // offsets of the marked rows, row order
  const uint8_t* pack_code; // [261] alpha code -> dense code 0..7, 0xff: not in the text
  const int64_t* pack_c;    // [16]: C[ch(code)] for code 0..7, then C[ch(code)+1]-1
  const uint64_t* ru;       // rank units of the table characters (ru_kernels.hip.hpp), two words each: [code - ru_nstop][unit], or NULL
  int64_t ru_stride;        // units per character
  int32_t ru_nstop;         // dense codes below this are <= SEOF: no unit vector
  const int64_t* ru_stop_rows;   // rows whose L is a stop character, per character in row order (ru_stop_step)
  int32_t ru_stop_off[4];   // stop character c's rows: entries ru_stop_off[c] .. ru_stop_off[c + 1] - 1
  int32_t pack_sa32;        // 1: pack_sa holds 32-bit offsets (indexes of fewer than 2^32 rows, half the bytes)
  int32_t ru_marks;         // 1: the units are the MARKED ones of 64 rows (ru_kernels.hip.hpp: mark bits of the rows holding the character)
  const int64_t* ktab2;     // level table of the first steps, heap-numbered over the table characters (direct_kernels.hip.hpp)
  int32_t kt2_syms;         // deepest level K
  int32_t kt2_base;         // t = number of table characters (characters of the text that are not <= SEOF)
  int32_t kt2_nstop;        // dense codes below this are <= SEOF: digit = dense code - kt2_nstop
  int32_t kt2_deep_big;     // deepest-level entries with at least this many rows store "recompute" (0xffffff by default)
  const uint64_t* kt2_deep; // the levels from kt2_cfrom on, compact: first (40 bits) | rows in the range (24 bits; 0xffffff: see ktab2_lookup)
  int64_t kt2_deep_off;     // heap position of level kt2_cfrom's first entry (= number of 16-byte entries of the levels above)
  int32_t kt2_cfrom;        // first compact level (K - 1, or K when K == 1)
  int32_t kt2_sa1;          // 1: a ONE-ROW entry of the compact levels also carries SA[first] (bit 63 | SA << 31 | first: indexes of at most
                            // 2^31 rows whose suffix array is resident when the table is built; direct_kernels.hip.hpp)
  uint32_t* trace;          // NULL, or the line bitmaps of femto_amd_trace_lines: bit trace_off[region] + line index
  unsigned long long* trace_reads;   // NULL, or [kTraceRegions] counters: every traced line READ (not only the distinct ones)
  int64_t trace_off[kTraceRegions];   // first bit of every traced region (kTrace* above)
  uint16_t pack_alpha[8];   // dense code -> alpha code
  int32_t pack_sigma;
  uint32_t pack_stop;       // bit c: alpha code of dense code c is <= SEOF (a locate walk stops there)
  // two-level 16-ary lines for byte alphabets (pack2_kernels.hip.hpp); null when more than 256 characters occur
  const uint32_t* p2_l1;    // 32 dwords per 64 rows
  const uint32_t* p2_l2;    // 32 dwords per 96 level-2 positions
  const int64_t* p2_base;   // [16] first level-2 line of every h
  const int64_t* p2_c;      // [512]: C[ch(code)] for code 0..255, then C[ch(code)+1]-1
  const uint16_t* p2_code;  // [261] alpha code -> dense code 0..255, 0xffff: not in the text
  const uint16_t* p2_alpha; // [256] dense code -> alpha code
  int32_t p2_sigma;
  uint32_t p2_stop_below;   // dense codes below this are <= SEOF (a locate walk stops there)
  uint32_t p2_single;       // S: the dense codes p2_stop_below .. p2_stop_below + S - 1 have a level-1 class of their own (pack2_kernels.hip.hpp: p2_hl)
  // long-pattern tail (text_kernels.hip.hpp); null when not derived
  const uint8_t* txt;       // dense character code of every text position
  const int64_t* isa8;      // row of the suffix at every (1 << isa_shift)-th text position
  const uint32_t* ind;      // per-character rank lines for byte alphabets (ind_kernels.hip.hpp), or NULL
  int64_t ind_stride;       // lines per character
  const uint64_t* ctx;      // context table (ctx_kernels.hip.hpp): {key, value} slots, or NULL
  int32_t ctx_log2;         // log2(slots)
  int32_t ctx_syms;         // H: symbols per key
  int32_t ctx_nstop;        // dense codes below this are characters <= SEOF
  int32_t ctx_bits;         // bits per key field
  const uint64_t* ctx2;     // wide context table: {key lo, key hi, value, 0} slots, or NULL
  int32_t sa32;             // 1: sa_full and isa8 hold 4-byte entries (indexes of fewer than 2^32 - 1 rows: half the bytes; 0xffffffff = -1): sa_at / isa_at
  int32_t ctx2_syms;        // H2 > ctx_syms
  int64_t ctx2_trace_off;   // its lines follow the narrow table's in the trace region
  uint64_t ctx2_slots;      // slots of the wide table (any number: slot = high half of hash x slots)
  const uint64_t* ctxm;     // a second wide table of a length between the two (ctx_syms < ctxm_syms < ctx2_syms), or NULL
  uint64_t ctxm_slots;
  int64_t ctxm_trace_off;
  int32_t ctxm_syms;
  int32_t ctxm_pad;
  const int64_t* sa_full;   // SA[row] of EVERY row when HBM allows (8 B/row), else NULL: locate is then one read, no walk
  int32_t isa_shift;        // 0: full inverse suffix array (8 B/row), 3: every 8th position
  int32_t row_free;        // 1: this launch is a locate that returns no rows (parallel_locate's contract, src/main/femto.c:331-400): a one-row range
                            // whose text tail consumes the pattern is located by the compare itself -- no inverse-suffix-array read, no (first, last)
  void* tail_items;         // TailItem work list of the current count launch
  int* tail_count;
  int32_t tail_min;         // hand a one-row range over when at least this many symbols remain
  int32_t tail_ones;        // ... and, inline tail only, the range has been one row for this many steps
  int32_t tail_rows;        // inline tail: ranges of up to this many rows ...
  int32_t tail_row_cost;    // ... when tail_min + tail_row_cost x (rows - 1) symbols remain
  int64_t total_length;
  int64_t total_buckets;
  int32_t b_size;
  int32_t b_shift;          // log2(b_size) when b_size is a power of two, else -1
  int32_t text_size_bits;
  int32_t walk_limit;       // a locate walk of a well-formed index ends within mark_period steps; beyond this many the
                            // index is inconsistent (e.g. an LF cycle without marks) and the row is reported as -1
};

}  // namespace femto_amd
