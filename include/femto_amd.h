/*
 * femto_amd.h -- C ABI of the MI355X-native FM-index query engine (drop-in for femto's batched
 * count/locate path).  Plain pointers and sizes only; no C++/torch/HIP types in any signature.
 *
 * The reference (femto-dev/femto v1.3.0) exposes no plugin ABI; its batch entry points are the C
 * functions of src/main/femto_internal.h.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference tree).  INTEGRATION.md shows the shim a femto
 * maintainer would add so that parallel_count()/parallel_locate() call into this library.
 *
 * Conventions kept from the reference:
 *   - patterns are arrays of alpha_t (uint16_t) = byte + 5 (src/main/index_types.h:61-69);
 *   - every function returns an err_code_t value (src/utils/error.h:25-39), 0 = OK;
 *   - calls are blocking; a handle may be used from several host threads at once (every call leases a
 *     scratch and a stream of its own from the handle: concurrent calls overlap on the GPU);
 *   - results are bit-exact with parallel_count / parallel_locate on the same index files.
 *
 * All compute runs in hand-written HIP kernels on the GPU; there is NO CPU fallback: if no HIP
 * device is usable, femto_amd_open fails with FEMTO_AMD_ERR_INVALID.
 */
#ifndef FEMTO_AMD_H
#define FEMTO_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* err_code_t of src/utils/error.h:25-39 (same numeric values) */
enum {
  FEMTO_AMD_OK = 0,
  FEMTO_AMD_ERR_MEM = 1,
  FEMTO_AMD_ERR_IO = 2,
  FEMTO_AMD_ERR_PARAM = 3,
  FEMTO_AMD_ERR_FORMAT = 4,
  FEMTO_AMD_ERR_BZ_DATA = 5,
  FEMTO_AMD_ERR_INVALID = 6,
  FEMTO_AMD_ERR_MISSING = 8,
  FEMTO_AMD_ERR_FULL = 10,
  FEMTO_AMD_ERR_OVERWORKED = 11,
  FEMTO_AMD_ERR_UNKNOWN = 12
};

#define FEMTO_AMD_ALPHA_SIZE 261       /* src/main/index_types.h:64-66 */
#define FEMTO_AMD_CHARACTER_OFFSET 5

typedef struct femto_amd_index femto_amd_index_t;   /* opaque; owns the device-resident index */

/* Replaces femto_start_server_err + femto_loc_for_path_err (src/main/femto.c:54,269) and the lazy
 * block faults of open_header_block/open_data_block (src/main/index.c:1482,1419): reads a femto
 * index (directory "<dir>/00","<dir>/01".. per src/main/block_storage.c:257-263, or a flattened
 * single file per src/main/index.c:2260), validates every block header (src/main/index.c:1348),
 * uploads the block images unchanged to HBM of HIP device `device` and builds the small side tables
 * (per-bucket Huffman leaf codes, wavelet-node directory, Occ bases). */
int femto_amd_open(const char* index_path, int device, femto_amd_index_t** out);
void femto_amd_close(femto_amd_index_t* ix);

/* like err_string() of src/utils/error.h:72; thread-local message of the last failing call */
const char* femto_amd_last_error(void);

/* index facts (block header fields, src/main/index.c:817-868) */
typedef struct {
  int64_t total_length;       /* rows of L == prepared text length */
  int64_t number_of_blocks;
  int64_t number_of_documents;
  int32_t block_size, bucket_size, mark_period, chunk_size;
  int32_t text_size_bits;     /* bits per mark-array record */
  int64_t total_buckets;
  int64_t image_bytes;        /* bytes of femto block images resident in HBM */
  int64_t table_bytes;        /* bytes of derived side tables resident in HBM */
} femto_amd_info_t;
int femto_amd_info(const femto_amd_index_t* ix, femto_amd_info_t* out);

/* ---- host-pointer batch API ------------------------------------------------------------- */

/* Replaces parallel_count (src/main/femto.c:275; do_string_query src/main/server.c:713):
 * for pattern i, [first[i], last[i]] is the inclusive row range of the backward search (first>last
 * when there is no match; the empty pattern gives [0, total_length-1]).  If last==NULL, first[i]
 * receives the match count last-first+1 (src/main/femto.c:313-318). */
int femto_amd_parallel_count(femto_amd_index_t* ix, int npats, const int* plen,
                             const uint16_t* const* pats, int64_t* first, int64_t* last);

/* Replaces parallel_locate (src/main/femto.c:331; do_locate_query src/main/server.c:4373):
 * noccs[i] rows are located per pattern, clamped as the reference does (when last-first >
 * max_occs_each only max_occs_each rows are located -- i.e. a range of exactly max_occs_each+1 rows
 * is returned whole, src/main/server.c:4411); offsets[i] is malloc()ed by the callee (NULL when
 * noccs[i]==0) and free()d by the caller; offsets[i][j] = SA[first[i]+j], row order. */
int femto_amd_parallel_locate(femto_amd_index_t* ix, int npats, const int* plen,
                              const uint16_t* const* pats, int max_occs_each,
                              int* noccs, int64_t** offsets);

/* Replaces parallel_locate_range (src/main/femto.c:481; setup_locate_range src/main/server.c:4047): the text offset
 * of every row in [first, last]; offsets must have room for last-first+1 entries.  (serial_locate, femto.c:402, is the
 * reference's single-threaded test twin of parallel_locate: call femto_amd_parallel_locate.) */
int femto_amd_parallel_locate_range(femto_amd_index_t* ix, int64_t first, int64_t last, int64_t* offsets);

/* Flat forms of the two calls above (no per-pattern pointers): pattern i is
 * pats[starts[i] .. starts[i]+plen[i]).  locate_flat writes out_starts[npats+1] (exclusive prefix
 * sum of noccs) and at most offsets_capacity offsets; *total_out receives sum(noccs) -- call with
 * offsets==NULL to size the buffer. */
int femto_amd_count_flat(femto_amd_index_t* ix, int64_t npats, const int32_t* plen,
                         const uint16_t* pats, const int64_t* starts, int64_t* first, int64_t* last);
int femto_amd_locate_flat(femto_amd_index_t* ix, int64_t npats, const int32_t* plen,
                          const uint16_t* pats, const int64_t* starts, int max_occs_each,
                          int32_t* noccs, int64_t* out_starts, int64_t* offsets,
                          int64_t offsets_capacity, int64_t* total_out);

/* One-pass form of locate_flat: *offsets_out receives ONE malloc()ed array holding all offsets (pattern i's are
 * [out_starts[i], out_starts[i+1]); NULL when *total_out == 0); the caller free()s it.  noccs / out_starts may be NULL. */
int femto_amd_locate_flat_alloc(femto_amd_index_t* ix, int64_t npats, const int32_t* plen,
                                const uint16_t* pats, const int64_t* starts, int max_occs_each,
                                int32_t* noccs, int64_t* out_starts, int64_t** offsets_out, int64_t* total_out);

/* Raw-byte convenience form: patterns given as bytes (each +5 -> alpha_t), as femto_search does
 * for literal patterns. */
int femto_amd_count_bytes(femto_amd_index_t* ix, int64_t npats, const int32_t* plen,
                          const uint8_t* bytes, const int64_t* starts, int64_t* first, int64_t* last);

/* Offset -> (document, offset in document): resolve_location (src/main/index.c:1587). */
int femto_amd_resolve_location(const femto_amd_index_t* ix, int64_t offset, int64_t* doc, int64_t* doc_offset);
/* The info string stored with a document (its path/URL): document_info (src/main/index.c:1768).  *info points
 * into memory owned by the handle (valid until femto_amd_close; NOT NUL-terminated), *len its length. */
int femto_amd_document_info(const femto_amd_index_t* ix, int64_t doc, const char** info, int64_t* len);

/* resolve_location for a batch (the reference resolves every located row: one header_loc_query_t each,
 * do_range_to_results_query src/main/server.c:4800-4823): doc[i] / doc_offset[i] for offsets[i]; either output may be NULL.
 * Host arrays; the search runs on the GPU (femto_amd_resolve_device below is the form without the copies). */
int femto_amd_resolve_batch(femto_amd_index_t* ix, int64_t n, const int64_t* offsets, int64_t* doc, int64_t* doc_offset);

/* ---- device-pointer batch API (inputs and outputs already resident in HBM) ---------------- */
/* All pointers are device pointers on the index's device; `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  Calls only enqueue work and return; the caller synchronises.
 * Nothing is reported to the host afterwards: a pattern holding a symbol >= 261 (which the host-pointer
 * calls reject with FEMTO_AMD_ERR_PARAM) has the empty range first = 0, last = -1.
 * READS AROUND THE SYMBOLS: the kernels fetch the patterns in aligned 16-byte pieces, so up to 14 bytes before the first and
 * behind the last symbol of d_pats[] are LOADED (never used, never written).  An aligned piece cannot cross a page, so this
 * cannot fault -- but a memory checker, or a sub-allocator with poisoned guard bytes next to d_pats, will see the reads:
 * give d_pats 16 bytes of slack on either side, or align its ends to 16 bytes, if that matters.  d_pats itself must be
 * 2-byte aligned (FEMTO_AMD_ERR_PARAM otherwise).  (femto_amd_locate_device
 * falls back to a host-synchronising path -- it reads the row total back on `stream` -- for rank modes 0 and 1; the
 * packed modes 3 / 4 are enqueue-only as stated.) */
int femto_amd_count_device(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen,
                           const uint16_t* d_pats, const int64_t* d_starts,
                           int64_t* d_first, int64_t* d_last, void* stream);

/* Locate on device: phase 1 (count + clamp + prefix sum) writes d_noccs[npats] and
 * d_out_starts[npats+1]; the caller reads d_out_starts[npats] (= total) to size d_offsets, then
 * phase 2 walks every row.  d_first must hold the ranges' first rows (phase 1 fills it). */
int femto_amd_locate_plan_device(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen,
                                 const uint16_t* d_pats, const int64_t* d_starts, int max_occs_each,
                                 int64_t* d_first, int64_t* d_last, int32_t* d_noccs,
                                 int64_t* d_out_starts, void* stream);
int femto_amd_locate_walk_device(femto_amd_index_t* ix, int64_t npats, const int64_t* d_first,
                                 const int64_t* d_out_starts, int64_t total, int64_t* d_offsets,
                                 void* stream);

/* The whole of parallel_locate (src/main/femto.c:331) as ONE enqueue-only call: count, the reference's clamp, prefix
 * sum and the locate walk run as one stream-ordered chain; nothing returns to the host in between.  d_offsets has room
 * for offsets_capacity rows; d_total[0] receives the number of rows (= d_out_starts[npats]) and d_total[1] is 1 when
 * that exceeds offsets_capacity (the offsets are then incomplete: call again with a larger buffer).
 * ROW-FREE FORM: d_first == d_last == NULL.  parallel_locate itself returns `noccs` and `offsets` and never a row
 * (src/main/femto.c:331-400); a caller that passes no row arrays gets exactly that -- d_noccs, d_out_starts, d_offsets,
 * d_total, bit-identical to the form with rows -- and the search is spared what only the rows need: on handles holding
 * the text, a pattern whose remaining symbols are compared against the text on ONE row is located by that compare (its
 * offset is where the compared text starts) without the inverse-suffix-array read (and, with sampled arrays, the LF steps)
 * that would turn the position back into a row; a compare that meets another character ends the pattern at once, where the
 * form with rows runs one more step for the (first, last) of the emptied range. */
int femto_amd_locate_device(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen,
                            const uint16_t* d_pats, const int64_t* d_starts, int max_occs_each,
                            int64_t* d_first, int64_t* d_last, int32_t* d_noccs, int64_t* d_out_starts,
                            int64_t* d_offsets, int64_t offsets_capacity, int64_t* d_total, void* stream);

/* ONE step of the locate walk (do_back_query, src/main/server.c:2228-2359) for n rows, enqueue-only: d_off[i] = the text offset of
 * d_rows[i] when the row is marked (its walk ends here), else -1 and d_next[i] = LF(row) -- -1 when L[row] is a character <= SEOF
 * (a walk does not cross a document start) or the row is out of range.  The unit a RANGE-SPLIT index exchanges walkers in
 * (each round every GPU advances the walkers whose rows it owns -- owner = row / block_size,
 * src/main/index.c:1613-1617 -- and sends each to the owner of its next row): femto_amd/parallel.py exchange_locate.  Marks are the
 * handle's own (the derived, denser ones in modes 3 / 4; femto's in modes 0 / 1): offsets are the same either way. */
int femto_amd_lf_steps_device(femto_amd_index_t* ix, int64_t n, const int64_t* d_rows, int64_t* d_next, int64_t* d_off, void* stream);

/* resolve_location (src/main/index.c:1587) on the device, one lane per offset: d_doc[i] (int64) and / or d_doc32[i] (int32:
 * indexes of fewer than 2^31 documents) = the document holding text offset d_offsets[i], d_doc_offset[i] = the offset inside
 * it; any of the three outputs may be NULL, and d_doc_offset may be d_offsets itself (in place).  d_n != NULL: only
 * min(n, *d_n) offsets are live -- pass the d_total of femto_amd_locate_device and its offsets_capacity as n to resolve the
 * located rows of an enqueue-only chain without a host round trip.  Enqueue-only.  The located offsets of a match are
 * inside a document; an offset at or beyond the last document's end resolves, as in the reference, to document
 * number_of_documents. */
int femto_amd_resolve_device(femto_amd_index_t* ix, const int64_t* d_offsets, int64_t n, const int64_t* d_n, int64_t* d_doc,
                             int32_t* d_doc32, int64_t* d_doc_offset, void* stream);

/* ---- leaf requests (the reference's block_request interface, src/main/index.h:300-394) ---- */
/* For rows[i] (global row numbers, host memory): ch_out = L[row] (BLOCK_REQUEST_CHAR),
 * occ_out = Occ-in-block(L[row] or ch_in[i], row) (BLOCK_REQUEST_OCCS; ch_in==NULL -> use L[row]),
 * off_out = mark offset or -1 (BLOCK_REQUEST_LOCATION).  Any output pointer may be NULL. */
int femto_amd_block_requests(femto_amd_index_t* ix, int64_t n, const int64_t* rows, const uint16_t* ch_in,
                             uint16_t* ch_out, int32_t* occ_out, int64_t* off_out);

/* One LF^-1 step per row (do_forward_query, src/main/server.c:2424; HDR_BSEARCH_C|HDR_BSEARCH_BLOCK_OCCS|
 * HDR_FORWARD then BLOCK_REQUEST_ROW|BLOCK_REQUEST_LOCATION, src/main/index.c:1698,1915): ch_out = first
 * character of the row, row_out = the row whose LF step leads here (-1 when ch <= SEOF), off_out = that
 * row's mark offset or -1.  Exercises bseq_select / wtree_select (src/main/wtree.c:770,1150) on the GPU. */
int femto_amd_forward_steps(femto_amd_index_t* ix, int64_t n, const int64_t* rows, uint16_t* ch_out,
                            int64_t* row_out, int64_t* off_out);

/* ---- range-split index (BASELINE.json configs[4]) ------------------------------
 * The reference scales an index past memory by block: a row belongs to data block row / block_size
 * (bsearch_block_rows, src/main/index.c:1613-1617) and blocks are faulted from disk one by one through
 * its block cache.  On one MI355X node the blocks are partitioned over the GPUs and a "block fault" is a plain
 * load: part p keeps the segment lines and block images of blocks [nblocks*p/nparts,
 * nblocks*(p+1)/nparts) in its own HBM, maps the other parts' slices into its address space
 * (hipIpcOpenMemHandle across processes, peer access inside one process) and the SAME lane kernels
 * read remote lines over xGMI.  The small per-bucket tables (Occ bases, tree shapes, block
 * directories) are replicated.  Every part answers any query; batches are sharded as with a
 * replicated index.
 *
 *   femto_amd_open_split(path, dev, part, nparts, &ix)      on every part
 *   femto_amd_split_export(ix, blob)                         128-byte blob, send to all other parts
 *   femto_amd_split_attach(ix, p, blob_of_p)                 for every other part p   (other process)
 *   femto_amd_split_attach_local(ix, handle_of_p)            ... or, same process
 *   femto_amd_split_commit(ix)                               then count/locate as usual
 * A part must stay open while any other part that attached it is in use.  Only the lane kernels
 * (mode 1) run on a range-split index; femto_amd_forward_steps is not available. */
int femto_amd_open_split(const char* index_path, int device, int part, int nparts, femto_amd_index_t** out);
int femto_amd_split_export(femto_amd_index_t* ix, void* blob128);
int femto_amd_split_attach(femto_amd_index_t* ix, int part, const void* blob128);
int femto_amd_split_attach_local(femto_amd_index_t* ix, femto_amd_index_t* owner);
int femto_amd_split_commit(femto_amd_index_t* ix);
/* bytes of this part's own slices (segment lines, block images) */
int femto_amd_split_info(const femto_amd_index_t* ix, int* part, int* nparts, int64_t* seg_bytes, int64_t* image_bytes);

/* ---- regular expressions and automata (do_regexp_query src/main/server.c:1656, nfa.c, compile_regexp.c) ----
 * The reference searches a regular expression by simulating an EPSILON-FREE automaton of the REVERSED pattern backwards over
 * the index (do_regexp_query): femto_amd_nfa_t is that automaton, field for field the reference's nfa_description_t
 * (src/main/nfa.h:62-88) with the arrays flattened -- node i's transitions are entries trans_start[i] .. trans_start[i+1]-1 of
 * trans_char[] (alpha codes, byte + 5) / trans_dest[]; is_start / is_final are the two bit sets; the four costs are
 * regexp_settings_t (src/main/index_types.h:147-162: cost_bound = largest allowed total cost + 1, 1 = exact matching).
 *
 * femto_amd_nfa_search_batch replaces setup_regexp_query_take_nfa (src/main/server.h:838) + do_regexp_query for nq automata at
 * once, each searched by one GPU workgroup (regexp_search.hip): for automaton q the results are entries
 * result_start[q] .. result_start[q+1]-1 of first_out / last_out (row range of a matched string: locate its rows with
 * femto_amd_parallel_locate_range), len_out (match_len: symbols of the matched string) and cost_out (errors), in the order
 * of the reference's sorted result list (regexp_result_list_sort, server.c:1528: first ascending, last descending, equal
 * ranges and ranges inside another result dropped) -- IDENTICAL to the reference's list for the same automaton, quirks
 * included (a range with a final state alive is a result and is NOT extended; a pending range reached again is merged).
 * status_out[q] (may be NULL): 0, FEMTO_AMD_ERR_OVERWORKED (more than MAX_REGEXP_ITERATIONS = 10^6 steps, server.c:40,1821:
 * the reference returns ERR_OVERWORKED and no results) or FEMTO_AMD_ERR_FULL (more pending ranges than option
 * "regexp_stack_cap", default 2^18, at most 2^22; the reference has no such bound -- it would run on to ERR_OVERWORKED).  The out arrays hold max_results entries for ALL
 * automata together; *n_out = results in total; max_results == 0 only counts (result_start[] and *n_out are filled, the
 * other out arrays may be NULL); more results than max_results is FEMTO_AMD_ERR_FULL with *n_out = the exact number
 * to call again with (raw result ranges are buffered by the library itself; only when even its buffer -- up to half of the
 * free HBM, 16 GB at most -- cannot hold them is *n_out the RAW count: an upper bound, before equal and nested ranges are dropped).  Limits: 2048 nodes, 2^22 transitions per
 * automaton, costs and cost_bound 1..255 (errors are counted in one byte, nfa.h:74-76). */
typedef struct femto_amd_nfa {
  int32_t num_nodes;
  int32_t num_transitions;
  const int32_t* trans_start;   /* [num_nodes + 1] */
  const int32_t* trans_char;    /* [num_transitions] alpha codes */
  const int32_t* trans_dest;    /* [num_transitions] */
  const uint8_t* is_start;      /* [num_nodes] */
  const uint8_t* is_final;      /* [num_nodes] */
  int32_t cost_bound, subst_cost, delete_cost, insert_cost;
} femto_amd_nfa_t;
int femto_amd_nfa_search_batch(femto_amd_index_t* ix, int64_t nq, const femto_amd_nfa_t* nfas, int64_t max_results,
                               int64_t* result_start /* nq + 1 */, int64_t* first_out, int64_t* last_out, int32_t* len_out,
                               int32_t* cost_out, int32_t* status_out, int64_t* n_out);
/* What the handle's last automaton batch did (the reference keeps many do_regexp_query state machines in flight,
 * src/main/server.c:3969-4001; here concurrent callers' batches share the GPU's workgroups, each kernel giving up its
 * workgroups beyond a fair share between automata): out8 = { automata, workgroups launched, entries popped in all, ... by the
 * longest search, busy workgroup-seconds, the span of the search passes in seconds (both on the device's wall clock),
 * occupancy = busy / (span x workgroups), seconds the longest search waited before a workgroup took it }.
 * ix == NULL: the calling thread's own last batch. */
int femto_amd_nfa_stats(femto_amd_index_t* ix, double* out8);
/* Pattern text -> automaton.  Pattern language: femto's own (src/main/QUERY_FORMAT.txt), restated token rule by token rule
 * and production by production from src/main/posix.flex.l and src/main/posix.bison.y (femto_amd/csrc/query_parser.hpp names
 * the corners): literal bytes, `.`, `[a-z]` / `[^...]`, `( )`, `|`, one of `*` `+` `?` `{m}` `{m,}` `{m,n}` per term, backslash
 * escapes (\n \t \xNN, \x-NN for the codes below the bytes), "double" and 'single' quotes, `{x 00 01}` hex strings, `#`
 * comments, runs of three or more letters as one term; unescaped whitespace separates terms and is ignored; a leading
 * APPROX [max_cost[:subst[:delete[:insert]]]] sets the costs.  The boolean operators (AND OR NOT THEN WITHIN: document-level
 * result sets) are recognised and refused.  The automaton is the position (Glushkov) automaton of the reversed pattern.
 * (The reference's generated front end -- flex/bison, then compile_regexp.c -- cannot be built in this image; search parity
 * is pinned at the automaton: the same femto_amd_nfa_t goes to the genuine do_regexp_query and to
 * femto_amd_nfa_search_batch; the grammar is pinned by the known answers of src/main/query_planning_test.c and by the
 * query set of src/test/test.pl.)  APPROX costs are validated as compile_regexp_from_ast does (compile_regexp.c:673-685:
 * three substitutions or insertions are refused).  Limits: 2^20 bytes of pattern text, parentheses nested 256 deep,
 * repeat counts <= 4096, 4096 Thompson states: beyond them FEMTO_AMD_ERR_PARAM, never a crash. */
typedef struct femto_amd_regexp femto_amd_regexp_t;
int femto_amd_regexp_compile(const uint8_t* regex, int64_t regex_len, int max_cost, int subst_cost, int delete_cost,
                             int insert_cost, femto_amd_regexp_t** out);
const femto_amd_nfa_t* femto_amd_regexp_nfa(const femto_amd_regexp_t* r);   /* valid until femto_amd_regexp_free */
/* A whole femto_search query, prepared as src/main_cc/search_tool.cc:716-751 prepares it: parse_string, then streamline_query
 * (src/main/query_planning.c:24: optional parts at either end of the pattern are dropped and repeats there trimmed to their
 * minimum -- "a*(bc|d)+" is searched as "(bc|d)"), simplify_query (ast.c:1239: a query without alternatives is ONE string),
 * and, with FEMTO_AMD_QUERY_ICASE, icase_ast (ast.c:556; femto_search --icase).  An APPROX prefix in the text sets the costs. */
#define FEMTO_AMD_QUERY_ICASE 1
#define FEMTO_AMD_QUERY_NO_STREAMLINE 2
int femto_amd_query_compile(const uint8_t* query, int64_t query_len, int flags, femto_amd_regexp_t** out);
/* 1 when the prepared query is one plain string (search it with femto_amd_count_flat / _locate_flat as femto_search runs a
 * string query, src/main/server.c:713): *syms / *n its alpha codes, owned by r; 0 otherwise */
int femto_amd_regexp_literal(const femto_amd_regexp_t* r, const uint16_t** syms, int64_t* n);
/* the prepared query printed back as ast_to_string(ast, 0, 1) prints it (src/main/ast.c:1122; femto_search --json "pattern") */
const char* femto_amd_regexp_echo(const femto_amd_regexp_t* r);
/* test hook for the known answers of src/main/query_planning_test.c and for tests/golden/query_ast_golden.json (the genuine
 * streamline_query / simplify_query / icase_ast / ast_to_string run on the tree this parser produced): parse, then flags bit 0
 * streamline, bit 1 simplify, bit 2 icase; print back with (1) or without (0) quotes into out[cap] -- usequotes 2: the parsed
 * tree in the text form oracle/ref_tool.c `ast` reads.  Returns the length, -1 on a syntax error or a buffer too small */
int femto_amd_query_echo(const uint8_t* query, int64_t query_len, int flags, int usequotes, char* out, int64_t cap);
void femto_amd_regexp_free(femto_amd_regexp_t* r);
/* compile + search, a batch of patterns with the same costs (max_cost = 0, costs 1: exact) */
int femto_amd_regexp_search_batch(femto_amd_index_t* ix, int64_t nq, const uint8_t* const* regex, const int64_t* regex_len,
                                  int max_cost, int subst_cost, int delete_cost, int insert_cost, int64_t max_results,
                                  int64_t* result_start, int64_t* first_out, int64_t* last_out, int32_t* len_out,
                                  int32_t* cost_out, int32_t* status_out, int64_t* n_out);
/* one pattern; a search that ends OVERWORKED / FULL returns that code */
int femto_amd_regexp_search(femto_amd_index_t* ix, const uint8_t* regex, int64_t regex_len, int64_t max_results,
                            int64_t* first_out, int64_t* last_out, int32_t* len_out, int64_t* n_out);
int femto_amd_regexp_search_approx(femto_amd_index_t* ix, const uint8_t* regex, int64_t regex_len, int max_cost,
                                   int subst_cost, int delete_cost, int insert_cost, int64_t max_results,
                                   int64_t* first_out, int64_t* last_out, int32_t* len_out, int32_t* cost_out,
                                   int64_t* n_out);
/* test hook: does the automaton built from `regex` accept exactly the byte string s?  1 yes, 0 no, -1 syntax error */
int femto_amd_regexp_match(const uint8_t* regex, int64_t regex_len, const uint8_t* s, int64_t len);

/* ---- several GPUs of one node ----------------------------------------------------------------------------------------
 * Queries are independent (each string_query_t is its own state machine, src/main/server.c:3969-4001), so a batch shards
 * with no exchange during the search.
 *
 * One process, several GPUs -- what the C caller of parallel_count (src/main/femto.c:275) needs: femto_amd_open_multi
 * opens the index on every listed device (replicated) and returns an ordinary handle; the HOST-pointer batch calls
 * (femto_amd_parallel_count / _parallel_locate / _count_flat / _locate_flat / _locate_flat_alloc /
 * _parallel_locate_range) then split the batch into contiguous shards, one host thread per GPU, and every GPU returns
 * its shard straight into the caller's arrays over its own PCIe link -- no gather at all.  Device-pointer calls are not
 * available on such a handle (FEMTO_AMD_ERR_INVALID).  INTEGRATION.md: the shim opens one when FEMTO_AMD_DEVICES lists
 * more than one device. */
int femto_amd_open_multi(const char* index_path, int ndev, const int* devices, femto_amd_index_t** out);
/* The same handle for an index LARGER than one GPU's HBM (BASELINE configs[4]; the reference partitions by block,
 * bsearch_block_rows src/main/index.c:1613-1617): every big array -- block images, segment lines, packed / two-level /
 * per-character lines, level table, suffix arrays, text -- is ONE address range whose pages live in the HBM of all the
 * listed GPUs in equal contiguous stripes (HIP virtual memory management: hipMemCreate per GPU, one hipMemMap'ed
 * range, access granted to every GPU), the small tables are copied to every GPU.  The kernels are unchanged: a line in
 * another GPU's stripe is an ordinary load that travels over xGMI -- the packed fast paths, unlike
 * femto_amd_open_split's wavelet-path kernels, keep working.  Batches shard over the GPUs as with femto_amd_open_multi. */
int femto_amd_open_multi_striped(const char* index_path, int ndev, const int* devices, femto_amd_index_t** out);
int femto_amd_device_count(const femto_amd_index_t* ix);
/* One process per GPU, results resident on the devices: the only exchange is the final gather of the shards' results
 * to one rank -- a grouped batch of ncclSend / ncclRecv over xGMI (RCCL is loaded on first use, it is not a link-time
 * dependency).  Bootstrap as with NCCL: one rank makes the 128-byte id, every rank passes it to femto_amd_comm_init
 * (any transport; bench.py --gather native broadcasts it through torch.distributed).  femto_amd_comm_gather enqueues on
 * `stream`: rank r's bytes_per_rank bytes arrive at d_recv + r * bytes_per_rank on `root` (d_recv is ignored elsewhere). */
int femto_amd_comm_unique_id(void* id128);
int femto_amd_comm_init(femto_amd_index_t* ix, const void* id128, int nranks, int rank);
int femto_amd_comm_gather(femto_amd_index_t* ix, const void* d_send, void* d_recv, int64_t bytes_per_rank, int root, void* stream);
/* the communicator's own view: ranks it spans and this handle's rank in it (ncclCommCount / ncclCommUserRank) */
int femto_amd_comm_info(femto_amd_index_t* ix, int* nranks, int* rank);

/* A striped index shared between PROCESSES (one process per GPU; BASELINE.json configs[4], "index range-split across 8
 * GPUs"; the owner of a stripe is position / chunk as the owner of a row is row / block_size in bsearch_block_rows,
 * src/main/index.c:1613).  The process that called femto_amd_open_multi_striped serves `nclients` other processes: every
 * physical stripe travels as a POSIX file descriptor over the Unix-domain socket `socket_path` (SCM_RIGHTS), with a
 * description of the handle.  femto_amd_open_striped_client (index_path: the same index, for its header and document
 * table) maps the stripes at the builder's addresses, copies the small tables to `device` and returns an ordinary
 * single-GPU handle -- every kernel family and fast path, remote lines over xGMI; it waits up to timeout_s seconds for the
 * builder's socket.  femto_amd_multi_child returns replica i of a multi-device handle (a borrowed single-GPU handle, valid
 * until the parent is closed): replica 0 of a striped handle is the builder's own. */
int femto_amd_striped_serve(femto_amd_index_t* ix, const char* socket_path, int nclients);
int femto_amd_open_striped_client(const char* index_path, const char* socket_path, int device, int timeout_s, femto_amd_index_t** out);
int femto_amd_multi_child(femto_amd_index_t* ix, int i, femto_amd_index_t** child);

/* ---- patterns as keys (device pointers) ---------------------------------------------------------------------------------
 * A pattern of at most max_syms symbols whose characters all occur in the text fits ONE 64-bit word: `bits`-bit fields, the
 * pattern's LAST symbol in the top field, field = field_of_alpha[alpha code] (1 + the character's dense code: its rank among the
 * text's characters -- by character value on small alphabets, by falling frequency behind the characters <= SEOF on byte
 * alphabets: take the table, do not derive it), 0 = end of pattern.  femto_amd_key_format reports bits / max_syms (3 bits, 21 symbols for DNA; 7 bits, 9
 * symbols for a 96-character text) and the field table; femto_amd_pack_keys_device packs a (plen, pats, starts) batch that
 * is already in HBM (*d_bad = patterns no key describes: they get key 0 = the empty pattern; such a batch belongs to the
 * symbol entry points).  femto_amd_locate_keys_device is femto_amd_count_device / femto_amd_locate_device on keys: the
 * same search (parallel_count / parallel_locate, src/main/femto.c:275,331), bit-identical results, 8 bytes of input per
 * pattern instead of 12 + 2 per symbol, and -- d_ranges32 != NULL, an index of fewer than 2^31 - 1 rows -- the ranges as
 * int32 (first, last) pairs instead of two int64 arrays.  d_noccs == NULL: count only (d_out_starts / d_offsets / d_total
 * unused).  Enqueue-only, like the calls it mirrors. */
int femto_amd_key_format(const femto_amd_index_t* ix, int* bits, int* max_syms, uint8_t* field_of_alpha /* [261] or NULL */);
/* An identity of the key fields of this handle (a hash of `bits` and field_of_alpha): keys are only meaningful to a handle -- and a
 * library build -- that reports the SAME id.  A caller that packs keys itself, caches them, or ships them to another process
 * compares ids first: the fields follow the handle's dense codes, which differ between indexes and have changed between library
 * versions (byte alphabets: by falling frequency since round 5); a key of another table searches for another string, silently. */
int femto_amd_key_table_id(const femto_amd_index_t* ix, uint64_t* id);
int femto_amd_pack_keys_device(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                               const int64_t* d_starts, uint64_t* d_keys, int64_t* d_bad, void* stream);
int femto_amd_locate_keys_device(femto_amd_index_t* ix, int64_t npats, const uint64_t* d_keys, int max_occs_each,
                                 int32_t* d_ranges32 /* [2 * npats] or NULL */, int64_t* d_first, int64_t* d_last,
                                 int32_t* d_noccs, int64_t* d_out_starts /* npats + 1 */, int64_t* d_offsets,
                                 int64_t offsets_capacity, int64_t* d_total /* 2 */, void* stream);

/* The match counts of a batch in the form a result gather sends ("one RCCL collective for results";
 * parallel_count's last == NULL form, src/main/femto.c:313-318, narrowed): d_counts8[i] = min(last[i] - first[i] + 1, 255),
 * 0 when there is no match; every pattern with 255 matches or more is appended to d_big as a pair (pattern index, count),
 * in no particular order, and *d_big_n receives how many there are (pairs beyond big_capacity are not written).  Lossless;
 * one byte per pattern on the links.  Enqueue-only, device pointers. */
int femto_amd_pack_counts_device(femto_amd_index_t* ix, int64_t npats, const int64_t* d_first, const int64_t* d_last,
                                 uint8_t* d_counts8, int64_t* d_big, int64_t big_capacity, int64_t* d_big_n, void* stream);

/* ---- options of an open handle ---------------------------------------------------------------------------------------------
 * What is derived at open, and how much HBM it may take, decided by the CALLER: fill the struct with femto_amd_options_init
 * (every field "auto") and set what matters.  -1 (auto) = the rule described with each field; the environment variables
 * named in brackets are read ONLY for fields left on auto -- they are test overrides, not the configuration interface.
 * DESIGN.md 3 has the measured table level_table_syms -> bytes -> ms per step a deployer picks a budget from. */
#define FEMTO_AMD_BUDGET_ALL (-2)
typedef struct femto_amd_options {
  uint32_t struct_size;          /* sizeof(femto_amd_options_t), set by femto_amd_options_init: versions the struct */
  int32_t rank_mode;             /* -1: the fastest that applies | 0 raw | 1 lane | 3 pack | 4 pack2   [FEMTO_AMD_RANK_MODE] */
  int64_t hbm_budget_bytes;      /* bytes this handle may HOLD in all (femto_amd_structures[13]): the optional structures are
                                  * declined (identical results on the slower path) once it is spent; the level table takes
                                  * what the lines, marks and rank units leave.  -1 (auto): the DEFAULT BOUND
                                  * min(free HBM / 4, max(8 x indexed text bytes, 2 GiB)) -- 8.6 GB for a 1 GiB text, where a
                                  * random 20-mer costs ~3 memory requests instead of ~1.3.  FEMTO_AMD_BUDGET_ALL (-2): whatever
                                  * is free on the device (dense suffix arrays, deepest tables: the benchmark's setting; DESIGN.md
                                  * 3 has bytes -> patterns/s).                                        [FEMTO_AMD_HBM_BUDGET] */
  int32_t packed_lines;          /* 0: skip mode 3's lines                                                [FEMTO_AMD_PACK] */
  int32_t two_level_lines;       /* 0: skip mode 4's lines | 1: build them for <= 8 characters too        [FEMTO_AMD_PACK2] */
  int32_t char_rank_lines;       /* 0: skip the per-character rank lines of byte alphabets                [FEMTO_AMD_IND] */
  int32_t text;                  /* 0: skip text + inverse suffix array (no text tail)                    [FEMTO_AMD_TEXT] */
  int32_t dense_arrays;          /* 0: sampled suffix / inverse suffix arrays only                        [FEMTO_AMD_DENSE] */
  int32_t mark_every;            /* derived marks every n-th text position, 0 = femto's own marks; auto 5 [FEMTO_AMD_MARK_EVERY] */
  int32_t level_table;           /* 0: no level table                                                     [FEMTO_AMD_KTAB] */
  int32_t level_table_syms;      /* K, the deepest level; auto: <= 4 entries per row within the budget   [FEMTO_AMD_KTAB_SYMS] */
  int64_t level_table_bytes;     /* cap of the level table; auto: a quarter (60 % for <= 1 entry per row) of the free HBM [.._KTAB_MB] */
  int32_t context_table;         /* 0: no context tables (byte alphabets)                                 [FEMTO_AMD_CTX] */
  int32_t context_syms;          /* H of the narrow table                                                 [FEMTO_AMD_CTX_SYMS] */
  int32_t context2_table;        /* 0: no wide table                                                      [FEMTO_AMD_CTX2] */
  int32_t context2_syms;         /* H2 of the wide table (<= 16)                                          [FEMTO_AMD_CTX2_SYMS] */
  int64_t context2_bytes;        /* cap of the wide table; auto: a quarter of the free HBM                [FEMTO_AMD_CTX2_MB] */
  int32_t tail_min, tail_ones, tail_rows, tail_row_cost;   /* thresholds of the text tail (direct_kernels.hip.hpp) [FEMTO_AMD_TAIL_*] */
  int32_t sort_queries;          /* 0: mode 1 does not order large batches by suffix                      [FEMTO_AMD_SORT] */
  int32_t host_threads;          /* staging threads of host-pointer batches; auto: half the hardware threads, <= 128, <= twice a cgroup CPU quota [.._HOST_THREADS] */
  int32_t host_pipeline;         /* 0: host-pointer batches are staged in one piece                       [FEMTO_AMD_HOST_PIPELINE] */
  int32_t host_keys;             /* 0: host-pointer batches travel as symbols, never as keys              [FEMTO_AMD_HOST_KEYS] */
  int32_t host_pipe_chunk_log2;  /* log2 patterns per pipeline stage; auto 20                             [FEMTO_AMD_PIPE_CHUNK_LOG2] */
  int32_t host_d2h_staged;       /* 0: located offsets return with one plain copy                         [FEMTO_AMD_D2H_STAGED] */
  int32_t rank_units;            /* the 16-byte rank units of small alphabets (ru_kernels.hip.hpp): 0 none, 1 auto, 2 plain (88 rows), 3 marked (64 rows + mark bits; auto picks them when the handle will not hold the suffix array) [FEMTO_AMD_RU] */
  int32_t marks_32bit;           /* 0: derived mark offsets stay 8 bytes; auto: 4 bytes when the index has < 2^32 rows [FEMTO_AMD_SA32] */
  int32_t context_mid_table;     /* 1: a third context table of the length half way between the other two; auto: none [FEMTO_AMD_CTXM] */
  int32_t wavelet_lines;         /* femto's OWN tables -- the block files as uploaded and its wavelet tree as segment lines (modes 0/1;
                                  * every derivation reads them, the derived layouts' kernels do not): 1 keep them in HBM | 0 release them
                                  * once the derived layouts stand; they come back, counted, for the calls that read them
                                  * (femto_amd_set_rank_mode(0/1), LOCATION leaf requests, femto_amd_forward_steps) and leave again when
                                  * they put the handle over its budget | auto: released on handles with a budget -- 1.3 GB of a 1 GiB DNA
                                  * index that then pays for rank units, marks, the suffix array  [FEMTO_AMD_WAVELET_LINES] */
} femto_amd_options_t;
void femto_amd_options_init(femto_amd_options_t* opts);
/* femto_amd_open with options (NULL = all auto = femto_amd_open) */
int femto_amd_open_opts(const char* index_path, int device, const femto_amd_options_t* opts, femto_amd_index_t** out);

/* ---- kernel family ---------------------------------------------------------------------------- */
/* Four kernel families, all bit-exact; the default at open is the fastest that applies (mode 3 for <= 8 distinct
 * characters, mode 4 for <= 256, else mode 1); FEMTO_AMD_RANK_MODE=pack|pack2|lane|raw overrides it.
 * mode 3 ("pack"): for indexes with at most 8 distinct characters (DNA) the loader derives, on the GPU, one self-contained
 *   128-byte line per 160 rows -- three bit planes of the dense character code, a "row is marked" plane, and C[ch]+Occ
 *   before the line for each character -- so that an Occ and a whole locate step each read ONE memory line
 *   (femto_amd/csrc/pack_kernels.hip.hpp).
 * mode 4 ("pack2"): 9..256 distinct characters -- a two-level 16-ary decomposition of the dense character code, one
 *   128-byte line per level (pack2_kernels.hip.hpp), plus per-character rank lines for the search steps
 *   (ind_kernels.hip.hpp) and hashed context tables of the text's H-grams (ctx_kernels.hip.hpp) when HBM allows.
 *   Modes 3 and 4 process a batch in the CALLER's order, one lane per pattern, the first steps from a level table
 *   precomputed at open (direct_kernels.hip.hpp); they also derive the full suffix array / inverse suffix array / text
 *   (HBM allowing) so that locate is one read and a long pattern's tail is compared with the text.
 * mode 1 ("lane"): one LANE per query on femto's own wavelet tree; a rank reads one 128-byte line per segment (segment +
 *   the counts before it) from tables derived at load time.  Alphabets of more than 256 characters, range-split indexes.
 * mode 0 ("raw"): one WAVEFRONT per query walking femto's own A0/A1/AP group tables and varbyte S sums with
 *   __ballot / ds_bpermute, no derived tables at all -- BASELINE.json north_star's sketch, kept as the documented
 *   reference kernel (5 % of the HBM roofline) and for indexes whose segments the derived tables cannot
 *   describe.
 * (Rounds 1-2 also carried a persistent-grid variant of mode 1 and suffix-sorted variants of modes 3/4; both measured
 * slower than what replaced them and were removed.) */
int femto_amd_set_rank_mode(femto_amd_index_t* ix, int mode);
/* Runtime switches of an open handle: "sort" (default 1: mode 1 orders large batches by pattern suffix),
 * "regexp_max_iterations" (default 10^6 = MAX_REGEXP_ITERATIONS), "regexp_stack_cap" (default 2^18). */
int femto_amd_set_option(femto_amd_index_t* ix, const char* name, int value);
/* What was derived (diagnostics; femto_amd/__init__.py pack_info() names every bit): *available bit 0 packed lines, 1 two-level lines,
 * 2 level table, 3 suffix array of every row, 4 inverse suffix array of every position, 5 per-character rank lines, 6 context table,
 * bits 8..11 / 12..16 / 24..28 the context tables' symbol counts, bit 20 rank units, bit 21 the MARKED rank units (64 rows + mark
 * bits: handles that walk to marks), bit 22 the suffix / inverse suffix arrays hold 4-byte entries (indexes of fewer than 2^32 - 1 rows). */
int femto_amd_pack_info(const femto_amd_index_t* ix, int* available, int64_t* bytes, double* build_ms, int* ktab_syms);
/* What the handle holds in HBM, in bytes -- the server's "what may a handle spend" made visible (the counterpart of
 * server_settings_t's cache sizes, src/main/server.c:3484-3602): out[0] the femto block files as uploaded, [1] packed lines,
 * [2] offsets of the derived marks, [3] rank units, [4] level table, [5] context tables, [6] per-character rank lines,
 * [7] text + suffix / inverse suffix arrays, [8] two-level lines, [9] everything derived (lane tables included),
 * [10] distance between derived marks (0: femto's own), [11] level-table depth K, [12] bytes per mark offset,
 * [14] the HBM budget in force (-1: everything that is free), [15] 1 when it is the default bound (hbm_budget_bytes auto),
 * [13] HBM the handle holds in all (every persistent allocation of the INDEX: what hbm_budget_bytes is counted against; the
 * scratch of batch calls -- patterns, results; in the host-pointer API also the staging buffers a call's scratch keeps for the
 * next call, up to ~0.4 GB pinned host + ~0.4 GB device per concurrently calling thread -- is not part of it).  n <= 16. */
int femto_amd_structures(const femto_amd_index_t* ix, int64_t* out, int n);
/* Where the LAST staged host-pointer batch call (femto_amd_count_flat / _parallel_count ... on >= 2^18 patterns) spent its
 * wall time, in ms: out8[0] staging threads packing the caller's patterns into pinned key / symbol chunks, [1] waiting for
 * a pinned input buffer, [2] enqueueing copies / kernels / events, [3] waiting for a chunk's results to arrive over PCIe,
 * [4] staging threads moving results into the caller's arrays, [5] the whole call; [6] chunks, [7] staging threads. */
int femto_amd_host_pipeline_stats(femto_amd_index_t* ix, double* out8);
/* The staging threads' packing loop on its own (no handle, no device; diagnostic): keys_out[i] = the 64-bit key of pattern i
 * (symbols flat[starts[i] .. starts[i] + plen[i]), dense[sym] = the symbol's field for sym < ndense, `bits` bits per field,
 * last symbol in the top field: what femto_amd_pack_keys_device computes on the GPU).  Returns 1 when every pattern is described
 * by its key, 0 when some pattern is not (longer than 63 / bits symbols, a symbol without a field: such a chunk travels as
 * symbols), -1 on bad arguments.  force_scalar != 0: the plain loop instead of the AVX-512 VBMI / BMI2 one the host-pointer
 * paths use where the CPU has them (*simd_used says which ran).  Replaces the per-pattern copy of do_parallel_query's setup
 * (src/main/server.c:691-695). */
int femto_amd_host_pack_keys(const uint8_t* dense, int ndense, int bits, int64_t npats, const int32_t* plen, const uint16_t* flat,
                             const int64_t* starts, int force_scalar, uint64_t* keys_out, int* simd_used);
int femto_amd_get_rank_mode(const femto_amd_index_t* ix);

/* ---- profiling hooks ---------------------------------------------------------------------- */
/* Average duration (ms) of the launches of the named kernel ("count", "locate", "resolve", "regexp") since the last
 * reset, measured with HIP events on the stream the kernel was launched on; n_launches out. */
int femto_amd_kernel_time_ms(femto_amd_index_t* ix, const char* kernel, double* avg_ms, int64_t* n_launches);
void femto_amd_kernel_time_reset(femto_amd_index_t* ix);
void femto_amd_kernel_time_enable(femto_amd_index_t* ix, int on);

/* Compulsory HBM traffic of a batch (bench.py's roofline): runs the batch once with a line trace and reports how many
 * DISTINCT 128-byte lines of each derived array the count phase (count_lines[10]) and the row expansion + locate walk
 * (locate_lines[10]) loaded; *rows_out = rows located.  Regions: 0 packed lines (mode 3), 1 level table, 2 suffix
 * array / offsets of the marked rows, 3 / 4 level-1 / level-2 lines (mode 4), 5 text, 6 inverse suffix array, 7 rank
 * units, 8 per-character rank lines, 9 context table.
 * Device pointers as in femto_amd_count_device; blocking; not to be called while other calls use the handle. */
int femto_amd_trace_lines(femto_amd_index_t* ix, int64_t npats, const int32_t* d_plen, const uint16_t* d_pats,
                          const int64_t* d_starts, int max_occs_each, int64_t* count_lines, int64_t* locate_lines,
                          int64_t* rows_out);
/* ... and how many lines of each region the two phases of that LAST trace READ in all (the same line read for two patterns
 * counts twice): what an unsorted batch fetches when the structure is far larger than the caches. */
int femto_amd_trace_reads(const femto_amd_index_t* ix, int64_t* count_reads /* [10] */, int64_t* locate_reads /* [10] */);

/* ---- index construction (femto block-file writer) ------------------------- */
/* Builds a femto index directory (byte-identical to index_documents(map=NULL),
 * src/main/construct.c:572) from `ndocs` documents given as raw bytes.  The suffix array of the
 * prepared text (bytes+5, one SEOF per document; src/main/bwt_prepare.c:227-311, ordered as the
 * reference's test sorter orders it, src/main/bwt_qsufsort.c:176-240) is computed on the GPU.
 * params: "block_size=..,bucket_size=..,mark_period=.." (src/main/index.c:185-219) or NULL. */
int femto_amd_build_index(const char* out_dir, int ndocs, const uint8_t* const* docs, const int64_t* doc_lens,
                          const char* const* doc_infos, const char* params, int device);
/* Same, from a caller-supplied suffix array of the prepared text (host arrays); no GPU needed. */
int femto_amd_build_index_from_sa(const char* out_dir, int ndocs, const uint8_t* const* docs,
                                  const int64_t* doc_lens, const char* const* doc_infos,
                                  const char* params, const int64_t* sa);

/* flatten_index (src/main/index.c:2260; the femto_flatten tool): directory index -> one flattened file that
 * both femto and femto_amd_open read; byte-identical to the reference's output. */
int femto_amd_flatten_index(const char* index_dir, const char* out_path);

/* Test hook: encodes one binary sequence exactly as bseq_construct_forcetype does
 * (src/main/wtree.c:365; force_type -1 literal only, 0 automatic, 1 RLE only).  out may be NULL to size. */
int femto_amd_bseq_encode(const uint8_t* bits_msb_first, int64_t bitlen, int force_type, uint8_t* out,
                          int64_t cap, int64_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
