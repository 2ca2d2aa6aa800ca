/*
 * oracle/femto_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; never part of the product).
 *
 * Plain-C restatement of the reference's (femto-dev/femto v1.3.0) count/locate path over an
 * unmodified femto index (directory of block files, or flattened single file).  Every function
 * cites the reference file:line whose behaviour it restates (paths relative to /root/reference).
 *
 * Pinning: this restatement is checked (tests/test_oracle_*.py, `-m "not gpu"`) against
 *   (1) the reference's own known-answer tests restated as data (gamma / varbyte / wavelet-tree
 *       worked example of src/main/wtree_test.c, test_construct constants of src/main/index_test.c),
 *   (2) golden vectors produced in the build container by the genuine reference (oracle/_ref,
 *       see oracle/Makefile) on committed fixture indexes (tests/golden/),
 *   (3) brute force over the fixture texts (the method of src/main/index_test.c:351-434).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 */
#ifndef FEMTO_ORACLE_H
#define FEMTO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO_ALPHA_SIZE 261       /* src/main/index_types.h:64-66 */
#define FO_SEOF 2               /* ESCAPE_CODE_SEOF, src/main/index_types.h:38-44 */
#define FO_CHARACTER_OFFSET 5

/* err_code_t values of src/utils/error.h:25-39 */
enum { FO_OK = 0, FO_ERR_MEM = 1, FO_ERR_IO = 2, FO_ERR_PARAM = 3, FO_ERR_FORMAT = 4,
       FO_ERR_BZ_DATA = 5, FO_ERR_INVALID = 6, FO_ERR_OVERWORKED = 11 };

typedef struct fo_index fo_index_t;

/* deterministic work counters used for the roofline's algorithmic bytes */
typedef struct {
  int64_t n_rank;        /* bseq_rank calls */
  int64_t n_occ;         /* wavelet Occ / rank walks (one per block_request) */
  int64_t n_mark;        /* mark-array records read */
  int64_t s_bytes;       /* varbyte bytes consumed through the target segment's pair */
  int64_t n_rle;         /* ranks that ended in an RLE segment */
  int64_t n_lit;         /* ranks that ended in a literal segment */
  int64_t n_gamma;       /* gamma codes decoded */
  int64_t n_lf;          /* LF steps in locate walks */
} fo_counters_t;

int  fo_open(const char* path, fo_index_t** out);
void fo_close(fo_index_t* ix);
int64_t fo_total_length(const fo_index_t* ix);
int64_t fo_num_blocks(const fo_index_t* ix);
int64_t fo_num_documents(const fo_index_t* ix);
int  fo_param(const fo_index_t* ix, int which); /* 0 block_size 1 b_size 2 mark_period 3 chunk_size 4 text_size_bits */

/* L1: src/main/wtree.c */
void fo_bseq_rank(const unsigned char* z, int index, int occs[2], int* bit, fo_counters_t* c);
void fo_bseq_rank_all(const unsigned char* z, int nbits, int64_t* out);
int  fo_wtree_occs(const unsigned char* wt, int leaf, int index, fo_counters_t* c);
void fo_wtree_rank(const unsigned char* wt, int index, int* leaf, int* count, fo_counters_t* c);
int  fo_decode_gamma(uint64_t word, unsigned int* out);
int  fo_decode_varbyte(const unsigned char* z, unsigned int* out);

/* L2: src/main/index.c */
int64_t fo_get_C(const fo_index_t* ix, int ch);
int64_t fo_get_block_occs(const fo_index_t* ix, int ch, int64_t block);
/* type bits: 1 CHAR, 2 OCCS, 4 LOCATION (block_request_type_t, src/main/index.h:300-321) */
int  fo_block_request(fo_index_t* ix, int64_t block, int type, int row_in_block,
                      int* ch, int* occs_in_block, int64_t* offset, fo_counters_t* c);
/* document resolution (resolve_location, src/main/index.c:1587) */
int  fo_resolve_location(const fo_index_t* ix, int64_t offset, int64_t* doc, int64_t* doc_offset);

/* L3/L4: src/main/server.c do_string_query / do_locate_query, src/main/femto.c:275,331.
   Patterns: flat alpha_t codes, pattern i = pats[starts[i] .. starts[i]+plen[i]). */
int  fo_count(fo_index_t* ix, int64_t npats, const int32_t* plen, const uint16_t* pats,
              const int64_t* starts, int64_t* first, int64_t* last, int nthreads, fo_counters_t* c);
/* offsets_flat must hold sum(noccs); pass NULL to size (noccs is filled either way) */
int  fo_locate(fo_index_t* ix, int64_t npats, const int32_t* plen, const uint16_t* pats,
               const int64_t* starts, int max_occs_each, int32_t* noccs, int64_t* offsets_flat,
               int nthreads, fo_counters_t* c);
/* single LF step with mark lookup: do_back_query, src/main/server.c:2228 */
void fo_bseq_select(const unsigned char* z, int bit, int rank1, int occs[2]);
int  fo_wtree_select(const unsigned char* wt, int leaf, int count);
/* single LF^-1 step with mark lookup: do_forward_query, src/main/server.c:2424 */
int  fo_forward_step(fo_index_t* ix, int64_t row, int64_t* new_row, int* ch, int64_t* offset, fo_counters_t* c);
int  fo_back_step(fo_index_t* ix, int64_t row, int64_t* new_row, int* ch, int64_t* offset, fo_counters_t* c);

/* do_regexp_query (src/main/server.c:1656-2163): backward simulation of one nfa_description_t (src/main/nfa.h:62-88, given
   flat: node i's transitions are entries trans_start[i] .. trans_start[i+1]-1; settings[4] = cost_bound, subst_cost,
   delete_cost, insert_cost) with the reference's stack-with-a-hash discipline (src/utils/queue_map.c), add_mapping
   (server.c:1558), the approx_* state functions (src/main/nfa.c:165-322) and regexp_result_list_sort (server.c:1528).
   out: up to cap results {first, last, match_len, cost}; returns the number of results, or -FO_ERR_* (OVERWORKED after
   max_iterations pops, server.c:40,1821).  Pinned against the genuine do_regexp_query: tests/golden/NAME_regexp.npz. */
int64_t fo_nfa_search(fo_index_t* ix, int num_nodes, const int32_t* trans_start, const int32_t* trans_char,
                      const int32_t* trans_dest, const uint8_t* is_start, const uint8_t* is_final, const int32_t* settings,
                      int64_t max_iterations, int64_t cap, int64_t* first, int64_t* last, int32_t* match_len, int32_t* cost);

#ifdef __cplusplus
}
#endif
#endif
