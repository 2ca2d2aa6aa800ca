/*
 * oracle/ref_tool.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A small driver of OUR OWN that calls the genuine reference (femto-dev/femto v1.3.0,
 * compiled by oracle/Makefile from /root/reference where it lies) through its own API:
 *
 *   build   : init_prepared_text / count_file / append_file_mem (src/main/bwt_prepare.c:115,192,227)
 *             -> save_prepared_bwt (src/main/bwt_creator.c:36, Larsson qsufsort test sorter)
 *             -> bwt_reader_open -> index_documents(map=NULL) (src/main/construct.c:572)
 *   dump    : open_header_block / open_data_block, header_occs_request, block_request
 *             (src/main/index.h:366-394) for every row: L[row], Occ(L[row],row), mark offset
 *   count   : parallel_count  (src/main/femto.c:275)
 *   locate  : parallel_locate (src/main/femto.c:331)
 *   bench   : the same two calls, timed (wall clock of the batch call only)
 *   forward : the leaf requests of do_forward_query (src/main/server.c:2424) for every row
 *   bseq    : bseq_construct_forcetype (src/main/wtree.c:365) -> encoded image
 *   flatten : flatten_index (src/main/index.c:2260)
 *   ast     : streamline_query / simplify_query / icase_ast / ast_to_string (src/main/query_planning.c, ast.c) on a query tree
 *             rebuilt with the reference's own constructors from the text form our parser dumps
 *   resolve : header_loc_request(HDR_LOC_RESOLVE_LOCATION) = resolve_location (src/main/index.c:1587) for every offset
 *   regexp_nfa : setup_regexp_query_take_nfa (src/main/server.h:838, server.c:1342) + femto_run_query: do_regexp_query
 *             (server.c:1656) on hand-fed nfa_description_t automata (filled as nfa_test.c:57-80 fills them); the
 *             reference's regex FRONT END (flex/bison) is not needed for this, and is not built
 *
 * Nothing here is part of the product; the product never links or executes this file.
 *
 * Pattern file ("FPAT", native little-endian): u32 magic 0x54415046, u32 npats,
 * i32 len[npats], then sum(len) alpha_t (u16 = byte+5) codes.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>
#include <sys/stat.h>

#include "index.h"
#include "femto_internal.h"
#include "construct.h"
#include "bwt_prepare.h"
#include "bwt_creator.h"
#include "bwt_reader.h"
#include "server.h"
#include "timing.h"
#include "wtree_funcs.h"
#include "nfa.h"
#include "bit_array.h"
#include "ast.h"
#include "query_planning.h"

#define FPAT_MAGIC 0x54415046u

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void die(const char* what, error_t err)
{
  fprintf(stderr, "ref_tool: %s failed", what);
  if (err) fprintf(stderr, ": %s", err_string(err));
  fprintf(stderr, "\n");
  exit(2);
}

static unsigned char* slurp(const char* path, int64_t* len)
{
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  *len = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char* d = malloc(*len ? *len : 1);
  if (*len && fread(d, 1, *len, f) != (size_t)*len) { perror("fread"); exit(2); }
  fclose(f);
  return d;
}

typedef struct {
  int npats;
  int* plen;
  alpha_t** pats;
  alpha_t* storage;
} patset_t;

static patset_t read_patterns(const char* path)
{
  patset_t ps;
  int64_t len;
  unsigned char* d = slurp(path, &len);
  uint32_t* w = (uint32_t*)d;
  if (len < 8 || w[0] != FPAT_MAGIC) { fprintf(stderr, "bad pattern file %s\n", path); exit(2); }
  ps.npats = (int)w[1];
  ps.plen = malloc(sizeof(int) * (ps.npats + 1));
  ps.pats = malloc(sizeof(alpha_t*) * (ps.npats + 1));
  memcpy(ps.plen, d + 8, sizeof(int) * ps.npats);
  ps.storage = (alpha_t*)(d + 8 + 4 * (size_t)ps.npats);
  size_t off = 0;
  for (int i = 0; i < ps.npats; i++) {
    ps.pats[i] = ps.storage + off;
    off += ps.plen[i];
  }
  return ps;
}

/* build <index_dir> <params|-> <doc files...>   (params e.g. "block_size=65536,bucket_size=8192,mark_period=20") */
static int cmd_build(int argc, char** argv)
{
  if (argc < 3) { fprintf(stderr, "build <index_dir> <params|-> <doc>...\n"); return 2; }
  const char* index_dir = argv[0];
  const char* params = argv[1];
  int ndocs = argc - 2;
  error_t err;
  index_block_param_t param;
  prepared_text_t p;
  char info_path[4096];
  char bwt_path[4096];

  set_default_param(&param);
  if (strcmp(params, "-") != 0) {
    err = parse_param(&param, params);
    if (err) die("parse_param", err);
  }
  mkdir(index_dir, 0777);
  snprintf(info_path, sizeof info_path, "%s/_build_info.tmp", index_dir);
  snprintf(bwt_path, sizeof bwt_path, "%s/_build_bwt.tmp", index_dir);

  err = init_prepared_text(&p, info_path);
  if (err) die("init_prepared_text", err);

  unsigned char** docs = malloc(sizeof(char*) * ndocs);
  int64_t* lens = malloc(sizeof(int64_t) * ndocs);
  for (int i = 0; i < ndocs; i++) {
    docs[i] = slurp(argv[2 + i], &lens[i]);
    err = count_file(&p, lens[i], 0, NULL, strlen(argv[2 + i]), (unsigned char*)argv[2 + i]);
    if (err) die("count_file", err);
  }
  for (int i = 0; i < ndocs; i++) {
    err = append_file_mem(&p, lens[i], docs[i], 0, NULL, NULL,
                          strlen(argv[2 + i]), (unsigned char*)argv[2 + i]);
    if (err) die("append_file_mem", err);
  }

  /* FEMTO_REF_WITH_MAP=1: also write the document map so that buckets carry document chunks
     (the production layout).  Only safe with one bucket per block (index_documents overruns
     chunks[] otherwise, construct.c:667). */
  int with_map = getenv("FEMTO_REF_WITH_MAP") != NULL;
  char map_path[4096];
  snprintf(map_path, sizeof map_path, "%s/_build_map.tmp", index_dir);
  double t0 = now_s();
  FILE* bf = fopen(bwt_path, "w+");
  if (!bf) { perror(bwt_path); return 2; }
  FILE* mf = with_map ? fopen(map_path, "w+") : NULL;
  start_clock(); /* save_prepared_bwt stops one clock more than it starts (bwt_creator.c:128) */
  err = save_prepared_bwt(&p, param.mark_period, bf, param.chunk_size, mf, 0);
  if (err) die("save_prepared_bwt", err);
  double t1 = now_s();
  rewind(bf);

  bwt_reader_t bwt;
  bwt_document_map_reader_t map;
  err = bwt_reader_open(&bwt, bf);
  if (err) die("bwt_reader_open", err);
  if (with_map) {
    rewind(mf);
    err = bwt_document_map_reader_open(&map, mf);
    if (err) die("bwt_document_map_reader_open", err);
  }
  err = index_documents(&bwt, with_map ? &map : NULL, &p.info_reader, &param, index_dir, NULL);
  if (err) die("index_documents", err);
  bwt_reader_close(&bwt);
  if (with_map) { bwt_document_map_reader_close(&map); remove(map_path); }
  double t2 = now_s();
  free_prepared_text(&p);
  remove(info_path);
  remove(bwt_path);
  printf("{\"bwt_s\": %.3f, \"index_s\": %.3f}\n", t1 - t0, t2 - t1);
  return 0;
}

/* dump <index> <out.bin>: i64 total_length, i64 nblocks, i64 C[262], i64 block_occs[261*nblocks],
   then per row: u16 L, i32 occ_in_block(L[row],row) , i64 offset   (packed arrays, one after another) */
static int cmd_dump(int argc, char** argv)
{
  if (argc < 2) return 2;
  error_t err;
  path_translator_t pt;
  index_locator_t loc;
  header_block_t hb;
  memset(&pt, 0, sizeof pt);
  err = path_translator_init(&pt);
  if (err) die("path_translator_init", err);
  err = path_translator_id_for_path(&pt, argv[0], &loc);
  if (err) die("id_for_path", err);
  err = open_header_block(&hb, &pt, loc);
  if (err) die("open_header_block", err);

  int64_t n = hb.hdr.total_length, nblocks = hb.hdr.number_of_blocks;
  FILE* out = fopen(argv[1], "wb");
  fwrite(&n, 8, 1, out);
  fwrite(&nblocks, 8, 1, out);
  for (int ch = 0; ch <= ALPHA_SIZE; ch++) {
    header_occs_request_t r; memset(&r, 0, sizeof r);
    r.ch = ch;
    err = header_occs_request(&hb, HDR_REQUEST_C, &r);
    if (err) die("HDR_REQUEST_C", err);
    fwrite(&r.occs, 8, 1, out);
  }
  for (int ch = 0; ch < ALPHA_SIZE; ch++)
    for (int64_t b = 0; b < nblocks; b++) {
      header_occs_request_t r; memset(&r, 0, sizeof r);
      r.ch = ch; r.block_num = b;
      err = header_occs_request(&hb, HDR_REQUEST_BLOCK_OCCS, &r);
      if (err) die("HDR_REQUEST_BLOCK_OCCS", err);
      fwrite(&r.occs, 8, 1, out);
    }
  uint16_t* L = malloc(2 * n);
  int32_t* occ = malloc(4 * n);
  int64_t* off = malloc(8 * n);
  int64_t row = 0;
  for (int64_t b = 0; b < nblocks; b++) {
    data_block_t blk;
    err = open_data_block(&blk, &pt, loc, b, 4);
    if (err) die("open_data_block", err);
    for (int r = 0; r < blk.hdr.size; r++, row++) {
      block_request_t q; memset(&q, 0, sizeof q);
      q.row_in_block = r; q.ch = INVALID_ALPHA;
      err = block_request(&blk, BLOCK_REQUEST_CHAR | BLOCK_REQUEST_OCCS | BLOCK_REQUEST_LOCATION, &q);
      if (err) die("block_request", err);
      L[row] = q.ch; occ[row] = q.occs_in_block; off[row] = q.offset;
    }
    close_data_block(&blk);
  }
  fwrite(L, 2, n, out); fwrite(occ, 4, n, out); fwrite(off, 8, n, out);
  fclose(out);
  close_header_block(&hb);
  return 0;
}

/* occs <index> <ch> <out.bin>: i32 Occ-in-block(ch,row) for every row (BLOCK_REQUEST_OCCS) */
static int cmd_occs(int argc, char** argv)
{
  if (argc < 3) return 2;
  error_t err;
  path_translator_t pt;
  index_locator_t loc;
  header_block_t hb;
  int ch = atoi(argv[1]);
  memset(&pt, 0, sizeof pt);
  err = path_translator_init(&pt);
  if (err) die("path_translator_init", err);
  err = path_translator_id_for_path(&pt, argv[0], &loc);
  if (err) die("id_for_path", err);
  err = open_header_block(&hb, &pt, loc);
  if (err) die("open_header_block", err);
  int64_t nblocks = hb.hdr.number_of_blocks;
  FILE* out = fopen(argv[2], "wb");
  for (int64_t b = 0; b < nblocks; b++) {
    data_block_t blk;
    err = open_data_block(&blk, &pt, loc, b, 4);
    if (err) die("open_data_block", err);
    for (int r = 0; r < blk.hdr.size; r++) {
      block_request_t q; memset(&q, 0, sizeof q);
      q.row_in_block = r; q.ch = ch;
      err = block_request(&blk, BLOCK_REQUEST_OCCS, &q);
      if (err) die("block_request", err);
      int32_t v = q.occs_in_block;
      fwrite(&v, 4, 1, out);
    }
    close_data_block(&blk);
  }
  fclose(out);
  return 0;
}

/* forward <index> <out.bin>: for every row the two leaf requests of do_forward_query
   (src/main/server.c:2424-2565): header HDR_BSEARCH_C|HDR_BSEARCH_BLOCK_OCCS|HDR_REQUEST_BLOCK_ROWS|HDR_FORWARD,
   then block BLOCK_REQUEST_ROW|BLOCK_REQUEST_LOCATION.  out: u16 chr[n], i64 new_row[n] (-1 when chr <= SEOF), i64 offset[n] */
static int cmd_forward(int argc, char** argv)
{
  if (argc < 2) return 2;
  error_t err;
  path_translator_t pt;
  index_locator_t loc;
  header_block_t hb;
  memset(&pt, 0, sizeof pt);
  err = path_translator_init(&pt);
  if (err) die("path_translator_init", err);
  err = path_translator_id_for_path(&pt, argv[0], &loc);
  if (err) die("id_for_path", err);
  err = open_header_block(&hb, &pt, loc);
  if (err) die("open_header_block", err);
  int64_t n = hb.hdr.total_length, nblocks = hb.hdr.number_of_blocks;
  uint16_t* chr = malloc(2 * n);
  int64_t* nrow = malloc(8 * n);
  int64_t* off = malloc(8 * n);
  data_block_t* blks = calloc(nblocks, sizeof(data_block_t));
  for (int64_t b = 0; b < nblocks; b++) {
    err = open_data_block(&blks[b], &pt, loc, b, 4);
    if (err) die("open_data_block", err);
  }
  for (int64_t row = 0; row < n; row++) {
    header_occs_request_t r; memset(&r, 0, sizeof r);
    r.ch = INVALID_ALPHA; r.occs = row;   /* setup_header_occs_query(..., chr, row=0, occs=row) */
    err = header_occs_request(&hb, HDR_BSEARCH_C | HDR_BSEARCH_BLOCK_OCCS | HDR_REQUEST_BLOCK_ROWS | HDR_FORWARD, &r);
    if (err) die("header_occs_request", err);
    chr[row] = r.ch; nrow[row] = -1; off[row] = -1;
    if (r.ch <= ESCAPE_CODE_SEOF) continue;
    block_request_t q; memset(&q, 0, sizeof q);
    q.ch = r.ch; q.occs_in_block = (int) r.occs;
    err = block_request(&blks[r.block_num], BLOCK_REQUEST_ROW | BLOCK_REQUEST_LOCATION, &q);
    if (err) die("block_request ROW", err);
    nrow[row] = r.row + q.row_in_block;
    off[row] = q.offset;
  }
  FILE* out = fopen(argv[1], "wb");
  fwrite(chr, 2, n, out); fwrite(nrow, 8, n, out); fwrite(off, 8, n, out);
  fclose(out);
  return 0;
}

static femto_server_t start_srv(int threads)
{
  femto_server_t srv;
  server_settings_t settings;
  error_t err = set_default_server_settings(&settings);
  if (err) die("set_default_server_settings", err);
  if (threads > 0) settings.num_threads = threads;  /* default is the reference's hard-wired 1 (server.c:3597) */
  err = start_server(&srv.state, &settings);
  if (err) die("start_server", err);
  return srv;
}

/* count <index> <patfile> <out.bin>: i64 first[n], i64 last[n] */
static int cmd_count(int argc, char** argv)
{
  if (argc < 3) return 2;
  femto_server_t srv = start_srv(0);
  index_locator_t loc;
  error_t err = femto_loc_for_path_err(&srv, argv[0], &loc);
  if (err) die("femto_loc_for_path_err", err);
  patset_t ps = read_patterns(argv[1]);
  int64_t* first = calloc(ps.npats + 1, 8);
  int64_t* last = calloc(ps.npats + 1, 8);
  err = parallel_count(&srv, loc, ps.npats, ps.plen, ps.pats, first, last);
  if (err) die("parallel_count", err);
  FILE* out = fopen(argv[2], "wb");
  fwrite(first, 8, ps.npats, out);
  fwrite(last, 8, ps.npats, out);
  fclose(out);
  femto_stop_server(&srv);
  return 0;
}

/* locate <index> <patfile> <max_occs> <out.bin>: i32 noccs[n], then all offsets i64 concatenated */
static int cmd_locate(int argc, char** argv)
{
  if (argc < 4) return 2;
  femto_server_t srv = start_srv(0);
  index_locator_t loc;
  error_t err = femto_loc_for_path_err(&srv, argv[0], &loc);
  if (err) die("femto_loc_for_path_err", err);
  patset_t ps = read_patterns(argv[1]);
  int max_occs = atoi(argv[2]);
  int* noccs = calloc(ps.npats + 1, sizeof(int));
  int64_t** offsets = calloc(ps.npats + 1, sizeof(int64_t*));
  err = parallel_locate(&srv, loc, ps.npats, ps.plen, ps.pats, max_occs, noccs, offsets);
  if (err) die("parallel_locate", err);
  FILE* out = fopen(argv[3], "wb");
  fwrite(noccs, 4, ps.npats, out);
  for (int i = 0; i < ps.npats; i++)
    if (noccs[i] > 0) fwrite(offsets[i], 8, noccs[i], out);
  fclose(out);
  femto_stop_server(&srv);
  return 0;
}

/* serial <index> <patfile> <max_occs> <out.bin>: serial_locate one pattern at a time, as query_tool.c:157 calls it;
   same output layout as locate */
static int cmd_serial(int argc, char** argv)
{
  if (argc < 4) return 2;
  femto_server_t srv = start_srv(0);
  index_locator_t loc;
  error_t err = femto_loc_for_path_err(&srv, argv[0], &loc);
  if (err) die("femto_loc_for_path_err", err);
  patset_t ps = read_patterns(argv[1]);
  int max_occs = atoi(argv[2]);
  int* noccs = calloc(ps.npats + 1, sizeof(int));
  int64_t** offsets = calloc(ps.npats + 1, sizeof(int64_t*));
  for (int i = 0; i < ps.npats; i++) {
    err = serial_locate(&srv, loc, 1, &ps.plen[i], &ps.pats[i], max_occs, &noccs[i], &offsets[i]);
    if (err) die("serial_locate", err);
  }
  FILE* out = fopen(argv[3], "wb");
  fwrite(noccs, 4, ps.npats, out);
  for (int i = 0; i < ps.npats; i++)
    if (noccs[i] > 0) fwrite(offsets[i], 8, noccs[i], out);
  fclose(out);
  femto_stop_server(&srv);
  return 0;
}

/* bench <index> <patfile> <count|locate> <max_occs> <threads> <reps> [out.bin]
   (out.bin, count mode: i64 first[n], i64 last[n] of the last repetition -- lets the caller check
   the very results that were timed) */
static int cmd_bench(int argc, char** argv)
{
  if (argc < 6) return 2;
  int locate = strcmp(argv[2], "locate") == 0;
  int max_occs = atoi(argv[3]);
  int threads = atoi(argv[4]);
  int reps = atoi(argv[5]);
  femto_server_t srv = start_srv(threads);
  index_locator_t loc;
  error_t err = femto_loc_for_path_err(&srv, argv[0], &loc);
  if (err) die("femto_loc_for_path_err", err);
  patset_t ps = read_patterns(argv[1]);
  int64_t* first = calloc(ps.npats + 1, 8);
  int64_t* last = calloc(ps.npats + 1, 8);
  int* noccs = calloc(ps.npats + 1, sizeof(int));
  int64_t** offsets = calloc(ps.npats + 1, sizeof(int64_t*));
  double best = 1e30, tot = 0;
  int64_t results = 0;
  /* one untimed warm-up pass pages the index in */
  for (int rep = -1; rep < reps; rep++) {
    double t0 = now_s();
    if (locate) err = parallel_locate(&srv, loc, ps.npats, ps.plen, ps.pats, max_occs, noccs, offsets);
    else err = parallel_count(&srv, loc, ps.npats, ps.plen, ps.pats, first, last);
    double dt = now_s() - t0;
    if (err) die("batch call", err);
    if (locate) {
      results = 0;
      for (int i = 0; i < ps.npats; i++) { results += noccs[i]; free(offsets[i]); offsets[i] = NULL; }
    } else {
      results = 0;
      for (int i = 0; i < ps.npats; i++) if (last[i] >= first[i]) results += last[i] - first[i] + 1;
    }
    if (rep >= 0) { tot += dt; if (dt < best) best = dt; }
  }
  if (argc >= 7 && !locate) {
    FILE* out = fopen(argv[6], "wb");
    fwrite(first, 8, ps.npats, out);
    fwrite(last, 8, ps.npats, out);
    fclose(out);
  }
  printf("{\"mode\": \"%s\", \"npats\": %d, \"threads\": %d, \"reps\": %d, \"best_s\": %.6f, \"mean_s\": %.6f, "
         "\"results\": %lld, \"patterns_per_s\": %.1f}\n",
         locate ? "locate" : "count", ps.npats, threads > 0 ? threads : 1, reps, best, tot / reps,
         (long long)results, ps.npats / best);
  femto_stop_server(&srv);
  return 0;
}

/* bseq <rawbits-file> <bitlen> <type -1|0|1> <out>: bseq_construct_forcetype (src/main/wtree.c:365) image */
static int cmd_bseq(int argc, char** argv)
{
  if (argc < 4) return 2;
  int64_t len;
  unsigned char* raw = slurp(argv[0], &len);
  int bitlen = atoi(argv[1]), type = atoi(argv[2]);
  int zlen = 0; unsigned char* z = NULL;
  error_t err = bseq_construct_forcetype(&zlen, &z, bitlen, raw, NULL, type);
  if (err) die("bseq_construct_forcetype", err);
  FILE* out = fopen(argv[3], "wb");
  fwrite(z, 1, zlen, out);
  fclose(out);
  return 0;
}

/* flatten <index_dir> <out_file>: flatten_index (src/main/index.c:2260) */
static int cmd_flatten(int argc, char** argv)
{
  if (argc < 2) return 2;
  error_t err = flatten_index(argv[0], argv[1]);
  if (err) die("flatten_index", err);
  return 0;
}

/* regexp_nfa <index> <nfas.bin> <out.bin>
   nfas.bin ("FNFA", native little-endian): u32 magic 0x41464e46, u32 nq, then per automaton
     i32 num_nodes, num_trans, cost_bound, subst_cost, delete_cost, insert_cost;
     i32 trans_start[num_nodes + 1]; i32 trans_char[num_trans] (alpha codes); i32 trans_dest[num_trans];
     u8 is_start[num_nodes]; u8 is_final[num_nodes]
   out.bin: per automaton i32 err_code (0 = ok), i32 nresults, then nresults x { i64 first, i64 last, i32 match_len,
   i32 cost } in the order of the query's sorted result list (regexp_result_list_sort, server.c:1528). */
static int cmd_regexp_nfa(int argc, char** argv)
{
  if (argc < 3) return 2;
  femto_server_t srv = start_srv(0);
  index_locator_t loc;
  error_t err = femto_loc_for_path_err(&srv, argv[0], &loc);
  if (err) die("femto_loc_for_path_err", err);
  int64_t flen;
  unsigned char* fb = slurp(argv[1], &flen);
  const unsigned char* p = fb;
  uint32_t magic, nq;
  memcpy(&magic, p, 4); p += 4;
  memcpy(&nq, p, 4); p += 4;
  if (magic != 0x41464e46u) { fprintf(stderr, "bad FNFA magic\n"); return 2; }
  FILE* out = fopen(argv[2], "wb");
  if (!out) { perror(argv[2]); return 2; }
  double query_s = 0;       /* wall time inside femto_run_query only (setup, index open and result writing excluded) */
  for (uint32_t qi = 0; qi < nq; qi++) {
    int32_t hd[6];
    memcpy(hd, p, 24); p += 24;
    const int nn = hd[0], nt = hd[1];
    const int32_t* tstart = (const int32_t*) p; p += 4 * (size_t)(nn + 1);
    const int32_t* tchar = (const int32_t*) p; p += 4 * (size_t) nt;
    const int32_t* tdest = (const int32_t*) p; p += 4 * (size_t) nt;
    const unsigned char* is_start = p; p += nn;
    const unsigned char* is_final = p; p += nn;
    nfa_description_t* nfa = malloc(sizeof(nfa_description_t));   /* freed by cleanup_regexp_query */
    err = init_nfa_description(nfa, nn);
    if (err) die("init_nfa_description", err);
    for (int i = 0; i < nn; i++) {
      const int k = tstart[i + 1] - tstart[i];
      nfa->nodes[i].num_transitions = k;
      nfa->nodes[i].transitions = k ? malloc(k * sizeof(nfa_transition_t)) : NULL;
      for (int j = 0; j < k; j++) {
        nfa->nodes[i].transitions[j].character = (alpha_t) tchar[tstart[i] + j];
        nfa->nodes[i].transitions[j].destination = tdest[tstart[i] + j];
      }
      if (is_start[i]) set_bit(nfa->bit_array_len, nfa->start_states_set, i);
      if (is_final[i]) set_bit(nfa->bit_array_len, nfa->final_states_set, i);
    }
    nfa->settings.cost_bound = hd[2];
    nfa->settings.subst_cost = hd[3];
    nfa->settings.delete_cost = hd[4];
    nfa->settings.insert_cost = hd[5];
    regexp_query_t* q = malloc(sizeof(regexp_query_t));
    err = setup_regexp_query_take_nfa(q, NULL, loc, nfa, 0);
    if (err) die("setup_regexp_query_take_nfa", err);
    const double tq = now_s();
    err = femto_run_query(&srv, (query_entry_t*) q);
    query_s += now_s() - tq;
    if (err) die("femto_run_query", err);
    int32_t code = q->proc.entry.err_code, nres = code ? 0 : q->results.num_results;
    fwrite(&code, 4, 1, out);
    fwrite(&nres, 4, 1, out);
    for (int i = 0; i < nres; i++) {
      const regexp_result_t* r = &q->results.results[i];
      int32_t ml = r->match_len, cost = r->cost;
      fwrite(&r->first, 8, 1, out);
      fwrite(&r->last, 8, 1, out);
      fwrite(&ml, 4, 1, out);
      fwrite(&cost, 4, 1, out);
    }
    cleanup_regexp_query(q);
    free(q);
  }
  fclose(out);
  free(fb);
  printf("{\"automata\": %u, \"query_s\": %.6f}\n", nq, query_s);
  femto_stop_server(&srv);
  return 0;
}

/* resolve <index> <out.bin>: header_loc_request(HDR_LOC_RESOLVE_LOCATION | HDR_LOC_REQUEST_DOC_LEN) = resolve_location
 * (src/main/index.c:1587) for EVERY logical offset 0 .. total_length - 1 (the end-of-document markers' offsets included):
 * i64 n, i64 ndocs, then (i64 doc, i64 offset in doc) x n, then i64 document_length x ndocs */
static int cmd_resolve(int argc, char** argv)
{
  if (argc < 2) return 2;
  error_t err;
  path_translator_t pt;
  index_locator_t loc;
  header_block_t hb;
  memset(&pt, 0, sizeof pt);
  err = path_translator_init(&pt);
  if (err) die("path_translator_init", err);
  err = path_translator_id_for_path(&pt, argv[0], &loc);
  if (err) die("id_for_path", err);
  err = open_header_block(&hb, &pt, loc);
  if (err) die("open_header_block", err);
  int64_t n = hb.hdr.total_length, ndocs = hb.hdr.number_of_documents;
  FILE* out = fopen(argv[1], "wb");
  if (!out) die("fopen", 0);
  fwrite(&n, 8, 1, out);
  fwrite(&ndocs, 8, 1, out);
  for (int64_t off = 0; off < n; off++) {
    header_loc_request_t r; memset(&r, 0, sizeof r);
    r.offset = off;
    err = header_loc_request(&hb, HDR_LOC_RESOLVE_LOCATION, &r);
    if (err) die("HDR_LOC_RESOLVE_LOCATION", err);
    fwrite(&r.loc.doc, 8, 1, out);
    fwrite(&r.loc.offset, 8, 1, out);
  }
  for (int64_t d = 0; d < ndocs; d++) {
    header_loc_request_t r; memset(&r, 0, sizeof r);
    r.loc.doc = d;
    err = header_loc_request(&hb, HDR_LOC_REQUEST_DOC_LEN, &r);
    if (err) die("HDR_LOC_REQUEST_DOC_LEN", err);
    fwrite(&r.doc_len, 8, 1, out);
  }
  fclose(out);
  close_header_block(&hb);
  return 0;
}

/* ast <trees.txt> <out.txt> [icase]: for every line of trees.txt -- a query as OUR parser (femto_amd/csrc/query_parser.hpp) parsed
 * it, in q_dump's text form -- rebuild the tree with the reference's own constructors (src/main/ast.h), then do what femto_search
 * does to a parsed query (src/main_cc/search_tool.cc:726-751): streamline_query (src/main/query_planning.c:24), simplify_query
 * (src/main/ast.c:1239), optionally icase_ast (ast.c:556); print ast_to_string(ast, 0, 1) and ast_to_string(ast, 0, 0), TAB-separated.
 * The generated parser (flex/bison) is not involved: this pins everything BEHIND it to the genuine code. */
static struct regexp_node* ast_read_regexp(char** p);
static long ast_int(char** p)
{
  while (**p == ' ') (*p)++;
  long v = strtol(*p, p, 10);
  return v;
}
static char ast_tag(char** p)
{
  while (**p == ' ') (*p)++;
  char c = **p;
  if (c) (*p)++;
  return c;
}
static struct atom_node* ast_read_atom(char** p)
{
  if (ast_tag(p) != 'A') die("ast: expected A", 0);
  range_t rep;
  rep.min = (int) ast_int(p);
  rep.max = (int) ast_int(p);
  struct ast_node* child = NULL;
  const char kind = ast_tag(p);
  if (kind == 'C') {
    child = (struct ast_node*) character_node_new((int) ast_int(p) - CHARACTER_OFFSET);
  } else if (kind == 'T') {
    struct set_node* sn = set_node_new();
    long n = ast_int(p);
    for (long i = 0; i < n; i++) set_node_set_one(sn, (alpha_t) ast_int(p));
    child = (struct ast_node*) sn;
  } else if (kind == 'G') {
    string_t str;
    str.len = (int) ast_int(p);
    str.chars = malloc(sizeof(alpha_t) * (str.len ? str.len : 1));
    for (int i = 0; i < str.len; i++) str.chars[i] = (alpha_t) ast_int(p);
    child = (struct ast_node*) string_node_new(str);
  } else if (kind == 'P') {
    child = (struct ast_node*) ast_read_regexp(p);
  } else die("ast: bad atom kind", 0);
  struct atom_node* a = atom_node_new(child);
  return atom_node_set_repeats(a, rep);
}
static struct regexp_node* ast_read_regexp(char** p)
{
  if (ast_tag(p) != 'R') die("ast: expected R", 0);
  regexp_settings_t st;
  st.cost_bound = (int) ast_int(p);
  st.subst_cost = (int) ast_int(p);
  st.delete_cost = (int) ast_int(p);
  st.insert_cost = (int) ast_int(p);
  long nch = ast_int(p);
  struct regexp_node* r = NULL;
  for (long c = 0; c < nch; c++) {
    if (ast_tag(p) != 'S') die("ast: expected S", 0);
    long na = ast_int(p);
    struct sequence_node* sq = sequence_node_new_empty();
    for (long a = 0; a < na; a++) sequence_node_add_atom(sq, ast_read_atom(p));
    r = r ? regexp_node_add_choice(r, sq) : regexp_node_new(sq);
  }
  if (!r) die("ast: regexp without choices", 0);
  r->s = st;
  return r;
}
static int cmd_ast(int argc, char** argv)
{
  if (argc < 2) return 2;
  const int icase = argc >= 3 && atoi(argv[2]);
  int64_t flen;
  unsigned char* fb = slurp(argv[0], &flen);
  FILE* out = fopen(argv[1], "wb");
  if (!out) die("fopen", 0);
  char* line = (char*) fb;
  char* end = (char*) fb + flen;
  while (line < end) {
    char* nl = memchr(line, '\n', (size_t)(end - line));
    if (!nl) nl = end;
    char saved = *nl;
    *nl = 0;
    if (*line) {
      char* p = line;
      struct ast_node* ast = (struct ast_node*) ast_read_regexp(&p);
      streamline_query(ast);
      error_t err = simplify_query(&ast);
      if (err) die("simplify_query", err);
      if (icase) icase_ast(&ast);
      char* quoted = ast_to_string(ast, 0, 1);
      char* plain = ast_to_string(ast, 0, 0);
      fprintf(out, "%s\t%s\n", quoted ? quoted : "", plain ? plain : "");
      free(quoted);
      free(plain);
      free_ast_node(ast);
    }
    *nl = saved;
    line = nl + 1;
  }
  fclose(out);
  free(fb);
  return 0;
}

int main(int argc, char** argv)
{
  if (argc < 2) {
    fprintf(stderr, "usage: ref_tool build|dump|occs|count|locate|bench ...\n");
    return 2;
  }
  const char* c = argv[1];
  if (!strcmp(c, "build")) return cmd_build(argc - 2, argv + 2);
  if (!strcmp(c, "dump")) return cmd_dump(argc - 2, argv + 2);
  if (!strcmp(c, "occs")) return cmd_occs(argc - 2, argv + 2);
  if (!strcmp(c, "count")) return cmd_count(argc - 2, argv + 2);
  if (!strcmp(c, "locate")) return cmd_locate(argc - 2, argv + 2);
  if (!strcmp(c, "serial")) return cmd_serial(argc - 2, argv + 2);
  if (!strcmp(c, "bench")) return cmd_bench(argc - 2, argv + 2);
  if (!strcmp(c, "forward")) return cmd_forward(argc - 2, argv + 2);
  if (!strcmp(c, "bseq")) return cmd_bseq(argc - 2, argv + 2);
  if (!strcmp(c, "flatten")) return cmd_flatten(argc - 2, argv + 2);
  if (!strcmp(c, "regexp_nfa")) return cmd_regexp_nfa(argc - 2, argv + 2);
  if (!strcmp(c, "resolve")) return cmd_resolve(argc - 2, argv + 2);
  if (!strcmp(c, "ast")) return cmd_ast(argc - 2, argv + 2);
  fprintf(stderr, "unknown command %s\n", c);
  return 2;
}
