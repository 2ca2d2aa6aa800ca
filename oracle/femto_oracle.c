/*
 * oracle/femto_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never part of the product).
 * See femto_oracle.h for scope and pinning.  Paths cited are relative to /root/reference.
 */
#define _GNU_SOURCE
#include "femto_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#define GROUP_SIZE 31          /* src/main/wtree.c:48 */
#define SEGMENT_WORDS 8        /* src/main/wtree_funcs.h:34 */
#define BLOCK_HEADER_SIZE 88   /* src/main/index.c:39-41 */
#define HEADER_BLOCK_START 0xb1177deau  /* src/main/index.h:182-186 */
#define DATA_BLOCK_START   0xb1501deau
#define BLOCK_VERSION 6
#define END_OF_HEADER 0xe0ffff4du
#define BUCKET_START 0xb140bcc7u
#define FLATTENED_START 0xb1497deau     /* src/main/block_storage.h:119-120 */
#define WTREE_SETTINGS (GROUP_SIZE + 0x1000 * SEGMENT_WORDS) /* src/main/wtree.c:50-53 */

static inline uint32_t be32(const void* p) { uint32_t v; memcpy(&v, p, 4); return __builtin_bswap32(v); }
static inline uint64_t be64(const void* p) { uint64_t v; memcpy(&v, p, 8); return __builtin_bswap64(v); }

/* ------------------------------------------------------------------ L1 */

/* decode_gamma, src/main/wtree_funcs.h:60-74: k leading zeros then a (k+1)-bit value */
int fo_decode_gamma(uint64_t word, unsigned int* out)
{
  int k = word ? __builtin_clzll(word) : 64;
  k = 2 * k + 1;
  *out = (unsigned int)(word >> (64 - k));
  return k;
}

/* decode_varbyte, src/main/wtree_funcs.h:458-479: little-endian 7-bit groups, LAST byte has 0x80 */
int fo_decode_varbyte(const unsigned char* z, unsigned int* out)
{
  int i = 0; unsigned int value = 0; unsigned char w;
  do {
    w = z[i];
    value |= (unsigned int)(w & 0x7f) << (7 * i);
    i++;
  } while (!(w & 0x80));
  *out = value;
  return i;
}

typedef struct { uint64_t w[SEGMENT_WORDS]; } seg_t;

/* bseq_segment, src/main/wtree_funcs.h:482-511: words past TOTAL_SEGMENT_WORDS read as zero */
static seg_t load_segment(const unsigned char* z, int seg)
{
  seg_t s;
  int total = (int)be32(z + 8);
  const unsigned char* D = z + be32(z + 12);
  for (int i = 0; i < SEGMENT_WORDS; i++) {
    int wi = SEGMENT_WORDS * seg + i;
    s.w[i] = wi < total ? be64(D + 8 * (size_t)wi) : 0;
  }
  return s;
}

/* 64 bits starting at bit position p of the 512-bit segment (advance_segs_reader,
   src/main/wtree_funcs.h:113-149), zero beyond the end */
static uint64_t seg_window(const seg_t* s, int p)
{
  int wi = p / 64, sh = p % 64;
  uint64_t v = 0;
  if (wi < SEGMENT_WORDS) v = s->w[wi] << sh;
  if (sh && wi + 1 < SEGMENT_WORDS) v |= s->w[wi + 1] >> (64 - sh);
  return v;
}

/* bseq_rank, src/main/wtree.c:635-763 (+ bsearch_A0A1 :609-629) */
void fo_bseq_rank(const unsigned char* z, int index1, int occs[2], int* bit_out, fo_counters_t* c)
{
  unsigned int index = (unsigned int)index1 - 1;
  int G = (int)be32(z + 4);
  const unsigned char* A0 = z + 16;
  const unsigned char* A1 = A0 + 4 * (size_t)G;
  const unsigned char* AP = A1 + 4 * (size_t)G;
  const unsigned char* S = AP + 4 * (size_t)G;
#define VAL(g) (be32(A0 + 4 * (size_t)(g)) + be32(A1 + 4 * (size_t)(g)))
  int a = 0, b = G - 1, group;
  if (index >= VAL(b)) group = b;
  else {
    while (b - a > 1) {
      int m = (a + b) / 2;
      if (index < VAL(m)) b = m; else a = m;
    }
    group = a;
  }
#undef VAL
  unsigned int o0 = be32(A0 + 4 * (size_t)group), o1 = be32(A1 + 4 * (size_t)group);
  const unsigned char* sums = S + be32(AP + 4 * (size_t)group);
  int segment = 0, i = 0;
  for (;;) {
    unsigned int s0, s1;
    i += fo_decode_varbyte(sums + i, &s0);
    i += fo_decode_varbyte(sums + i, &s1);
    if (o0 + s0 + o1 + s1 <= index) { o0 += s0; o1 += s1; segment++; }
    else break;
  }
  if (c) { c->n_rank++; c->s_bytes += i; }
  segment += GROUP_SIZE * group;

  seg_t sg = load_segment(z, segment);
  int bit;
  if (sg.w[0] >> 63) {                       /* RLE segment, wtree.c:690-712 */
    bit = (int)((sg.w[0] >> 62) & 1);
    int p = 2;
    if (c) c->n_rle++;
    for (;;) {
      unsigned int v;
      p += fo_decode_gamma(seg_window(&sg, p), &v);
      if (c) c->n_gamma++;
      if (o0 + o1 + v <= index) { if (bit) o1 += v; else o0 += v; bit = !bit; }
      else { unsigned int r = 1 + index - (o0 + o1); if (bit) o1 += r; else o0 += r; break; }
    }
  } else {                                   /* literal segment, wtree.c:713-759 */
    int nb = (int)(1 + index - o0 - o1);
    int word_idx = nb / 64, bit_idx = nb % 64;
    if (c) c->n_lit++;
    for (int k = 0; k < word_idx; k++) {
      int ones = __builtin_popcountll(sg.w[k]);
      int zeros = 64 - ones;
      if (k == 0) zeros--;
      o0 += zeros; o1 += ones;
    }
    uint64_t tmp = sg.w[word_idx];
    if (word_idx == 0) { tmp <<= 1; bit_idx--; }
    int num_bits = bit_idx + 1;
    tmp >>= 64 - num_bits;
    int ones = __builtin_popcountll(tmp);
    o0 += num_bits - ones; o1 += ones;
    bit = (int)(tmp & 1);
  }
  occs[0] = (int)o0; occs[1] = (int)o1; *bit_out = bit;
}

/* bseq_select, src/main/wtree.c:770-885 (+ bsearch_A0A1 with one array, :609-629): position of the
   rank'th (1-based) occurrence of `bit`; returns occs[] like the reference (index = occs[0]+occs[1]) */
void fo_bseq_select(const unsigned char* z, int bit, int rank1, int occs[2])
{
  unsigned int rank = (unsigned int)rank1 - 1;
  int G = (int)be32(z + 4);
  const unsigned char* A0 = z + 16;
  const unsigned char* A1 = A0 + 4 * (size_t)G;
  const unsigned char* AP = A1 + 4 * (size_t)G;
  const unsigned char* S = AP + 4 * (size_t)G;
  const unsigned char* Ab = bit ? A1 : A0;
#define VAL(g) (be32(Ab + 4 * (size_t)(g)))
  int a = 0, b = G - 1, group;
  if (rank >= VAL(b)) group = b;
  else {
    while (b - a > 1) { int m = (a + b) / 2; if (rank < VAL(m)) b = m; else a = m; }
    group = a;
  }
#undef VAL
  unsigned int o[2];
  o[0] = be32(A0 + 4 * (size_t)group);
  o[1] = be32(A1 + 4 * (size_t)group);
  const unsigned char* sums = S + be32(AP + 4 * (size_t)group);
  int segment = 0, i = 0;
  for (;;) {
    unsigned int s[2];
    i += fo_decode_varbyte(sums + i, &s[0]);
    i += fo_decode_varbyte(sums + i, &s[1]);
    if (o[bit] + s[bit] <= rank) { o[0] += s[0]; o[1] += s[1]; segment++; }
    else break;
  }
  segment += GROUP_SIZE * group;
  seg_t sg = load_segment(z, segment);
  if (sg.w[0] >> 63) {
    int rb = (int)((sg.w[0] >> 62) & 1);
    int p = 2;
    for (;;) {
      unsigned int v;
      p += fo_decode_gamma(seg_window(&sg, p), &v);
      if (bit != rb || o[bit] + v <= rank) { o[rb] += v; rb = !rb; }
      else { o[bit] += 1 + rank - o[bit]; break; }
    }
  } else {
    int word_idx;
    for (word_idx = 0; word_idx < SEGMENT_WORDS; word_idx++) {
      unsigned int cnt[2];
      cnt[1] = (unsigned)__builtin_popcountll(sg.w[word_idx]);
      cnt[0] = 64 - cnt[1];
      if (word_idx == 0) cnt[0]--;
      if (o[bit] + cnt[bit] <= rank) { o[0] += cnt[0]; o[1] += cnt[1]; }
      else break;
    }
    uint64_t tmp = word_idx < SEGMENT_WORDS ? sg.w[word_idx] : 0;
    if (word_idx == 0) tmp <<= 1;
    for (int k = 0; k < 64; k++) {
      int rb = (int)(tmp >> 63);
      if (bit != rb || o[bit] + 1 <= rank) o[rb] += 1;
      else { o[bit] += 1 + rank - o[bit]; break; }
      tmp <<= 1;
    }
  }
  occs[0] = (int)o[0]; occs[1] = (int)o[1];
}

/* test helper: rank at every index 1..nbits -> out[3*i] = {occs0, occs1, bit} */
void fo_bseq_rank_all(const unsigned char* z, int nbits, int64_t* out)
{
  for (int i = 0; i < nbits; i++) {
    int occs[2], bit;
    fo_bseq_rank(z, i + 1, occs, &bit, NULL);
    out[3 * (size_t)i] = occs[0]; out[3 * (size_t)i + 1] = occs[1]; out[3 * (size_t)i + 2] = bit;
  }
}

/* wtree_bseq + stored_num_for_node_num, src/main/wtree_funcs.h:584-626 */
static const unsigned char* wtree_node(const unsigned char* wt, unsigned int node)
{
  int n = (int)be32(wt);
  const unsigned char* dir = wt + 4;
  int a = 0, b = n - 1;
  int found = -1;
  if (n <= 0) return NULL;
#define NODE(k) be32(dir + 8 * (size_t)(k))
  if (node < NODE(a)) return NULL;
  else if (node == NODE(a)) found = a;
  else if (node > NODE(b)) return NULL;
  else if (node == NODE(b)) found = b;
  else {
    while (b - a > 1) {
      int m = (a + b) / 2;
      if (node < NODE(m)) b = m;
      else if (node == NODE(m)) { found = m; break; }
      else a = m;
    }
  }
#undef NODE
  if (found < 0) return NULL;
  return wt + be32(dir + 8 * (size_t)found + 4);
}

/* wtree_occs, src/main/wtree.c:1081-1115 */
int fo_wtree_occs(const unsigned char* wt, int leaf, int index, fo_counters_t* c)
{
  int leaf_len = 31 - __builtin_clz((unsigned)leaf);
  int node = 1;
  for (int i = 1;; i++) {
    const unsigned char* bs = wtree_node(wt, (unsigned)node);
    if (!bs) break;
    int occs[2], bit;
    fo_bseq_rank(bs, index, occs, &bit, c);
    node = leaf >> (leaf_len - i);
    index -= occs[!(node & 1)];
    if (index == 0) break;
  }
  return index;
}

/* wtree_rank, src/main/wtree.c:1117-1148 */
void fo_wtree_rank(const unsigned char* wt, int index, int* leaf, int* count, fo_counters_t* c)
{
  int node = 1;
  for (;;) {
    const unsigned char* bs = wtree_node(wt, (unsigned)node);
    if (!bs) break;
    int occs[2], bit;
    fo_bseq_rank(bs, index, occs, &bit, c);
    node = (node << 1) | bit;
    index -= occs[!bit];
  }
  *leaf = node; *count = index;
}

/* wtree_select, src/main/wtree.c:1150-1178: index (1-based) of the count'th occurrence of `leaf` */
int fo_wtree_select(const unsigned char* wt, int leaf, int count)
{
  int rank = count, node = leaf;
  while (node > 1) {
    int bit = node & 1;
    node >>= 1;
    const unsigned char* bs = wtree_node(wt, (unsigned)node);
    int occs[2];
    fo_bseq_select(bs, bit, rank, occs);
    rank = occs[0] + occs[1];
  }
  return rank;
}

/* ------------------------------------------------------------------ L2 */

typedef struct {
  const unsigned char* data;
  size_t len;
  int mapped;
} blob_t;

typedef struct {
  uint32_t bucket_offset, wtree_offset, mark_tables_offset, mark_arrays_offset; /* within block */
  int nInUse;
  unsigned char inUse[FO_ALPHA_SIZE];
  uint16_t seqToUnseq[FO_ALPHA_SIZE + 1];
  uint16_t unseqToSeq[FO_ALPHA_SIZE];
  int32_t leaf[FO_ALPHA_SIZE + 1];    /* huff_code with the leading 1 (index.c:290-300) */
  unsigned char len[FO_ALPHA_SIZE + 1];
} bucket_t;

typedef struct {
  blob_t blob;
  int64_t block_number;
  int32_t num_buckets, size;
  bucket_t* buckets;
} dblock_t;

struct fo_index {
  blob_t header;
  int64_t total_length, nblocks, ndocs;
  int32_t block_size, b_size, mark_period, chunk_size, buckets_per_block;
  int text_size_bits;
  dblock_t* blocks;
};

static int map_file_range(const char* path, int64_t start, int64_t len, blob_t* out)
{
  int fd = open(path, O_RDONLY);
  if (fd < 0) return FO_ERR_IO;
  if (len < 0) {
    struct stat st;
    if (fstat(fd, &st)) { close(fd); return FO_ERR_IO; }
    len = st.st_size - start;
  }
  long pg = sysconf(_SC_PAGESIZE);
  int64_t astart = start / pg * pg;
  unsigned char* p = mmap(NULL, (size_t)(len + (start - astart)), PROT_READ, MAP_PRIVATE, fd, astart);
  close(fd);
  if (p == MAP_FAILED) return FO_ERR_IO;
  out->data = p + (start - astart);
  out->len = (size_t)len;
  out->mapped = 1;
  return FO_OK;
}

/* read_block_header, src/main/index.c:1348-1404 */
static int check_block_header(const blob_t* b, uint32_t magic)
{
  if (b->len < BLOCK_HEADER_SIZE) return FO_ERR_FORMAT;
  if (be32(b->data) != magic) return FO_ERR_FORMAT;
  if (be32(b->data + 4) != BLOCK_VERSION) return FO_ERR_FORMAT;
  if (be32(b->data + 76) != WTREE_SETTINGS) return FO_ERR_FORMAT;
  if (be32(b->data + 80) != FO_ALPHA_SIZE) return FO_ERR_FORMAT;
  if (be32(b->data + 84) != END_OF_HEADER) return FO_ERR_FORMAT;
  return FO_OK;
}

static int num_bits64(int64_t v) { return v > 0 ? 64 - __builtin_clzll((uint64_t)v) : 1; } /* bit_funcs.h:191 */

typedef struct { const unsigned char* p; size_t bit; } bitrd_t;
static unsigned rd_bits(bitrd_t* r, int n) /* MSB-first, as bsR24 (src/utils/buffer_funcs.h:136) */
{
  unsigned v = 0;
  for (int i = 0; i < n; i++, r->bit++)
    v = (v << 1) | ((r->p[r->bit >> 3] >> (7 - (r->bit & 7))) & 1);
  return v;
}

/* BZ2_hbAssignCodes, src/main/huffman.c:152-167 + leading 1 (index.c:290-300) */
static void assign_codes(int32_t* code, const unsigned char* len, int minLen, int maxLen, int alphaSize)
{
  int32_t vec = 0;
  for (int n = minLen; n <= maxLen; n++) {
    for (int i = 0; i < alphaSize; i++) if (len[i] == n) { code[i] = vec; vec++; }
    vec <<= 1;
  }
  for (int i = 0; i < alphaSize; i++) code[i] |= (1 << len[i]);
}

/* b_fault, src/main/index.c:1222-1342 */
static int parse_bucket(const dblock_t* blk, int id, bucket_t* e)
{
  const unsigned char* d = blk->blob.data;
  uint32_t boff = be32(d + BLOCK_HEADER_SIZE + 4 * (size_t)id);
  if (boff + 24 > blk->blob.len) return FO_ERR_FORMAT;
  if (be32(d + boff) != BUCKET_START) return FO_ERR_FORMAT;
  e->bucket_offset = boff;
  uint32_t map_off = boff + be32(d + boff + 4);
  e->wtree_offset = boff + be32(d + boff + 8);
  e->mark_tables_offset = boff + be32(d + boff + 12);
  e->mark_arrays_offset = boff + be32(d + boff + 16);
  if ((e->bucket_offset & 7) || (e->wtree_offset & 7) || (e->mark_tables_offset & 7) || (e->mark_arrays_offset & 7))
    return FO_ERR_FORMAT;
  bitrd_t r = { d + map_off, 0 };
  unsigned char inUse16[17];
  for (int i = 0; i < 17; i++) inUse16[i] = (unsigned char)rd_bits(&r, 1);
  memset(e->inUse, 0, sizeof e->inUse);
  for (int i = 0; i < 17; i++)
    if (inUse16[i])
      for (int j = 0; j < 16; j++) {
        unsigned uc = rd_bits(&r, 1);
        if (uc && i * 16 + j < FO_ALPHA_SIZE) e->inUse[i * 16 + j] = 1;
      }
  e->nInUse = 0;
  for (int i = 0; i < FO_ALPHA_SIZE; i++) {
    if (e->inUse[i]) { e->seqToUnseq[e->nInUse] = (uint16_t)i; e->unseqToSeq[i] = (uint16_t)e->nInUse; e->nInUse++; }
    else e->unseqToSeq[i] = 0xffff;
  }
  int alphaSize = e->nInUse + 1;
  int curr = (int)rd_bits(&r, 5);
  for (int i = 0; i < alphaSize; i++) {
    for (;;) {
      if (curr < 1 || curr > 20) return FO_ERR_BZ_DATA;
      if (rd_bits(&r, 1) == 0) break;
      if (rd_bits(&r, 1) == 0) curr++; else curr--;
    }
    e->len[i] = (unsigned char)curr;
  }
  int minLen = 32, maxLen = 0;
  for (int i = 0; i < alphaSize; i++) {
    if (e->len[i] > maxLen) maxLen = e->len[i];
    if (e->len[i] < minLen) minLen = e->len[i];
  }
  assign_codes(e->leaf, e->len, minLen, maxLen, alphaSize);
  return FO_OK;
}

static int open_dblock(dblock_t* blk, fo_index_t* ix, int64_t number)
{
  int rc = check_block_header(&blk->blob, DATA_BLOCK_START);
  if (rc) return rc;
  const unsigned char* d = blk->blob.data;
  blk->block_number = (int64_t)be64(d + 8);
  if (blk->block_number != number) return FO_ERR_FORMAT;
  blk->num_buckets = (int32_t)be32(d + 40);
  blk->size = (int32_t)be32(d + 44);
  blk->buckets = calloc((size_t)blk->num_buckets ? blk->num_buckets : 1, sizeof(bucket_t));
  if (!blk->buckets) return FO_ERR_MEM;
  (void)ix;
  for (int i = 0; i < blk->num_buckets; i++) {
    rc = parse_bucket(blk, i, &blk->buckets[i]);
    if (rc) return rc;
  }
  return FO_OK;
}

int fo_open(const char* path, fo_index_t** out)
{
  struct stat st;
  if (stat(path, &st)) return FO_ERR_IO;
  fo_index_t* ix = calloc(1, sizeof *ix);
  if (!ix) return FO_ERR_MEM;
  int rc;
  int flat = S_ISREG(st.st_mode);
  blob_t dir = {0};
  const unsigned char* offs = NULL;
  if (flat) {                                  /* src/main/block_storage.c:157-190, 496-540 */
    rc = map_file_range(path, 0, -1, &dir);
    if (rc) { free(ix); return rc; }
    if (dir.len < 16 || be32(dir.data) != FLATTENED_START || be32(dir.data + 4) != 6) { free(ix); return FO_ERR_FORMAT; }
    offs = dir.data + 16;
    int64_t s = (int64_t)be64(offs), e = (int64_t)be64(offs + 8);
    ix->header.data = dir.data + s; ix->header.len = (size_t)(e - s); ix->header.mapped = 0;
  } else {                                     /* "%s/%02x", src/main/block_storage.c:257-263 */
    char fn[4096];
    snprintf(fn, sizeof fn, "%s/%02x", path, 0);
    rc = map_file_range(fn, 0, -1, &ix->header);
    if (rc) { free(ix); return rc; }
  }
  rc = check_block_header(&ix->header, HEADER_BLOCK_START);
  if (rc) { free(ix); return rc; }
  const unsigned char* h = ix->header.data;
  ix->nblocks = (int64_t)be64(h + 16);
  ix->total_length = (int64_t)be64(h + 24);
  ix->ndocs = (int64_t)be64(h + 32);
  ix->block_size = (int32_t)be32(h + 52);
  ix->b_size = (int32_t)be32(h + 56);
  ix->mark_period = (int32_t)be32(h + 60);
  ix->chunk_size = (int32_t)be32(h + 72);
  if (ix->block_size <= 0 || ix->b_size <= 0 || ix->block_size % ix->b_size) { free(ix); return FO_ERR_PARAM; }
  ix->buckets_per_block = ix->block_size / ix->b_size;
  ix->text_size_bits = num_bits64(ix->total_length);   /* index.c:1441 */
  ix->blocks = calloc((size_t)ix->nblocks ? ix->nblocks : 1, sizeof(dblock_t));
  for (int64_t b = 0; b < ix->nblocks; b++) {
    dblock_t* blk = &ix->blocks[b];
    if (flat) {
      int64_t s = (int64_t)be64(offs + 8 * (b + 1)), e = (int64_t)be64(offs + 8 * (b + 2));
      blk->blob.data = dir.data + s; blk->blob.len = (size_t)(e - s);
    } else {
      char fn[4096];
      snprintf(fn, sizeof fn, "%s/%02llx", path, (unsigned long long)(b + 1));
      rc = map_file_range(fn, 0, -1, &blk->blob);
      if (rc) return rc;
    }
    rc = open_dblock(blk, ix, b);
    if (rc) return rc;
  }
  *out = ix;
  return FO_OK;
}

void fo_close(fo_index_t* ix)
{
  if (!ix) return;
  for (int64_t b = 0; b < ix->nblocks; b++) free(ix->blocks[b].buckets);
  free(ix->blocks);
  free(ix); /* mappings are left to process exit: test infrastructure */
}

int64_t fo_total_length(const fo_index_t* ix) { return ix->total_length; }
int64_t fo_num_blocks(const fo_index_t* ix) { return ix->nblocks; }
int64_t fo_num_documents(const fo_index_t* ix) { return ix->ndocs; }
int fo_param(const fo_index_t* ix, int which)
{
  switch (which) {
    case 0: return ix->block_size; case 1: return ix->b_size; case 2: return ix->mark_period;
    case 3: return ix->chunk_size; case 4: return ix->text_size_bits; default: return -1;
  }
}

/* get_C, src/main/index.c:1538-1554 (C[ALPHA_SIZE] == total_length) */
int64_t fo_get_C(const fo_index_t* ix, int ch)
{
  if (ch >= FO_ALPHA_SIZE) return ix->total_length;
  return (int64_t)be64(ix->header.data + BLOCK_HEADER_SIZE + 8 * (size_t)ch);
}

/* get_block_occs, src/main/index.c:1556-1569 */
int64_t fo_get_block_occs(const fo_index_t* ix, int ch, int64_t block)
{
  const unsigned char* arr = ix->header.data + BLOCK_HEADER_SIZE + 8 * FO_ALPHA_SIZE;
  return (int64_t)be64(arr + 8 * ((size_t)ch * (size_t)ix->nblocks + (size_t)block));
}

/* resolve_location, src/main/index.c:1587-1611 (bsearch_int64_ntoh_arr, src/utils/util.c:346) */
int fo_resolve_location(const fo_index_t* ix, int64_t offset, int64_t* doc, int64_t* doc_offset)
{
  const unsigned char* arr = ix->header.data + BLOCK_HEADER_SIZE + 8 * FO_ALPHA_SIZE * (1 + (size_t)ix->nblocks);
  int64_t prev = -1;  /* largest i with doc_ends[i] <= offset */
  for (int64_t i = 0; i < ix->ndocs; i++) if ((int64_t)be64(arr + 8 * i) <= offset) prev = i; else break;
  if (prev < 0) { *doc = 0; *doc_offset = offset; }
  else { *doc = prev + 1; *doc_offset = offset - (int64_t)be64(arr + 8 * prev); }
  return FO_OK;
}

/* get_bucket_occs, src/main/index.c:1828-1843 */
static int get_bucket_occs(const fo_index_t* ix, const dblock_t* blk, int ch, int bucket)
{
  const unsigned char* arr = blk->blob.data + BLOCK_HEADER_SIZE + 4 * ((size_t)ix->buckets_per_block + 1);
  return (int)be32(arr + 4 * ((size_t)ch * (size_t)blk->num_buckets + (size_t)bucket));
}

/* block_request CHAR|OCCS|LOCATION, src/main/index.c:1973-2144 */
int fo_block_request(fo_index_t* ix, int64_t block, int type, int row, int* ch_io, int* occs_out,
                     int64_t* offset_out, fo_counters_t* c)
{
  if (block < 0 || block >= ix->nblocks) return FO_ERR_PARAM;
  dblock_t* blk = &ix->blocks[block];
  if (type == 0 || type > 7) return FO_ERR_PARAM;
  if (row < 0 || row >= blk->size) return FO_ERR_PARAM;
  int bucket = row / ix->b_size, row_in_bucket = row % ix->b_size;
  bucket_t* e = &blk->buckets[bucket];
  const unsigned char* d = blk->blob.data;
  const unsigned char* wt = d + e->wtree_offset;
  int64_t occ = 0;
  int location_seq = -1, location_occ = 0;
  if (c) c->n_occ++;
  if (type & (1 | 4)) {
    int leaf, count;
    fo_wtree_rank(wt, row_in_bucket + 1, &leaf, &count, c);
    int seq = -1;
    for (int s = 0; s <= e->nInUse; s++) if (e->leaf[s] == leaf) { seq = s; break; } /* huff_perm decode, index.c:2051-2065 */
    if (seq < 0) return FO_ERR_BZ_DATA;
    location_seq = seq; location_occ = count;
    if (type & 1) {
      *ch_io = e->seqToUnseq[seq];
      if (type & 2) occ += count;
    }
  }
  if (type & 2) {
    int ch = *ch_io;
    if (!(type & 1)) {
      if (ch < 0 || ch >= FO_ALPHA_SIZE) return FO_ERR_PARAM;
      if (e->inUse[ch]) occ += fo_wtree_occs(wt, e->leaf[e->unseqToSeq[ch]], row_in_bucket + 1, c);
    }
    occ += get_bucket_occs(ix, blk, ch, bucket);
    *occs_out = (int)occ;
  }
  if (type & 4) {
    if (location_seq < 0 || location_seq > e->nInUse) return FO_ERR_INVALID;
    uint32_t table_off = be32(d + e->mark_tables_offset + 4 * (size_t)location_seq);
    if (table_off & 7) return FO_ERR_FORMAT;
    const unsigned char* table = d + e->mark_tables_offset + table_off;
    int occs[2], bit;
    fo_bseq_rank(table, location_occ, occs, &bit, c);
    if (bit) {
      uint32_t arr_off = be32(d + e->mark_arrays_offset + 4 * (size_t)location_seq);
      int64_t mark_offset = occs[1] - 1;
      bitrd_t r = { d + e->mark_arrays_offset + arr_off, (size_t)(ix->text_size_bits * mark_offset) };
      uint64_t v = 0;
      for (int i = 0; i < ix->text_size_bits; i++) v = (v << 1) | rd_bits(&r, 1);
      *offset_out = (int64_t)v;
      if (c) c->n_mark++;
    } else *offset_out = -1;
  }
  return FO_OK;
}

/* ------------------------------------------------------------------ L3/L4 */

/* Occ(ch,row) part that do_string_query assembles from a header request (HDR_BSEARCH_BLOCK_ROWS |
   HDR_REQUEST_C | HDR_REQUEST_BLOCK_OCCS | HDR_BACK, src/main/index.c:1698-1765) and a
   BLOCK_REQUEST_OCCS (src/main/server.c:847-897) */
static int c_plus_occ(fo_index_t* ix, int ch, int64_t row, int64_t* out, fo_counters_t* c)
{
  int64_t block = row / ix->block_size;               /* bsearch_block_rows, index.c:1613 */
  int64_t block_row = block * (int64_t)ix->block_size; /* get_block_row, index.c:1644 */
  int64_t hdr = fo_get_C(ix, ch) + fo_get_block_occs(ix, ch, block);
  int occ = 0, chv = ch;
  int rc = fo_block_request(ix, block, 2, (int)(row - block_row), &chv, &occ, NULL, c);
  if (rc) return rc;
  *out = hdr + occ;
  return FO_OK;
}

/* do_string_query, src/main/server.c:713-946 */
static int count_one(fo_index_t* ix, int plen, const uint16_t* pat, int64_t* first_out, int64_t* last_out, fo_counters_t* c)
{
  int64_t first, last;
  if (plen == 0) { *first_out = 0; *last_out = ix->total_length - 1; return FO_OK; } /* server.c:782-808 */
  int i = plen - 1;
  int ch = pat[i];
  if (ch >= FO_ALPHA_SIZE) return FO_ERR_PARAM;
  first = fo_get_C(ix, ch);
  last = fo_get_C(ix, ch + 1) - 1;
  while (!(first > last || i == 0)) {
    ch = pat[i - 1];
    if (ch >= FO_ALPHA_SIZE) return FO_ERR_PARAM;
    int64_t nf, nl;
    int rc;
    if (first == 0) nf = fo_get_C(ix, ch);           /* server.c:838-843, 884-888 */
    else { rc = c_plus_occ(ix, ch, first - 1, &nf, c); if (rc) return rc; }
    rc = c_plus_occ(ix, ch, last, &nl, c); if (rc) return rc;
    first = nf; last = nl - 1;
    i--;
  }
  *first_out = first; *last_out = last;
  return FO_OK;
}

/* do_back_query, src/main/server.c:2228-2359 */
int fo_back_step(fo_index_t* ix, int64_t row, int64_t* new_row, int* ch_out, int64_t* offset, fo_counters_t* c)
{
  int64_t block = row / ix->block_size;
  int64_t block_row = block * (int64_t)ix->block_size;
  int ch = 0x1ff, occ = 0;
  int rc = fo_block_request(ix, block, 1 | 2 | 4, (int)(row - block_row), &ch, &occ, offset, c);
  if (rc) return rc;
  int64_t r = (int64_t)occ - 1;
  r += fo_get_C(ix, ch) + fo_get_block_occs(ix, ch, block);
  if (ch <= FO_SEOF) r = -1;
  *new_row = r; *ch_out = ch;
  if (c) c->n_lf++;
  return FO_OK;
}

/* do_forward_query, src/main/server.c:2424-2565: one LF^-1 step.  Header part
   (HDR_BSEARCH_C | HDR_BSEARCH_BLOCK_OCCS | HDR_REQUEST_BLOCK_ROWS | HDR_FORWARD, index.c:1698-1765),
   then block_request_row (index.c:1915-1966) + LOCATION at the row found. */
int fo_forward_step(fo_index_t* ix, int64_t row, int64_t* new_row, int* ch_out, int64_t* offset, fo_counters_t* c)
{
  /* bsearch_C: largest ch with C[ch] <= row (bsearch_int64_ntoh_arr, src/utils/util.c:346) */
  int ch = -1;
  for (int k = 0; k < FO_ALPHA_SIZE; k++) if (fo_get_C(ix, k) <= row) ch = k; else break;
  *ch_out = ch;
  *new_row = -1;
  *offset = -1;
  if (ch <= FO_SEOF) return FO_OK;
  int64_t occs = row + 1 - fo_get_C(ix, ch);
  /* bsearch_block_occs: largest block with block_occs[ch][block] <= occs - 1 (index.c:1571) */
  int64_t block = -1;
  for (int64_t b = 0; b < ix->nblocks; b++) if (fo_get_block_occs(ix, ch, b) <= occs - 1) block = b; else break;
  if (block < 0) return FO_ERR_INVALID;
  int occs_in_block = (int)(occs - fo_get_block_occs(ix, ch, block));
  dblock_t* blk = &ix->blocks[block];
  /* bsearch_bucket_occs: largest bucket with bucket_occs < occs_in_block (index.c:1847) */
  int bucket = -1;
  for (int b = 0; b < blk->num_buckets; b++) if (get_bucket_occs(ix, blk, ch, b) <= occs_in_block - 1) bucket = b; else break;
  if (bucket < 0) return FO_ERR_INVALID;
  bucket_t* e = &blk->buckets[bucket];
  if (!e->inUse[ch]) return 8; /* ERR_MISSING */
  int count = occs_in_block - get_bucket_occs(ix, blk, ch, bucket);
  int index = fo_wtree_select(blk->blob.data + e->wtree_offset, e->leaf[e->unseqToSeq[ch]], count);
  int row_in_block = bucket * ix->b_size + index - 1;
  *new_row = block * (int64_t)ix->block_size + row_in_block;
  int chv = 0x1ff, occ = 0;
  return fo_block_request(ix, block, 4, row_in_block, &chv, &occ, offset, c);
}

/* do_context_query with LOCATE_STRONG, no context (src/main/server.c:2627-2795): the reference
   walks backward AND forward until either meets a marked row; both give SA[row], and because
   document offset 0 is always marked (should_mark, src/main/index_types.h:134-144) the backward
   walk alone always terminates on a mark before it would cross a document start. */
static int locate_row(fo_index_t* ix, int64_t row, int64_t* out, fo_counters_t* c)
{
  int64_t steps = 0;
  while (row >= 0) {
    int64_t off = -1, nr; int ch;
    int rc = fo_back_step(ix, row, &nr, &ch, &off, c);
    if (rc) return rc;
    if (off != -1) { *out = off + steps; return FO_OK; }
    row = nr; steps++;
  }
  *out = -1;
  return FO_OK;
}

typedef struct {
  fo_index_t* ix; int64_t lo, hi; const int32_t* plen; const uint16_t* pats; const int64_t* starts;
  int64_t* first; int64_t* last; fo_counters_t c; int rc;
  /* locate */
  const int64_t* out_starts; int64_t* offsets;
} job_t;

static void* count_worker(void* p)
{
  job_t* j = p;
  for (int64_t i = j->lo; i < j->hi; i++) {
    int rc = count_one(j->ix, j->plen[i], j->pats + j->starts[i], &j->first[i], &j->last[i], &j->c);
    if (rc) { j->rc = rc; break; }
  }
  return NULL;
}

static void add_counters(fo_counters_t* d, const fo_counters_t* s)
{
  d->n_rank += s->n_rank; d->n_occ += s->n_occ; d->n_mark += s->n_mark; d->s_bytes += s->s_bytes;
  d->n_rle += s->n_rle; d->n_lit += s->n_lit; d->n_gamma += s->n_gamma; d->n_lf += s->n_lf;
}

static int run_jobs(void* (*fn)(void*), job_t* jobs, int nt)
{
  pthread_t th[256];
  if (nt > 256) nt = 256;
  for (int t = 1; t < nt; t++) pthread_create(&th[t], NULL, fn, &jobs[t]);
  fn(&jobs[0]);
  for (int t = 1; t < nt; t++) pthread_join(th[t], NULL);
  for (int t = 0; t < nt; t++) if (jobs[t].rc) return jobs[t].rc;
  return FO_OK;
}

/* parallel_count, src/main/femto.c:275-329 */
int fo_count(fo_index_t* ix, int64_t npats, const int32_t* plen, const uint16_t* pats, const int64_t* starts,
             int64_t* first, int64_t* last, int nthreads, fo_counters_t* c)
{
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  job_t* jobs = calloc((size_t)nthreads, sizeof(job_t));
  for (int t = 0; t < nthreads; t++) {
    jobs[t].ix = ix; jobs[t].lo = npats * t / nthreads; jobs[t].hi = npats * (t + 1) / nthreads;
    jobs[t].plen = plen; jobs[t].pats = pats; jobs[t].starts = starts; jobs[t].first = first; jobs[t].last = last;
  }
  int rc = run_jobs(count_worker, jobs, nthreads);
  if (c) for (int t = 0; t < nthreads; t++) add_counters(c, &jobs[t].c);
  free(jobs);
  return rc;
}

static void* locate_worker(void* p)
{
  job_t* j = p;
  for (int64_t i = j->lo; i < j->hi; i++) {
    int64_t n = j->out_starts[i + 1] - j->out_starts[i];
    for (int64_t k = 0; k < n; k++) {
      int rc = locate_row(j->ix, j->first[i] + k, &j->offsets[j->out_starts[i] + k], &j->c);
      if (rc) { j->rc = rc; return NULL; }
    }
  }
  return NULL;
}

/* parallel_locate, src/main/femto.c:331-400 + do_locate_query clamp, src/main/server.c:4405-4415
   (note the reference's `last-first > max_occs` test: a range of exactly max_occs+1 rows is NOT clamped) */
int fo_locate(fo_index_t* ix, int64_t npats, const int32_t* plen, const uint16_t* pats, const int64_t* starts,
              int max_occs_each, int32_t* noccs, int64_t* offsets_flat, int nthreads, fo_counters_t* c)
{
  int64_t* first = malloc(8 * (size_t)(npats + 1));
  int64_t* last = malloc(8 * (size_t)(npats + 1));
  int64_t* out_starts = malloc(8 * (size_t)(npats + 1));
  int rc = fo_count(ix, npats, plen, pats, starts, first, last, nthreads, c);
  if (rc) goto done;
  out_starts[0] = 0;
  for (int64_t i = 0; i < npats; i++) {
    int64_t n;
    if (first[i] > last[i]) n = 0;
    else if (last[i] - first[i] > (int64_t)max_occs_each) n = max_occs_each;
    else n = last[i] - first[i] + 1;
    noccs[i] = (int32_t)n;
    out_starts[i + 1] = out_starts[i] + n;
  }
  if (offsets_flat) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t* jobs = calloc((size_t)nthreads, sizeof(job_t));
    for (int t = 0; t < nthreads; t++) {
      jobs[t].ix = ix; jobs[t].lo = npats * t / nthreads; jobs[t].hi = npats * (t + 1) / nthreads;
      jobs[t].first = first; jobs[t].out_starts = out_starts; jobs[t].offsets = offsets_flat;
    }
    rc = run_jobs(locate_worker, jobs, nthreads);
    if (c) for (int t = 0; t < nthreads; t++) add_counters(c, &jobs[t].c);
    free(jobs);
  }
done:
  free(first); free(last); free(out_starts);
  return rc;
}

/* ---- do_regexp_query (src/main/server.c:1656-2163) ---------------------------------------------------------------------- */
typedef struct {
  int64_t first, last;
  int32_t len, cost;
  uint8_t* states;          /* one error count per node (nfa_errcnt_t) */
  int64_t seq;
} rx_entry_t;

static int rx_result_cmp(const void* ap, const void* bp)   /* regexp_result_cmp, server.c:1485; ties: append order (stable qsort) */
{
  const rx_entry_t* a = ap; const rx_entry_t* b = bp;
  if (a->first < b->first) return -1;
  if (a->first > b->first) return 1;
  if (a->last < b->last) return 1;
  if (a->last > b->last) return -1;
  return a->seq < b->seq ? -1 : (a->seq > b->seq ? 1 : 0);
}

int64_t fo_nfa_search(fo_index_t* ix, int nn, const int32_t* tstart, const int32_t* tchar, const int32_t* tdest,
                      const uint8_t* is_start, const uint8_t* is_final, const int32_t* settings, int64_t max_iterations,
                      int64_t cap, int64_t* first_out, int64_t* last_out, int32_t* len_out, int32_t* cost_out)
{
  const int bound = settings[0], subst = settings[1], del = settings[2], ins = settings[3];
  rx_entry_t* stack = NULL; int64_t sp = 0, scap = 0;      /* queue_map: a stack ... */
  rx_entry_t* res = NULL; int64_t nres = 0, rcap = 0;
  uint8_t* cur = malloc(nn), *tmp = malloc(nn), *sub = malloc(nn), *child = malloc(nn);
  int64_t iters = 0, rc = 0, seq = 0;
#define RX_PUSH(arr, n, c, e) do { if (n == c) { c = c ? 2 * c : 64; arr = realloc(arr, c * sizeof(rx_entry_t)); } arr[n++] = (e); } while (0)
  {
    rx_entry_t e = { 0, ix->total_length - 1, 0, 0, malloc(nn), 0 };
    for (int i = 0; i < nn; i++) e.states[i] = is_start[i] ? 0 : 255;          /* approx_get_start_states, nfa.c:220 */
    RX_PUSH(stack, sp, scap, e);
  }
  for (;;) {
    if (iters > max_iterations) { rc = -FO_ERR_OVERWORKED; break; }            /* server.c:1821 */
    if (sp == 0) break;
    rx_entry_t var = stack[--sp];
    int fin = -1;
    for (int i = 0; i < nn; i++)                                               /* approx_is_final_state, nfa.c:205 */
      if (var.states[i] < bound && is_final[i]) { fin = i; break; }
    if (fin >= 0) {
      var.cost = var.states[fin];
      var.seq = seq++;
      free(var.states); var.states = NULL;
      RX_PUSH(res, nres, rcap, var);
      continue;
    }
    /* deletions (approx_get_reachable_states_allchars, nfa.c:234, with delete_cost), merged in */
    memset(tmp, 255, nn);
    if (bound > 1)
      for (int i = 0; i < nn; i++)
        if (var.states[i] + del < bound)
          for (int j = tstart[i]; j < tstart[i + 1]; j++)
            if (tchar[j] >= FO_CHARACTER_OFFSET && tchar[j] < FO_ALPHA_SIZE && var.states[i] + del < tmp[tdest[j]]) tmp[tdest[j]] = var.states[i] + del;
    for (int i = 0; i < nn; i++) cur[i] = var.states[i] < tmp[i] ? var.states[i] : tmp[i];
    free(var.states);
    uint8_t min_err = (uint8_t) bound;                                         /* nfa_errcnt_t arithmetic, server.c:1868-1873 */
    for (int i = 0; i < nn; i++) if (cur[i] < min_err) min_err = cur[i];
    min_err = (uint8_t) ((min_err + subst) < (min_err + ins) ? (min_err + subst) : (min_err + ins));
    uint8_t rc_set[FO_ALPHA_SIZE];                                             /* approx_get_reachable_characters, nfa.c:165 */
    memset(rc_set, 0, sizeof rc_set);
    for (int i = 0; i < nn; i++)
      if (cur[i] < bound)
        for (int j = tstart[i]; j < tstart[i + 1]; j++) rc_set[tchar[j]] = 1;
    if (min_err < bound && iters > 0)
      for (int c = FO_CHARACTER_OFFSET; c < FO_ALPHA_SIZE; c++) rc_set[c] = 1;
    int64_t nf[FO_ALPHA_SIZE], nl[FO_ALPHA_SIZE];
    for (int c = 0; c < FO_ALPHA_SIZE; c++) {
      if (!rc_set[c]) continue;
      int64_t a, b;                                                            /* server.c:2050-2056 */
      if (var.first == 0) a = fo_get_C(ix, c);
      else if (c_plus_occ(ix, c, var.first - 1, &a, NULL)) { rc = -FO_ERR_FORMAT; goto done; }
      if (c_plus_occ(ix, c, var.last, &b, NULL)) { rc = -FO_ERR_FORMAT; goto done; }
      nf[c] = a; nl[c] = b - 1;
    }
    memset(sub, 255, nn);                                                      /* substitutions, server.c:2107-2110 */
    if (bound > 1)
      for (int i = 0; i < nn; i++)
        if (cur[i] + subst < bound)
          for (int j = tstart[i]; j < tstart[i + 1]; j++)
            if (tchar[j] >= FO_CHARACTER_OFFSET && tchar[j] < FO_ALPHA_SIZE && cur[i] + subst < sub[tdest[j]]) sub[tdest[j]] = cur[i] + subst;
    /* the three add_mapping loops (server.c:2114-2141): pass 0 substitution states for characters >= CHARACTER_OFFSET,
       pass 1 the states after reading ch, pass 2 the insertion states */
    for (int pass = 0; pass < 3; pass++)
      for (int c = pass == 0 ? FO_CHARACTER_OFFSET : 0; c < FO_ALPHA_SIZE; c++) {
        if (!rc_set[c] || nl[c] < nf[c]) continue;                             /* add_mapping ignores empty ranges */
        if (pass == 0) memcpy(child, sub, nn);
        else if (pass == 1) {                                                  /* approx_get_reachable_states, nfa.c:273 */
          memset(child, 255, nn);
          for (int i = 0; i < nn; i++)
            if (cur[i] < bound)
              for (int j = tstart[i]; j < tstart[i + 1]; j++)
                if (tchar[j] == c && cur[i] < child[tdest[j]]) child[tdest[j]] = cur[i];
        } else {                                                               /* approx_add_error_allchars, nfa.c:305 */
          for (int i = 0; i < nn; i++) child[i] = (uint8_t) (cur[i] + ins < 255 ? cur[i] + ins : 255);
        }
        int64_t k;
        for (k = 0; k < sp; k++) if (stack[k].first == nf[c] && stack[k].last == nl[c]) break;
        if (k < sp) {                                                          /* found: union the states, keep the longer match */
          for (int i = 0; i < nn; i++) if (child[i] < stack[k].states[i]) stack[k].states[i] = child[i];
          if (var.len + 1 > stack[k].len) stack[k].len = var.len + 1;
        } else {
          rx_entry_t e = { nf[c], nl[c], var.len + 1, 0, malloc(nn), 0 };
          memcpy(e.states, child, nn);
          RX_PUSH(stack, sp, scap, e);
        }
      }
    iters++;
  }
done:
  for (int64_t k = 0; k < sp; k++) free(stack[k].states);
  if (rc == 0) {                                                               /* regexp_result_list_sort, server.c:1528-1573 */
    qsort(res, nres, sizeof(rx_entry_t), rx_result_cmp);
    int64_t n = 0;
    for (int64_t k = 0; k < nres; k++) {
      if (n && res[k].first == res[n - 1].first && res[k].last == res[n - 1].last) continue;
      res[n++] = res[k];
    }
    nres = n;
    if (nres > 0) {
      int64_t f = res[0].first, l = res[0].last, i = 1;
      for (int64_t k = 1; k < nres; k++) {
        if (res[k].first >= f && res[k].last <= l) continue;
        f = res[k].first; l = res[k].last;
        res[i++] = res[k];
      }
      nres = i;
    }
    for (int64_t k = 0; k < nres && k < cap; k++) {
      first_out[k] = res[k].first; last_out[k] = res[k].last; len_out[k] = res[k].len; cost_out[k] = res[k].cost;
    }
    rc = nres;
  }
  free(stack); free(res); free(cur); free(tmp); free(sub); free(child);
  return rc;
#undef RX_PUSH
}
