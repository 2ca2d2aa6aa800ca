/* femto_amd_shim.c -- routes femto's batch count/locate (src/main/femto.c:275,331,402,481) to the MI355X engine.
 *
 * This is the file a femto maintainer adds to src/main/.  The reference's callers (index_test.c:356-410,
 * query_tool.c:133-206, ...) keep calling parallel_count / parallel_locate / serial_locate / parallel_locate_range with the signatures
 * of src/main/femto_internal.h:63-75; the bodies below forward to the C ABI of include/femto_amd.h.  The reference's own
 * CPU implementations stay linked under the names femto_cpu_* (femto.c compiled with
 * -Dparallel_count=femto_cpu_parallel_count ... -- see oracle/Makefile; in the reference tree that is an
 * #ifdef USE_FEMTO_AMD around the three definitions).
 *
 * Build: gcc -std=gnu99 -DUSE_FEMTO_AMD -I<femto>/src/main -I<femto>/src/utils -I<femto_amd>/include -c femto_amd_shim.c
 * Link : ... -lfemto_amd          (plain C ABI; no C++ / HIP headers are needed on this side)
 *
 * tests/test_integration.py compiles this file against /root/reference (CPU) and runs the reference's own callers
 * (oracle/ref_tool.c count/locate and the reference's index_test.c) linked with it on the GPU.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "server.h"            /* shared_server_state_t: path_to_id (src/main/server.h:641-651) */
#include "femto_internal.h"    /* femto_server_t, index_locator_t, alpha_t, error_t, the three prototypes */
#include "femto_amd.h"

#ifdef USE_FEMTO_AMD

/* One GPU-resident index per (server, index_locator_t id).  path_translator ids are small integers that start
 * at 1 and are handed out per server (src/main/block_storage.c:104-155), so the table is indexed by id and is
 * emptied when the server stops. */
typedef struct {
  struct shared_server_state* owner;
  femto_amd_index_t* ix;
} shim_slot_t;

static pthread_mutex_t shim_lock = PTHREAD_MUTEX_INITIALIZER;
static shim_slot_t* shim_slots = NULL;
static intptr_t shim_nslots = 0;

static error_t amd_err(int code)   /* err_code_t values are shared (src/utils/error.h:25-39) */
{
  if( code == FEMTO_AMD_OK ) return ERR_NOERR;
  return ERR_MAKE_STR((err_code_t) code, femto_amd_last_error());
}

/* FEMTO_AMD_DEVICES="0,1,2,3": the index is opened on every listed GPU and each batch is sharded over them
 * (femto_amd_open_multi); one entry, or FEMTO_AMD_DEVICE=n, or nothing (device 0): a single GPU. */
static int shim_open(const char* path, femto_amd_index_t** out)
{
  int devs[64];
  int n = 0;
  const char* list = getenv("FEMTO_AMD_DEVICES");
  if( list ) {
    const char* p = list;
    while( *p && n < 64 ) {
      char* end = NULL;
      long v = strtol(p, &end, 10);
      if( end == p ) break;
      devs[n++] = (int) v;
      p = end;
      while( *p == ',' || *p == ' ' ) p++;
    }
  }
  if( n == 0 ) {
    const char* d = getenv("FEMTO_AMD_DEVICE");
    devs[n++] = d ? atoi(d) : 0;
  }
  if( n == 1 ) return femto_amd_open(path, devs[0], out);
  return femto_amd_open_multi(path, n, devs, out);
}

/* id -> path.  block_storage.h:62 declares path_translator_path_for_id() but block_storage.c never defines it (only the
 * static path_translator_path_for_id_unlocked, block_storage.c:237, which its block readers call under the read lock),
 * so the shim does the same lookup under the same lock and copies the string. */
static error_t shim_path_for_id(path_translator_t* t, index_locator_t id, char** out)
{
  error_t err = ERR_NOERR;
  int rc = pthread_rwlock_rdlock(&t->rwlock);
  if( rc ) return ERR_MAKE_STR(ERR_CODE_INVALID, "path translator lock");
  if( id.id <= 0 || id.id >= t->next_id || ! t->id_to_path[id.id].path ) err = ERR_INVALID;
  else {
    *out = strdup(t->id_to_path[id.id].path);
    if( ! *out ) err = ERR_MEM;
  }
  pthread_rwlock_unlock(&t->rwlock);
  return err;
}

static error_t amd_get(femto_server_t* srv, index_locator_t loc, femto_amd_index_t** out)
{
  error_t err = ERR_NOERR;
  char* path = NULL;

  if( ! srv || ! srv->state ) return ERR_PARAM;
  if( ! index_locator_is_valid(loc) ) return ERR_PARAM;

  pthread_mutex_lock(&shim_lock);
  if( loc.id >= shim_nslots ) {
    intptr_t n = shim_nslots ? shim_nslots : 16;
    shim_slot_t* grown;
    while( n <= loc.id ) n *= 2;
    grown = realloc(shim_slots, n * sizeof(shim_slot_t));
    if( ! grown ) { err = ERR_MEM; goto done; }
    memset(grown + shim_nslots, 0, (n - shim_nslots) * sizeof(shim_slot_t));
    shim_slots = grown;
    shim_nslots = n;
  }
  if( shim_slots[loc.id].ix && shim_slots[loc.id].owner != srv->state ) {
    /* an id of a server that is gone */
    femto_amd_close(shim_slots[loc.id].ix);
    shim_slots[loc.id].ix = NULL;
  }
  if( ! shim_slots[loc.id].ix ) {
    int rc;
    /* the path given to femto_loc_for_path_err (src/main/femto.c:269); directory or flattened file */
    err = shim_path_for_id(&srv->state->path_to_id, loc, &path);
    if( err ) goto done;
    rc = shim_open(path, &shim_slots[loc.id].ix);
    free(path);
    if( rc ) { shim_slots[loc.id].ix = NULL; err = amd_err(rc); goto done; }
    shim_slots[loc.id].owner = srv->state;
  }
  *out = shim_slots[loc.id].ix;
done:
  pthread_mutex_unlock(&shim_lock);
  return err;
}

/* closes the GPU-resident indexes of a server; femto_stop_server (src/main/femto.c:60) calls this first */
void femto_amd_shim_forget(femto_server_t* srv)
{
  intptr_t i;
  pthread_mutex_lock(&shim_lock);
  for( i = 0; i < shim_nslots; i++ ) {
    if( shim_slots[i].ix && ( ! srv || shim_slots[i].owner == srv->state ) ) {
      femto_amd_close(shim_slots[i].ix);
      shim_slots[i].ix = NULL;
      shim_slots[i].owner = NULL;
    }
  }
  pthread_mutex_unlock(&shim_lock);
}

/* src/main/femto.c:275 -- last==NULL keeps femto's "first[i] = count" form (femto.c:313-318) */
error_t parallel_count(femto_server_t* srv, index_locator_t loc, int npats, int* plen, alpha_t** pats,
                       int64_t* first, int64_t* last)
{
  femto_amd_index_t* ix = NULL;
  error_t err = amd_get(srv, loc, &ix);
  if( err ) return err;
  /* alpha_t is uint16_t (src/main/index_types.h:69) */
  return amd_err(femto_amd_parallel_count(ix, npats, plen, (const uint16_t* const*) pats, first, last));
}

/* src/main/femto.c:331 -- offsets[i] is malloc()ed by the callee and free()d by the caller, NULL when noccs[i]==0
 * (femto.c:372-386) */
error_t parallel_locate(femto_server_t* srv, index_locator_t loc,
                        int npats, int* plen, alpha_t** pats,
                        int max_occs_each,
                        int* noccs, int64_t** offsets)
{
  femto_amd_index_t* ix = NULL;
  error_t err = amd_get(srv, loc, &ix);
  if( err ) return err;
  return amd_err(femto_amd_parallel_locate(ix, npats, plen, (const uint16_t* const*) pats, max_occs_each, noccs, offsets));
}

/* src/main/femto.c:402 -- parallel_locate's test twin (query_tool.c:157): one pattern at a time, and its OWN clamp,
 * "1+last-first >= max_occs_each" (femto.c:427-428; parallel_locate's is "last-first > max_occs", server.c:4411), so the
 * two differ when a pattern has exactly max_occs_each + 1 rows.  Reproduced as written: the range from the GPU's count,
 * the offsets of its first noccs rows from the GPU's range locate. */
error_t serial_locate(femto_server_t* srv, index_locator_t loc,
                      int npats, int* plen, alpha_t** pats,
                      int max_occs_each,
                      int* noccs, int64_t** offsets)
{
  femto_amd_index_t* ix = NULL;
  error_t err;
  int i;
  if( ! srv ) return ERR_PARAM;
  err = amd_get(srv, loc, &ix);
  if( err ) return err;
  for( i = 0; i < npats; i++ ) {
    int64_t first = -1, last = -1;
    int rc = femto_amd_parallel_count(ix, 1, &plen[i], (const uint16_t* const*) &pats[i], &first, &last);
    if( rc ) return amd_err(rc);
    if( 1+last-first >= max_occs_each ) noccs[i] = max_occs_each;
    else noccs[i] = (int) (1+last-first);
    offsets[i] = NULL;
    if( noccs[i] > 0 ) {
      offsets[i] = malloc(sizeof(int64_t) * noccs[i]);
      if( ! offsets[i] ) return ERR_MEM;
      rc = femto_amd_parallel_locate_range(ix, first, first + noccs[i] - 1, offsets[i]);
      if( rc ) return amd_err(rc);
    }
  }
  return ERR_NOERR;
}

/* src/main/femto.c:481 */
error_t parallel_locate_range(femto_server_t* srv, index_locator_t loc,
                              int64_t first, int64_t last,
                              int64_t* offsets)
{
  femto_amd_index_t* ix = NULL;
  error_t err = amd_get(srv, loc, &ix);
  if( err ) return err;
  return amd_err(femto_amd_parallel_locate_range(ix, first, last, offsets));
}

/* src/main/femto.c:60: the GPU-resident indexes go with the server */
void femto_cpu_stop_server(femto_server_t* srv);
void femto_stop_server(femto_server_t* srv)
{
  femto_amd_shim_forget(srv);
  femto_cpu_stop_server(srv);
}

#endif /* USE_FEMTO_AMD */
