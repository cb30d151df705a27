/* A stand-in for librccl.so that works on HOST pointers, for tests only (tests/test_dist.py): the four entry points
 * libloromerge.so resolves with dlsym (lm_capi_impl.h lmcomm::api) — ncclGetUniqueId, ncclCommInitRank, ncclAllGather,
 * ncclCommDestroy — with the real library's signatures.  Ranks are processes of one machine; a collective is a rendezvous through
 * files in a directory named after the unique id (every rank writes its contribution under a temporary name, renames it, and reads
 * the others').  The kernel-logic harness (tests/emu: "device" memory is host memory) loads it through LM_RCCL_LIB, so
 * lm_comm_init / lm_summary_allgather / lm_summary_allgather_device run end to end between two processes without a GPU.
 * Never linked into, nor looked for by, the product library unless LM_RCCL_LIB names it. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sys/stat.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct { int rank, world; unsigned long seq; char dir[200]; } stub_comm;

static size_t dtype_size(int dt) {
  switch (dt) { case 0: case 1: return 1; case 2: case 3: return 4; case 4: case 5: return 8; case 6: return 2; case 7: return 4; case 8: return 8; default: return 0; }
}

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id->internal, 0, sizeof id->internal);
  const char* base = getenv("LM_RCCL_STUB_DIR");
  snprintf(id->internal, sizeof id->internal, "%s/lmstub-%ld-%ld", base ? base : "/tmp", (long)getpid(), (long)time(NULL));
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || rank < 0 || rank >= nranks || id.internal[sizeof id.internal - 1] != 0) return 1;
  stub_comm* c = (stub_comm*)calloc(1, sizeof(stub_comm));
  if (!c) return 1;
  c->rank = rank; c->world = nranks; c->seq = 0;
  snprintf(c->dir, sizeof c->dir, "%s", id.internal);
  mkdir(c->dir, 0700);   /* (every rank tries; EEXIST is fine) */
  *comm = c;
  return 0;
}

static int read_all(const char* path, void* dst, size_t n) {
  for (int tries = 0; tries < 60000; tries++) {   /* up to ~60 s */
    FILE* f = fopen(path, "rb");
    if (f) {
      size_t got = fread(dst, 1, n, f);
      fclose(f);
      if (got == n) return 0;
    }
    usleep(1000);
  }
  return 1;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, void* stream) {
  (void)stream;
  stub_comm* c = (stub_comm*)comm;
  size_t es = dtype_size(dtype);
  if (!c || !es) return 1;
  size_t n = count * es;
  char tmp[300], path[300];
  snprintf(tmp, sizeof tmp, "%s/s%lu_r%d.tmp", c->dir, c->seq, c->rank);
  snprintf(path, sizeof path, "%s/s%lu_r%d.bin", c->dir, c->seq, c->rank);
  FILE* f = fopen(tmp, "wb");
  if (!f) return 1;
  if (n && fwrite(send, 1, n, f) != n) { fclose(f); return 1; }
  fclose(f);
  if (rename(tmp, path) != 0) return 1;
  for (int r = 0; r < c->world; r++) {
    snprintf(path, sizeof path, "%s/s%lu_r%d.bin", c->dir, c->seq, r);
    if (r == c->rank) { memcpy((char*)recv + (size_t)r * n, send, n); continue; }
    if (n == 0) continue;
    if (read_all(path, (char*)recv + (size_t)r * n, n) != 0) return 1;
  }
  c->seq++;
  return 0;
}

int ncclCommDestroy(void* comm) { free(comm); return 0; }
