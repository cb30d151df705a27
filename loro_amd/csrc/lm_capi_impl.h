// C-ABI glue shared by the product library (lm_hip.cpp) and the kernel-logic test harness (tests/emu).
// LM_API(name) expands to the exported symbol name.
#pragma once
#include "lm_pipeline.h"

struct lm_ctx_impl {
  lm::Engine eng;
  std::string err;
};

extern "C" {

typedef struct lm_doc_in_c { const uint8_t* const* blobs; const size_t* blob_lens; size_t n_blobs; } lm_doc_in_c;
typedef struct lm_doc_out_c { int32_t status; const uint8_t* json; size_t json_len; const uint8_t* vv; size_t vv_len; uint64_t pending_ops; } lm_doc_out_c;
typedef struct lm_run_stats_c { uint64_t n_docs, n_blobs, in_bytes, out_bytes, device_bytes_allocated; uint32_t n_kernels; } lm_run_stats_c;

void* LM_API(create)(int device) {
  if (!lmbe::init(device)) return nullptr;
  return new lm_ctx_impl();
}
void LM_API(destroy)(void* c) { delete (lm_ctx_impl*)c; }
const char* LM_API(last_error)(void* c) { return c ? ((lm_ctx_impl*)c)->err.c_str() : "no context (HIP device unavailable)"; }

int LM_API(stage)(void* c, const lm_doc_in_c* docs, size_t n) {
  auto* x = (lm_ctx_impl*)c;
  try {
    std::vector<lm::Engine::DocIn> v(n);
    for (size_t i = 0; i < n; i++) v[i] = lm::Engine::DocIn{docs[i].blobs, docs[i].blob_lens, docs[i].n_blobs};
    x->eng.stage(v.data(), n);
    return 0;
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
int LM_API(run)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  try { x->eng.run(); return 0; } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
int LM_API(fetch)(void* c, lm_doc_out_c* outs) {
  auto* x = (lm_ctx_impl*)c;
  try {
    x->eng.fetch();
    for (uint32_t i = 0; i < x->eng.n_docs; i++) {
      const lm::DocResult& r = x->eng.results[i];
      outs[i].status = r.status;
      outs[i].json = x->eng.h_out.data() + r.json_off; outs[i].json_len = (size_t)r.json_len;
      outs[i].vv = x->eng.h_vv.data() + r.vv_off; outs[i].vv_len = (size_t)r.vv_len;
      outs[i].pending_ops = r.pending;
    }
    return 0;
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
int LM_API(merge_batch)(void* c, const lm_doc_in_c* docs, size_t n, lm_doc_out_c* outs) {
  int rc = LM_API(stage)(c, docs, n);
  if (rc) return rc;
  rc = LM_API(run)(c);
  if (rc) return rc;
  return LM_API(fetch)(c, outs);
}
// per-document result metadata of the last lm_run without copying the rendered bytes back
int LM_API(result_meta)(void* c, int32_t* status, uint64_t* json_len, uint64_t* vv_len, uint64_t* pending) {
  auto* x = (lm_ctx_impl*)c;
  if (!x->eng.ran) { x->err = "lm_result_meta before lm_run"; return -1; }
  for (uint32_t i = 0; i < x->eng.n_docs; i++) {
    const lm::DocResult& r = x->eng.results[i];
    if (status) status[i] = r.status;
    if (json_len) json_len[i] = r.json_len;
    if (vv_len) vv_len[i] = r.vv_len;
    if (pending) pending[i] = r.pending;
  }
  return 0;
}
int LM_API(get_stats)(void* c, lm_run_stats_c* s) {
  auto* x = (lm_ctx_impl*)c;
  s->n_docs = x->eng.n_docs; s->n_blobs = x->eng.n_blobs; s->in_bytes = x->eng.in_bytes;
  s->out_bytes = x->eng.payload_bytes;
  s->device_bytes_allocated = lmbe::allocated_bytes();
  s->n_kernels = (uint32_t)x->eng.times.size();
  return 0;
}
// sizing diagnostics of the last run: max over documents of (leaves used, leaf capacity, elements)
int LM_API(sizing)(void* c, uint32_t* out3) {  // out3[3] = documents re-run with the worst-case directory
  auto* x = (lm_ctx_impl*)c;
  out3[0] = out3[1] = out3[2] = 0;
  out3[3] = x->eng.last_retries;
  for (auto& m : x->eng.h_doc) { if (m.pad0 > out3[0]) out3[0] = m.pad0; if (m.leaf_cap > out3[1]) out3[1] = m.leaf_cap; if (m.n_elems > out3[2]) out3[2] = m.n_elems; }
  return 0;
}
// returns the number of mismatches of the wave-primitive self test (0 = ok)
int LM_API(selftest)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  try {
    return x->eng.selftest();
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
// LM_PROF builds only: sum over documents of the 16 cycle-accounting slots of k_integrate
int LM_API(prof_sum)(void* c, uint64_t* out16) {
  auto* x = (lm_ctx_impl*)c;
  for (int i = 0; i < 16; i++) out16[i] = 0;
  for (size_t k = 0; k < x->eng.h_prof.size(); k++) out16[k % 16] += x->eng.h_prof[k];
  return x->eng.h_prof.empty() ? -1 : 0;
}
int LM_API(set_profiling)(void* c, int en) { ((lm_ctx_impl*)c)->eng.profiling = en != 0; return 0; }
int LM_API(kernel_time)(void* c, uint32_t i, const char** name, double* ms) {
  auto* x = (lm_ctx_impl*)c;
  if (i >= x->eng.times.size()) return -1;
  *name = x->eng.times[i].name.c_str();
  *ms = x->eng.times[i].ms;
  return 0;
}
}
