// Host orchestration of the batched merge: staging, table sizing, kernel sequence, result fetch.
// Mirrors the reference's import flow at batch granularity (crates/loro-internal/src/loro.rs:568-649,
// 720-851: parse → decode changes → apply to oplog → diff_calc → state) with every stage on the device.
//
// The including translation unit provides the backend:
//   lmbe::dalloc/dfree/dmemset/h2d/d2h/sync/halloc/hfree, lmbe::tic()/toc(name) and the macro
//   LM_LAUNCH(kernel, grid, block, args...).
#pragma once
#include <algorithm>
#include <map>
#include <mutex>
#include <thread>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <stdexcept>
#include "lm_k_scan.h"
#include "lm_k_emit.h"
#include "lm_k_lww_doc.h"
#include "lm_k_map_fused.h"
#include "lm_k_fuse.h"
#include "lm_k_lca.h"
#include "lm_k_richtext.h"
#include "lm_snapshot.h"
#include "lm_export.h"
#include "lm_snapshot_base.h"

namespace lm {

struct DBuf {  // grow-only device buffer
  void* p = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    if (p) lmbe::dfree(p);
    size_t want = n + n / 8 + 256;
    p = lmbe::dalloc(want);
    if (!p) throw std::runtime_error("device allocation failed");
    cap = want;
  }
  // grow and keep the first `keep` bytes (resident documents: arenas that outlive a run); stream-ordered copy
  void ensure_keep(size_t n, size_t keep) {
    if (n <= cap) return;
    size_t want = n + n / 2 + 256;
    void* q = lmbe::dalloc(want);
    if (!q) throw std::runtime_error("device allocation failed");
    if (p && keep) { lmbe::d2d(q, p, keep < cap ? keep : cap); lmbe::sync(); }
    if (p) lmbe::dfree(p);
    p = q;
    cap = want;
  }
  void release() { if (p) lmbe::dfree(p); p = nullptr; cap = 0; }
  template <class T> T* as() { return (T*)p; }
};

struct DocResult {
  int32_t status;
  uint64_t json_off, json_len, vv_off, vv_len, pending;
  uint64_t json_xxh64;   // xxh64 (seed 0) of the JSON bytes, computed on the device (0 for failed documents)
};

struct KernelTime { std::string name; double ms; };

// ---- pinned host regions handed out by lm_host_alloc (include/loro_merge.h): blobs that already live in one of them are copied to
// the device straight from where they are (Engine::stage "direct"), without the gather into the engine's own pinned staging buffer
struct HostRegions {
  std::mutex mu;
  std::map<const uint8_t*, size_t> r;     // base -> bytes
  void add(const void* p, size_t n) { std::lock_guard<std::mutex> g(mu); r[(const uint8_t*)p] = n; }
  bool remove(const void* p) { std::lock_guard<std::mutex> g(mu); return r.erase((const uint8_t*)p) != 0; }
  // the region that holds [p, p + n), or nullptr
  const uint8_t* find(const uint8_t* p, size_t n, size_t* bytes) {
    std::lock_guard<std::mutex> g(mu);
    auto it = r.upper_bound(p);
    if (it == r.begin()) return nullptr;
    --it;
    if (p < it->first || (size_t)(p - it->first) + n > it->second) return nullptr;
    if (bytes) *bytes = it->second;
    return it->first;
  }
};
inline HostRegions& host_regions() { static HostRegions h; return h; }

struct Engine {
  // staged input
  uint32_t n_docs = 0, n_blobs = 0;
  uint64_t data_bytes = 0, in_bytes = 0;
  std::vector<uint64_t> h_blob_off, h_front_off, h_froot_off;
  std::vector<uint8_t> h_front_bytes;             // the staged checkout frontiers, back to back (h_front_off)
  std::vector<uint8_t> h_froot;                   // per document: the root containers its first snapshot's state section holds (h_froot_off)
  std::vector<uint32_t> h_blob_len, h_doc_blob, h_blob_doc;
  DBuf b_data, b_blob_off, b_blob_len, b_doc_blob, b_blob_doc, b_front, b_front_off, b_froot, b_froot_off, b_blob_hash, b_big;
  // work buffers
  DBuf b_blob_status, b_blob_nblk, b_blob_blk0, b_tile, b_tot;
  DBuf b_blk, b_bcnt, b_boff, b_blk_kind, b_doc_fused, b_mf_docs, b_mf_key0;
  std::vector<uint8_t> h_fused;                   // per document: decoded by k_map_fused (lm_k_map_fused.h)
  uint32_t n_fused = 0;
  DBuf b_chg, b_dep_peer, b_dep_ctr, b_dep_ci, b_op, b_op_val, b_op_blk, b_key_off, b_key_len, b_cid_raw, b_cid_map, b_peer_raw, b_peer_map;
  DBuf b_doc, b_peer_uniq, b_peer_end, b_peer_ext, b_peer_base, b_peer_end_all, b_elem_base, b_peer_chg0, b_peer_chg1, b_cont;
  DBuf b_chg_mask, b_chg_sorted, b_chg_lamport, b_chg_skip, b_chg_flag, b_node_first, b_node_last, b_node_order, b_vvh;
  DBuf b_blk_sorted, b_chg_node, b_node_done, b_node_lam;
  DBuf b_cp, b_loc, b_tb, b_fuse, b_dcnt, b_posdel;
  DBuf b_it, b_dir_out, b_lf_chunk;
  DBuf b_cont_root0, b_cont_nroot;
  DBuf b_ht_key, b_ht_pfx, b_ht_best, b_ht0, b_ht_cap, b_ht_list, b_ht_cnt;
  DBuf b_out, b_out_off, b_vv_out, b_vv_off, b_hash, b_order, b_prof, b_slab, b_vslab, b_slab_off, b_vslab_off, b_slab2, b_slab2_off;
  uint64_t payload_bytes = 0;   // Σ json_len + Σ vv_len of the last run (without alignment padding)
  std::vector<uint64_t> h_prof, h_hash, h_ht0;
  std::vector<uint32_t> h_ht_cap, h_ht_full;   // LWW tables of the run: first slot, slots, slots of a table sized for the document's Map rows
  // results
  std::vector<DocMeta> h_doc;
  std::vector<DocResult> results;
  std::vector<uint8_t> h_out, h_vv;
  std::vector<uint64_t> h_out_off, h_vv_off;
  uint64_t out_bytes = 0, vv_bytes = 0;
  bool ran = false, fetched = false;
  // lm_richtext (lm_k_richtext.h): the tables of the last run, the rendered richtext values
  Dev last_d;
  DBuf b_rt_out, b_rt_off, b_rt_len;
  std::vector<uint8_t> h_rt;
  std::vector<uint64_t> h_rt_off;
  std::vector<uint32_t> h_rt_len;
  std::vector<int32_t> h_rt_status;
  bool rt_ran = false;
  int rt_launches = 0;                            // k_richtext launches of the last lm_richtext (2: a slab was too small)
  std::vector<KernelTime> times;
  bool profiling = false;
  std::string last_error;
  uint64_t device_bytes = 0;
  uint32_t last_retries = 0, last_reemits = 0, last_posdel = 0;
  lmbe::StreamCtx* sc = nullptr;   // this engine's HIP stream + timing events
  long long* sum_rows = nullptr;   // lm_summary_layout: where this engine's documents' summary rows go (device), first id, stride
  long long sum_id0 = 0, sum_stride = 1;

  // ---- resident documents (lm_import; SURVEY §8f N2 — diff_calc.rs:62-68 DiffCalculatorRetainMode::Persist, loro.rs:568-649
  // import on a document that already holds history).  The blobs stay in `b_data`, used as an append-only arena; every
  // document keeps the list of its blobs, and the trackers of its sequence containers stay in HBM between runs (leaf pool,
  // both directory words — double-buffered, a run reads the previous run's and writes its own — and one record per document
  // in `b_tk`, lm_k_integrate_span.h).  A run after lm_import decodes the document's blobs again (old and new: the tables are
  // laid out per batch) but integrates only the changes the trackers have not applied; a run that only changes the rendered
  // versions reuses the tables as well.
  struct BlobRef { uint64_t off; uint32_t len; };
  bool resident = false;
  uint32_t shared_mode = 0;                       // lm_capi_impl.h "shared replay": 1 = the run that imports, 2 = a run that renders one checkout of it (k_res_exists)
  std::vector<std::vector<BlobRef>> r_blobs;      // per document: its blobs in import order (arena offsets)
  std::vector<uint32_t> r_step;                   // per document: blobs appended since the last run (dropped again if that run fails for it)
  std::vector<std::vector<uint8_t>> r_front;      // per document: checkout frontiers of the next run (empty = latest)
  uint64_t arena_top = 0;
  bool tables_valid = false;                      // the decoded tables belong to the current blob lists
  std::vector<uint32_t> tk_leaf0, tk_leaf_cap, tk_pcap, tk_ccap, tk_elem_cap, h_old_blobs;
  std::vector<uint64_t> tk_off, tk_elem0;
  uint64_t elem_top = 0;                          // element slots handed out in the resident element arena (cp[] / loc[])
  DBuf b_elem_cap, b_old_blobs, b_prev_doc, b_prev_uniq, b_prev_end, b_lca_out, b_lca_scratch, b_lca_off;
  bool have_prev = false;                         // the tables of a previous resident run exist (what k_import_lca measures the import against)
  std::vector<uint32_t> h_lca;                    // per document LCA_OUT words of the last run: DiffMode + common ancestors of its import
  std::vector<uint8_t> tk_reset;
  void upload_res() {   // the per-document records of this run (tracker record, capacities, reset flag)
    std::vector<ResDoc> hres(n_docs);
    for (uint32_t i = 0; i < n_docs; i++) hres[i] = ResDoc{tk_off[i], tk_pcap[i], tk_ccap[i], tk_reset[i], tk_elem_cap[i]};
    b_res.ensure((size_t)n_docs * sizeof(ResDoc));
    lmbe::h2d(b_res.p, hres.data(), (size_t)n_docs * sizeof(ResDoc));
  }
  uint64_t tk_top = 0;                            // words used in b_tk
  uint32_t leaf_top = 0;                          // leaves handed out in the resident pool
  int dir_parity = 0;                             // which directory buffers the next run writes
  DBuf b_tk, b_res, b_dir_out2, b_dir_b, b_dir_b2, b_doc_saved;
  uint32_t last_fresh = 0;
  // what a run that reuses the tables needs from the run that built them
  struct Saved {
    Dev d; DevDag g;
    uint32_t NB = 0, NC = 0, NO = 0, NCID = 0, NP = 0, dir_cap = 0, dir_opt = 0, pmax = 0;
    uint64_t ht = 0;
    bool any_ml = false, any_common = false;
    std::vector<DocMeta> h_doc;
  } sv;

  // environment knobs (A/B measurements and tests): read once per staged batch / import, not inside the run
  struct Knobs { bool span = true, decode_wave = true, no_opt_dir = false, loc_memset = true; int plain = 2; uint32_t dec_slot = 1264, dec_slot_big = 4096, dec_big_mode = 1, dir_opt_max = 0; size_t lds_pad = 0; long long slab_cap = -1; uint32_t ht_opt = 2048, cut_min_rows = 2048, vs_row_cost = 300; bool lww_lds = true, fuse_rows = true, version_sweep = true, span_auto = true, linear = true, posdel = true, reclass = true, redo = true, map_fused = true, snapshot_state = true, snapshot_state_force = false, stage_direct = true; uint32_t mf_min_rows = 2048, mf_chg_ratio = 4, pd_state_pieces = 2; } kn;
  void read_knobs() {
    Knobs k;
    if (const char* e = getenv("LM_SPAN")) k.span = atoi(e) != 0;
    if (const char* e = getenv("LM_PLAIN")) k.plain = atoi(e);
    if (const char* e = getenv("LM_DECODE")) k.decode_wave = atoi(e) != 0;
    if (const char* e = getenv("LM_RECLASS")) k.reclass = atoi(e) != 0;
    if (const char* e = getenv("LM_DEC_SLOT")) k.dec_slot = ((uint32_t)atoi(e) + 15u) & ~15u;   // LDS bytes per block for everything before its value payloads (per 5k configs[1] documents: 512 5.4 ms, 1024 4.5, 1280 4.2, 1536 4.9, 2048 5.7 — occupancy against the share of heads that fit; larger heads are read from HBM)
    if (const char* e = getenv("LM_DEC_SLOT_BIG")) k.dec_slot_big = ((uint32_t)atoi(e) + 15u) & ~15u;   // slot of the second decoder launch (groups with a head beyond LM_DEC_SLOT); 0: one launch, as in rounds 2-3
    if (const char* e = getenv("LM_DEC_BIG_MODE")) k.dec_big_mode = (uint32_t)atoi(e);
    if (const char* e = getenv("LM_LDS_PAD")) k.lds_pad = (size_t)atoi(e);                       // occupancy experiments only
    k.no_opt_dir = getenv("LM_NO_OPT_DIR") != nullptr;
    if (const char* e = getenv("LM_LOC_MEMSET")) k.loc_memset = atoi(e) != 0;
    if (const char* e = getenv("LM_DIR_OPT_MAX")) k.dir_opt_max = (uint32_t)atoi(e);             // tests: force the retry launch
    if (const char* e = getenv("LM_SLAB_CAP")) k.slab_cap = atoll(e);                            // tests: force the re-emit pass
    if (const char* e = getenv("LM_HT_OPT")) k.ht_opt = (uint32_t)atoi(e);                       // slots of a document's optimistic LWW table (a power of two; 0: sized for its Map rows at once; tests: 64 forces the second pass)
    if (const char* e = getenv("LM_CUT_MIN_ROWS")) k.cut_min_rows = (uint32_t)atoll(e);           // op rows from which a document's nodes are cut at cross-peer dependency targets and replayed largest peer first (tests: 0)
    if (const char* e = getenv("LM_SPAN_AUTO")) k.span_auto = atoi(e) != 0;                      // 0: the span-granular kernels for every batch (rounds 2-4b), whatever its documents' sizes
    if (const char* e = getenv("LM_VS_ROW_COST")) k.vs_row_cost = (uint32_t)atoi(e);             // ts_sweep_pays_batch (A/B)
    if (const char* e = getenv("LM_VERSION_SWEEP")) k.version_sweep = atoi(e) != 0;              // 0: resident trackers move row by row (rounds 3-4a: delete rows undone / redone one by one)
    if (const char* e = getenv("LM_FUSE_ROWS")) k.fuse_rows = atoi(e) != 0;                      // 0: one-change-per-keystroke documents are replayed row by row, as in rounds 1-3
    if (const char* e = getenv("LM_POSDEL")) k.posdel = atoi(e) != 0;                            // 0: a delete row whose target ids are not the elements at its position is LM_DATA_CORRUPTION (rounds 3-4) instead of being applied by position
    if (const char* e = getenv("LM_LINEAR")) k.linear = atoi(e) != 0;                            // 0: no linear prefix in the plain batch kernels (rounds 1-4: every node through the tracker)
    if (const char* e = getenv("LM_LWW_LDS")) k.lww_lds = atoi(e) != 0;                          // 0: every document's Map rows go through the HBM tables (k_map_lww), as in rounds 1-3
    if (const char* e = getenv("LM_REDO")) k.redo = atoi(e) != 0;                                // 0: no second replay of the documents a kernel flags DF_REDO (round 5: their verdict is that kernel's LM_DATA_CORRUPTION)
    // a side engine (lm_capi_impl.h redo) replays the documents another configuration flagged DF_REDO: span-granular batch kernels, whatever the batch's statistics say
    if (const char* e = getenv("LM_MAP_FUSED")) k.map_fused = atoi(e) != 0;                      // 0: LWW Map documents go through the row tables like every other document (rounds 1-5)
    if (const char* e = getenv("LM_MF_MIN_ROWS")) k.mf_min_rows = (uint32_t)atoi(e);             // rows from which a Map document gets a workgroup of k_map_fused (tests: 1)
    if (const char* e = getenv("LM_STAGE_DIRECT")) k.stage_direct = atoi(e) != 0;                // 0: blobs inside an lm_host_alloc region are gathered into the engine's staging buffer like any others
    if (const char* e = getenv("LM_PD_STATE_PIECES")) k.pd_state_pieces = (uint32_t)atoi(e);          // by-position list of a document staged on a snapshot's state: pieces per op row (2; tests: 0 = overflow -> replayed from the snapshot's history)
    if (const char* e = getenv("LM_SNAPSHOT_STATE")) { k.snapshot_state = atoi(e) != 0; k.snapshot_state_force = atoi(e) == 2; }   // (2: snapshot + updates on the state even where the history is expected to be cheaper, lm_snapshot_base.h `pays`)
               // 0: a document given as one snapshot is replayed from its ChangeStore (rounds 2-5) instead of rendered from its state section
    if (const char* e = getenv("LM_MF_CHG_RATIO")) k.mf_chg_ratio = (uint32_t)atoi(e);           // … and rows per change it needs on average (tests: 0)
    if (force_span) { k.span = true; k.span_auto = false; k.posdel = true; k.redo = false; k.map_fused = false; }
    kn = k;
  }
  bool force_span = false;
  // ---- documents staged from a snapshot's STATE section (lm_snapshot.h snapshot_state_to_updates): their version vector,
  // and their ChangeStore in updates form — what they are staged from again when lm_import asks for the history after all
  bool allow_state = true;                            // (the context switches it off for a folded batch: its entries are checked out)
  std::vector<uint8_t> h_vvo;
  std::vector<uint64_t> h_vvo_off;
  std::vector<std::vector<std::vector<uint8_t>>> st_hist;   // per document: empty, or its blobs in history form (the snapshot through its ChangeStore, the updates as they came)
  uint32_t n_state_docs = 0;
  bool stage_history_only = false;                    // (restage_history: this stage call takes every snapshot through its ChangeStore)
  DBuf b_vvo, b_vvo_off, b_pd_off, b_pd_row;
  std::vector<uint8_t> redo_hist;                 // per document: 1 = flagged in redo_docs AND replayed from its snapshot's HISTORY (st_hist), not from the staged state
  // ---- DF_REDO: what lm_stage left in the pinned staging buffer (the blobs as the device sees them: snapshots already reframed) —
  // a side engine stages the documents to replay from there (stage_from); lm_import with new blobs reuses the buffer and ends that
  bool st_valid = false;
  std::vector<uint32_t> st_doc_blob, st_blob_len;
  std::vector<uint64_t> st_blob_off, st_froot_off;
  std::vector<uint8_t> st_froot;
  std::vector<uint32_t> redo_docs;                // documents of the last run flagged DF_REDO (local indices)
  struct RedoItem { uint32_t doc; const uint8_t* front; size_t front_len; };
  bool redo_by_history(uint32_t doc) const { return doc < redo_hist.size() && redo_hist[doc] != 0; }
  void stage_from(const std::vector<Engine*>& parents, const std::vector<RedoItem>& items) {
    for (Engine* pe : parents) if (!pe->st_valid) throw std::runtime_error("redo: the staged blobs are gone");
    std::vector<std::vector<const uint8_t*>> bp(items.size());
    std::vector<std::vector<size_t>> bl(items.size());
    std::vector<DocIn> in(items.size());
    for (size_t k = 0; k < items.size(); k++) {
      const uint32_t i = items[k].doc;
      Engine& parent = *parents[k];
      if (parent.redo_by_history(i)) for (auto& hb : parent.st_hist[i]) { bp[k].push_back(hb.data()); bl[k].push_back(hb.size()); }
      else
      for (uint32_t b = parent.st_doc_blob[i]; b < parent.st_doc_blob[i + 1]; b++) { bp[k].push_back(parent.st_base + parent.st_blob_off[b]); bl[k].push_back(parent.st_blob_len[b]); }
      in[k] = DocIn{bp[k].data(), bl[k].data(), bp[k].size(), items[k].front, items[k].front_len};
    }
    stage_history_only = true;   // (a document replayed from its snapshot's history arrives as the snapshot itself: through its ChangeStore this time)
    try { stage(in.data(), in.size()); } catch (...) { stage_history_only = false; throw; }
    stage_history_only = false;
    // the state-section roots of the documents' snapshots (lm_stage read them from the mode-3 blobs, which the buffer no longer holds)
    h_froot.clear();
    h_froot_off.assign(items.size() + 1, 0);
    for (size_t k = 0; k < items.size(); k++) {
      h_froot_off[k] = h_froot.size();
      const uint32_t i = items[k].doc;
      Engine& parent = *parents[k];
      if ((size_t)i + 1 < parent.st_froot_off.size()) h_froot.insert(h_froot.end(), parent.st_froot.begin() + parent.st_froot_off[i], parent.st_froot.begin() + parent.st_froot_off[i + 1]);
    }
    h_froot_off[items.size()] = h_froot.size();
    // … and the version vectors of documents staged from a snapshot's state section (their staged blob is the synthetic one)
    h_vvo.clear(); h_vvo_off.assign(items.size() + 1, 0); n_state_docs = 0;
    for (size_t k = 0; k < items.size(); k++) {
      h_vvo_off[k] = h_vvo.size();
      Engine& parent = *parents[k];
      const uint32_t i = items[k].doc;
      if (parent.n_state_docs && (size_t)i + 1 < parent.h_vvo_off.size() && parent.h_vvo_off[i + 1] > parent.h_vvo_off[i] && !parent.redo_by_history(i)) {
        h_vvo.insert(h_vvo.end(), parent.h_vvo.begin() + parent.h_vvo_off[i], parent.h_vvo.begin() + parent.h_vvo_off[i + 1]);
        n_state_docs++;
      }
    }
    h_vvo_off[items.size()] = h_vvo.size();
    lmbe::bind(sc);
    if (n_state_docs) {
      b_vvo.ensure(h_vvo.size() + 16); lmbe::h2d(b_vvo.p, h_vvo.data(), h_vvo.size());
      b_vvo_off.ensure((items.size() + 1) * 8); lmbe::h2d(b_vvo_off.p, h_vvo_off.data(), (items.size() + 1) * 8);
    }
    b_froot.ensure(h_froot.size() + 16); if (!h_froot.empty()) lmbe::h2d(b_froot.p, h_froot.data(), h_froot.size());
    b_froot_off.ensure((items.size() + 1) * 8); lmbe::h2d(b_froot_off.p, h_froot_off.data(), (items.size() + 1) * 8);
    lmbe::sync();
  }

  explicit Engine(int device) { sc = lmbe::stream_create(device); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  uint8_t* h_stage = nullptr;      // pinned staging buffer of lm_stage (grow-only)
  const uint8_t* st_base = nullptr;   // where the staged blobs are on the host: h_stage, or the caller's lm_host_alloc region (direct staging) — st_blob_off counts from here
  bool staged_direct = false;
  size_t h_stage_cap = 0;
  ~Engine() { release_all(); if (h_stage) lmbe::hfree(h_stage); lmbe::stream_destroy(sc); }
  void release_all() {
    DBuf* all[] = {&b_front, &b_front_off, &b_froot, &b_froot_off, &b_blob_hash, &b_big, &b_data, &b_blob_off, &b_blob_len, &b_doc_blob, &b_blob_doc, &b_blob_status, &b_blob_nblk, &b_blob_blk0, &b_tile, &b_tot,
                   &b_vvo, &b_vvo_off, &b_blk, &b_bcnt, &b_boff, &b_blk_kind, &b_doc_fused, &b_mf_docs, &b_mf_key0, &b_chg, &b_dep_peer, &b_dep_ctr, &b_dep_ci, &b_op, &b_op_val, &b_op_blk, &b_key_off, &b_key_len,
                   &b_cid_raw, &b_cid_map, &b_peer_raw, &b_peer_map, &b_doc, &b_peer_uniq, &b_peer_end, &b_peer_ext, &b_peer_base, &b_peer_end_all, &b_elem_base,
                   &b_peer_chg0, &b_peer_chg1, &b_cont, &b_chg_mask, &b_chg_sorted, &b_chg_lamport, &b_chg_skip, &b_chg_flag, &b_node_first,
                   &b_node_last, &b_node_order, &b_vvh, &b_blk_sorted, &b_chg_node, &b_node_done, &b_node_lam, &b_cp, &b_loc, &b_tb, &b_it,
                   &b_dir_out, &b_lf_chunk, &b_fuse, &b_dcnt, &b_posdel, &b_pd_off, &b_pd_row,
                   &b_cont_root0, &b_cont_nroot, &b_prof, &b_hash, &b_order, &b_ht_key, &b_ht_pfx, &b_ht_best, &b_ht0, &b_ht_cap, &b_ht_list, &b_ht_cnt, &b_slab, &b_vslab, &b_slab_off, &b_vslab_off, &b_slab2, &b_slab2_off, &b_out, &b_out_off,
                   &b_vv_out, &b_vv_off, &b_tk, &b_res, &b_dir_out2, &b_dir_b, &b_dir_b2, &b_doc_saved, &b_elem_cap, &b_old_blobs, &b_prev_doc, &b_prev_uniq, &b_prev_end, &b_lca_out, &b_lca_scratch, &b_lca_off};
    for (DBuf* b : all) b->release();
  }

  // ---- stage: pack the blobs (16-byte aligned starts) and upload
  struct DocIn { const uint8_t* const* blobs; const size_t* lens; size_t n; const uint8_t* front; size_t front_len; uint8_t state_root = 0; };   // state_root: one shallow snapshot, shown at its shallow root (lm_capi_impl.h stage)
  void stage(const DocIn* docs, size_t nd) {
    lmbe::bind(sc);
    read_knobs();
    n_docs = (uint32_t)nd;
    h_doc_blob.assign(nd + 1, 0);
    size_t nb = 0;
    for (size_t i = 0; i < nd; i++) { h_doc_blob[i] = (uint32_t)nb; nb += docs[i].n; }
    h_doc_blob[nd] = (uint32_t)nb;
    n_blobs = (uint32_t)nb;
    h_blob_off.assign(nb + 1, 0);
    h_blob_len.assign(nb, 0);
    h_blob_doc.assign(nb, 0);
    uint64_t off = 0;
    in_bytes = 0;
    size_t b = 0;
    // FastSnapshot blobs (mode 3) are turned into the FastUpdates framing of their ChangeStore here, on the host
    // (lm_snapshot.h); everything behind this point only ever sees mode 4.  A shallow snapshot stays as it is (the device
    // reports LM_UNSUPPORTED for mode 3); a damaged one becomes a stub that fails the way its damage would.
    std::vector<const uint8_t*> bsrc(nb);
    std::vector<size_t> blen(nb);
    std::vector<std::vector<uint8_t>> conv;
    std::vector<uint8_t>& froot = h_froot;      // per document: root containers of the state section of the snapshot that initialises it
    froot.clear();
    size_t best_changes = 0;
    bool have_snapshot = false;
    h_froot_off.assign(nd + 1, 0);
    static const uint8_t stub_decode[22] = {'l', 'o', 'r', 'o', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xff, 0xff};
    static const uint8_t stub_checksum[22] = {'l', 'o', 'r', 'o', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4};
    std::vector<std::vector<uint8_t>> vvo_tmp(nd);
    st_hist.assign(nd, {});
    n_state_docs = 0;
    const bool state_off = stage_history_only;
    auto is_mode = [](const uint8_t* p, size_t l, uint8_t mode) { return l >= 22 && memcmp(p, "loro", 4) == 0 && p[20] == 0 && p[21] == mode; };
    for (size_t i = 0; i < nd; i++) {
      // ONE snapshot + updates that continue its history, rendered at the latest version: the snapshot is staged as its STATE and the
      // updates are hung on it (lm_snapshot_base.h) — the snapshot's history is neither uploaded nor decoded nor replayed.  Declined
      // (updates concurrent with part of the snapshot, …): every blob through the loop below, the snapshot through its ChangeStore
      if (docs[i].n >= 2 && !docs[i].front && kn.snapshot_state && allow_state && !state_off) {
        size_t ks = docs[i].n, n3 = 0, n4 = 0;
        for (size_t k = 0; k < docs[i].n; k++) { if (is_mode(docs[i].blobs[k], docs[i].lens[k], 3)) { n3++; ks = k; } else if (is_mode(docs[i].blobs[k], docs[i].lens[k], 4)) n4++; }
        std::vector<uint8_t> o2, vv2, roots2;
        lmsnap::StateBase sb;
        std::vector<std::pair<const uint8_t*, size_t>> U;
        std::vector<std::vector<uint8_t>> outU;
        bool pays = true;
        if (n3 == 1 && n3 + n4 == docs[i].n) for (size_t k = 0; k < docs[i].n; k++) if (k != ks) U.emplace_back(docs[i].blobs[k], docs[i].lens[k]);
        if (n3 == 1 && n3 + n4 == docs[i].n && lmsnap::snapshot_state_to_updates(docs[i].blobs[ks], docs[i].lens[ks], o2, vv2, &roots2, false, &sb) &&
            lmsnap::rebase_updates_on_state(sb, U, outU, &pays) && (pays || kn.snapshot_state_force)) {
          h_froot_off[i] = froot.size();
          froot.insert(froot.end(), roots2.begin(), roots2.end());
          // (the history form is kept as the snapshot came: a copy — the caller's buffers need not outlive lm_stage — that is taken through
          // its ChangeStore only if the document is ever staged as history: restage_history, a failed state replay)
          size_t u = 0;
          for (size_t k = 0; k < docs[i].n; k++, b++) {
            if (k == ks) { st_hist[i].emplace_back(docs[i].blobs[ks], docs[i].blobs[ks] + docs[i].lens[ks]); conv.push_back(std::move(o2)); }
            else { st_hist[i].emplace_back(docs[i].blobs[k], docs[i].blobs[k] + docs[i].lens[k]); conv.push_back(std::move(outU[u++])); }
            bsrc[b] = conv.back().data(); blen[b] = conv.back().size();
          }
          vvo_tmp[i].resize(8);
          for (int x = 0; x < 8; x++) vvo_tmp[i][x] = (uint8_t)(sb.synth_peer >> (8 * x));
          vvo_tmp[i].insert(vvo_tmp[i].end(), vv2.begin(), vv2.end());
          n_state_docs++;
          if (const char* dump = getenv("LM_DUMP_STATE_BASE")) {   // (debugging: the blobs as staged — <prefix><doc>_<blob>.bin)
            for (size_t k = 0; k < docs[i].n; k++) {
              std::string fn = std::string(dump) + std::to_string(i) + "_" + std::to_string(k) + ".bin";
              if (FILE* f = fopen(fn.c_str(), "wb")) { fwrite(bsrc[b - docs[i].n + k], 1, blen[b - docs[i].n + k], f); fclose(f); }
            }
          }
          continue;
        }
      }
      for (size_t k = 0; k < docs[i].n; k++, b++) {
        const uint8_t* p = docs[i].blobs[k];
        size_t l = docs[i].lens[k];
        if (k == 0) h_froot_off[i] = froot.size();
        if (k == 0) { best_changes = 0; have_snapshot = false; }
        if (l >= 22 && memcmp(p, "loro", 4) == 0 && p[20] == 0 && p[21] == 3) {
          // a document that IS one snapshot, rendered at its latest version: its state section is the value (fast_snapshot.rs:168-258)
          // — staged as the state itself, never as the history (lm_snapshot.h snapshot_state_to_updates); a shallow snapshot has no
          // other way to be rendered.  Declined (a state this reader cannot take): the ChangeStore, as for every other snapshot
          if (docs[i].n == 1 && !docs[i].front && kn.snapshot_state && (allow_state || docs[i].state_root) && !state_off) {
            std::vector<uint8_t> o2, vv2, roots2;
            lmsnap::StateBase sb;
            if (lmsnap::snapshot_state_to_updates(p, l, o2, vv2, &roots2, docs[i].state_root != 0, &sb)) {
              st_hist[i].emplace_back(p, p + l);            // (as it came: through its ChangeStore only when the batch is staged as history — a shallow snapshot is LM_UNSUPPORTED there, as before)
              conv.push_back(std::move(o2)); p = conv.back().data(); l = conv.back().size();
              froot.resize(h_froot_off[i]); froot.insert(froot.end(), roots2.begin(), roots2.end());
              vvo_tmp[i].resize(8);                         // (Dev::vvo: the synthetic peer, then the snapshot's version vector)
              for (int x = 0; x < 8; x++) vvo_tmp[i][x] = (uint8_t)(sb.synth_peer >> (8 * x));
              vvo_tmp[i].insert(vvo_tmp[i].end(), vv2.begin(), vv2.end());
              n_state_docs++;
              bsrc[b] = p; blen[b] = l;
              continue;
            }
          }
          // import_batch imports snapshots first, the one with the most changes first (loro.rs:1445-1449): THAT snapshot's state
          // section initialises the empty document's state store; the others arrive as updates
          std::vector<uint8_t> o, roots;
          size_t n_changes = 0;
          int st = lmsnap::snapshot_to_updates(p, l, o, &roots, &n_changes);
          if (st == lmsnap::SN_OK) {
            conv.push_back(std::move(o)); p = conv.back().data(); l = conv.back().size();
            if (!have_snapshot || n_changes > best_changes) { froot.resize(h_froot_off[i]); froot.insert(froot.end(), roots.begin(), roots.end()); best_changes = n_changes; have_snapshot = true; }
          }
          else if (st == lmsnap::SN_CHECKSUM) { p = stub_checksum; l = 22; }
          else if (st == lmsnap::SN_DECODE) { p = stub_decode; l = 22; }
        }
        bsrc[b] = p; blen[b] = l;
      }
    }
    h_vvo.clear(); h_vvo_off.assign(nd + 1, 0);
    for (size_t i = 0; i < nd; i++) { h_vvo_off[i] = h_vvo.size(); h_vvo.insert(h_vvo.end(), vvo_tmp[i].begin(), vvo_tmp[i].end()); }
    h_vvo_off[nd] = h_vvo.size();
    if (n_state_docs) {
      b_vvo.ensure(h_vvo.size() + 16); lmbe::h2d(b_vvo.p, h_vvo.data(), h_vvo.size());
      b_vvo_off.ensure((nd + 1) * 8); lmbe::h2d(b_vvo_off.p, h_vvo_off.data(), (nd + 1) * 8);
    }
    for (size_t i = 0; i < nd; i++) if (docs[i].n == 0) h_froot_off[i] = froot.size();
    h_froot_off[nd] = froot.size();
    for (size_t i = nd; i-- > 0;) if (h_froot_off[i] > h_froot_off[i + 1]) h_froot_off[i] = h_froot_off[i + 1];
    b_froot.ensure(froot.size() + 16); if (!froot.empty()) lmbe::h2d(b_froot.p, froot.data(), froot.size());
    b_froot_off.ensure((nd + 1) * 8); lmbe::h2d(b_froot_off.p, h_froot_off.data(), (nd + 1) * 8);
    // Direct staging: every blob of this part lies — 16-byte aligned, in staging order, without overlap — inside ONE region of
    // lm_host_alloc (pinned): the span [first blob, end of the last) is copied to the device as it is, chunk by chunk, and a blob's
    // device offset is its distance from the first; no byte is touched by the host.  (A span much larger than its blobs — a caller
    // that scattered them over the region — is not worth copying: gathered like pageable memory.)
    bool direct = kn.stage_direct && nb > 0 && conv.empty();
    if (direct) {
      uint64_t sum = 0;
      for (size_t j = 0; j < nb && direct; j++) {
        sum += blen[j];
        if (((uintptr_t)bsrc[j] & 15) || (j && bsrc[j] < bsrc[j - 1] + blen[j - 1])) direct = false;
      }
      const uint64_t span = direct ? (uint64_t)(bsrc[nb - 1] + blen[nb - 1] - bsrc[0]) : 0;
      if (direct && (span > 2 * sum + (1u << 20) || !host_regions().find(bsrc[0], (size_t)span, nullptr))) direct = false;
    }
    staged_direct = direct;
    b = 0;
    for (size_t i = 0; i < nd; i++)
      for (size_t k = 0; k < docs[i].n; k++, b++) {
        if (blen[b] > 0xfffffff0ull) throw std::runtime_error("blob larger than 4 GiB");
        if (direct) off = (uint64_t)(bsrc[b] - bsrc[0]);
        h_blob_off[b] = off;
        h_blob_len[b] = (uint32_t)blen[b];
        h_blob_doc[b] = (uint32_t)i;
        in_bytes += blen[b];
        off += (blen[b] + 15) & ~(uint64_t)15;
      }
    h_blob_off[nb] = off;
    data_bytes = off + 64;
    // Pinned staging buffer, kept across batches (pinning a fresh gigabyte per batch cost more than the copy).  The
    // blobs are gathered into it by a few worker threads in ≈16 MB chunks of destination, and every chunk is handed
    // to the copy engine as soon as it is complete: the gather of chunk c+1 runs beside the DMA of chunk c.
    if (!direct && data_bytes > h_stage_cap) {
      if (h_stage) lmbe::hfree(h_stage);
      h_stage_cap = data_bytes + data_bytes / 4 + 4096;
      h_stage = (uint8_t*)lmbe::halloc(h_stage_cap);
      if (!h_stage) { h_stage_cap = 0; throw std::runtime_error("host staging allocation failed"); }
    }
    b_data.ensure(data_bytes);
    st_base = direct ? bsrc[0] : h_stage;
    if (direct) {
      // (the last blob's padding and the 64 bytes of slack behind it are the device's to clear: the caller's span ends with the blob)
      const uint64_t span = (uint64_t)(bsrc[nb - 1] + blen[nb - 1] - bsrc[0]);
      uint64_t CHUNK = 64ull << 20;
      for (uint64_t o = 0; o < span; o += CHUNK) lmbe::h2d_async((uint8_t*)b_data.p + o, bsrc[0] + o, span - o < CHUNK ? span - o : CHUNK);
      lmbe::dmemset((uint8_t*)b_data.p + span, 0, data_bytes - span);
    } else {
      uint8_t* host = h_stage;
      const std::vector<const uint8_t*>& src = bsrc;
      std::vector<size_t> cut{0};           // chunk c = blobs [cut[c], cut[c+1])
      uint64_t CHUNK = 16ull << 20;
      if (const char* e = getenv("LM_STAGE_CHUNK_MB")) { long v = atol(e); if (v >= 1 && v <= 1024) CHUNK = (uint64_t)v << 20; }   // (A/B of the staging pipeline)
      for (size_t j = 0; j < nb; j++) if (h_blob_off[j + 1] - h_blob_off[cut.back()] >= CHUNK && j + 1 < nb) cut.push_back(j + 1);
      cut.push_back(nb);
      size_t nc = cut.size() - 1;
      std::vector<std::atomic<int>> done(nc);
      for (auto& x : done) x.store(0);
      std::atomic<size_t> next(0);
      auto gather = [&]() {
        for (;;) {
          size_t c = next.fetch_add(1);
          if (c >= nc) return;
          for (size_t j = cut[c]; j < cut[c + 1]; j++) {
            uint64_t o = h_blob_off[j], l = h_blob_len[j], pad = h_blob_off[j + 1] - o - l;
            memcpy(host + o, src[j], l);
            if (pad) memset(host + o + l, 0, pad);
          }
          done[c].store(1, std::memory_order_release);
        }
      };
      size_t nt = nc < 2 ? 0 : (nc < 6 ? nc - 1 : 6);
      if (const char* e = getenv("LM_STAGE_THREADS")) { long v = atol(e); if (v >= 1 && v <= 64 && nc >= 2) nt = (size_t)v < nc - 1 ? (size_t)v : nc - 1; }
      std::vector<std::thread> th;
      for (size_t t = 0; t < nt; t++) th.emplace_back(gather);
      if (!nt) gather();
      memset(host + off, 0, 64);
      try {
        for (size_t c = 0; c < nc; c++) {
          while (!done[c].load(std::memory_order_acquire)) std::this_thread::yield();
          uint64_t o0 = h_blob_off[cut[c]], o1 = c + 1 == nc ? data_bytes : h_blob_off[cut[c + 1]];
          lmbe::h2d_async((uint8_t*)b_data.p + o0, host + o0, o1 - o0);
        }
      } catch (...) {   // a failed copy must not unwind past joinable threads (std::terminate): stop the workers, join, report
        next.store(nc);
        for (auto& t : th) t.join();
        throw;
      }
      for (auto& t : th) t.join();
      if (!nb) lmbe::h2d_async(b_data.p, host, data_bytes);
    }
    b_blob_off.ensure((nb + 1) * 8); lmbe::h2d(b_blob_off.p, h_blob_off.data(), (nb + 1) * 8);
    b_blob_len.ensure(nb * 4 + 4); if (nb) lmbe::h2d(b_blob_len.p, h_blob_len.data(), nb * 4);
    b_doc_blob.ensure((nd + 1) * 4); lmbe::h2d(b_doc_blob.p, h_doc_blob.data(), (nd + 1) * 4);
    b_blob_doc.ensure(nb * 4 + 4); if (nb) lmbe::h2d(b_blob_doc.p, h_blob_doc.data(), nb * 4);
    // optional checkout frontiers, back to back
    h_front_off.assign(nd + 1, 0);
    std::vector<uint8_t>& fr = h_front_bytes;
    fr.clear();
    for (size_t i = 0; i < nd; i++) {
      h_front_off[i] = fr.size();
      if (docs[i].front) {
        if (docs[i].front_len == 0) throw std::runtime_error("checkout_frontiers with zero length (the empty version is the byte 00)");
        fr.insert(fr.end(), docs[i].front, docs[i].front + docs[i].front_len);
      }
    }
    h_front_off[nd] = fr.size();
    b_front.ensure(fr.size() + 16); if (!fr.empty()) lmbe::h2d(b_front.p, fr.data(), fr.size());
    b_front_off.ensure((nd + 1) * 8); lmbe::h2d(b_front_off.p, h_front_off.data(), (nd + 1) * 8);
    lmbe::sync();   // (leaving the copies of a directly staged batch in flight and returning at once was measured: 335k against 356k docs/s end to end, tests/tools/gpu_e2e.py — the host thread is not what bounds the rotation)
    st_valid = true; st_doc_blob = h_doc_blob; st_blob_len = h_blob_len; st_blob_off = h_blob_off; st_froot_off = h_froot_off; st_froot = h_froot;
    redo_docs.clear();
    ran = fetched = false;
    resident = false; tables_valid = false;   // a new batch: whatever was resident is gone
    have_prev = false; h_lca.clear();         // (lm_import_modes / lm_import_lca: nothing is known about the new batch's imports)
  }

  // ---- lm_import: more blobs (and / or other checkouts) for the documents of the resident batch
  static const uint8_t* snapshot_or_same(const uint8_t* p, size_t& l, std::vector<std::vector<uint8_t>>& conv) {
    static const uint8_t stub_decode[22] = {'l', 'o', 'r', 'o', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0xff, 0xff};
    static const uint8_t stub_checksum[22] = {'l', 'o', 'r', 'o', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4};
    if (l >= 22 && memcmp(p, "loro", 4) == 0 && p[20] == 0 && p[21] == 3) {
      // a snapshot imported into a document that already holds history contributes its ChangeStore only (fast_snapshot.rs:326-344)
      std::vector<uint8_t> o;
      int st = lmsnap::snapshot_to_updates(p, l, o, nullptr);
      if (st == lmsnap::SN_OK) { conv.push_back(std::move(o)); p = conv.back().data(); l = conv.back().size(); }
      else if (st == lmsnap::SN_CHECKSUM) { p = stub_checksum; l = 22; }
      else if (st == lmsnap::SN_DECODE) { p = stub_decode; l = 22; }
    }
    return p;
  }
  void restage_history() {
    if (!st_valid) throw std::runtime_error("lm_import: the staged blobs of a batch rendered from snapshot states are gone");
    const size_t nd = n_docs;
    std::vector<std::vector<const uint8_t*>> bp(nd);
    std::vector<std::vector<size_t>> bl(nd);
    std::vector<DocIn> in(nd);
    std::vector<std::vector<std::vector<uint8_t>>> hist = std::move(st_hist);
    std::vector<uint8_t> fronts = h_front_bytes;
    std::vector<uint64_t> foff = h_front_off;
    // (the other documents' blobs are read from the staging buffer while stage() gathers INTO it: copied out first)
    std::vector<std::vector<uint8_t>> keep(n_blobs);
    for (size_t i = 0; i < nd; i++) {
      if (!hist[i].empty()) for (auto& hb : hist[i]) { bp[i].push_back(hb.data()); bl[i].push_back(hb.size()); }
      else for (uint32_t b = st_doc_blob[i]; b < st_doc_blob[i + 1]; b++) { keep[b].assign(st_base + st_blob_off[b], st_base + st_blob_off[b] + st_blob_len[b]); bp[i].push_back(keep[b].data()); bl[i].push_back(keep[b].size()); }
      const bool hf = foff.size() > i + 1 && foff[i + 1] > foff[i];
      in[i] = DocIn{bp[i].data(), bl[i].data(), bp[i].size(), hf ? fronts.data() + foff[i] : nullptr, hf ? (size_t)(foff[i + 1] - foff[i]) : 0};
    }
    std::vector<uint8_t> froot_keep = st_froot;
    std::vector<uint64_t> froot_off_keep = st_froot_off;
    stage_history_only = true;
    try { stage(in.data(), nd); } catch (...) { stage_history_only = false; throw; }
    stage_history_only = false;
    // (documents that were NOT snapshots any more in the buffer — already reframed — lost their state-section roots in this second pass: put back)
    bool differs = false;
    // (… and so did the snapshots restaged in their history form: the root set of the state section is what the first pass read)
    auto kept = [&](size_t i) { return froot_off_keep.size() > i + 1 && froot_off_keep[i + 1] > froot_off_keep[i]; };
    for (size_t i = 0; i < nd && !differs; i++) differs = kept(i);
    if (differs) {
      std::vector<uint8_t> fr; std::vector<uint64_t> fo(nd + 1, 0);
      for (size_t i = 0; i < nd; i++) {
        fo[i] = fr.size();
        if (kept(i)) fr.insert(fr.end(), froot_keep.begin() + froot_off_keep[i], froot_keep.begin() + froot_off_keep[i + 1]);
        else fr.insert(fr.end(), h_froot.begin() + h_froot_off[i], h_froot.begin() + h_froot_off[i + 1]);
      }
      fo[nd] = fr.size();
      h_froot = fr; h_froot_off = fo; st_froot = fr; st_froot_off = fo;
      lmbe::bind(sc);
      b_froot.ensure(h_froot.size() + 16); if (!h_froot.empty()) lmbe::h2d(b_froot.p, h_froot.data(), h_froot.size());
      b_froot_off.ensure((nd + 1) * 8); lmbe::h2d(b_froot_off.p, h_froot_off.data(), (nd + 1) * 8);
      lmbe::sync();
    }
  }
  void adopt_resident() {   // the staged batch becomes the first generation of the resident documents
    r_blobs.assign(n_docs, {});
    for (uint32_t i = 0; i < n_docs; i++)
      for (uint32_t b = h_doc_blob[i]; b < h_doc_blob[i + 1]; b++) r_blobs[i].push_back(BlobRef{h_blob_off[b], h_blob_len[b]});
    arena_top = h_blob_off[n_blobs];
    r_step.assign(n_docs, 0);
    r_front.assign(n_docs, {});
    for (uint32_t i = 0; i < n_docs; i++) r_front[i].assign(h_front_bytes.begin() + h_front_off[i], h_front_bytes.begin() + h_front_off[i + 1]);
    tk_leaf0.assign(n_docs, 0); tk_leaf_cap.assign(n_docs, 0); tk_pcap.assign(n_docs, 0); tk_ccap.assign(n_docs, 0);
    tk_off.assign(n_docs, 0); tk_reset.assign(n_docs, 1);
    tk_elem0.assign(n_docs, 0); tk_elem_cap.assign(n_docs, 0); elem_top = 0;
    tk_top = 0; leaf_top = 0; dir_parity = 0;
    resident = true; tables_valid = false; have_prev = false;
  }
  // The resident batch becomes a batch of src.size() documents, document i holding the blobs of the present document src[i] — the
  // SAME bytes of the arena — and rendered at fronts[i] by the next run: what lm_import needs from a batch that was staged folded
  // (lm_capi_impl.h "shared replay": entries with the same blobs were uploaded once).  Trackers start from the empty version.
  void expand(const std::vector<uint32_t>& src, const std::vector<std::vector<uint8_t>>& fronts) {
    lmbe::bind(sc);
    if (!resident) adopt_resident();
    std::vector<std::vector<BlobRef>> old = r_blobs;
    std::vector<uint64_t> old_froot_off = h_froot_off;
    std::vector<uint8_t> old_froot = h_froot;
    n_docs = (uint32_t)src.size();
    r_blobs.assign(n_docs, {});
    r_front.assign(n_docs, {});
    h_froot.clear();
    h_froot_off.assign((size_t)n_docs + 1, 0);
    for (uint32_t i = 0; i < n_docs; i++) {
      r_blobs[i] = old[src[i]];
      r_front[i] = fronts[i];
      h_froot_off[i] = h_froot.size();
      if (src[i] + 1 < old_froot_off.size()) h_froot.insert(h_froot.end(), old_froot.begin() + old_froot_off[src[i]], old_froot.begin() + old_froot_off[src[i] + 1]);
    }
    h_froot_off[n_docs] = h_froot.size();
    b_froot.ensure(h_froot.size() + 16); if (!h_froot.empty()) lmbe::h2d(b_froot.p, h_froot.data(), h_froot.size());
    b_froot_off.ensure(((size_t)n_docs + 1) * 8); lmbe::h2d(b_froot_off.p, h_froot_off.data(), ((size_t)n_docs + 1) * 8);
    r_step.assign(n_docs, 0);
    tk_leaf0.assign(n_docs, 0); tk_leaf_cap.assign(n_docs, 0); tk_pcap.assign(n_docs, 0); tk_ccap.assign(n_docs, 0);
    tk_off.assign(n_docs, 0); tk_reset.assign(n_docs, 1);
    tk_elem0.assign(n_docs, 0); tk_elem_cap.assign(n_docs, 0); elem_top = 0;
    tk_top = 0; leaf_top = 0; dir_parity = 0;
    tk_new.clear();
    tables_valid = false; have_prev = false; h_lca.clear();
    shared_mode = 0;
    rebuild_blob_tables();
    lmbe::sync();
    ran = fetched = false;
  }
  void rebuild_blob_tables() {
    size_t nd = n_docs, nb = 0;
    h_doc_blob.assign(nd + 1, 0);
    for (size_t i = 0; i < nd; i++) { h_doc_blob[i] = (uint32_t)nb; nb += r_blobs[i].size(); }
    h_doc_blob[nd] = (uint32_t)nb;
    n_blobs = (uint32_t)nb;
    h_blob_off.assign(nb + 1, 0); h_blob_len.assign(nb, 0); h_blob_doc.assign(nb, 0);
    in_bytes = 0;
    size_t b = 0;
    for (size_t i = 0; i < nd; i++) {
      // list values are addressed by 32-bit offsets from the document's first blob (cp[], lm_k_dag.h k_elem_fill)
      if (!r_blobs[i].empty() && r_blobs[i].back().off + r_blobs[i].back().len - r_blobs[i].front().off >= 0xfffffff0ull)
        throw std::runtime_error("lm_import: a resident document's blobs span more than 4 GiB of the arena (stage the batch again)");
      for (const BlobRef& r : r_blobs[i]) { h_blob_off[b] = r.off; h_blob_len[b] = r.len; h_blob_doc[b] = (uint32_t)i; in_bytes += r.len; b++; }
    }
    h_blob_off[nb] = arena_top;
    data_bytes = arena_top + 64;
    b_blob_off.ensure((nb + 1) * 8); lmbe::h2d(b_blob_off.p, h_blob_off.data(), (nb + 1) * 8);
    b_blob_len.ensure(nb * 4 + 4); if (nb) lmbe::h2d(b_blob_len.p, h_blob_len.data(), nb * 4);
    b_doc_blob.ensure((nd + 1) * 4); lmbe::h2d(b_doc_blob.p, h_doc_blob.data(), (nd + 1) * 4);
    b_blob_doc.ensure(nb * 4 + 4); if (nb) lmbe::h2d(b_blob_doc.p, h_blob_doc.data(), nb * 4);
    h_front_off.assign(nd + 1, 0);
    h_front_bytes.clear();
    for (size_t i = 0; i < nd; i++) { h_front_off[i] = h_front_bytes.size(); h_front_bytes.insert(h_front_bytes.end(), r_front[i].begin(), r_front[i].end()); }
    h_front_off[nd] = h_front_bytes.size();
    b_front.ensure(h_front_bytes.size() + 16); if (!h_front_bytes.empty()) lmbe::h2d(b_front.p, h_front_bytes.data(), h_front_bytes.size());
    b_front_off.ensure((nd + 1) * 8); lmbe::h2d(b_front_off.p, h_front_off.data(), (nd + 1) * 8);
  }
  // a (larger) tracker record for document i; the sticky "the state store holds this container" words move over, the tracker
  // itself is not carried (the document is replayed from the empty version by the next run).  The new records are collected
  // in tk_new and uploaded together by tk_flush (one copy per run, not one per document)
  std::vector<uint32_t> tk_new;
  uint64_t tk_new_at = 0;
  void tk_grow(uint32_t i, uint32_t P, uint32_t C) {
    uint32_t pcap = P + P / 2 + 2, ccap = C + C / 2 + 4;
    uint64_t words = (uint64_t)TK_HDR + 5ull * pcap + (uint64_t)ccap * (TK_CW + pcap);
    if (tk_new.empty()) tk_new_at = tk_top;
    size_t at = tk_new.size();
    tk_new.resize(at + words, 0u);
    if (tk_pcap[i]) {
      uint32_t op = tk_pcap[i], oc = tk_ccap[i];
      uint64_t ow = (uint64_t)TK_HDR + 5ull * op + (uint64_t)oc * (TK_CW + op);
      std::vector<uint32_t> old(ow);
      lmbe::d2h(old.data(), b_tk.as<uint32_t>() + tk_off[i], ow * 4);
      for (uint32_t c = 0; c < oc && c < ccap; c++)
        tk_new[at + TK_HDR + 5ull * pcap + (uint64_t)c * (TK_CW + pcap) + 3] = old[TK_HDR + 5ull * op + (uint64_t)c * (TK_CW + op) + 3] == 1u ? 1u : 0u;
    }
    tk_off[i] = tk_top; tk_top += words; tk_pcap[i] = pcap; tk_ccap[i] = ccap; tk_reset[i] = 1;
  }
  void tk_flush() {
    if (tk_new.empty()) return;
    b_tk.ensure_keep((tk_top + 16) * 4, tk_new_at * 4);
    lmbe::h2d(b_tk.as<uint32_t>() + tk_new_at, tk_new.data(), tk_new.size() * 4);
    tk_new.clear();
  }
  void import_more(const DocIn* docs, size_t nd) {
    lmbe::bind(sc);
    if (nd != n_docs) throw std::runtime_error("lm_import: the document count differs from the resident batch");
    for (size_t i = 0; i < nd; i++)
      if (docs[i].front && docs[i].front_len == 0) throw std::runtime_error("checkout_frontiers with zero length (the empty version is the byte 00)");
    read_knobs();
    if (!resident && n_state_docs) {
      // documents staged from their snapshots' STATE sections hold no history a later import could build on (their ids are the
      // synthetic peer's): the batch is staged once more, every snapshot through its ChangeStore this time (the blobs of the other
      // documents are still in the pinned staging buffer)
      const bool was_run = ran;
      restage_history();
      ran = was_run;
    }
    if (!resident) {
      // The staged batch becomes the first generation of the resident documents — as ITS OWN import: `lm_stage; [lm_run;]
      // lm_import(b); lm_run` is import_batch(staged) followed by import(b), two diffs for the state store (a root whose text the
      // second import deletes stays, empty; loro.rs:568-649, state.rs:621-849), not one import_batch of everything.  A batch run
      // (lm_run straight after lm_stage) records neither trackers nor the "state store holds this container" words, so the
      // staged blobs are run once more here as the resident documents' first step.
      // (lm_import straight after lm_stage, with no lm_run in between, asks for no state in between: staged and imported blobs
      // then form the documents' first step together — one import_batch.)
      const bool was_run = ran;
      adopt_resident();
      if (was_run && n_blobs) {
        for (uint32_t i = 0; i < n_docs; i++) r_step[i] = (uint32_t)r_blobs[i].size();   // a document whose batch import failed is empty again afterwards (loro.rs:780-838)
        run();
        lmbe::bind(sc);
      }
    }
    // Everything that can fail happens on temporaries: a rejected lm_import leaves the documents' blob lists, their checkouts and
    // the arena exactly as they were (a half-recorded one handed the next import arena offsets that other documents' bytes
    // already occupied).
    std::vector<std::vector<uint8_t>> conv;
    struct Src { const uint8_t* p; size_t l; uint64_t off; };
    std::vector<Src> src;
    std::vector<std::vector<BlobRef>> add_refs(nd);
    uint64_t top = arena_top;
    for (size_t i = 0; i < nd; i++) {
      for (size_t k = 0; k < docs[i].n; k++) {
        size_t l = docs[i].lens[k];
        const uint8_t* p = snapshot_or_same(docs[i].blobs[k], l, conv);
        if (l > 0xfffffff0ull) throw std::runtime_error("blob larger than 4 GiB");
        src.push_back(Src{p, l, top});
        add_refs[i].push_back(BlobRef{top, (uint32_t)l});
        top += (l + 15) & ~(uint64_t)15;
      }
      // list values are addressed by 32-bit offsets from the document's first blob (cp[], lm_k_dag.h k_elem_fill)
      if (!add_refs[i].empty()) {
        uint64_t first = r_blobs[i].empty() ? add_refs[i].front().off : r_blobs[i].front().off;
        if (add_refs[i].back().off + add_refs[i].back().len - first >= 0xfffffff0ull)
          throw std::runtime_error("lm_import: a resident document's blobs span more than 4 GiB of the arena (stage the batch again)");
      }
    }
    uint64_t add = top - arena_top;
    if (add) {
      st_valid = false;
      if (add + 64 > h_stage_cap) {
        uint8_t* nh = (uint8_t*)lmbe::halloc(add + add / 4 + 4096);
        if (!nh) throw std::runtime_error("host staging allocation failed");
        if (h_stage) lmbe::hfree(h_stage);
        h_stage = nh; h_stage_cap = add + add / 4 + 4096;
      }
      for (const Src& x : src) {
        uint64_t o = x.off - arena_top, pad = ((x.l + 15) & ~(uint64_t)15) - x.l;
        memcpy(h_stage + o, x.p, x.l);
        if (pad) memset(h_stage + o + x.l, 0, pad);
      }
      memset(h_stage + add, 0, 64);
      b_data.ensure_keep(top + 64, arena_top);
      lmbe::h2d_async((uint8_t*)b_data.p + arena_top, h_stage, add + 64);
    }
    // commit
    for (size_t i = 0; i < nd; i++) {
      r_blobs[i].insert(r_blobs[i].end(), add_refs[i].begin(), add_refs[i].end());
      r_step[i] += (uint32_t)add_refs[i].size();
      std::vector<uint8_t> f;
      if (docs[i].front) f.assign(docs[i].front, docs[i].front + docs[i].front_len);
      r_front[i].swap(f);
    }
    if (add) { arena_top = top; tables_valid = false; }
    rebuild_blob_tables();
    lmbe::sync();
    ran = fetched = false;
  }

  void scan(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t ncomp) {
    uint32_t tiles = (n + 1 + SCAN_TILE - 1) / SCAN_TILE;
    b_tile.ensure((size_t)tiles * ncomp * 4 + 64);
    b_tot.ensure(64 * 4);
    LM_LAUNCH(k_scan_tile, tiles, SCAN_TILE, in, out, b_tile.as<uint32_t>(), n, ncomp);
    LM_LAUNCH(k_scan_sums, 1, 64, b_tile.as<uint32_t>(), tiles, ncomp, b_tot.as<uint32_t>());
    LM_LAUNCH(k_scan_add, tiles, SCAN_TILE, out, b_tile.as<uint32_t>(), b_tot.as<uint32_t>(), n, ncomp);
  }

  int selftest() {
    lmbe::bind(sc);
    DBuf b;
    b.ensure(64 * 4);
    LM_LAUNCH(k_selftest, 64, 64, b.as<uint32_t>(), 200u);
    uint32_t h[64];
    lmbe::d2h(h, b.p, sizeof h);
    b.release();
    int bad = 0;
    for (int i = 0; i < 64; i++) bad += (int)h[i];
    return bad;
  }

  static uint32_t cdiv(uint64_t a, uint32_t b) { return (uint32_t)((a + b - 1) / b); }

  // ---- run: the device pipeline over the staged batch
  void run() {
    lmbe::bind(sc);
    times.clear();
    lmbe::reset_times();
    Dev d;
    memset(&d, 0, sizeof d);
    d.data = b_data.as<uint8_t>();
    d.front = b_front.as<uint8_t>(); d.front_off = b_front_off.as<uint64_t>();
    d.froot = b_froot.as<uint8_t>(); d.froot_off = b_froot_off.as<uint64_t>();
    d.blob_off = b_blob_off.as<uint64_t>();
    d.blob_len = b_blob_len.as<uint32_t>();
    d.doc_blob = b_doc_blob.as<uint32_t>();
    d.n_blobs = n_blobs;
    d.n_docs = n_docs;
    if (n_state_docs && !resident) { d.vvo = b_vvo.as<uint8_t>(); d.vvo_off = b_vvo_off.as<uint64_t>(); }
    results.assign(n_docs, DocResult{0, 0, 0, 0, 0, 0});
    if (n_docs == 0) { ran = true; return; }
    // resident documents whose blob lists did not change since the tables were built: only the rendered versions differ —
    // the decode / DAG stages are skipped, the saved tables are rendered again (k_dag_b's checkout part onwards)
    const bool reuse = resident && tables_valid;
    if (resident && have_prev) {
      // the version every document was at before this run (k_import_lca: LCA + DiffMode of the import, dag.rs:487-765): the
      // previous run's document records, peers and applied ends, copied before this run's kernels rebuild the tables
      b_prev_doc.ensure((size_t)n_docs * sizeof(DocMeta)); lmbe::d2d(b_prev_doc.p, b_doc_saved.p, (size_t)n_docs * sizeof(DocMeta));
      b_prev_uniq.ensure((size_t)(sv.NP + 1) * 8); lmbe::d2d(b_prev_uniq.p, b_peer_uniq.p, (size_t)(sv.NP + 1) * 8);
      b_prev_end.ensure((size_t)(sv.NP + 1) * 4); lmbe::d2d(b_prev_end.p, b_peer_end_all.p, (size_t)(sv.NP + 1) * 4);
    }
    uint32_t NB = 0, NC = 0, NO = 0, NCID = 0, NP = 0, sv_NK = 0;
    DevDag g;
    memset(&g, 0, sizeof g);
    bool span = kn.span;   // (a batch of small documents for the common kernel switches to the element-granular one below, once the documents' records are known)
    if (resident && !span) throw std::runtime_error("resident documents need the span-granular integrate kernel (LM_SPAN=0 is set)");
    uint64_t ht = 0;
    uint32_t dir_cap = 64, dir_opt = 64, pmax = 2;
    uint32_t DIR_CAP_MAX = span ? 18000u : 36000u;
    // DF_PLAIN (k_dag_a) survives only with the span kernel, for documents rendered at the latest version.  Such documents are
    // replayed by k_integrate_span_plain_sweep (default, = LM_PLAIN=2; measured -9 % against the common kernel on configs[1],
    // profiles/r02_ab_prepared.log); LM_PLAIN=1 selects k_integrate_span_plain, LM_PLAIN=0 the common kernel for every document
    int plain_mode = !span ? 0 : kn.plain;   // (resident documents: 0 = the general kernel for all, otherwise k_integrate_span_res_plain for the DF_PLAIN ones)
    bool plain_on = plain_mode == 1 || plain_mode == 2;
    bool any_plain = false, any_fused = false, want_dcnt = false;
    if (reuse) {
      d = sv.d; g = sv.g;
      d.front = b_front.as<uint8_t>(); d.front_off = b_front_off.as<uint64_t>();
      d.blob_off = b_blob_off.as<uint64_t>(); d.blob_len = b_blob_len.as<uint32_t>(); d.doc_blob = b_doc_blob.as<uint32_t>(); d.blob_doc = b_blob_doc.as<uint32_t>();
      NB = sv.NB; NC = sv.NC; NO = sv.NO; NCID = sv.NCID; NP = sv.NP; ht = sv.ht; dir_cap = sv.dir_cap; dir_opt = sv.dir_opt; pmax = sv.pmax;
      h_doc = sv.h_doc;
      lmbe::d2d(b_doc.p, b_doc_saved.p, (size_t)n_docs * sizeof(DocMeta));   // the documents' records as they were in front of the checkout
      if (ht) { lmbe::dmemset(b_ht_key.p, 0xff, ht * 8); lmbe::dmemset(b_ht_best.p, 0, ht * 8); }
      lmbe::dmemset(b_ht_cnt.p, 0, (size_t)n_docs * 4 + 4);
      if (shared_mode == 2 && NCID) LM_LAUNCH(k_cont_untouch, cdiv(NCID, 256), 256, d, NCID);
    } else {
    // 1. envelope / checksum / block count
    b_blob_status.ensure((size_t)n_blobs * 4 + 4);
    b_blob_nblk.ensure((size_t)(n_blobs + 1) * 4);
    b_blob_blk0.ensure((size_t)(n_blobs + 1) * 4);
    d.blob_status = b_blob_status.as<int32_t>();
    d.blob_nblk = b_blob_nblk.as<uint32_t>();
    d.blob_blk0 = b_blob_blk0.as<uint32_t>();
    d.blob_doc = b_blob_doc.as<uint32_t>();
    lmbe::tic(profiling);
    {
      std::vector<uint32_t> big;
      for (uint32_t i = 0; i < n_blobs; i++) if (h_blob_len[i] >= BIG_BLOB) big.push_back(i);
      // longest first: xxh32 is a serial chain per blob (≈2 ms for 2.4 MB) — a long one that starts last IS the stage's duration
      std::stable_sort(big.begin(), big.end(), [&](uint32_t a, uint32_t b) { return h_blob_len[a] > h_blob_len[b]; });
      b_blob_hash.ensure((size_t)n_blobs * 4 + 4);
      d.blob_hash = b_blob_hash.as<uint32_t>();
      if (!big.empty()) {
        b_big.ensure(big.size() * 4);
        lmbe::h2d(b_big.p, big.data(), big.size() * 4);
        LM_LAUNCH(k_hash_big_blobs, cdiv(big.size(), HASH_G), 64, d, (const uint32_t*)b_big.as<uint32_t>(), (uint32_t)big.size());
      }
    }
    if (n_blobs) LM_LAUNCH(k_frame_count, cdiv(n_blobs, 64), 64, d);
    lmbe::toc("k_frame_count", times, profiling);
    scan(d.blob_nblk, d.blob_blk0, n_blobs, 1);
    lmbe::d2h(&NB, d.blob_blk0 + n_blobs, 4);
    d.n_blocks = NB;
    // 2. block descriptors + row counts
    b_blk.ensure((size_t)(NB + 1) * sizeof(BlockDesc));
    b_bcnt.ensure((size_t)(NB + 1) * BCN * 4);
    b_boff.ensure((size_t)(NB + 1) * BCN * 4);
    d.blk = b_blk.as<BlockDesc>();
    d.bcnt = b_bcnt.as<uint32_t>();
    d.boff = b_boff.as<uint32_t>();
    lmbe::tic(profiling);
    if (n_blobs) LM_LAUNCH(k_frame_fill, cdiv(n_blobs, 64), 64, d);
    if (NB) LM_LAUNCH(k_block_desc, cdiv(NB, 64), 64, d);
    b_tot.ensure(64 * 4);
    d.dec_stat = b_tot.as<uint32_t>() + 48; d.dec_slot = kn.dec_slot; d.cut_min_rows = kn.cut_min_rows; d.vs_row_cost = kn.vs_row_cost;
    lmbe::dmemset(d.dec_stat, 0, 12);
    // LWW Map documents whose blocks hold scalar writes only are decoded WITHOUT op rows (lm_k_map_fused.h): found here, in front of
    // the row counts (their blocks ask for no op rows and for key rows of their root containers' names only).  Batches only, and only
    // while the side engine can take a document the fused kernel gives up on (DF_REDO)
    n_fused = 0;
    h_fused.assign(n_docs, 0);
    const bool mf_on = kn.map_fused && kn.redo && kn.lww_lds && kn.ht_opt && kn.ht_opt <= LWW_LDS_CAP && !resident && st_valid && NB;
    if (mf_on) {
      b_blk_kind.ensure((size_t)(NB + 1) * 4); b_doc_fused.ensure((size_t)n_docs + 16);
      d.blk_kind = b_blk_kind.as<uint32_t>();
      LM_LAUNCH(k_block_kind, cdiv(NB, 64), 64, d);
      LM_LAUNCH(k_doc_kind, cdiv(n_docs, 64), 64, d, b_doc_fused.as<uint8_t>(), kn.mf_min_rows, kn.mf_chg_ratio);
      d.doc_fused = b_doc_fused.as<uint8_t>();
    }
    if (NB) LM_LAUNCH(k_block_count, cdiv(NB, 64), 64, d);
    lmbe::toc("k_frame_fill+k_block_count", times, profiling);
    scan(d.bcnt, d.boff, NB, BCN);
    uint32_t tot[BCN];
    lmbe::d2h(tot, d.boff + (uint64_t)NB * BCN, sizeof tot);
    if (mf_on) { lmbe::d2h(h_fused.data(), b_doc_fused.p, n_docs); for (uint32_t i = 0; i < n_docs; i++) n_fused += h_fused[i]; }
    if (!n_fused) d.doc_fused = nullptr;
    const uint32_t NF = n_fused ? tot[BC_MAPOP] : 0u;    // record rows of the fused documents, behind the op rows
    const uint32_t KF = n_fused * LWW_LDS_CAP;           // their slot-numbered key rows, behind the key rows
    uint32_t dec_stat[3] = {0, 0, 0};
    lmbe::d2h(dec_stat, d.dec_stat, 12);
    uint32_t ND = tot[BC_DEP], NK = tot[BC_KEY];
    NC = tot[BC_CHG]; NO = tot[BC_OP]; NCID = tot[BC_CID]; NP = tot[BC_PEER];
    // 3. row tables
    b_chg.ensure((size_t)(NC + 1) * sizeof(ChangeRow));
    b_dep_peer.ensure((size_t)(ND + 1) * 4); b_dep_ctr.ensure((size_t)(ND + 1) * 4); b_dep_ci.ensure((size_t)(ND + 1) * 4);
    if ((uint64_t)NO + NF > 0xfffffff0ull || (uint64_t)NK + KF > 0xfffffff0ull) throw std::runtime_error("batch too large for 32-bit row indices");
    b_op.ensure((size_t)(NO + NF + 1) * sizeof(OpRow)); b_op_val.ensure((size_t)(NO + NF + 1) * 8); b_op_blk.ensure((size_t)(NO + NF + 1) * 4);
    b_key_off.ensure((size_t)(NK + KF + 1) * 8); b_key_len.ensure((size_t)(NK + KF + 1) * 4);
    d.n_op_rows = NO; sv_NK = NK;
    b_cid_raw.ensure((size_t)(NCID + 1) * 16); b_cid_map.ensure((size_t)(NCID + 1) * 4);
    b_peer_raw.ensure((size_t)(NP + 1) * 8); b_peer_map.ensure((size_t)(NP + 1) * 4);
    b_doc.ensure((size_t)n_docs * sizeof(DocMeta));
    b_peer_uniq.ensure((size_t)(NP + 1) * 8);
    for (DBuf* b : {&b_peer_end, &b_peer_ext, &b_peer_base, &b_peer_end_all, &b_elem_base, &b_peer_chg0, &b_peer_chg1}) b->ensure((size_t)(NP + 1) * 4);
    b_cont.ensure((size_t)(NCID + 1) * sizeof(ContRow));
    for (DBuf* b : {&b_chg_sorted, &b_chg_lamport, &b_chg_skip, &b_chg_flag, &b_node_first, &b_node_last, &b_node_order, &b_chg_node,
                    &b_node_done, &b_node_lam})
      b->ensure((size_t)(NC + 1) * 4);
    b_blk_sorted.ensure((size_t)(NB + 1) * 4);
    b_chg_mask.ensure((size_t)(NC + 1) * 8);
    b_cont_root0.ensure((size_t)(NCID + 1) * 4); b_cont_nroot.ensure((size_t)(NCID + 1) * 4);
    d.chg = b_chg.as<ChangeRow>(); d.dep_peer = b_dep_peer.as<uint32_t>(); d.dep_ctr = b_dep_ctr.as<uint32_t>(); d.dep_ci = b_dep_ci.as<uint32_t>();
    d.op = b_op.as<OpRow>(); d.op_val = b_op_val.as<uint64_t>(); d.op_blk = b_op_blk.as<uint32_t>();
    d.key_off = b_key_off.as<uint64_t>(); d.key_len = b_key_len.as<uint32_t>();
    d.cid_raw = b_cid_raw.as<uint32_t>(); d.cid_map = b_cid_map.as<uint32_t>();
    d.peer_raw = b_peer_raw.as<uint64_t>(); d.peer_map = b_peer_map.as<uint32_t>();
    d.doc = b_doc.as<DocMeta>(); d.peer_uniq = b_peer_uniq.as<uint64_t>();
    d.peer_end = b_peer_end.as<uint32_t>(); d.peer_ext = b_peer_ext.as<uint32_t>(); d.peer_base = b_peer_base.as<uint32_t>(); d.peer_end_all = b_peer_end_all.as<uint32_t>(); d.elem_base = b_elem_base.as<uint32_t>();
    d.peer_chg0 = b_peer_chg0.as<uint32_t>(); d.peer_chg1 = b_peer_chg1.as<uint32_t>();
    d.cont = b_cont.as<ContRow>();
    d.chg_sorted = b_chg_sorted.as<uint32_t>(); d.chg_lamport = b_chg_lamport.as<uint32_t>();
    d.chg_skip = b_chg_skip.as<uint32_t>(); d.chg_flag = b_chg_flag.as<uint32_t>(); d.chg_mask = b_chg_mask.as<uint32_t>();
    d.node_first = b_node_first.as<uint32_t>(); d.node_last = b_node_last.as<uint32_t>(); d.node_order = b_node_order.as<uint32_t>();
    d.cont_root0 = b_cont_root0.as<uint32_t>(); d.cont_nroot = b_cont_nroot.as<uint32_t>();
    g.blk_sorted = b_blk_sorted.as<uint32_t>(); g.chg_node = b_chg_node.as<uint32_t>();
    g.node_done = b_node_done.as<uint32_t>(); g.node_lam = b_node_lam.as<uint32_t>();
    lmbe::dmemset(b_chg_flag.p, 0, (size_t)(NC + 1) * 4);
    lmbe::dmemset(b_chg_mask.p, 0, (size_t)(NC + 1) * 8);
    lmbe::dmemset(b_chg_skip.p, 0, (size_t)(NC + 1) * 4);
    lmbe::dmemset(b_chg_lamport.p, 0, (size_t)(NC + 1) * 4);
    lmbe::tic(profiling);
    // one wave per group of DEC_G blocks, staged through LDS (lm_k_decode_wave.h); LM_DECODE=0 selects the one-lane-per-block
    // decoder (kept as the second, independently structured implementation the parity suites also run)
    if (NB) {
#if defined(LM_PROF_DEC) || defined(LM_PROF_DAG)   // experiment builds: cycle accounting of the decoder's phases / of k_dag_a's passes (tests/tools/gpu_prof_dec.py, gpu_prof_dag.py)
      b_prof.ensure((size_t)n_docs * 16 * 8); d.prof = b_prof.as<unsigned long long>(); lmbe::dmemset(b_prof.p, 0, (size_t)n_docs * 16 * 8);
#endif
      if (n_fused) LM_LAUNCH(k_block_head, cdiv(NB, 64), 64, d);   // the fused documents' blocks: header, meta, container ids
      if (n_fused == n_docs) {}                                    // (nothing left for the row decoders)
      else if (!kn.decode_wave) LM_LAUNCH(k_block_decode, cdiv(NB, 64), 64, d);
      else {
        uint32_t slot_cap = kn.dec_slot;
        // heads beyond the default slot: those groups get a launch of their own.  LM_DEC_BIG_MODE=1 (default): slots sized for their
        // op / delete-start COLUMNS (the head is parsed from HBM by role 0) — the decoder is bound by its dependent round trips at
        // the occupancy its LDS allows, and a Map block's columns (≈650 bytes) need half the default slot: measured on configs[2],
        // 2,048 documents: one launch 50.7 ms; slots of 4 KB for the whole heads 116 ms (4 waves per CU).  =2: slots sized for the
        // largest head, capped by LM_DEC_SLOT_BIG
        uint32_t slot_big = 0;
        // (only when the batch is MADE of such blocks: a group that mixes one of them with ordinary blocks needs the ordinary slot —
        // the MovableList batch, where a few blocks carry a 300-item insert, decoded in 7.7 ms with the second launch and 3.9 without)
        if (dec_stat[0] && kn.dec_slot_big && (kn.dec_big_mode == 2 || (uint64_t)dec_stat[0] * 10 >= (uint64_t)NB * 9)) {
          if (kn.dec_big_mode == 2) { slot_big = (dec_stat[1] + 15u) & ~15u; if (slot_big > kn.dec_slot_big) slot_big = kn.dec_slot_big; }
          else { slot_big = (dec_stat[2] + 15u) & ~15u; if (slot_big < 64) slot_big = 64; if (slot_big >= slot_cap) slot_big = 0; }
        }
        if (!(slot_big && dec_stat[0] == NB))   // (a batch in which EVERY block is of the second launch's kind needs no first)
          LM_LAUNCH_DYN(k_block_decode_wave, cdiv(NB, DEC_G), 64, (size_t)DEC_G * slot_cap + DEC_LDS_FIXED + DEC_G * DEC_KINDS, d, slot_cap, 0u, slot_big ? slot_cap : 0xffffffffu);
        if (slot_big) LM_LAUNCH_DYN(k_block_decode_wave_map, cdiv(NB, DEC_G), 64, (size_t)DEC_G * slot_big + DEC_LDS_FIXED + DEC_G * DEC_KINDS, d, slot_big, slot_cap, 0xffffffffu);
      }
      // blocks a decoder rejected with DecodeError are read once more, sequentially, by the decode that names corruption where it stands
      // (lm_k_decode.h EXACT): the reference's error for an undefined value tag / a nested key index beyond the key table is
      // DecodeDataCorruptionError.  One lane per block; every healthy block leaves at once.  LM_RECLASS=0 switches it off.
      if (kn.reclass) LM_LAUNCH(k_block_reclassify, cdiv(NB, 64), 64, d);
    }
    lmbe::toc("k_block_decode", times, profiling);
    lmbe::tic(profiling);
    LM_LAUNCH(k_doc_ranges, cdiv(n_docs, 64), 64, d);
    LM_LAUNCH(k_doc_tables, n_docs, 64, d);
    uint32_t nmax = NO > NC ? NO : NC;
    if (nmax) LM_LAUNCH(k_remap, cdiv(nmax, 256), 256, d, NO, NC);
    lmbe::toc("k_doc_tables+k_remap", times, profiling);
    lmbe::tic(profiling);
    LM_LAUNCH(k_dag_a, n_docs, 64, d, g);
    lmbe::toc("k_dag_a", times, profiling);
    // 4. per-doc pool sizing on the host (one small round trip)
    h_doc.resize(n_docs);
    lmbe::d2h(h_doc.data(), d.doc, (size_t)n_docs * sizeof(DocMeta));
    // integrate kernel: span-granular leaves (lm_k_integrate_span.h) by default; LM_SPAN=0 selects the element-granular
    // kernel (lm_k_integrate.h), kept as the second implementation the parity suites also run
    // (directory entries that fit the 160 KiB LDS of a CU next to 3·MAX_PEERS words: one word per entry in the
    // element-granular kernel, two in the span-granular one: DIR_CAP_MAX)
    if (span && !resident && kn.span_auto) {
      // SMALL documents that the common span kernel would replay — style anchors, sliced changes, checkouts, MovableLists; a few
      // thousand op rows from several peers that sync every few dozen ops — are replayed faster by the element-granular kernel
      // (lm_k_integrate.h): its leaves are plain arrays of elements, a tracker move touches no run bookkeeping.  Measured on one
      // box (profiles/r04_small_documents_element_kernel.log): configs[3] integrate 48.9 -> 39.9 ms per 12,500 documents, the
      // MovableList batch 15.2 -> 13.4 ms per 4,096.  The mode is the batch's (leaf record, Text payload layout): taken when no
      // document is large, the common kernel's documents hold at least half of the batch's op rows …
      uint64_t rows_all = 0, rows_common = 0, atoms_common = 0;
      uint32_t rows_max = 0, elems_max = 0;
      for (uint32_t i = 0; i < n_docs; i++) {
        const DocMeta& m = h_doc[i];
        if (m.status != ST_OK) continue;
        const bool checked_out = h_front_off.size() > i + 1 && h_front_off[i + 1] > h_front_off[i];
        rows_all += m.n_op;
        if (!(m.flags & DF_PLAIN) || checked_out || !plain_on) { rows_common += m.n_op; atoms_common += m.atoms; }
        rows_max = m.n_op > rows_max ? m.n_op : rows_max;
        elems_max = m.n_elems > elems_max ? m.n_elems : elems_max;
      }
      // … and their op rows are SHORT — fewer than three op ids per row: single list items, one- or two-character edits.  (Rich text of
      // the same size typed in runs — 5-6 ids per row, tests/tools/gpu_small_docs.py — stays 20 % faster in the span kernel.)
      if (rows_all && rows_max < 4096 && elems_max < 65536 && rows_common * 2 >= rows_all && atoms_common < 3 * rows_common) {
        span = false; DIR_CAP_MAX = 36000u; plain_mode = 0; plain_on = false;
      }
    }
    d.span = span ? 1u : 0u;
    d.res_vis = resident ? 1u : 0u;
    uint64_t elem = 0, leaves = 0, vvh = 0;
    h_ht0.assign(n_docs, 0); h_ht_cap.assign(n_docs, 0); h_ht_full.assign(n_docs, 0);
    for (uint32_t i = 0; i < n_docs; i++) {
      DocMeta& m = h_doc[i];
      bool ok = m.status == ST_OK;
      // (resident documents keep the flag: which version is rendered changes from run to run — the kernels look at the frontiers)
      if (!plain_on || !ok || (!resident && h_front_off.size() > i + 1 && h_front_off[i + 1] > h_front_off[i])) m.flags &= ~(DF_PLAIN | DF_FUSED);
      if (resident || plain_mode != 2 || !kn.fuse_rows) m.flags &= ~DF_FUSED;
      any_plain |= (m.flags & DF_PLAIN) != 0;
      any_fused |= (m.flags & DF_FUSED) != 0;
      want_dcnt |= ok && !(m.flags & (DF_PLAIN | DF_MOVABLE));   // replayed by the common kernel: its tracker may move by a version pass (ts_sweep_version)
      m.elem0_lo = (uint32_t)elem; m.elem0_hi = (uint32_t)(elem >> 32);
      if (resident) {
        // the document's slice of the element arena stays where it is while it is large enough (loc[] and the payload slots of
        // earlier runs stay valid, k_res_layout); a larger one is handed out with room to grow
        // (k_res_layout gives every peer a region of extent x 1.5 + 64 slots: a fresh layout needs at most 1.5 x atoms + 68 per peer)
        uint64_t need = (uint64_t)m.atoms + m.atoms / 2 + 68ull * (m.n_peers + 1) + 8;
        if (ok && need > tk_elem_cap[i]) {
          uint64_t cap = ((need + need / 2 + 80ull * 8) + 15) & ~15ull;   // (slices start at multiples of 16 element slots)
          if (cap > 0xfffffff0ull) throw std::runtime_error("resident element slice beyond 32-bit indices");
          tk_elem0[i] = elem_top; tk_elem_cap[i] = (uint32_t)cap; elem_top += cap;
        }
        m.elem0_lo = (uint32_t)tk_elem0[i]; m.elem0_hi = (uint32_t)(tk_elem0[i] >> 32);
      }
      // every split leaves both halves with >= 32 elements; each container starts with one (possibly small) leaf
      uint32_t lc = ok ? m.n_elems / 32 + 2 * m.n_cont + 2 : 0;
      if (ok && span) {
        // span-granular leaves: items <= one per insert row + one per boundary that ever cuts a run (op boundaries, checkout
        // cuts); a split leaves both halves with 32 items
        uint64_t items = 3ull * m.n_op + 2ull * m.n_nodes * m.n_peers + 64;
        uint64_t lcs = items / 32 + 2 * m.n_cont + 2;
        if (lcs < lc) lc = (uint32_t)lcs;
      }
      if (leaves + lc > 0xfffffff0ull) throw std::runtime_error("batch too large for 32-bit pool indices");
      m.leaf0 = (uint32_t)leaves; m.leaf_cap = lc;
      if (resident) {
        // the document's region of the resident leaf pool and its tracker record: kept while they are large enough, handed out
        // anew (with room to grow) otherwise — the stored tracker is then not used, the document is replayed from the empty version
        if (ok && lc > tk_leaf_cap[i]) {
          uint32_t cap = lc + lc / 2 + 8;
          if ((uint64_t)leaf_top + cap > 0xfffffff0ull) throw std::runtime_error("resident leaf pool beyond 32-bit indices");
          tk_leaf0[i] = leaf_top; tk_leaf_cap[i] = cap; leaf_top += cap; tk_reset[i] = 1;
        }
        m.leaf0 = tk_leaf0[i]; m.leaf_cap = ok ? tk_leaf_cap[i] : 0;
        if (ok && (tk_pcap[i] == 0 || m.n_peers > tk_pcap[i] || m.n_cont > tk_ccap[i])) tk_grow(i, m.n_peers, m.n_cont);   // (a record for every document that runs, an empty one included)
      }
      m.vvh0_lo = (uint32_t)vvh; m.vvh0_hi = (uint32_t)(vvh >> 32);
      if (ok) { elem += ((uint64_t)m.atoms + 15) & ~15ull; leaves += lc; vvh += (uint64_t)m.n_nodes * m.n_peers; }   // element slices start at multiples of 16 slots (k_integrate_span clears loc[] four entries per store; tb[] slices start 16-byte aligned)
      if (lc > dir_cap) dir_cap = lc;
      // optimistic LDS directory: leaves are ≈3/4 full in practice (≈48 elements); sized for 40 per leaf
      uint32_t lo = ok ? m.n_elems / 40 + 2 * m.n_cont + 16 : 0;
      if (ok && span) { uint32_t los = (uint32_t)((3ull * m.n_op / 2) / 40) + 2 * m.n_cont + 16; if (los < lo) lo = los; }
      if (lo > lc) lo = lc;
      if (lo > dir_opt) dir_opt = lo;
      if (ok && m.n_peers > pmax) pmax = m.n_peers;
      // LWW table: 2× the doc's Map op rows rounded up to a power of two
      uint32_t cap = 0;
      if (ok && m.n_mapop) { cap = 64; while (cap < 2 * m.n_mapop) cap <<= 1; }
      h_ht_full[i] = cap;
      // … optimistically capped (k_map_lww): an LWW history writes few keys many times — 1,024 keys in 160,000 rows on configs[2],
      // 8 MB of table per document against 128 KB; a document with more keys than half the cap gets the full table in a second pass.
      // (Not for resident documents, whose tables outlive the run, nor for MovableLists, whose elements enter after the integrate stage.)
      if (kn.ht_opt && !resident && !(m.flags & DF_MOVABLE) && cap > kn.ht_opt) cap = kn.ht_opt;
      h_ht0[i] = ht; h_ht_cap[i] = cap;
      ht += cap;
    }
    if (resident) tk_flush();
    if (dir_cap > DIR_CAP_MAX) dir_cap = DIR_CAP_MAX;  // larger documents are reported LM_UNSUPPORTED by k_integrate
    // longest first: workgroups are dispatched in index order, so the integrate stage takes its documents by descending op
    // rows — a batch of mixed sizes does not end with a few long replays that started in the last round
    {
      std::vector<uint32_t> order(n_docs);
      for (uint32_t i = 0; i < n_docs; i++) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return h_doc[a].n_op > h_doc[b].n_op; });
      b_order.ensure((size_t)n_docs * 4 + 4);
      lmbe::h2d(b_order.p, order.data(), (size_t)n_docs * 4);
      d.doc_order = b_order.as<uint32_t>();
    }
    lmbe::h2d(d.doc, h_doc.data(), (size_t)n_docs * sizeof(DocMeta));
    if (resident) {
      // arenas that outlive the run: element payloads (a MovableList move keeps the id of the item it deleted in its own slot),
      // the leaf pool and the two generations of the leaf directories
      b_cp.ensure_keep((elem_top + 1) * 4, b_cp.cap); b_loc.ensure_keep((elem_top + 4) * 4, b_loc.cap); b_dcnt.ensure((elem_top + 1) * 4); b_tb.ensure_keep(elem_top + 16, b_tb.cap);
      b_it.ensure_keep(((size_t)leaf_top + 1) * SP_REC * 4, b_it.cap);
      for (DBuf* b : {&b_dir_out, &b_dir_out2, &b_dir_b, &b_dir_b2}) b->ensure_keep(((size_t)leaf_top + 1) * 4, b->cap);
    } else {
      b_cp.ensure((elem + 1) * 4); b_loc.ensure((elem + 4) * 4); b_tb.ensure(elem + 16);
      if (LM_BATCH_VSWEEP && want_dcnt && span && kn.version_sweep) b_dcnt.ensure((elem + 1) * 4);   // (the batch kernels' only reader is ts_goto's version pass, compiled out by default: ADVICE r4)
      b_it.ensure((leaves + 1) * (span ? (size_t)SP_REC : 256) * 4);
      b_dir_out.ensure((leaves + 1) * 4);
    }
    b_lf_chunk.ensure(leaves + 64);
    b_vvh.ensure((vvh + 1) * 4);
    b_prof.ensure((size_t)n_docs * 16 * 8);
    d.prof = b_prof.as<unsigned long long>();
#if !defined(LM_PROF_DEC) && !defined(LM_PROF_DAG)
    lmbe::dmemset(b_prof.p, 0, (size_t)n_docs * 16 * 8);
#endif
    b_ht_key.ensure((ht + 1) * 8); b_ht_best.ensure((ht + 1) * 8); b_ht_pfx.ensure((ht + 1) * 8); b_ht_list.ensure((ht + 1) * 8);   // per doc 2·cap entries: claimed slots [0, cap/2) + sort scratch
    b_ht0.ensure((size_t)n_docs * 8); b_ht_cap.ensure((size_t)n_docs * 4); b_ht_cnt.ensure((size_t)n_docs * 4 + 4);
    lmbe::h2d(b_ht0.p, h_ht0.data(), (size_t)n_docs * 8);
    lmbe::h2d(b_ht_cap.p, h_ht_cap.data(), (size_t)n_docs * 4);
    d.cp = b_cp.as<uint32_t>(); d.loc = b_loc.as<uint32_t>(); d.dcnt = ((resident || (LM_BATCH_VSWEEP && want_dcnt && span)) && kn.version_sweep) ? b_dcnt.as<uint32_t>() : nullptr; d.tb = b_tb.as<uint8_t>();
    d.it = b_it.as<uint32_t>();
    d.dir_out = b_dir_out.as<uint32_t>();
    d.lf_chunk = b_lf_chunk.as<uint8_t>();
    d.vvh = b_vvh.as<uint32_t>();
    d.ht_key = b_ht_key.as<unsigned long long>(); d.ht_best = b_ht_best.as<unsigned long long>(); d.ht_pfx = b_ht_pfx.as<unsigned long long>();
    d.ht0 = b_ht0.as<uint64_t>(); d.ht_cap = b_ht_cap.as<uint32_t>();
    d.ht_list = b_ht_list.as<uint32_t>(); d.ht_cnt = b_ht_cnt.as<uint32_t>();
    lmbe::dmemset(b_ht_cnt.p, 0, (size_t)n_docs * 4 + 4);
    // loc[] := NONE.  A batch: one memset on the stream, in front of the payload fill — it runs beside the other stream's /
    // context's issue-bound integrate kernel instead of inside this one (LM_LOC_MEMSET=0: every document's wave clears its own
    // slice, as resident documents always do — they keep loc[] between runs).  cp[] needs no fill: every element that can be
    // placed was written by k_elem_fill
    d.loc_cleared = (!resident && span && kn.loc_memset) ? 1u : 0u;
    d.no_linear = kn.linear ? 0u : 1u;
    d.posdel_redo = (kn.redo && kn.posdel && st_valid && (resident ? shared_mode != 0 : !span)) ? 1u : 0u;
    d.posdel_off = nullptr; d.pd_row_idx = nullptr;
    if (span && !resident && kn.posdel) {
      // a document staged on a snapshot's state (lm_snapshot_base.h) deletes base content by position as a matter of course — every
      // delete row of its updates that names an element of the base: its list holds two pieces per op row (+ a row index, so a
      // version move finds a row's pieces without a scan); everybody else: PD_CAP pieces, damaged rows only.  A list that overflows
      // fails the document, and a state document that fails is replayed from its history (redo_hist below)
      uint64_t pieces = 0;
      bool big = false;
      std::vector<uint64_t> h_pd_off(n_docs + 1);
      for (uint32_t i = 0; i < n_docs; i++) {
        h_pd_off[i] = pieces;
        const bool on_state = n_state_docs && h_vvo_off.size() > (size_t)i + 1 && h_vvo_off[i + 1] > h_vvo_off[i] && h_doc[i].n_op > PD_CAP / 2;
        pieces += on_state ? (uint64_t)kn.pd_state_pieces * h_doc[i].n_op + PD_CAP + 1 : (uint64_t)PD_CAP;
        big |= on_state;
      }
      h_pd_off[n_docs] = pieces;
      b_posdel.ensure((size_t)pieces * 3 * 4 + 16); d.posdel = b_posdel.as<uint32_t>();
      if (big) {
        b_pd_off.ensure(((size_t)n_docs + 1) * 8); lmbe::h2d(b_pd_off.p, h_pd_off.data(), ((size_t)n_docs + 1) * 8);
        b_pd_row.ensure(((size_t)NO + 1) * 4); lmbe::dmemset(b_pd_row.p, 0xff, ((size_t)NO + 1) * 4);
        d.posdel_off = b_pd_off.as<uint64_t>(); d.pd_row_idx = b_pd_row.as<uint32_t>();
      }
    }
    if (d.loc_cleared && elem) {
      // (k_fill_words, not hipMemsetAsync: its fill kernel writes at half of what HBM takes; the buffers are padded to 16 bytes.  Clearing
      // speculatively on a second stream beside the decode stage — as many words as the last run needed, joined here — was built and
      // measured: the step got SLOWER, 23.0 against 22.7 ms (k_frame_count 0.30 -> 0.72 ms under the 2 GB of stores; profiles/r06_side_fill_ab.log))
      const uint64_t n4 = ((uint64_t)elem + 3) / 4;
      const uint32_t blocks = (uint32_t)std::min<uint64_t>((n4 + 255) / 256, 4096);
      LM_LAUNCH(k_fill_words, blocks, 256, b_loc.as<uint32_t>(), n4, NONE, blocks * 256u);
    }
    if (ht) { lmbe::dmemset(b_ht_key.p, 0xff, ht * 8); lmbe::dmemset(b_ht_best.p, 0, ht * 8); }
      if (resident) {
        sv.d = d; sv.g = g; sv.NB = NB; sv.NC = NC; sv.NO = NO; sv.NCID = NCID; sv.NP = NP; sv.ht = ht;
        sv.dir_cap = dir_cap; sv.dir_opt = dir_opt; sv.pmax = pmax; sv.h_doc = h_doc;
      }
    }   // (!reuse)
    lmbe::dmemset(b_cont_root0.p, 0, (size_t)(NCID + 1) * 4);
    lmbe::dmemset(b_cont_nroot.p, 0, (size_t)(NCID + 1) * 4);
    // 5. causal order, element payloads, LWW, integrate
    DevRes rs;
    memset(&rs, 0, sizeof rs);
    if (resident) {
      upload_res();
      rs.doc = b_res.as<ResDoc>(); rs.tk = b_tk.as<uint32_t>();
      if (!reuse) {
        h_old_blobs.resize(n_docs);
        for (uint32_t i = 0; i < n_docs; i++) h_old_blobs[i] = (uint32_t)r_blobs[i].size() - r_step[i];
        b_old_blobs.ensure((size_t)n_docs * 4 + 4);
        lmbe::h2d(b_old_blobs.p, h_old_blobs.data(), (size_t)n_docs * 4);
        d.res_old_blobs = b_old_blobs.as<uint32_t>();
        sv.d.res_old_blobs = d.res_old_blobs;
      }
      if (!reuse) { b_elem_cap.ensure((size_t)(NP + 1) * 4); d.elem_cap = b_elem_cap.as<uint32_t>(); sv.d.elem_cap = d.elem_cap; }
      LM_LAUNCH(k_res_layout, n_docs, 64, d, rs);   // (a run that reuses its tables: the layout is the previous run's, unless that run failed for the document)
    }
    lmbe::tic(profiling);
    d.fuse = nullptr;
    if (!resident && any_fused && NO) {
      // one-change-per-keystroke documents: rows of one node chained into runs (lm_k_fuse.h); their replay is k_integrate_span_plain_fuse's
      b_fuse.ensure(((size_t)NO + 1) * 8);
      d.fuse = b_fuse.as<uint32_t>();
      LM_LAUNCH(k_fuse_rows, cdiv(NO, 256), 256, d, NO);
    }
    if (!resident) LM_LAUNCH(k_dag_b, n_docs, 64, d, g, 0u);
    else {
      if (!reuse) {
        LM_LAUNCH(k_dag_b, n_docs, 64, d, g, 1u);    // the DAG pass; the documents' records are kept as they are in front of the checkout
        b_doc_saved.ensure((size_t)n_docs * sizeof(DocMeta));
        lmbe::d2d(b_doc_saved.p, b_doc.p, (size_t)n_docs * sizeof(DocMeta));
      }
      LM_LAUNCH(k_dag_b, n_docs, 64, d, g, 2u);      // the checkout of this run
    }
    lmbe::toc("k_dag_b", times, profiling);
    if (resident) {
      std::vector<uint64_t> loff(n_docs + 1, 0);
      for (uint32_t i = 0; i < n_docs; i++) { uint64_t N = h_doc[i].status == ST_OK ? h_doc[i].n_nodes : 0; loff[i + 1] = loff[i] + 3 * (8 * N + 64) + 3 * (4 * N + 64) + N + 16; }
      b_lca_off.ensure((size_t)(n_docs + 1) * 8); lmbe::h2d(b_lca_off.p, loff.data(), (size_t)(n_docs + 1) * 8);
      b_lca_scratch.ensure((loff[n_docs] + 16) * 4);
      b_lca_out.ensure((size_t)n_docs * LCA_OUT * 4 + 16);
      DevLca lc;
      lc.out = b_lca_out.as<uint32_t>(); lc.scratch = b_lca_scratch.as<uint32_t>(); lc.scratch_off = b_lca_off.as<uint64_t>();
      lc.prev_doc = have_prev ? b_prev_doc.as<DocMeta>() : nullptr; lc.prev_uniq = b_prev_uniq.as<uint64_t>(); lc.prev_end = b_prev_end.as<uint32_t>();
      lmbe::tic(profiling);
      LM_LAUNCH(k_import_lca, n_docs, 64, d, g, lc);
      lmbe::toc("k_import_lca", times, profiling);
    }
    lmbe::tic(profiling);
    // (a batch without a single sequence element — LWW Map documents — has no payload slot to fill: one wave per block that finds
    // nothing was 8 of configs[2]'s 243 ms)
    bool any_elems = resident;
    for (uint32_t i = 0; i < n_docs && !any_elems; i++) any_elems = h_doc[i].status == ST_OK && h_doc[i].n_elems != 0;
    if (NB && !reuse && any_elems) LM_LAUNCH(k_elem_fill, NB, 64, d);
    lmbe::toc("k_elem_fill", times, profiling);
    lmbe::tic(profiling);
    b_tot.ensure(64 * 4);
    uint32_t* retry_cnt = b_tot.as<uint32_t>() + 32;
    lmbe::dmemset(retry_cnt, 0, 16);   // [0] documents to re-run with the worst-case directory, [1] resident documents replayed from the empty version, [2] documents whose optimistic LWW table filled up, [3] documents with a delete row that does not match its position (k_integrate_span_pos)
    if (n_fused && ht && !reuse) {
      // workgroup -> fused document, most rows first; every document's slot-numbered key rows
      std::vector<uint32_t> mfd, mfk(n_docs, 0);
      uint32_t rank = 0;
      for (uint32_t i = 0; i < n_docs; i++) if (h_fused[i]) { mfk[i] = sv_NK + rank * LWW_LDS_CAP; rank++; if (h_doc[i].status == ST_OK && h_doc[i].n_mapop) mfd.push_back(i); }
      std::stable_sort(mfd.begin(), mfd.end(), [&](uint32_t a, uint32_t b) { return h_doc[a].n_op > h_doc[b].n_op; });
      b_mf_docs.ensure(mfd.size() * 4 + 4); b_mf_key0.ensure((size_t)n_docs * 4 + 4);
      if (!mfd.empty()) lmbe::h2d(b_mf_docs.p, mfd.data(), mfd.size() * 4);
      lmbe::h2d(b_mf_key0.p, mfk.data(), (size_t)n_docs * 4);
      DevMf mf; mf.docs = b_mf_docs.as<uint32_t>(); mf.doc_fused = d.doc_fused; mf.key0 = b_mf_key0.as<uint32_t>(); mf.stop_after = 0;
      if (const char* e = getenv("LM_MF_STOP")) mf.stop_after = (uint32_t)atoi(e);   // (timing experiments: the results are not the documents')
      if (!mfd.empty()) LM_LAUNCH_DYN(k_map_fused, (uint32_t)mfd.size(), MF_WG, (size_t)MF_LDS, d, mf, retry_cnt);
    }
    if (NO && ht) {   // (ht == 0: no document holds a Map / MovableList-LWW / out-of-scope row, k_dag_a)
      // documents whose (optimistic) table fits LDS are resolved by a workgroup each (lm_k_lww_doc.h); the others — resident
      // documents, MovableLists, tables sized for every row — one row per lane in their HBM tables
      bool any_lds = false, any_hbm = false;
      for (uint32_t i = 0; i < n_docs; i++) if (h_doc[i].status == ST_OK && h_doc[i].n_mapop && !h_fused[i]) {
        if (kn.lww_lds && !resident && h_ht_cap[i] <= LWW_LDS_CAP && !(h_doc[i].flags & DF_MOVABLE)) any_lds = true; else any_hbm = true;
      }
      if (any_lds) LM_LAUNCH_DYN(k_map_lww_doc, n_docs, LWW_WG, (size_t)LWW_LDS_CAP * 24 + (MAX_PEERS + MAX_CONTAINERS / 32 + 8) * 4, d, retry_cnt);
      if (any_hbm) LM_LAUNCH(k_map_lww, cdiv(NO, 256), 256, d, NO, retry_cnt, 0u, any_lds ? 1u : 0u);
    }
    lmbe::toc("k_map_lww", times, profiling);
    lmbe::tic(profiling);
    dir_cap = (dir_cap + 3) & ~3u;
    dir_opt = (dir_opt + 3) & ~3u;
    if (dir_opt > DIR_CAP_MAX) dir_opt = DIR_CAP_MAX;
    size_t lds_pad = kn.lds_pad;
    if (kn.no_opt_dir) dir_opt = dir_cap;
    if (kn.dir_opt_max) { uint32_t mx = kn.dir_opt_max; if (mx >= 4 && mx < dir_opt) dir_opt = mx & ~3u; }
    const size_t dir_words = span ? 2 : 1;   // LDS words per directory entry
    // documents that hold a MovableList are replayed by the kernel that knows move rows (k_integrate_span_ml), the others by
    // the common one; each kernel's waves leave the other's documents at once
    bool any_ml = false, any_common = false;
    if (resident) any_plain = false;
    for (uint32_t i = 0; i < n_docs; i++) if (h_doc[i].status == ST_OK) {
      uint32_t fl = h_doc[i].flags;
      if (resident && (fl & DF_PLAIN) && h_front_off[i + 1] > h_front_off[i]) fl &= ~DF_PLAIN;   // rendered at a checked-out version: the general instantiation
      any_ml |= (fl & DF_MOVABLE) != 0; any_common |= (fl & (DF_MOVABLE | DF_PLAIN)) == 0;
      if (resident) any_plain |= (fl & (DF_MOVABLE | DF_PLAIN)) == DF_PLAIN;
    }
    if (resident) {
      // every document's tracker record (uploaded above) and its two directory generations: this run reads what the previous run wrote
      DBuf& wa = dir_parity ? b_dir_out2 : b_dir_out; DBuf& ra = dir_parity ? b_dir_out : b_dir_out2;
      DBuf& wb = dir_parity ? b_dir_b2 : b_dir_b;     DBuf& rb = dir_parity ? b_dir_b : b_dir_b2;
      d.dir_out = wa.as<uint32_t>(); rs.dir_a_prev = ra.as<uint32_t>(); rs.dir_b = wb.as<uint32_t>(); rs.dir_b_prev = rb.as<uint32_t>();
      const size_t lds = (size_t)(2 * dir_opt + (dir_opt >> SD_BSH) + 2 + PD_LDS + 6 * pmax) * 4 + lds_pad;
      if (any_plain)
        LM_LAUNCH_DYN(k_integrate_span_res_plain, n_docs, 64, lds, d, g, dir_opt, pmax, (const OpRow*)d.op, (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted,
                      (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt, rs);
      if (any_common || !(any_ml || any_plain))
        LM_LAUNCH_DYN(k_integrate_span_res, n_docs, 64, lds, d, g, dir_opt, pmax, (const OpRow*)d.op, (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted,
                      (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt, rs);
      if (any_ml)
        LM_LAUNCH_DYN(k_integrate_span_res_ml, n_docs, 64, lds, d, g, dir_opt, pmax, (const OpRow*)d.op, (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted,
                      (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt, rs);
    } else if (span) {
      if (any_plain && plain_mode == 2) {
        LM_LAUNCH_DYN(k_integrate_span_plain_sweep, n_docs, 64, (size_t)(dir_words * dir_opt + (span ? (dir_opt >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4 + lds_pad, d, g, dir_opt, pmax, (const OpRow*)d.op,
                      (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt);
        if (d.fuse)
          LM_LAUNCH_DYN(k_integrate_span_plain_fuse, n_docs, 64, (size_t)(dir_words * dir_opt + (span ? (dir_opt >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4 + lds_pad, d, g, dir_opt, pmax, (const OpRow*)d.op,
                        (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt);
      }
      else if (any_plain)
        LM_LAUNCH_DYN(k_integrate_span_plain, n_docs, 64, (size_t)(dir_words * dir_opt + (span ? (dir_opt >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4 + lds_pad, d, g, dir_opt, pmax, (const OpRow*)d.op,
                      (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt);
      if (any_common || !(any_ml || any_plain))
        LM_LAUNCH_DYN(k_integrate_span, n_docs, 64, (size_t)(dir_words * dir_opt + (span ? (dir_opt >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4 + lds_pad, d, g, dir_opt, pmax, (const OpRow*)d.op,
                      (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt);
      if (any_ml)
        LM_LAUNCH_DYN(k_integrate_span_ml, n_docs, 64, (size_t)(dir_words * dir_opt + (span ? (dir_opt >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4 + lds_pad, d, g, dir_opt, pmax, (const OpRow*)d.op,
                      (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt);
    } else {
      LM_LAUNCH_DYN(k_integrate, n_docs, 64, (size_t)(dir_opt + 3 * pmax) * 4 + lds_pad, d, g, dir_opt, pmax, (const OpRow*)d.op,
                    (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 0u, retry_cnt);
    }
    uint32_t n_retry = 0, n_lww = 0, n_posdel = 0;
    {
      uint32_t four[4] = {0, 0, 0, 0};
      lmbe::d2h(four, retry_cnt, 16);
      n_retry = four[0]; n_lww = four[2]; n_posdel = four[3];
      if (resident) last_fresh = four[1];
    }
    if (n_lww) {
      // documents with more keys than their optimistic LWW table holds: full-size tables behind the others', their Map rows again
      std::vector<DocMeta> hd(n_docs);
      lmbe::d2h(hd.data(), d.doc, (size_t)n_docs * sizeof(DocMeta));
      std::vector<uint32_t> h_cnt(n_docs);
      lmbe::d2h(h_cnt.data(), d.ht_cnt, (size_t)n_docs * 4);
      const uint64_t ht_old = ht;
      for (uint32_t i = 0; i < n_docs; i++) if (hd[i].flags & DF_LWW_RETRY) { h_ht0[i] = ht; h_ht_cap[i] = h_ht_full[i]; ht += h_ht_full[i]; h_cnt[i] = 0; }
      b_ht_key.ensure_keep((ht + 1) * 8, (ht_old + 1) * 8); b_ht_best.ensure_keep((ht + 1) * 8, (ht_old + 1) * 8);
      b_ht_pfx.ensure_keep((ht + 1) * 8, 0); b_ht_list.ensure_keep((ht + 1) * 8, (ht_old + 1) * 8);
      d.ht_key = b_ht_key.as<unsigned long long>(); d.ht_best = b_ht_best.as<unsigned long long>(); d.ht_pfx = b_ht_pfx.as<unsigned long long>();
      d.ht_list = b_ht_list.as<uint32_t>();
      lmbe::dmemset((uint8_t*)b_ht_key.p + ht_old * 8, 0xff, (ht - ht_old) * 8);
      lmbe::dmemset((uint8_t*)b_ht_best.p + ht_old * 8, 0, (ht - ht_old) * 8);
      lmbe::h2d(b_ht0.p, h_ht0.data(), (size_t)n_docs * 8);
      lmbe::h2d(b_ht_cap.p, h_ht_cap.data(), (size_t)n_docs * 4);
      lmbe::h2d(b_ht_cnt.p, h_cnt.data(), (size_t)n_docs * 4);
      LM_LAUNCH(k_map_lww, cdiv(NO, 256), 256, d, NO, retry_cnt, 1u, 0u);
    }
    if (n_retry) {  // rare: re-run the overflowed documents with the worst-case directory
      if (resident) {
        const size_t lds = (size_t)(2 * dir_cap + (dir_cap >> SD_BSH) + 2 + PD_LDS + 6 * pmax) * 4;
        if (any_plain)
          LM_LAUNCH_DYN(k_integrate_span_res_plain, n_docs, 64, lds, d, g, dir_cap, pmax, (const OpRow*)d.op, (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted,
                        (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt, rs);
        if (any_common || !(any_ml || any_plain))
          LM_LAUNCH_DYN(k_integrate_span_res, n_docs, 64, lds, d, g, dir_cap, pmax, (const OpRow*)d.op, (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted,
                        (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt, rs);
        if (any_ml)
          LM_LAUNCH_DYN(k_integrate_span_res_ml, n_docs, 64, lds, d, g, dir_cap, pmax, (const OpRow*)d.op, (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted,
                        (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt, rs);
      } else if (span) {
        LM_LAUNCH_DYN(k_integrate_span, n_docs, 64, (size_t)(dir_words * dir_cap + (span ? (dir_cap >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4, d, g, dir_cap, pmax, (const OpRow*)d.op,
                      (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt);
        if (any_ml)
          LM_LAUNCH_DYN(k_integrate_span_ml, n_docs, 64, (size_t)(dir_words * dir_cap + (span ? (dir_cap >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4, d, g, dir_cap, pmax, (const OpRow*)d.op,
                        (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt);
        if (any_plain && plain_mode == 2) {
          LM_LAUNCH_DYN(k_integrate_span_plain_sweep, n_docs, 64, (size_t)(dir_words * dir_cap + (span ? (dir_cap >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4, d, g, dir_cap, pmax, (const OpRow*)d.op,
                        (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt);
          if (d.fuse)
            LM_LAUNCH_DYN(k_integrate_span_plain_fuse, n_docs, 64, (size_t)(dir_words * dir_cap + (span ? (dir_cap >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4, d, g, dir_cap, pmax, (const OpRow*)d.op,
                          (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt);
        }
        else if (any_plain)
          LM_LAUNCH_DYN(k_integrate_span_plain, n_docs, 64, (size_t)(dir_words * dir_cap + (span ? (dir_cap >> SD_BSH) + 2 + PD_LDS : 0) + 4 * pmax + 1) * 4, d, g, dir_cap, pmax, (const OpRow*)d.op,
                        (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt);
      } else {
        LM_LAUNCH_DYN(k_integrate, n_docs, 64, (size_t)(dir_cap + 3 * pmax) * 4, d, g, dir_cap, pmax, (const OpRow*)d.op,
                      (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 1u, retry_cnt);
      }
    }
    last_retries = n_retry;
    if (n_retry && span && !resident && d.posdel) lmbe::d2h(&n_posdel, retry_cnt + 3, 4);   // (the retry launch may have met such a row too)
    if (n_posdel && span && !resident) {
      // Documents with a delete row whose target ids are not the elements at its position — damaged input; the reference applies
      // every delete by position (crdt_rope.rs:256-335) — left the launches above with ST_POSDEL: replayed once more by the kernel
      // that finishes such rows by position and remembers what they deleted (ts_del_positional), with the worst-case directory
      // Many such documents (a batch staged on snapshot states: every delete of base content goes this way): first with the optimistic
      // directory — the worst-case one leaves a few waves per CU — and once more, worst case, for the documents that overflow it
      uint32_t n_left = n_posdel;
#ifdef LM_EMU_TRACE
      if (getenv("LM_EMU_BASE")) fprintf(stderr, "POS documents %u, directory %u optimistic / %u worst case\n", n_posdel, dir_opt, dir_cap);
#endif
      if (n_posdel >= 64 && dir_opt < dir_cap) {
        lmbe::dmemset(retry_cnt + 3, 0, 4);
        LM_LAUNCH_DYN(k_integrate_span_pos, n_docs, 64, (size_t)(2 * dir_opt + (dir_opt >> SD_BSH) + 2 + PD_LDS + 5 * pmax + 1) * 4, d, g, dir_opt, pmax, (const OpRow*)d.op,
                      (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 3u, retry_cnt);
        LM_LAUNCH_DYN(k_integrate_span_pos_plain, n_docs, 64, (size_t)(2 * dir_opt + (dir_opt >> SD_BSH) + 2 + PD_LDS + 5 * pmax + 1) * 4, d, g, dir_opt, pmax, (const OpRow*)d.op,
                      (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 3u, retry_cnt);
        lmbe::d2h(&n_left, retry_cnt + 3, 4);
#ifdef LM_EMU_TRACE
        if (getenv("LM_EMU_BASE")) fprintf(stderr, "POS pass with the optimistic directory (%u of %u): %u documents, %u left for the worst-case pass\n", dir_opt, dir_cap, n_posdel, n_left);
#endif
      }
      if (n_left)
      {
      LM_LAUNCH_DYN(k_integrate_span_pos, n_docs, 64, (size_t)(2 * dir_cap + (dir_cap >> SD_BSH) + 2 + PD_LDS + 5 * pmax + 1) * 4, d, g, dir_cap, pmax, (const OpRow*)d.op,
                    (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 2u, retry_cnt);
      LM_LAUNCH_DYN(k_integrate_span_pos_plain, n_docs, 64, (size_t)(2 * dir_cap + (dir_cap >> SD_BSH) + 2 + PD_LDS + 5 * pmax + 1) * 4, d, g, dir_cap, pmax, (const OpRow*)d.op,
                    (const ChangeRow*)d.chg, (const uint32_t*)d.chg_sorted, (const uint32_t*)d.chg_skip, (const uint32_t*)d.vvh, 2u, retry_cnt);
      }
    }
    last_posdel = n_posdel;
    if (!resident && h_front_off.size() > n_docs && h_front_off[n_docs] > 0) LM_LAUNCH(k_seq_alive_latest, n_docs, 64, d);   // checked-out documents only (a resident tracker knows both versions)
    if (h_froot_off.size() > n_docs && h_froot_off[n_docs] > 0) LM_LAUNCH(k_state_roots, n_docs, 64, d);        // documents initialised from a snapshot only
    // documents holding a MovableList only: element → item maxima and the items' elements (loc[] is free from here on)
    if (any_ml) LM_LAUNCH(k_mlist_post, n_docs, 64, d, g);
    if (resident) LM_LAUNCH(k_res_exists, n_docs, 64, d, rs, shared_mode);   // the state store keeps what it once held: OR over the document's runs
#ifdef LM_EMU_TRACE
    if (getenv("LM_EMU_DUMP")) {  // kernel-logic harness only: leaves of document 0 in document order
      lmbe::d2h(h_doc.data(), b_doc.p, (size_t)n_docs * sizeof(DocMeta));
      const DocMeta& m0 = h_doc[0];
      for (uint32_t c = 0; c < m0.n_cont; c++) {
        uint32_t r0 = d.cont_root0[m0.cid0 + c], nr = d.cont_nroot[m0.cid0 + c];
        for (uint32_t q = 0; q < nr; q++) {
          uint32_t e = d.dir_out[m0.leaf0 + r0 + q], L = de_leaf(e), n = de_n(e);
          for (uint32_t i = 0; i < n; i++) {
            const uint32_t* rec = d.it + ((uint64_t)m0.leaf0 + L) * (span ? (size_t)SP_REC : 256);
            uint32_t id0 = rec[i], ln = span ? rec[64 + i] : 1u, ol0 = rec[(span ? 128 : 64) + i], orr = rec[(span ? 192 : 128) + i], st = rec[(span ? 256 : 192) + i];
            for (uint32_t k = 0; k < ln; k++) {
              uint32_t id = id0 + k, ol = k ? id - 1 : ol0;
              fprintf(stderr, "DUMP c%u %u:%u ol=%d:%d or=%d:%d st=%x\n", c, id >> 24, id & 0xffffff,
                      ol == NONE ? -1 : (int)(ol >> 24), ol == NONE ? -1 : (int)(ol & 0xffffff),
                      orr == NONE ? -1 : (int)(orr >> 24), orr == NONE ? -1 : (int)(orr & 0xffffff), st);
            }
          }
        }
      }
    }
#endif
    {
      // the stage is named after the kernel that ran when there was only one (so the name matches rocprofv3's)
      const bool l_plain = span && any_plain, l_common = span && (any_common || !(any_ml || any_plain));
      const int n_launched = (int)l_plain + (int)l_common + (int)(span && any_ml);
      const char* stage = resident ? ((int)any_ml + (int)any_common + (int)any_plain > 1 ? "k_integrate_span_res (several instantiations)"
                                      : any_ml ? "k_integrate_span_res_ml" : any_plain ? "k_integrate_span_res_plain" : "k_integrate_span_res")
                          : !span ? "k_integrate"
                          : n_launched > 1 ? "k_integrate_span (several instantiations)"
                          : l_plain ? (plain_mode == 2 ? "k_integrate_span_plain_sweep" : "k_integrate_span_plain")
                          : any_ml ? "k_integrate_span_ml" : "k_integrate_span";
      lmbe::toc(stage, times, profiling);
    }
    // 6. emit in one pass into optimistic slabs (2 output bytes per input byte: the JSON of a text document is shorter than
    // its blobs), then compact.  The emitter never writes beyond a slab: a document whose JSON is longer (map-typed values
    // re-render their keys, child maps repeat them) comes back flagged DF_REEMIT with its exact size and is rendered again
    // into an exactly-sized slab.
    std::vector<uint64_t> slab_off(n_docs + 1, 0), vslab_off(n_docs + 1, 0);
    {
      for (uint32_t i = 0; i < n_docs; i++) {
        uint64_t in_b = 0;
        for (uint32_t b = h_doc_blob[i]; b < h_doc_blob[i + 1]; b++) in_b += h_blob_len[b];
        bool ok = h_doc[i].status == ST_OK;
        uint64_t cap = ok ? 2 * in_b + 64ull * h_doc[i].n_cont + 256 : 0;
        if (kn.slab_cap >= 0) cap = ok ? (uint64_t)kn.slab_cap : 0;
        uint64_t vcap = ok ? 16ull * h_doc[i].n_peers + 16 : 0;
        if (ok && n_state_docs && !resident && h_vvo_off.size() > (size_t)i + 1) vcap += h_vvo_off[i + 1] - h_vvo_off[i];
        slab_off[i + 1] = slab_off[i] + ((cap + 15) & ~15ull);
        vslab_off[i + 1] = vslab_off[i] + ((vcap + 15) & ~15ull);
      }
      b_slab.ensure(slab_off[n_docs] + 64); b_vslab.ensure(vslab_off[n_docs] + 64);
      b_slab_off.ensure((size_t)(n_docs + 1) * 8); b_vslab_off.ensure((size_t)(n_docs + 1) * 8);
      lmbe::h2d(b_slab_off.p, slab_off.data(), (size_t)(n_docs + 1) * 8);
      lmbe::h2d(b_vslab_off.p, vslab_off.data(), (size_t)(n_docs + 1) * 8);
      d.out = b_slab.as<uint8_t>(); d.out_off = b_slab_off.as<uint64_t>();
      d.vv_out = b_vslab.as<uint8_t>(); d.vv_off = b_vslab_off.as<uint64_t>();
    }
    lmbe::tic(profiling);
    LM_LAUNCH(k_emit_text, n_docs, 64, d, 1, 0);
    LM_LAUNCH(k_emit_any, n_docs, 64, d, 1, 0);
    lmbe::toc("k_emit", times, profiling);
    lmbe::d2h(h_doc.data(), d.doc, (size_t)n_docs * sizeof(DocMeta));
    std::vector<uint64_t> src_addr(n_docs + 1, 0);   // absolute device address of every document's rendered JSON
    for (uint32_t i = 0; i < n_docs; i++) src_addr[i] = (uint64_t)(uintptr_t)b_slab.p + slab_off[i];
    last_reemits = 0;
    for (uint32_t i = 0; i < n_docs; i++) if (h_doc[i].status == ST_OK && (h_doc[i].flags & DF_REEMIT)) last_reemits++;
    if (last_reemits) {
      // second slab set: the overflowed documents at their exact sizes, the others keep an empty range (they are skipped)
      std::vector<uint64_t> off2(n_docs + 1, 0);
      for (uint32_t i = 0; i < n_docs; i++) {
        bool re = h_doc[i].status == ST_OK && (h_doc[i].flags & DF_REEMIT);
        off2[i + 1] = off2[i] + (re ? ((uint64_t)h_doc[i].out_len + 15) & ~15ull : 0);
      }
      b_slab2.ensure(off2[n_docs] + 64); b_slab2_off.ensure((size_t)(n_docs + 1) * 8);
      lmbe::h2d(b_slab2_off.p, off2.data(), (size_t)(n_docs + 1) * 8);
      d.out = b_slab2.as<uint8_t>(); d.out_off = b_slab2_off.as<uint64_t>();
      LM_LAUNCH(k_emit_text, n_docs, 64, d, 1, 1);
      LM_LAUNCH(k_emit_any, n_docs, 64, d, 1, 1);
      // splice: documents rendered in the second pass are compacted from the second slab
      std::vector<DocMeta> h2(n_docs);
      lmbe::d2h(h2.data(), d.doc, (size_t)n_docs * sizeof(DocMeta));
      for (uint32_t i = 0; i < n_docs; i++) {
        bool re = h_doc[i].status == ST_OK && (h_doc[i].flags & DF_REEMIT);
        if (re && (h2[i].status != ST_OK || (h2[i].flags & DF_REEMIT) || h2[i].out_len != h_doc[i].out_len)) h2[i].status = h2[i].status != ST_OK ? h2[i].status : ST_INTERNAL;
      }
      for (uint32_t i = 0; i < n_docs; i++) if (h2[i].status != h_doc[i].status) lmbe::h2d(&d.doc[i].status, &h2[i].status, 4);
      for (uint32_t i = 0; i < n_docs; i++) if (h_doc[i].status == ST_OK && (h_doc[i].flags & DF_REEMIT)) src_addr[i] = (uint64_t)(uintptr_t)b_slab2.p + off2[i];
      h_doc = h2;
    }
    lmbe::h2d(b_slab_off.p, src_addr.data(), (size_t)n_docs * 8);   // the slab offsets are no longer needed: reuse as k_compact's source table
    h_out_off.assign(n_docs + 1, 0);
    h_vv_off.assign(n_docs + 1, 0);
    for (uint32_t i = 0; i < n_docs; i++) {
      bool ok = h_doc[i].status == ST_OK;
      h_out_off[i + 1] = h_out_off[i] + (ok ? ((uint64_t)h_doc[i].out_len + 15) & ~15ull : 0);
      h_vv_off[i + 1] = h_vv_off[i] + (ok ? ((uint64_t)h_doc[i].vv_len + 15) & ~15ull : 0);
    }
    out_bytes = h_out_off[n_docs];
    vv_bytes = h_vv_off[n_docs];
    b_out.ensure(out_bytes + 64); b_vv_out.ensure(vv_bytes + 64);
    b_out_off.ensure((size_t)(n_docs + 1) * 8); b_vv_off.ensure((size_t)(n_docs + 1) * 8);
    lmbe::h2d(b_out_off.p, h_out_off.data(), (size_t)(n_docs + 1) * 8);
    lmbe::h2d(b_vv_off.p, h_vv_off.data(), (size_t)(n_docs + 1) * 8);
    {
      const uint64_t* vo = b_vslab_off.as<uint64_t>();
      const uint8_t* vs = b_vslab.as<uint8_t>();
      d.out = b_out.as<uint8_t>(); d.out_off = b_out_off.as<uint64_t>();
      d.vv_out = b_vv_out.as<uint8_t>(); d.vv_off = b_vv_off.as<uint64_t>();
      lmbe::tic(profiling);
      LM_LAUNCH(k_compact, n_docs, 64, d, (const uint64_t*)b_slab_off.as<uint64_t>(), vo, vs);
      lmbe::toc("k_compact", times, profiling);
      b_hash.ensure((size_t)n_docs * 8 + 8);
      LM_LAUNCH(k_hash_json, cdiv(n_docs, HASH_DOCS), 64, d, b_hash.as<uint64_t>());
      if (sum_rows) LM_LAUNCH(k_summary_rows, cdiv(n_docs, 256), 256, d, (const uint64_t*)b_hash.as<uint64_t>(), sum_rows, sum_id0, sum_stride);
      h_hash.resize(n_docs);
      lmbe::d2h(h_hash.data(), b_hash.p, (size_t)n_docs * 8);
    }
    payload_bytes = 0;
    for (uint32_t i = 0; i < n_docs; i++) if (h_doc[i].status == ST_OK) payload_bytes += (uint64_t)h_doc[i].out_len + h_doc[i].vv_len;
#if defined(LM_PROF) || defined(LM_PROF_DEC) || defined(LM_PROF_EMIT) || defined(LM_PROF_DAG) || defined(LM_PROF_MF)
    h_prof.resize((size_t)n_docs * 16);
    lmbe::d2h(h_prof.data(), d.prof, (size_t)n_docs * 16 * 8);
#endif
    lmbe::sync();
    for (uint32_t i = 0; i < n_docs; i++) {
      DocResult& r = results[i];
      r.status = h_doc[i].status;
      bool ok = r.status == ST_OK;
      // out-of-scope containers met: everything in scope is rendered (they appear as null) and the document is flagged
      if (ok && (h_doc[i].flags & DF_SOFT_UNSUPPORTED)) r.status = ST_UNSUPPORTED;
      if (ok && (h_doc[i].flags & DF_FRONT_ERR)) { r.status = h_doc[i].front_err; ok = false; }   // resident: the import went through, the checkout was refused
      r.json_off = h_out_off[i]; r.json_len = ok ? h_doc[i].out_len : 0;
      r.vv_off = h_vv_off[i]; r.vv_len = ok ? h_doc[i].vv_len : 0;
      r.pending = ok ? (((uint64_t)h_doc[i].pending_hi << 32) | h_doc[i].pending_lo) : 0;
      r.json_xxh64 = ok ? h_hash[i] : 0;
    }
    redo_docs.clear();
    // (a fused Map document that FAILED anywhere: its op columns are validated late — behind the DAG stages — where the row tables'
    // decoders speak first; the failure's code is theirs to give, so the document is replayed through them)
    // (… and a document staged on a snapshot's STATE that failed in any way — an engine limit of the by-position path, a damaged update:
    // replayed from the snapshot's history, where every verdict is the row decoders' / the tracker's on real ids)
    redo_hist.assign(n_docs, 0);
    if (kn.redo && st_valid) for (uint32_t i = 0; i < n_docs; i++) {
      const bool on_state = !resident && n_state_docs && h_vvo_off.size() > (size_t)i + 1 && h_vvo_off[i + 1] > h_vvo_off[i] && i < st_hist.size() && !st_hist[i].empty() &&
                            !(h_doc[i].status == ST_OK) && !force_span;
      if ((h_doc[i].flags & DF_REDO) || (i < h_fused.size() && h_fused[i] && h_doc[i].status != ST_OK) || on_state) { redo_docs.push_back(i); redo_hist[i] = on_state ? 1 : 0; }
    }
    lmbe::flush_times(times);
    if (resident) {
      // a document whose run failed keeps what it held before: the blobs of this step are dropped again (reference import is
      // atomic per document, loro.rs:780-838) and its tracker — possibly half moved — is not used by the next run
      bool dropped = false;
      for (uint32_t i = 0; i < n_docs; i++) {
        if (h_doc[i].status != ST_OK) {
          if (r_step[i]) { r_blobs[i].resize(r_blobs[i].size() - r_step[i]); dropped = true; }
          tk_reset[i] = 1;
        } else tk_reset[i] = 0;
        r_step[i] = 0;
      }
      h_lca.resize((size_t)n_docs * LCA_OUT);
      lmbe::d2h(h_lca.data(), b_lca_out.p, (size_t)n_docs * LCA_OUT * 4);
      have_prev = true;
      dir_parity ^= 1;
      tables_valid = !dropped;
      if (dropped) { rebuild_blob_tables(); lmbe::sync(); }
    }
    ran = true;
    fetched = false;
    last_d = d;
    rt_ran = false;
  }

  // ---- lm_richtext: get_richtext_value of every Text container of every document, from the trackers the last run left.  One launch
  // of k_richtext into optimistic slabs (twice the document's JSON + a few KB: a span costs ~40 bytes on top of its text); the kernel
  // never writes beyond a slab and always reports the exact size, so a document that needs more sends the batch through a second
  // launch at exact sizes (as the JSON renderer's DF_REEMIT does).  LM_RT_SLAB=<bytes> overrides the slab size (tests: 0 forces the
  // second launch).
  void richtext() {
    lmbe::bind(sc);
    if (!ran) throw std::runtime_error("lm_richtext before lm_run");
    if (shared_mode != 0) throw std::runtime_error("lm_richtext: not available on a batch folded by shared replay (stage with LM_SHARE_REPLAY=0)");
    h_rt_off.assign(n_docs + 1, 0);
    h_rt_len.assign(n_docs, 0);
    h_rt_status.assign(n_docs, 0);
    h_rt.assign(1, 0);
    rt_ran = true;
    rt_launches = 0;
    if (n_docs == 0) return;
    Dev d = last_d;
    b_rt_len.ensure((size_t)n_docs * 12 + 8);
    uint32_t* len = b_rt_len.as<uint32_t>();
    int32_t* st = (int32_t*)(b_rt_len.as<uint32_t>() + n_docs);
    uint32_t* cnt = b_rt_len.as<uint32_t>() + 2 * (size_t)n_docs;
    std::vector<uint32_t> h_cnt(n_docs, 0);
    const char* slab_env = getenv("LM_RT_SLAB");
    for (uint32_t i = 0; i < n_docs; i++) {
      uint64_t cap = h_doc[i].status == ST_OK ? (slab_env ? (uint64_t)atoll(slab_env) : 2ull * h_doc[i].out_len + 64ull * h_doc[i].n_cont + 4096) : 0;
      h_rt_off[i + 1] = h_rt_off[i] + ((cap + 15) & ~15ull);
    }
    for (int attempt = 0; attempt < 2; attempt++) {
      b_rt_out.ensure(h_rt_off[n_docs] + 64);
      b_rt_off.ensure((size_t)(n_docs + 1) * 8);
      lmbe::h2d(b_rt_off.p, h_rt_off.data(), (size_t)(n_docs + 1) * 8);
      LM_LAUNCH(k_richtext, n_docs, 64, d, b_rt_out.as<uint8_t>(), (const uint64_t*)b_rt_off.as<uint64_t>(), len, st, cnt, 1);
      rt_launches++;
      lmbe::d2h(h_rt_len.data(), len, (size_t)n_docs * 4);
      lmbe::d2h(h_cnt.data(), cnt, (size_t)n_docs * 4);
      lmbe::d2h(h_rt_status.data(), st, (size_t)n_docs * 4);
      lmbe::sync();
      bool over = false;
      for (uint32_t i = 0; i < n_docs; i++) over |= h_rt_status[i] == ST_OK && h_rt_len[i] > h_rt_off[i + 1] - h_rt_off[i];
      if (!over) break;
      if (attempt == 1) { for (uint32_t i = 0; i < n_docs; i++) if (h_rt_status[i] == ST_OK && h_rt_len[i] > h_rt_off[i + 1] - h_rt_off[i]) h_rt_status[i] = ST_INTERNAL; break; }
      for (uint32_t i = 0; i < n_docs; i++) h_rt_off[i + 1] = h_rt_off[i] + (h_rt_status[i] == ST_OK ? ((uint64_t)h_rt_len[i] + 15) & ~15ull : 0);
    }
    h_rt.resize(h_rt_off[n_docs] + 1);
    if (h_rt_off[n_docs]) lmbe::d2h(h_rt.data(), b_rt_out.p, h_rt_off[n_docs]);
    lmbe::sync();
    // canonical member order: a document that lists several Text containers has its members put into the bytewise order of their
    // JSON-encoded keys (the kernel writes them in container-table order); bytes are moved, never changed
    std::vector<uint8_t> tmp;
    for (uint32_t i = 0; i < n_docs; i++) {
      if (h_rt_status[i] != ST_OK || h_cnt[i] < 2 || h_rt_len[i] < 2) continue;
      uint8_t* p = h_rt.data() + h_rt_off[i];
      const size_t n = h_rt_len[i];
      std::vector<std::pair<size_t, size_t>> mem;   // [begin, end) of every member inside the outer braces
      size_t b = 1;
      int depth = 0;
      bool in_str = false;
      for (size_t k = 1; k + 1 < n; k++) {
        const uint8_t c = p[k];
        if (in_str) { if (c == '\\') k++; else if (c == '"') in_str = false; continue; }
        if (c == '"') in_str = true;
        else if (c == '[' || c == '{') depth++;
        else if (c == ']' || c == '}') depth--;
        else if (c == ',' && depth == 0) { mem.emplace_back(b, k); b = k + 1; }
      }
      mem.emplace_back(b, n - 1);
      if (mem.size() != h_cnt[i]) { h_rt_status[i] = ST_INTERNAL; continue; }
      auto key_end = [&](const std::pair<size_t, size_t>& m) { size_t k = m.first + 1; while (k < m.second && p[k] != '"') k += p[k] == '\\' ? 2 : 1; return k; };
      std::stable_sort(mem.begin(), mem.end(), [&](const std::pair<size_t, size_t>& x, const std::pair<size_t, size_t>& y) {
        const size_t xe = key_end(x), ye = key_end(y), xl = xe - x.first, yl = ye - y.first;
        const int c = memcmp(p + x.first, p + y.first, xl < yl ? xl : yl);
        return c != 0 ? c < 0 : xl < yl;
      });
      tmp.assign(1, '{');
      for (size_t m = 0; m < mem.size(); m++) { if (m) tmp.push_back(','); tmp.insert(tmp.end(), p + mem[m].first, p + mem[m].second); }
      tmp.push_back('}');
      if (tmp.size() == n) memcpy(p, tmp.data(), n); else h_rt_status[i] = ST_INTERNAL;
    }
  }

  // ---- lm_export: the updates document `i` holds beyond `from_vv` (lm_export.h).  The blobs come back from the arena, the
  // applied version from the tables of the last run
  lmexp::Bytes export_doc(uint32_t i, const uint8_t* from_vv, size_t from_len) {
    lmbe::bind(sc);
    if (!ran) throw std::runtime_error("lm_export before lm_run");
    if (i >= n_docs) throw std::runtime_error("lm_export: no such document");
    const DocMeta& m = h_doc[i];
    if (m.status != ST_OK) throw std::runtime_error("lm_export: the document failed to import");
    std::map<uint64_t, uint32_t> applied;
    if (m.n_peers) {
      std::vector<uint64_t> ids(m.n_peers);
      std::vector<uint32_t> ends(m.n_peers);
      bool checked_out = h_front_off.size() > i + 1 && h_front_off[i + 1] > h_front_off[i];
      lmbe::d2h(ids.data(), b_peer_uniq.as<uint64_t>() + m.praw0, (size_t)m.n_peers * 8);
      lmbe::d2h(ends.data(), ((resident || checked_out) ? b_peer_end_all.as<uint32_t>() : b_peer_end.as<uint32_t>()) + m.praw0, (size_t)m.n_peers * 4);
      for (uint32_t p = 0; p < m.n_peers; p++) if (ends[p]) applied[ids[p]] = ends[p];
    }
    std::vector<lmexp::Block> blocks;
    std::vector<uint8_t> buf;
    for (uint32_t b = h_doc_blob[i]; b < h_doc_blob[i + 1]; b++) {
      buf.resize(h_blob_len[b]);
      if (h_blob_len[b]) lmbe::d2h(buf.data(), b_data.as<uint8_t>() + h_blob_off[b], h_blob_len[b]);
      lmexp::blocks_of_blob(buf.data(), buf.size(), blocks);
    }
    return lmexp::export_updates(blocks, lmexp::decode_vv(from_vv, from_len), applied);
  }

  // ---- fetch: copy the rendered states back to the host
  void fetch() {
    lmbe::bind(sc);
    if (!ran) throw std::runtime_error("lm_fetch before lm_run");
    h_out.resize(out_bytes + 1);
    h_vv.resize(vv_bytes + 1);
    if (out_bytes) lmbe::d2h(h_out.data(), b_out.p, out_bytes);
    if (vv_bytes) lmbe::d2h(h_vv.data(), b_vv_out.p, vv_bytes);
    lmbe::sync();
    fetched = true;
  }
};

}  // namespace lm
