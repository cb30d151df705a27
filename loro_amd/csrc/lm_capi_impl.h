// C-ABI glue shared by the product library (lm_hip.cpp) and the kernel-logic test harness (tests/emu).
// LM_API(name) expands to the exported symbol name.
#pragma once
#include <memory>
#include <thread>
#include <map>
#include <mutex>
#include "lm_pipeline.h"
#include "lm_encode.h"

// One lm_ctx drives up to LM_MAX_PARTS engines, each on its own HIP stream over a contiguous range of the batch's
// documents.  lm_run launches the parts from separate host threads so the kernels of one part (latency-bound
// integrate, LDS-limited occupancy) overlap with the decode / emit kernels of the other.  Documents never span parts.
struct lm_ctx_impl {
  static constexpr uint32_t LM_MAX_PARTS = 8;
  std::vector<std::unique_ptr<lm::Engine>> parts;
  std::vector<uint32_t> first;     // first[p] = first document of part p; first[n_parts] = n_docs
  uint32_t n_docs = 0;
  uint32_t want_parts = 2, part_min_docs = 128;
  int profiling = 0;               // 0 off | 1 stage events, streams one after the other | 2 stage events, streams overlapped
  bool ran = false;
  std::vector<lm::KernelTime> times;
  std::string err;
  std::thread runner;              // lm_run_async: the run in flight
  std::string async_err;
  bool in_flight = false;

  // ---- shared replay (loro.rs:1625-1760 — a live LoroDoc serves every checkout from ONE imported history and its
  // persistent DiffCalculator).  Entries of a staged batch that name the same blobs (same pointers, same lengths) and differ only in
  // checkout_frontiers are one DOCUMENT rendered at several versions: the document is staged once, imported once per lm_run (the
  // resident machinery of lm_import: decode, DAG, replay from the empty version), and every rendering is a move of its trackers
  // (ts_sweep_version) + the emit stage — instead of one replay of the whole history per entry.  Results are per entry, as if every
  // entry had been import_batch + checkout on its own (k_res_exists modes 1 / 2 keep the state store's view per rendering).
  struct Shared {
    bool on = false;
    uint32_t n_entries = 0, G = 0;                     // API-level documents; rendering runs behind the import run
    std::vector<uint32_t> uniq_of, slot_of;            // entry -> staged document, entry -> the run that renders it (0 = the import run: latest version)
    std::vector<std::vector<uint32_t>> by_slot;        // by_slot[s][u] = the entry document u renders in run s (NONE = none)
    std::vector<std::vector<uint8_t>> fronts;          // entry -> its checkout frontiers (copied: inputs are borrowed for lm_stage only)
    std::vector<lm::DocResult> res;                    // per entry; json_off / vv_off index the arenas of the entry's part
    std::vector<uint32_t> part_of;                     // entry -> part
    std::vector<lm::DBuf> d_out, d_vv;                 // per part: device arenas of the rendered bytes of all its runs
    std::vector<uint64_t> out_top, vv_top;
    std::vector<std::vector<uint8_t>> h_out, h_vv;     // per part, after lm_fetch
  } sh;
  uint32_t api_docs() const { return sh.on ? sh.n_entries : n_docs; }
  // lm_import on a batch that was staged folded: the documents are unfolded in HBM (Engine::expand — every entry becomes a resident
  // document of the part that holds its blobs) and the API's document order is no longer a part's contiguous range
  bool mapped = false;
  std::vector<std::pair<uint32_t, uint32_t>> emap;   // API document -> (part, document of the part)
  std::vector<std::vector<uint32_t>> inv;            // part -> its documents' API indices
  void unfold() {
    uint32_t np = n_parts();
    std::vector<std::vector<uint32_t>> src(np);
    std::vector<std::vector<std::vector<uint8_t>>> fr(np);
    emap.assign(sh.n_entries, {0u, 0u});
    inv.assign(np, {});
    for (uint32_t i = 0; i < sh.n_entries; i++) {
      uint32_t p = sh.part_of[i];
      emap[i] = {p, (uint32_t)src[p].size()};
      inv[p].push_back(i);
      src[p].push_back(sh.uniq_of[i] - first[p]);
      fr[p].push_back(sh.fronts[i]);
    }
    const bool was_run = ran;
    for (uint32_t p = 0; p < np; p++) parts[p]->expand(src[p], fr[p]);
    for (uint32_t p = 0; p < np; p++) first[p + 1] = first[p] + (uint32_t)src[p].size();
    n_docs = sh.n_entries;
    sh = Shared();
    mapped = true;
    rd.on = false;   // (documents replayed in the side engine are the parts' resident documents again)
    if (was_run) {   // lm_stage; lm_run; lm_import: the staged blobs are the resident documents' first step, an import of its own (lm_pipeline.h import_more)
      for (uint32_t p = 0; p < np; p++) for (uint32_t i = 0; i < parts[p]->n_docs; i++) parts[p]->r_step[i] = (uint32_t)parts[p]->r_blobs[i].size();
      run_parts();
    }
    ran = false;
  }

  // ---- DF_REDO (lm_types.h): documents the configuration the batch ran under has no path for — a delete row that does not match its
  // position met by the element-granular kernel (the batch's statistics picked it, LM_SPAN_AUTO) or by the resident kernels of a folded
  // batch; a Map document the fused decode→LWW kernel bailed out of — are replayed ONCE MORE, in a side engine that stages them from
  // the pinned staging buffer lm_stage filled (nothing is read from the caller again) and runs the span-granular batch pipeline,
  // whose k_integrate_span_pos finishes such rows by position like the reference (crdt_rope.rs:256-335).  Results, rendered bytes,
  // hashes, summary rows, richtext values and exports of those documents then come from the side engine: a document's verdict does
  // not depend on its neighbours in the batch nor on whether the host reused blob pointers.
  struct Redo {
    bool on = false;
    std::unique_ptr<lm::Engine> eng;
    std::vector<int32_t> of;               // API document / entry -> document of the side engine, -1 = none
    std::vector<lm::DocResult> res;        // per side-engine document (a folded document whose whole-history import fails fails for every entry)
    std::vector<std::vector<uint8_t>> fr;  // frontiers the items point at
    uint32_t n = 0;
  } rd;
  void redo_pass(const std::vector<std::vector<uint32_t>>& flagged) {   // flagged[p] = documents of part p to replay
    rd.on = false; rd.n = 0;
    size_t total = 0;
    for (auto& v : flagged) total += v.size();
    if (!total || mapped) return;
    std::vector<lm::Engine::RedoItem> items;
    std::vector<lm::Engine*> parents;
    std::vector<int32_t> verdict_of;        // item -> the item that carries its document's whole-history verdict (folded batches), -1 = itself
    rd.of.assign(api_docs(), -1);
    rd.fr.clear();
    // (frontier bytes are copied first: the items keep pointers into rd.fr, which must not grow afterwards)
    struct Want { uint32_t p, doc; int32_t api; const std::vector<uint8_t>* fr; int32_t verdict; };
    std::vector<Want> want;
    for (uint32_t p = 0; p < n_parts() && p < flagged.size(); p++)
      for (uint32_t i : flagged[p]) {
        lm::Engine& e = *parts[p];
        if (!sh.on) {
          rd.fr.emplace_back();
          if (e.h_front_off.size() > (size_t)i + 1 && e.h_front_off[i + 1] > e.h_front_off[i]) rd.fr.back().assign(e.h_front_bytes.begin() + e.h_front_off[i], e.h_front_bytes.begin() + e.h_front_off[i + 1]);
          want.push_back(Want{p, i, (int32_t)(first[p] + i), &rd.fr.back(), -1});
        } else {
          // a folded document: the whole history once (its verdict is every entry's, as LoroDoc::import in front of LoroDoc::checkout), then every entry at its version
          const int32_t v = (int32_t)want.size();
          rd.fr.emplace_back();
          want.push_back(Want{p, i, -1, &rd.fr.back(), -1});
          for (uint32_t en = 0; en < sh.n_entries; en++) if (sh.uniq_of[en] == first[p] + i) {
            if (sh.fronts[en].empty()) { rd.of[en] = v; continue; }
            rd.fr.push_back(sh.fronts[en]);
            want.push_back(Want{p, i, (int32_t)en, nullptr, v});
          }
        }
      }
    {   // rd.fr is complete: hand out the pointers (entries of folded documents took their frontiers in push order)
      size_t k = 0;
      for (auto& w : want) { w.fr = &rd.fr[k++]; }
    }
    for (auto& w : want) {
      items.push_back(lm::Engine::RedoItem{w.doc, w.fr->empty() ? nullptr : w.fr->data(), w.fr->size()});
      parents.push_back(parts[w.p].get());
      verdict_of.push_back(w.verdict);
    }
    if (!rd.eng) rd.eng.reset(new lm::Engine(device));
    lm::Engine& a = *rd.eng;
    a.force_span = true;
    a.profiling = profiling != 0;
    a.stage_from(parents, items);
    a.run();
    rd.res = a.results;
    for (size_t k = 0; k < want.size(); k++) {
      if (verdict_of[k] >= 0) {
        const int32_t vs = a.results[verdict_of[k]].status;
        if ((vs != lm::ST_OK && vs != lm::ST_UNSUPPORTED) || (vs == lm::ST_UNSUPPORTED && a.results[verdict_of[k]].json_len == 0)) rd.res[k] = lm::DocResult{vs, 0, 0, 0, 0, 0, 0};
      }
      if (want[k].api >= 0) rd.of[want[k].api] = (int32_t)k;
    }
    rd.on = true; rd.n = (uint32_t)want.size();
    for (auto& t : a.times) times.push_back(lm::KernelTime{t.name + " (redo)", t.ms});
    // the summary rows of the replayed documents (k_summary_rows wrote the first verdict): patched in place
    if (sum_rows_padded && !sh.on) {
      lmbe::bind(parts[0]->sc);
      for (size_t k = 0; k < want.size(); k++) if (want[k].api >= 0) {
        const lm::DocResult& r = rd.res[k];
        const bool ok = r.status == lm::ST_OK || r.status == lm::ST_UNSUPPORTED;
        long long w[6] = {sum_first + (long long)want[k].api * sum_stride, r.status, ok ? (long long)r.pending : 0, ok ? (long long)r.json_len : 0, ok ? (long long)r.vv_len : 0, ok ? (long long)r.json_xxh64 : 0};
        lmbe::h2d(sum_buf.as<long long>() + (size_t)want[k].api * 6, w, sizeof w);
      }
      lmbe::sync();
    }
  }

  int device = 0;                  // HIP device of this context: every engine (stream, buffers) is created on it
  // lm_summary_layout: the summary rows of this context's documents, written on the device by every run (k_summary_rows)
  lm::DBuf sum_buf, sum_all;       // rows_padded x 6 int64 (this rank's), world x rows_padded x 6 (gathered)
  size_t sum_rows_padded = 0;
  long long sum_first = 0, sum_stride = 1;
  void sum_bind() {                // every part writes its own documents' rows
    for (uint32_t p = 0; p < n_parts() && p < parts.size(); p++) {
      // (a folded batch — entries that share their blobs — or one unfolded by lm_import: the API's documents are not the parts' contiguous
      // ranges; their rows are written from the host after the run, sum_host_rows)
      parts[p]->sum_rows = sum_rows_padded && !sh.on && !mapped ? sum_buf.as<long long>() + (size_t)first[p] * 6 : nullptr;
      parts[p]->sum_id0 = sum_first + (long long)first[p] * sum_stride; parts[p]->sum_stride = sum_stride;
    }
  }

  void sum_host_rows() {
    if (!sum_rows_padded || !(sh.on || mapped)) return;
    std::vector<long long> rows((size_t)api_docs() * 6, -1);
    for_docs([&](uint32_t i, lm::Engine&, const lm::DocResult& r) {
      const bool ok = r.status == lm::ST_OK || r.status == lm::ST_UNSUPPORTED;
      long long* w = rows.data() + (size_t)i * 6;
      w[0] = sum_first + (long long)i * sum_stride; w[1] = r.status; w[2] = ok ? (long long)r.pending : 0;
      w[3] = ok ? (long long)r.json_len : 0; w[4] = ok ? (long long)r.vv_len : 0; w[5] = ok ? (long long)r.json_xxh64 : 0;
    });
    lmbe::bind(parts[0]->sc);
    if (!rows.empty()) lmbe::h2d(sum_buf.p, rows.data(), rows.size() * 8);
    lmbe::sync();
  }

  explicit lm_ctx_impl(int dev) : device(dev) {
    if (const char* e = getenv("LM_STREAMS")) { int v = atoi(e); if (v >= 1 && v <= (int)LM_MAX_PARTS) want_parts = (uint32_t)v; }
    if (const char* e = getenv("LM_PART_MIN_DOCS")) { int v = atoi(e); if (v >= 1) part_min_docs = (uint32_t)v; }
    parts.emplace_back(new lm::Engine(device));
    first = {0, 0};
  }
  ~lm_ctx_impl() { if (runner.joinable()) runner.join(); }
  uint32_t n_parts() const { return (uint32_t)first.size() - 1; }

  void stage(const lm::Engine::DocIn* docs_api, size_t n_api) {
    // entries that share their blobs: staged once (see Shared above).  LM_SHARE_REPLAY=0: every entry is its own document (rounds 1-4)
    std::vector<lm::Engine::DocIn> udocs;
    // an entry that is ONE shallow snapshot checked out at exactly its shallow root (checkout_frontiers == the `fr` entry of the
    // snapshot's third section; loro_js_interop.rs:141-147): that section IS the state at that version — staged from it, as a document
    // of its own at "its latest version" (lm_snapshot.h snapshot_state_to_updates, root_only).  Any other version of a shallow
    // snapshot needs ops replayed over a state base: LM_UNSUPPORTED as before
    std::vector<lm::Engine::DocIn> pre;
    {
      const char* e = getenv("LM_SNAPSHOT_STATE");
      std::vector<uint8_t> fr;
      for (size_t i = 0; !(e && atoi(e) == 0) && i < n_api; i++) {
        const lm::Engine::DocIn& d = docs_api[i];
        if (d.n != 1 || !d.front || d.lens[0] < 22 || d.blobs[0][21] != 3 || memcmp(d.blobs[0], "loro", 4) != 0) continue;
        fr.clear();
        if (!lmsnap::snapshot_shallow_root_frontiers(d.blobs[0], d.lens[0], fr) || fr.size() != d.front_len || memcmp(fr.data(), d.front, fr.size()) != 0) continue;
        if (pre.empty()) pre.assign(docs_api, docs_api + n_api);
        pre[i].front = nullptr; pre[i].front_len = 0; pre[i].state_root = 1;
      }
      if (!pre.empty()) docs_api = pre.data();
    }
    const lm::Engine::DocIn* docs = docs_api;
    size_t n = n_api;
    sh.on = false;
    {
      const char* e = getenv("LM_SHARE_REPLAY");
      const char* sp = getenv("LM_SPAN");
      bool want = !(e && atoi(e) == 0) && !(sp && atoi(sp) == 0);   // (the resident machinery is the span-granular kernels')
      // (folded: groups of two or more entries of which at least one asks for a checkout — entries that all render the latest
      // version stay documents of their own: a caller may stage a batch of equal documents and lm_import different updates into each)
      std::map<std::vector<uint64_t>, std::vector<uint32_t>> groups;
      std::vector<uint32_t> uniq_of(n_api);
      std::vector<uint64_t> key;
      for (size_t i = 0; want && i < n_api; i++) {
        key.clear();
        key.push_back(docs_api[i].n);
        for (size_t b = 0; b < docs_api[i].n; b++) { key.push_back((uint64_t)(uintptr_t)docs_api[i].blobs[b]); key.push_back((uint64_t)docs_api[i].lens[b]); }
        if (docs_api[i].state_root) { key.push_back(~0ull); key.push_back(i); }   // (a document of its own: never folded with the entries that share its bytes)
        groups[key].push_back((uint32_t)i);
      }
      // LM_CHECKOUT_FULL=1: EVERY entry with a checkout takes this path, alone in its group too — the whole history is imported (what
      // LoroDoc::import does before LoroDoc::checkout, loro.rs:568-649,1625-1760) and the version is reached by moving the trackers,
      // instead of the batch entry's replay of the version's causal closure only (DESIGN §7 "Checkout").  On healthy documents the two
      // give the same bytes; on DAMAGED ones the closure replay never meets damage that lies outside the rendered version — it renders
      // where the reference's import fails (≈1 % of a damaged corpus with checkouts) or, rarely, renders another value.  The default since
      // round 6: a checked-out entry costs the replay of its whole history, as in the reference.
      const char* cf = getenv("LM_CHECKOUT_FULL");
      const bool full = !(cf && atoi(cf) == 0);   // (round 6: the default — LM_CHECKOUT_FULL=0 keeps the closure replay of rounds 1-5 for single checkouts)
      bool any_fold = false;
      if (want) {
        std::vector<uint8_t> fold(n_api, 0);
        for (auto& kv : groups) {
          bool any_front = false;
          for (uint32_t i : kv.second) any_front |= docs_api[i].front != nullptr;
          if ((kv.second.size() >= 2 || full) && any_front && docs_api[kv.second[0]].n) for (uint32_t i : kv.second) { fold[i] = 1; any_fold = true; }
        }
        std::map<std::vector<uint64_t>, uint32_t> seen;
        for (size_t i = 0; i < n_api; i++) {   // staged documents in the order of their first entries
          uint32_t u = (uint32_t)udocs.size();
          if (fold[i]) {
            key.clear();
            key.push_back(docs_api[i].n);
            for (size_t b = 0; b < docs_api[i].n; b++) { key.push_back((uint64_t)(uintptr_t)docs_api[i].blobs[b]); key.push_back((uint64_t)docs_api[i].lens[b]); }
            auto it = seen.find(key);
            if (it != seen.end()) { uniq_of[i] = it->second; continue; }
            seen.emplace(key, u);
          }
          udocs.push_back(lm::Engine::DocIn{docs_api[i].blobs, docs_api[i].lens, docs_api[i].n, nullptr, 0, docs_api[i].state_root});
          uniq_of[i] = u;
        }
      }
      if (want && (udocs.size() < n_api || (full && any_fold))) {
        for (size_t i = 0; i < n_api; i++)
          if (docs_api[i].front && docs_api[i].front_len == 0) throw std::runtime_error("checkout_frontiers with zero length (the empty version is the byte 00)");
        sh = Shared();
        sh.on = true; sh.n_entries = (uint32_t)n_api; sh.uniq_of = uniq_of;
        sh.slot_of.assign(n_api, 0); sh.fronts.assign(n_api, {});
        std::vector<uint32_t> used(udocs.size(), 0);
        for (size_t i = 0; i < n_api; i++) {
          if (!docs_api[i].front) continue;                      // the latest version: what the import run itself renders
          sh.fronts[i].assign(docs_api[i].front, docs_api[i].front + docs_api[i].front_len);
          sh.slot_of[i] = ++used[uniq_of[i]];
          if (sh.slot_of[i] > sh.G) sh.G = sh.slot_of[i];
        }
        sh.by_slot.assign(sh.G + 1, std::vector<uint32_t>(udocs.size(), lm::NONE));
        for (size_t i = 0; i < n_api; i++) sh.by_slot[sh.slot_of[i]][uniq_of[i]] = (uint32_t)i;   // (several entries at the latest version: the last one is rendered, the others copy it)
        docs = udocs.data(); n = udocs.size();
      }
    }
    n_docs = (uint32_t)n;
    ran = false;
    mapped = false;
    rd.on = false;
    sum_rows_padded = 0;             // (a new batch: lm_summary_layout is called again for it)
    for (auto& pt : parts) pt->sum_rows = nullptr;
    // split into contiguous ranges of about equal blob bytes; small batches stay in one part
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) for (size_t b = 0; b < docs[i].n; b++) total += docs[i].lens[b];
    uint32_t np = want_parts;
    while (np > 1 && (n < (size_t)np * part_min_docs)) np--;
    first.assign(1, 0);
    uint64_t acc = 0;
    for (size_t i = 0; i < n && first.size() < np; i++) {
      for (size_t b = 0; b < docs[i].n; b++) acc += docs[i].lens[b];
      if (acc * np >= total * first.size() && i + 1 < n) first.push_back((uint32_t)i + 1);
    }
    first.push_back((uint32_t)n);
    while (parts.size() < n_parts()) parts.emplace_back(new lm::Engine(device));
    // every part gathers and uploads its share on its own host thread and stream
    uint32_t np2 = n_parts();
    std::vector<std::string> errs(np2);
    for (uint32_t p = 0; p < np2; p++) parts[p]->allow_state = !sh.on;   // (a folded batch's entries are checked out: the history is what they need)
    auto body = [&](uint32_t p) { try { parts[p]->stage(docs + first[p], first[p + 1] - first[p]); } catch (const std::exception& e) { errs[p] = e.what(); if (errs[p].empty()) errs[p] = "error"; } };
    std::vector<std::thread> th;
    for (uint32_t p = 1; p < np2; p++) th.emplace_back(body, p);
    body(0);
    for (auto& t : th) t.join();
    for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
    if (sh.on) {
      // the staged documents become resident at once (lm_import straight after lm_stage: one import_batch, lm_pipeline.h)
      std::vector<lm::Engine::DocIn> none(n_docs, lm::Engine::DocIn{nullptr, nullptr, 0, nullptr, 0});
      import_parts(none.data(), n_docs);
      sh.part_of.assign(sh.n_entries, 0);
      for (uint32_t i = 0; i < sh.n_entries; i++) { uint32_t p = 0; while (p + 1 < n_parts() && sh.uniq_of[i] >= first[p + 1]) p++; sh.part_of[i] = p; }
      sh.d_out.resize(n_parts()); sh.d_vv.resize(n_parts()); sh.out_top.assign(n_parts(), 0); sh.vv_top.assign(n_parts(), 0);
      sh.h_out.assign(n_parts(), {}); sh.h_vv.assign(n_parts(), {});
    }
  }
  // lm_import: more blobs / other checkouts for the documents of the resident batch — each part takes its own documents
  void import_more(const lm::Engine::DocIn* docs, size_t n) {
    if (n != api_docs()) throw std::runtime_error("lm_import: the document count differs from the resident batch");
    for (size_t i = 0; i < n; i++)
      if (docs[i].front && docs[i].front_len == 0) throw std::runtime_error("checkout_frontiers with zero length (the empty version is the byte 00)");
    if (sh.on) unfold();
    import_parts(docs, n);
  }
  void import_parts(const lm::Engine::DocIn* docs, size_t n) {
    if (n != n_docs) throw std::runtime_error("lm_import: the document count differs from the resident batch");
    ran = false;
    if (!sh.on) rd.on = false;
    uint32_t np2 = n_parts();
    std::vector<std::string> errs(np2);
    std::vector<std::vector<lm::Engine::DocIn>> per;
    if (mapped) {
      per.resize(np2);
      for (uint32_t p = 0; p < np2; p++) per[p].resize(inv[p].size());
      for (size_t i = 0; i < n; i++) per[emap[i].first][emap[i].second] = docs[i];
    }
    auto body = [&](uint32_t p) { try { if (mapped) parts[p]->import_more(per[p].data(), per[p].size()); else parts[p]->import_more(docs + first[p], first[p + 1] - first[p]); } catch (const std::exception& e) { errs[p] = e.what(); if (errs[p].empty()) errs[p] = "error"; } };
    std::vector<std::thread> th;
#ifdef LM_PARALLEL_PARTS
    for (uint32_t p = 1; p < np2; p++) th.emplace_back(body, p);
    body(0);
#else
    for (uint32_t p = 0; p < np2; p++) body(p);
#endif
    for (auto& t : th) t.join();
    for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
  }
  void run() {
    if (!sh.on) {
      run_parts();
      std::vector<std::vector<uint32_t>> fl(n_parts());
      for (uint32_t p = 0; p < n_parts(); p++) if (!parts[p]->resident) fl[p] = parts[p]->redo_docs;
      redo_pass(fl);
      sum_host_rows();
      return;
    }
    rd.on = false;
    std::vector<std::vector<uint32_t>> redo_flagged(n_parts());
    // every lm_run is the whole job: the import (decode, DAG, replay from the empty version) and one rendering run per checkout slot
    uint32_t np = n_parts();
    std::vector<lm::KernelTime> all_times;
    sh.res.assign(sh.n_entries, lm::DocResult{0, 0, 0, 0, 0, 0, 0});
    std::vector<int32_t> import_status(n_docs, 0);
    {   // the import run renders the latest version (and the blob tables are those of the documents as staged)
      std::vector<lm::Engine::DocIn> none(n_docs, lm::Engine::DocIn{nullptr, nullptr, 0, nullptr, 0});
      import_parts(none.data(), n_docs);
    }
    for (uint32_t p = 0; p < np; p++) {
      lm::Engine& e = *parts[p];
      e.tables_valid = false; e.have_prev = false;
      std::fill(e.tk_reset.begin(), e.tk_reset.end(), (uint8_t)1);
      for (uint32_t i = 0; i < e.n_docs; i++) e.r_step[i] = (uint32_t)e.r_blobs[i].size();   // (a document whose import fails is empty afterwards, loro.rs:780-838)
      sh.out_top[p] = sh.vv_top[p] = 0;
    }
    std::vector<std::vector<std::vector<lm::Engine::BlobRef>>> blobs0(np);
    for (uint32_t p = 0; p < np; p++) blobs0[p] = parts[p]->r_blobs;
    for (uint32_t s = 0; s <= sh.G; s++) {
      if (s) {
        // the trackers move to this slot's versions (a document without an entry in the slot stays where it is)
        std::vector<lm::Engine::DocIn> in(n_docs, lm::Engine::DocIn{nullptr, nullptr, 0, nullptr, 0});
        for (uint32_t u = 0; u < n_docs; u++) {
          uint32_t en = sh.by_slot[s][u];
          for (uint32_t s2 = s; en == lm::NONE && s2-- > 1;) en = sh.by_slot[s2][u];
          if (en != lm::NONE && sh.slot_of[en]) { in[u].front = sh.fronts[en].data(); in[u].front_len = sh.fronts[en].size(); }
        }
        import_parts(in.data(), n_docs);
      }
      for (uint32_t p = 0; p < np; p++) parts[p]->shared_mode = s ? 2u : 1u;
      run_parts();
      for (auto& t : times) all_times.push_back(t);
      for (uint32_t p = 0; p < np; p++) {
        lm::Engine& e = *parts[p];
        lmbe::bind(e.sc);
        // this run's rendered bytes go behind the earlier runs' in the part's arenas (one copy each, on the part's stream)
        if (s == 0) { sh.d_out[p].ensure((e.out_bytes + 64) * (sh.G + 1) + (e.out_bytes >> 2) + 256); sh.d_vv[p].ensure((e.vv_bytes + 64) * (sh.G + 1) + 256); }
        sh.d_out[p].ensure_keep(sh.out_top[p] + e.out_bytes + 64, sh.out_top[p]);
        sh.d_vv[p].ensure_keep(sh.vv_top[p] + e.vv_bytes + 64, sh.vv_top[p]);
        if (e.out_bytes) lmbe::d2d((uint8_t*)sh.d_out[p].p + sh.out_top[p], e.b_out.p, e.out_bytes);
        if (e.vv_bytes) lmbe::d2d((uint8_t*)sh.d_vv[p].p + sh.vv_top[p], e.b_vv_out.p, e.vv_bytes);
        lmbe::sync();
        for (uint32_t i = 0; i < e.n_docs; i++) {
          uint32_t u = first[p] + i;
          if (s == 0) import_status[u] = e.results[i].status;
          // (LM_UNSUPPORTED without a rendering — a limit of the engine, a shallow snapshot — is the document's verdict at every version;
          // LM_UNSUPPORTED WITH one only says that out-of-scope containers ride along as null)
          if (s == 0 && e.results[i].status == lm::ST_UNSUPPORTED && e.results[i].json_len == 0) import_status[u] = -(int32_t)lm::ST_UNSUPPORTED;
          if (s == 0 && i == 0) redo_flagged[p] = e.redo_docs;   // (delete rows the resident kernels cannot finish by position: replayed below)
          uint32_t en = sh.by_slot[s][u];
          if (en == lm::NONE) continue;
          lm::DocResult r = e.results[i];
          // a document whose import failed fails for every entry (its later runs would render the empty document it is left as)
          if (s && import_status[u] != lm::ST_OK && import_status[u] != lm::ST_UNSUPPORTED) { r = lm::DocResult{import_status[u] < 0 ? -import_status[u] : import_status[u], 0, 0, 0, 0, 0, 0}; }
          r.json_off += sh.out_top[p]; r.vv_off += sh.vv_top[p];
          sh.res[en] = r;
        }
        sh.out_top[p] += e.out_bytes; sh.vv_top[p] += e.vv_bytes;
      }
    }
    // entries at the latest version beyond the one that was rendered: the same bytes
    for (uint32_t i = 0; i < sh.n_entries; i++) if (sh.slot_of[i] == 0) { uint32_t en = sh.by_slot[0][sh.uniq_of[i]]; if (en != i) sh.res[i] = sh.res[en]; }
    // the documents are as they were staged (a failed import dropped its blobs from the resident lists)
    for (uint32_t p = 0; p < np; p++) { parts[p]->r_blobs = blobs0[p]; parts[p]->shared_mode = 0; }
    times = all_times;
    ran = true;
    redo_pass(redo_flagged);
    sum_host_rows();
  }
  void run_parts() {
    uint32_t np = n_parts();
    for (uint32_t p = 0; p < np; p++) parts[p]->profiling = profiling != 0;
    std::vector<std::string> errs(np);
    auto body = [&](uint32_t p) { try { parts[p]->run(); } catch (const std::exception& e) { errs[p] = e.what(); if (errs[p].empty()) errs[p] = "error"; } };
#ifdef LM_PARALLEL_PARTS
    if (profiling == 1) {
      // stage timing: one part after the other, so a kernel's duration is its own and not a function of whatever the
      // other stream happened to run beside it (overlapped, the same kernel measures anywhere between 26 and 33 ms)
      for (uint32_t p = 0; p < np; p++) body(p);
    } else {
      std::vector<std::thread> th;
      for (uint32_t p = 1; p < np; p++) th.emplace_back(body, p);
      body(0);
      for (auto& t : th) t.join();
    }
#else
    for (uint32_t p = 0; p < np; p++) body(p);
#endif
    for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
    times.clear();
    for (uint32_t p = 0; p < np; p++) for (auto& t : parts[p]->times) times.push_back(t);
    ran = true;
  }
  bool redone(size_t i) const { return rd.on && i < rd.of.size() && rd.of[i] >= 0; }
  template <class F> void for_docs(F f) {
    if (sh.on) { for (uint32_t i = 0; i < sh.n_entries; i++) { if (redone(i)) f(i, *rd.eng, rd.res[rd.of[i]]); else f(i, *parts[sh.part_of[i]], sh.res[i]); } return; }
    if (rd.on) {
      for (uint32_t p = 0; p < n_parts(); p++)
        for (uint32_t i = 0; i < parts[p]->n_docs; i++) { if (redone(first[p] + i)) f(first[p] + i, *rd.eng, rd.res[rd.of[first[p] + i]]); else f(first[p] + i, *parts[p], parts[p]->results[i]); }
      return;
    }
    if (mapped) { for (uint32_t i = 0; i < n_docs; i++) f(i, *parts[emap[i].first], parts[emap[i].first]->results[emap[i].second]); return; }
    for (uint32_t p = 0; p < n_parts(); p++)
      for (uint32_t i = 0; i < parts[p]->n_docs; i++) f(first[p] + i, *parts[p], parts[p]->results[i]);
  }
};

extern "C" {

typedef struct lm_doc_in_c { const uint8_t* const* blobs; const size_t* blob_lens; size_t n_blobs; const uint8_t* checkout_frontiers; size_t checkout_len; } lm_doc_in_c;
typedef struct lm_doc_out_c { int32_t status; const uint8_t* json; size_t json_len; const uint8_t* vv; size_t vv_len; uint64_t pending_ops; } lm_doc_out_c;
typedef struct lm_run_stats_c { uint64_t n_docs, n_blobs, in_bytes, out_bytes, device_bytes_allocated; uint32_t n_kernels; } lm_run_stats_c;

void* LM_API(create)(int device) {
  if (!lmbe::init(device)) return nullptr;
  try { return new lm_ctx_impl(device); } catch (const std::exception&) { return nullptr; }
}
void LM_API(destroy)(void* c);
const char* LM_API(last_error)(void* c) { return c ? ((lm_ctx_impl*)c)->err.c_str() : "no context (HIP device unavailable)"; }

int LM_API(stage)(void* c, const lm_doc_in_c* docs, size_t n) {
  auto* x = (lm_ctx_impl*)c;
  try {
    std::vector<lm::Engine::DocIn> v(n);
    for (size_t i = 0; i < n; i++) v[i] = lm::Engine::DocIn{docs[i].blobs, docs[i].blob_lens, docs[i].n_blobs, docs[i].checkout_frontiers, docs[i].checkout_len};
    x->stage(v.data(), n);
    return 0;
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
// Resident documents: more blobs and / or other checkout versions for the documents of the batch staged last (same count, same
// order; n_blobs may be 0).  The next lm_run imports them into the documents the context already holds.
int LM_API(import)(void* c, const lm_doc_in_c* docs, size_t n) {
  auto* x = (lm_ctx_impl*)c;
  try {
    std::vector<lm::Engine::DocIn> v(n);
    for (size_t i = 0; i < n; i++) v[i] = lm::Engine::DocIn{docs[i].blobs, docs[i].blob_lens, docs[i].n_blobs, docs[i].checkout_frontiers, docs[i].checkout_len};
    x->import_more(v.data(), n);
    return 0;
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
// DiffMode of the last lm_run's import per resident document (0 Checkout, 1 Import, 2 ImportGreaterUpdates, 3 Linear; -1 = not
// computed: no resident run yet, a failed document, more than 16 common-ancestor ids), computed on the device by k_import_lca
int LM_API(import_modes)(void* c, int32_t* modes) {
  auto* x = (lm_ctx_impl*)c;
  for (size_t i = 0; i < x->api_docs(); i++) modes[i] = -1;
  if (x->sh.on) return 0;
  for (uint32_t p = 0; p < x->n_parts(); p++) {
    lm::Engine& e = *x->parts[p];
    for (uint32_t i = 0; i < e.n_docs; i++)
      if ((size_t)(i + 1) * lm::LCA_OUT <= e.h_lca.size()) { uint32_t m = e.h_lca[(size_t)i * lm::LCA_OUT]; modes[x->mapped ? x->inv[p][i] : x->first[p] + i] = m == lm::DM_UNKNOWN ? -1 : (int32_t)m; }
  }
  return 0;
}
// the common ancestors that import was measured from (dag.rs:487-765) as Frontiers::encode() bytes (postcard Vec<ID>, sorted);
// returns the length, or -1 when unknown / `cap` too small
long LM_API(import_lca)(void* c, size_t doc, uint8_t* buf, size_t cap) {
  auto* x = (lm_ctx_impl*)c;
  if (x->sh.on || doc >= x->n_docs) return -1;
  for (uint32_t p = 0; p < x->n_parts(); p++) {
    if (x->mapped ? x->emap[doc].first != p : (doc < x->first[p] || doc >= x->first[p + 1])) continue;
    lm::Engine& e = *x->parts[p];
    size_t i = x->mapped ? x->emap[doc].second : doc - x->first[p];
    if ((i + 1) * lm::LCA_OUT > e.h_lca.size()) return -1;
    const uint32_t* o = e.h_lca.data() + i * lm::LCA_OUT;
    if (o[0] == lm::DM_UNKNOWN) return -1;
    std::vector<uint8_t> b;
    lmenc::put_uleb(b, o[1]);
    for (uint32_t k = 0; k < o[1]; k++) {
      uint64_t peer = ((uint64_t)o[3 + 3 * k] << 32) | o[2 + 3 * k];
      lmenc::put_uleb(b, peer);
      lmenc::put_uleb(b, (uint64_t)o[4 + 3 * k] << 1);   // zigzag of a non-negative counter
    }
    if (b.size() > cap) return -1;
    memcpy(buf, b.data(), b.size());
    return (long)b.size();
  }
  return -1;
}
// documents of the last run that were replayed from the empty version (no usable resident tracker); diagnostics
int LM_API(resident_fresh)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  int n = 0;
  for (uint32_t p = 0; p < x->n_parts(); p++) n += (int)x->parts[p]->last_fresh;
  return n;
}
// diagnostics of the shared replay: the number of documents the batch staged last was folded into (0: every entry is its own document)
int LM_API(shared_documents)(void* c) { auto* x = (lm_ctx_impl*)c; return x->sh.on ? (int)x->n_docs : 0; }
int LM_API(fused_documents)(void* c) { auto* x = (lm_ctx_impl*)c; int n = 0; for (uint32_t p = 0; p < x->n_parts(); p++) n += (int)x->parts[p]->n_fused; return n; }
int LM_API(redo_documents)(void* c) { auto* x = (lm_ctx_impl*)c; return x->rd.on ? (int)x->rd.n : 0; }
int LM_API(state_documents)(void* c) { auto* x = (lm_ctx_impl*)c; int n = 0; for (uint32_t p = 0; p < x->n_parts(); p++) n += (int)x->parts[p]->n_state_docs; return n; }
int LM_API(run)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  try { x->run(); return 0; } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
// Asynchronous form of lm_run: returns at once, the pipeline runs on the context's own host threads and HIP streams.
// A server keeps two contexts in flight (double buffering): while one batch is in its integrate kernels the next
// batch's decode stages run beside it.  lm_wait blocks until the run is finished and reports its result.
int LM_API(run_async)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  if (x->in_flight) { x->err = "lm_run_async: a run is already in flight"; return -1; }
  if (x->runner.joinable()) x->runner.join();
  x->async_err.clear();
  x->in_flight = true;
  auto job = [x]() { try { x->run(); } catch (const std::exception& e) { x->async_err = e.what(); if (x->async_err.empty()) x->async_err = "error"; } };
#ifdef LM_PARALLEL_PARTS
  try { x->runner = std::thread(job); } catch (const std::exception& e) { x->in_flight = false; x->err = e.what(); return -1; }
#else
  job();   // the kernel-logic test harness is single-threaded: the run completes here, lm_wait only reports it
#endif
  return 0;
}
int LM_API(wait)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  if (!x->in_flight) return 0;
  if (x->runner.joinable()) x->runner.join();
  x->in_flight = false;
  if (!x->async_err.empty()) { x->err = x->async_err; return -1; }
  return 0;
}
int LM_API(fetch)(void* c, lm_doc_out_c* outs) {
  auto* x = (lm_ctx_impl*)c;
  try {
    if (!x->ran) throw std::runtime_error("lm_fetch before lm_run");
    if (x->sh.on) {
      for (uint32_t p = 0; p < x->n_parts(); p++) {
        lmbe::bind(x->parts[p]->sc);
        x->sh.h_out[p].resize(x->sh.out_top[p] + 1); x->sh.h_vv[p].resize(x->sh.vv_top[p] + 1);
        if (x->sh.out_top[p]) lmbe::d2h(x->sh.h_out[p].data(), x->sh.d_out[p].p, x->sh.out_top[p]);
        if (x->sh.vv_top[p]) lmbe::d2h(x->sh.h_vv[p].data(), x->sh.d_vv[p].p, x->sh.vv_top[p]);
        lmbe::sync();
      }
      if (x->rd.on) x->rd.eng->fetch();
      for (uint32_t i = 0; i < x->sh.n_entries; i++) {
        if (x->redone(i)) {
          const lm::DocResult& r = x->rd.res[x->rd.of[i]];
          outs[i].status = r.status;
          outs[i].json = x->rd.eng->h_out.data() + r.json_off; outs[i].json_len = (size_t)r.json_len;
          outs[i].vv = x->rd.eng->h_vv.data() + r.vv_off; outs[i].vv_len = (size_t)r.vv_len;
          outs[i].pending_ops = r.pending;
          continue;
        }
        const lm::DocResult& r = x->sh.res[i];
        uint32_t p = x->sh.part_of[i];
        outs[i].status = r.status;
        outs[i].json = x->sh.h_out[p].data() + r.json_off; outs[i].json_len = (size_t)r.json_len;
        outs[i].vv = x->sh.h_vv[p].data() + r.vv_off; outs[i].vv_len = (size_t)r.vv_len;
        outs[i].pending_ops = r.pending;
      }
      return 0;
    }
    for (uint32_t p = 0; p < x->n_parts(); p++) x->parts[p]->fetch();
    if (x->rd.on) x->rd.eng->fetch();
    x->for_docs([&](uint32_t i, lm::Engine& e, const lm::DocResult& r) {
      outs[i].status = r.status;
      outs[i].json = e.h_out.data() + r.json_off; outs[i].json_len = (size_t)r.json_len;
      outs[i].vv = e.h_vv.data() + r.vv_off; outs[i].vv_len = (size_t)r.vv_len;
      outs[i].pending_ops = r.pending;
    });
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
  if (!x->ran) { x->err = "lm_result_meta before lm_run"; return -1; }
  x->for_docs([&](uint32_t i, lm::Engine&, const lm::DocResult& r) {
    if (status) status[i] = r.status;
    if (json_len) json_len[i] = r.json_len;
    if (vv_len) vv_len[i] = r.vv_len;
    if (pending) pending[i] = r.pending;
  });
  return 0;
}
// xxh64 (seed 0) of every document's JSON, computed on the device by the last lm_run (0 for failed documents)
int LM_API(result_hashes)(void* c, uint64_t* json_xxh64) {
  auto* x = (lm_ctx_impl*)c;
  if (!x->ran) { x->err = "lm_result_hashes before lm_run"; return -1; }
  x->for_docs([&](uint32_t i, lm::Engine&, const lm::DocResult& r) { json_xxh64[i] = r.json_xxh64; });
  return 0;
}
// ---- export / encode side (host only): lm_encode.h
int LM_API(encode_block)(const lm_block_tables* t, uint8_t** out, size_t* out_len) {
  try {
    lmenc::Bytes b = lmenc::encode_block(*t);
    *out = (uint8_t*)malloc(b.size() ? b.size() : 1);
    if (!*out) return -1;
    memcpy(*out, b.data(), b.size());
    *out_len = b.size();
    return 0;
  } catch (...) {
    return -1;
  }
}
int LM_API(encode_updates)(const uint8_t* const* blocks, const size_t* lens, size_t n, uint8_t** out, size_t* out_len) {
  try {
    lmenc::Bytes b = lmenc::encode_updates(blocks, lens, n);
    *out = (uint8_t*)malloc(b.size());
    if (!*out) return -1;
    memcpy(*out, b.data(), b.size());
    *out_len = b.size();
    return 0;
  } catch (...) {
    return -1;
  }
}
// LoroDoc::export(ExportMode::Updates{from}) for document `doc` of the batch the context holds (after lm_run): the changes beyond
// `from_vv` (VersionVector::encode bytes; NULL / 0 = everything) as a FastUpdates blob.  malloc'ed; release with lm_free_bytes.
int LM_API(export)(void* c, size_t doc, const uint8_t* from_vv, size_t from_len, uint8_t** out, size_t* out_len) {
  auto* x = (lm_ctx_impl*)c;
  try {
    if (!x->ran) throw std::runtime_error("lm_export before lm_run");
    if (x->redone(doc)) {
      lmenc::Bytes b = x->rd.eng->export_doc((uint32_t)x->rd.of[doc], from_vv, from_len);
      *out = (uint8_t*)malloc(b.size() ? b.size() : 1);
      if (!*out) throw std::runtime_error("out of memory");
      memcpy(*out, b.data(), b.size());
      *out_len = b.size();
      return 0;
    }
    if (x->sh.on) { if (doc >= x->sh.n_entries) throw std::runtime_error("lm_export: no such document"); doc = x->sh.uniq_of[doc]; }
    if (x->mapped && doc >= x->n_docs) throw std::runtime_error("lm_export: no such document");
    for (uint32_t p = 0; p < x->n_parts(); p++) {
      if (x->mapped ? x->emap[doc].first != p : (doc < x->first[p] || doc >= x->first[p + 1])) continue;
      lmenc::Bytes b = x->parts[p]->export_doc(x->mapped ? x->emap[doc].second : (uint32_t)(doc - x->first[p]), from_vv, from_len);
      *out = (uint8_t*)malloc(b.size() ? b.size() : 1);
      if (!*out) throw std::runtime_error("out of memory");
      memcpy(*out, b.data(), b.size());
      *out_len = b.size();
      return 0;
    }
    throw std::runtime_error("lm_export: no such document");
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
void LM_API(free_bytes)(uint8_t* p) { free(p); }
// pinned host memory for the blobs a host hands to lm_stage (include/loro_merge.h "Direct staging")
void* LM_API(host_alloc)(size_t bytes) {
  if (!bytes) return nullptr;
  void* p = lmbe::halloc(bytes);
  if (p) lm::host_regions().add(p, bytes);
  return p;
}
void LM_API(host_free)(void* p) { if (p && lm::host_regions().remove(p)) lmbe::hfree(p); }
int LM_API(staged_direct)(void* c) { auto* x = (lm_ctx_impl*)c; int n = 0; for (uint32_t p = 0; p < x->n_parts(); p++) n += x->parts[p]->staged_direct ? 1 : 0; return n == (int)x->n_parts() ? 1 : 0; }
// Richtext values of the documents of the last lm_run (lm_k_richtext.h): lm_richtext renders them on the device and copies them
// back, lm_richtext_result hands out one document's bytes (valid until the next lm_richtext / lm_stage / lm_destroy).
int LM_API(richtext)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  try {
    if (!x->ran) throw std::runtime_error("lm_richtext before lm_run");
    {
      // documents staged from their snapshots' STATE sections hold neither marks nor the real ids of their child containers (the keys of
      // the richtext result): the batch is staged once more, every snapshot through its ChangeStore, and run again — the results of
      // lm_fetch / lm_result_meta are that run's from here on (the same bytes; a shallow snapshot, which has no history to give, is
      // LM_UNSUPPORTED in both from here on)
      bool any = false;
      for (uint32_t p = 0; p < x->n_parts(); p++) if (x->parts[p]->n_state_docs && !x->parts[p]->resident) { x->parts[p]->restage_history(); any = true; }
      if (any) { x->run(); x->sum_host_rows(); }
    }
    if (x->sh.on) {
      // a folded batch (entries that share their blobs, or any checked-out entry: one import, the trackers moved from version to
      // version): k_richtext reads the trackers as a run left them, so every entry becomes a resident document of its own over the
      // bytes already uploaded (Engine::expand) and is replayed and moved to ITS version — the price of asking for richtext values of
      // checked-out entries; the results of lm_fetch / lm_result_meta are those of that run from here on (the same bytes)
      x->unfold();
      x->ran = true;
      x->sum_host_rows();
    }
    for (uint32_t p = 0; p < x->n_parts(); p++) x->parts[p]->richtext();
    if (x->rd.on) x->rd.eng->richtext();
    return 0;
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
int LM_API(richtext_result)(void* c, size_t doc, int32_t* status, const uint8_t** json, size_t* json_len) {
  auto* x = (lm_ctx_impl*)c;
  try {
    if (x->sh.on || doc >= x->api_docs()) throw std::runtime_error("lm_richtext_result: no such document");
    if (x->redone(doc)) {   // (replayed in the side engine, DF_REDO)
      lm::Engine& e = *x->rd.eng;
      if (!e.rt_ran) throw std::runtime_error("lm_richtext_result before lm_richtext");
      const uint32_t i = (uint32_t)x->rd.of[doc];
      const bool ok = e.h_rt_status[i] == lm::ST_OK;
      if (status) *status = e.h_rt_status[i];
      if (json) *json = e.h_rt.data() + e.h_rt_off[i];
      if (json_len) *json_len = ok ? (size_t)e.h_rt_len[i] : 0;
      return 0;
    }
    for (uint32_t p = 0; p < x->n_parts(); p++) {
      if (x->mapped ? x->emap[doc].first != p : (doc < x->first[p] || doc >= x->first[p + 1])) continue;
      lm::Engine& e = *x->parts[p];
      if (!e.rt_ran) throw std::runtime_error("lm_richtext_result before lm_richtext");
      const uint32_t i = x->mapped ? x->emap[doc].second : (uint32_t)(doc - x->first[p]);
      const bool ok = e.h_rt_status[i] == lm::ST_OK;
      if (status) *status = e.h_rt_status[i];
      if (json) *json = e.h_rt.data() + e.h_rt_off[i];
      if (json_len) *json_len = ok ? (size_t)e.h_rt_len[i] : 0;
      return 0;
    }
    throw std::runtime_error("lm_richtext_result: no such document");
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
int LM_API(get_stats)(void* c, lm_run_stats_c* s) {
  auto* x = (lm_ctx_impl*)c;
  s->n_docs = x->api_docs(); s->n_blobs = 0; s->in_bytes = 0; s->out_bytes = 0;
  if (x->sh.on) {   // per ENTRY, as for a batch whose entries are documents of their own: its blobs in, its rendering out
    for (uint32_t i = 0; i < x->sh.n_entries; i++) {
      lm::Engine& e = *x->parts[x->sh.part_of[i]];
      uint32_t u = x->sh.uniq_of[i] - x->first[x->sh.part_of[i]];
      s->n_blobs += e.h_doc_blob[u + 1] - e.h_doc_blob[u];
      for (uint32_t b = e.h_doc_blob[u]; b < e.h_doc_blob[u + 1]; b++) s->in_bytes += e.h_blob_len[b];
      if (i < x->sh.res.size()) s->out_bytes += x->sh.res[i].json_len + x->sh.res[i].vv_len;
    }
  } else
  for (uint32_t p = 0; p < x->n_parts(); p++) {
    s->n_blobs += x->parts[p]->n_blobs; s->in_bytes += x->parts[p]->in_bytes; s->out_bytes += x->parts[p]->payload_bytes;
  }
  s->device_bytes_allocated = lmbe::allocated_bytes();
  s->n_kernels = (uint32_t)x->times.size();
  return 0;
}
// sizing diagnostics of the last run: max over documents of (leaves used, leaf capacity, elements)
int LM_API(sizing)(void* c, uint32_t* out3) {  // out3[3] = documents re-run with the worst-case directory, out3[4] = documents re-rendered at their exact size
  auto* x = (lm_ctx_impl*)c;
  out3[0] = out3[1] = out3[2] = out3[3] = out3[4] = 0;
  for (uint32_t p = 0; p < x->n_parts(); p++) {
    out3[3] += x->parts[p]->last_retries;
    out3[4] += x->parts[p]->last_reemits;
    for (auto& m : x->parts[p]->h_doc) { if (m.pad0 > out3[0]) out3[0] = m.pad0; if (m.leaf_cap > out3[1]) out3[1] = m.leaf_cap; if (m.n_elems > out3[2]) out3[2] = m.n_elems; }
  }
  return 0;
}
// returns the number of mismatches of the wave-primitive self test (0 = ok)
int LM_API(selftest)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  try {
    return x->parts[0]->selftest();
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
// LM_PROF builds only: sum over documents of the 16 cycle-accounting slots of k_integrate
int LM_API(prof_sum)(void* c, uint64_t* out16) {
  auto* x = (lm_ctx_impl*)c;
  for (int i = 0; i < 16; i++) out16[i] = 0;
  bool any = false;
  for (uint32_t p = 0; p < x->n_parts(); p++) {
    auto& h = x->parts[p]->h_prof;
    for (size_t k = 0; k < h.size(); k++) out16[k % 16] += h[k];
    any |= !h.empty();
  }
  return any ? 0 : -1;
}
int LM_API(set_profiling)(void* c, int en) { ((lm_ctx_impl*)c)->profiling = en < 0 ? 0 : (en > 2 ? 2 : en); return 0; }
int LM_API(kernel_time)(void* c, uint32_t i, const char** name, double* ms) {
  auto* x = (lm_ctx_impl*)c;
  if (i >= x->times.size()) return -1;
  *name = x->times[i].name.c_str();
  *ms = x->times[i].ms;
  return 0;
}
// ---- the exchange step of a sharded deployment for a host without Python / torch: every rank contributes the
// summary of its own documents — 6 words each: document id, status, pending ops, JSON length, VV length, xxh64 of the JSON
// (computed on the device, k_hash_json) — and receives the table of all documents, by ONE RCCL all-gather over xGMI (after a
// one-word all-gather of the shard sizes).  RCCL is loaded at lm_comm_init (dlopen): the library itself does not depend on it.
// (The kernel-logic harness compiles the same exchange code: its "device" memory is host memory and there is no stream, so a
// collective library that works on host pointers — tests/emu/rccl_stub.c, named through LM_RCCL_LIB — exercises lm_comm_init,
// lm_summary_allgather and lm_summary_allgather_device end to end between two processes without a GPU.  LM_RCCL_LIB also lets a
// deployment name its librccl.so explicitly.)
#include <dlfcn.h>
namespace lmcomm {
#ifndef LM_EMU
typedef hipStream_t Stream;
inline Stream cur_stream() { return lmbe::cur->s; }
#else
typedef void* Stream;
inline Stream cur_stream() { return nullptr; }
#endif
typedef struct { char internal[128]; } UniqueId;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, Stream);
typedef int (*CommDestroyFn)(void*);
struct Api { void* h = nullptr; GetUniqueIdFn get_id = nullptr; CommInitRankFn init = nullptr; AllGatherFn all_gather = nullptr; CommDestroyFn destroy = nullptr; };
inline Api& api() {
  static Api a;
  if (!a.h) {
    if (const char* e = getenv("LM_RCCL_LIB")) a.h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
#ifndef LM_EMU
    if (!a.h) a.h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!a.h) a.h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
#endif
    if (a.h) {
      a.get_id = (GetUniqueIdFn)dlsym(a.h, "ncclGetUniqueId"); a.init = (CommInitRankFn)dlsym(a.h, "ncclCommInitRank");
      a.all_gather = (AllGatherFn)dlsym(a.h, "ncclAllGather"); a.destroy = (CommDestroyFn)dlsym(a.h, "ncclCommDestroy");
    }
  }
  return a;
}
}  // namespace lmcomm
struct lm_comm_state { int rank = 0, world = 1; void* comm = nullptr; };
static std::map<void*, lm_comm_state>& lm_comms() { static std::map<void*, lm_comm_state> m; return m; }
static std::mutex& lm_comms_mu() { static std::mutex m; return m; }   // contexts may be created / destroyed from several threads

// rank 0 creates the id and hands it to the other ranks out of band (128 bytes)
int LM_API(comm_unique_id)(uint8_t* out128) {
  auto& a = lmcomm::api();
  if (!a.get_id) return -1;
  lmcomm::UniqueId id;
  if (a.get_id(&id) != 0) return -1;
  memcpy(out128, id.internal, 128);
  return 0;
}
// world == 1 needs no communicator (and no RCCL); otherwise every rank calls this with the same id, on its own context
int LM_API(comm_init)(void* c, int rank, int world, const uint8_t* id128) {
  auto* x = (lm_ctx_impl*)c;
  if (world < 1 || rank < 0 || rank >= world) { x->err = "lm_comm_init: rank / world"; return -1; }
  lm_comm_state st;
  st.rank = rank; st.world = world;
  if (world > 1) {
    auto& a = lmcomm::api();
    if (!a.init || !a.all_gather) { x->err = "lm_comm_init: librccl.so could not be loaded"; return -1; }
#ifndef LM_EMU
    (void)hipSetDevice(x->device);
#endif
    lmcomm::UniqueId id;
    memcpy(id.internal, id128, 128);
    if (a.init(&st.comm, world, id, rank) != 0) { x->err = "ncclCommInitRank failed"; return -1; }
  }
  { std::lock_guard<std::mutex> lk(lm_comms_mu()); lm_comms()[c] = st; }
  return 0;
}
// ---- the exchange as ONE collective on device memory.  lm_summary_layout (after lm_stage): this context's
// document i has the global id first_id + i * stride, and every rank contributes `rows_padded` rows (>= its document count; with
// documents dealt `doc % world` that is ceil(total / world) — computed, not exchanged).  From then on every lm_run writes the rows
// on the device (k_summary_rows; rows beyond the context's documents stay -1); lm_summary_rows_device is that buffer (a host that
// drives its own collective — torch.distributed in bench.py — sends it as it is), lm_summary_allgather_device issues the single
// ncclAllGather of rows_padded x 6 int64 per rank and leaves the gathered table (world x rows_padded rows, rank-major, -1 rows =
// padding) on the device until somebody asks for it.
int LM_API(summary_layout)(void* c, int64_t first_id, int64_t stride, size_t rows_padded) {
  auto* x = (lm_ctx_impl*)c;
  try {
    if (rows_padded < x->api_docs()) throw std::runtime_error("lm_summary_layout: fewer rows than staged documents");
    if (x->in_flight) throw std::runtime_error("lm_summary_layout while a run is in flight");
    lm::Engine& e = *x->parts[0];
    lmbe::bind(e.sc);
    x->sum_buf.ensure((rows_padded ? rows_padded : 1) * 48);
    lmbe::dmemset(x->sum_buf.p, 0xff, (rows_padded ? rows_padded : 1) * 48);   // every word -1
    lmbe::sync();
    x->sum_rows_padded = rows_padded; x->sum_first = first_id; x->sum_stride = stride;
    x->sum_bind();
    return 0;
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
const int64_t* LM_API(summary_rows_device)(void* c) {
  auto* x = (lm_ctx_impl*)c;
  return x->sum_rows_padded ? (const int64_t*)x->sum_buf.p : nullptr;
}
long LM_API(summary_allgather_device)(void* c, const int64_t** table_dev) {
  auto* x = (lm_ctx_impl*)c;
  try {
    if (!x->ran) throw std::runtime_error("lm_summary_allgather_device before lm_run");
    if (!x->sum_rows_padded) throw std::runtime_error("lm_summary_allgather_device before lm_summary_layout");
    lm_comm_state st;
    { std::lock_guard<std::mutex> lk(lm_comms_mu()); auto it = lm_comms().find(c); if (it != lm_comms().end()) st = it->second; }
    if (st.world == 1) { *table_dev = (const int64_t*)x->sum_buf.p; return (long)x->sum_rows_padded; }
    auto& a = lmcomm::api();
    lm::Engine& e = *x->parts[0];
    lmbe::bind(e.sc);
    x->sum_all.ensure((size_t)st.world * x->sum_rows_padded * 48);
    if (a.all_gather(x->sum_buf.p, x->sum_all.p, x->sum_rows_padded * 6, /*ncclInt64*/ 4, st.comm, lmcomm::cur_stream()) != 0) throw std::runtime_error("ncclAllGather (summaries) failed");
    lmbe::sync();
    *table_dev = (const int64_t*)x->sum_all.p;
    return (long)((size_t)st.world * x->sum_rows_padded);
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}
// doc_ids[n_docs] = the global ids of this context's documents; table receives rows of 6 int64 (layout above = loro_amd/dist.py
// SUMMARY_WORDS) for the documents of ALL ranks, sorted by document id; returns the number of rows, -1 on error
// (the host-table form: shard sizes are exchanged first, the rows pass through host memory; lm_summary_layout +
// lm_summary_allgather_device is the one-collective form)
long LM_API(summary_allgather)(void* c, const int64_t* doc_ids, int64_t* table, size_t cap_rows) {
  auto* x = (lm_ctx_impl*)c;
  try {
    if (!x->ran) throw std::runtime_error("lm_summary_allgather before lm_run");
    lm_comm_state st;
    { std::lock_guard<std::mutex> lk(lm_comms_mu()); auto it = lm_comms().find(c); if (it != lm_comms().end()) st = it->second; }
    size_t n = x->api_docs();
    std::vector<int64_t> local(n * 6);
    x->for_docs([&](uint32_t i, lm::Engine&, const lm::DocResult& r) {
      int64_t* w = local.data() + (size_t)i * 6;
      w[0] = doc_ids[i]; w[1] = r.status; w[2] = (int64_t)r.pending; w[3] = (int64_t)r.json_len; w[4] = (int64_t)r.vv_len; w[5] = (int64_t)r.json_xxh64;
    });
    std::vector<int64_t> all;
    if (st.world == 1) all = local;
    else {
      auto& a = lmcomm::api();
      lm::Engine& e = *x->parts[0];
      lmbe::bind(e.sc);
      // shard sizes, then the tables padded to the largest shard: the same two collectives loro_amd/dist.py issues
      lm::DBuf sz_in, sz_out, tb_in, tb_out;
      sz_in.ensure(8); sz_out.ensure((size_t)st.world * 8);
      int64_t nn = (int64_t)n;
      lmbe::h2d(sz_in.p, &nn, 8);
      if (a.all_gather(sz_in.p, sz_out.p, 1, /*ncclInt64*/ 4, st.comm, lmcomm::cur_stream()) != 0) throw std::runtime_error("ncclAllGather (sizes) failed");
      std::vector<int64_t> sizes(st.world);
      lmbe::d2h(sizes.data(), sz_out.p, (size_t)st.world * 8);
      int64_t n_max = 0;
      for (int64_t s2 : sizes) n_max = s2 > n_max ? s2 : n_max;
      std::vector<int64_t> padded((size_t)n_max * 6, -1);
      memcpy(padded.data(), local.data(), local.size() * 8);
      tb_in.ensure((size_t)n_max * 48 + 8); tb_out.ensure((size_t)st.world * n_max * 48 + 8);
      lmbe::h2d(tb_in.p, padded.data(), padded.size() * 8);
      if (a.all_gather(tb_in.p, tb_out.p, (size_t)n_max * 6, 4, st.comm, lmcomm::cur_stream()) != 0) throw std::runtime_error("ncclAllGather (summaries) failed");
      std::vector<int64_t> raw((size_t)st.world * n_max * 6);
      lmbe::d2h(raw.data(), tb_out.p, raw.size() * 8);
      for (int r = 0; r < st.world; r++) all.insert(all.end(), raw.begin() + (size_t)r * n_max * 6, raw.begin() + ((size_t)r * n_max + sizes[r]) * 6);
      sz_in.release(); sz_out.release(); tb_in.release(); tb_out.release();
    }
    size_t rows = all.size() / 6;
    if (rows > cap_rows) throw std::runtime_error("lm_summary_allgather: the table does not fit");
    std::vector<size_t> order(rows);
    for (size_t i = 0; i < rows; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t p, size_t q) { return all[p * 6] < all[q * 6]; });
    for (size_t i = 0; i < rows; i++) memcpy(table + i * 6, all.data() + order[i] * 6, 48);
    return (long)rows;
  } catch (const std::exception& e) { x->err = e.what(); return -1; }
}

void LM_API(destroy)(void* c) {
  lm_comm_state st;
  {
    std::lock_guard<std::mutex> lk(lm_comms_mu());
    auto it = lm_comms().find(c);
    if (it != lm_comms().end()) { st = it->second; lm_comms().erase(it); }
  }
  if (st.comm && lmcomm::api().destroy) (void)lmcomm::api().destroy(st.comm);
  delete (lm_ctx_impl*)c;
}
// number of engine parts (HIP streams) the last staged batch was split into
int LM_API(n_streams)(void* c) { return (int)((lm_ctx_impl*)c)->n_parts(); }
}
