/* loro_merge.h — C ABI of the MI355X batched CRDT merge engine.
 *
 * Drop-in boundary for Loro's import → diff_calc → state path.  The reference has no FFI on this path
 * (SURVEY.md §8b); the boundary therefore sits at the byte interface the Rust host already owns:
 *
 *   input   exactly what `LoroDoc::export(ExportMode::Updates{..})` (or ExportMode::Snapshot, see "Limits") produces and
 *           `LoroDoc::import()` consumes
 *           (crates/loro/src/lib.rs:710,1306; crates/loro-internal/src/encoding.rs:334-373,399-405;
 *           docs/encoding.md §2,§6-10).  The blobs of one document are imported in order, like
 *           `LoroDoc::import_batch` (crates/loro-internal/src/loro.rs:1432-1523), into an empty document.
 *   output  what `doc.get_deep_value().to_json_value()` (crates/loro/src/lib.rs:937,
 *           crates/loro-internal/src/state.rs:1294-1329) and `doc.oplog_vv().encode()`
 *           (crates/loro/src/lib.rs:887, crates/loro-internal/src/version.rs:962-964) produce:
 *           canonical JSON (UTF-8, object keys sorted bytewise, no whitespace) and a postcard map
 *           peer→exclusive counter with entries sorted by peer.
 *
 * Errors are per document and mirror LoroError (crates/loro-common/src/error.rs): a bad document never
 * fails the batch (reference import is per-document atomic, loro.rs:780-838).  Changes whose dependencies
 * are missing are not errors: they are reported as pending (ImportStatus.pending, encoding.rs:227-231) and
 * excluded from state and version vector.
 *
 * Threading: a context is single-caller; several contexts may be used concurrently.
 * Ownership: inputs are borrowed for the duration of the call; outputs belong to the context and stay
 * valid until the next lm_merge_batch / lm_run / lm_destroy on it.
 */
#ifndef LORO_MERGE_H
#define LORO_MERGE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
  LM_OK = 0,
  LM_DECODE_ERROR = 1,       /* LoroError::DecodeError */
  LM_CHECKSUM_MISMATCH = 2,  /* LoroError::DecodeChecksumMismatchError */
  LM_DATA_CORRUPTION = 3,    /* LoroError::DecodeDataCorruptionError */
  LM_UNSUPPORTED = 4,        /* outside the device path's scope — see "Limits" below */
  LM_INTERNAL = 5,
  LM_FRONTIERS_NOT_FOUND = 6 /* LoroError::FrontiersNotFound: a checkout id the imported history does not hold */
};

/* Limits of the device path.  A document beyond one of them is reported LM_UNSUPPORTED — never a guessed value — and
 * does not disturb the other documents of the batch (tests: `documented_limits_are_reported_not_guessed`, emu + GPU):
 *   - container kinds: Map, List, Text, MovableList (root or child; MovableList: insert / delete / move / set, per-element
 *     last-writer-wins of position and value — diff_calc.rs:1669-1993, history_cache.rs:754-1003).  A document that also
 *     holds Tree / Counter containers is rendered with those as null and reported LM_UNSUPPORTED *together with* its JSON
 *     and VV;
 *   - blobs: EncodeMode::FastUpdates (mode 4) and FastSnapshot (mode 3).  A snapshot is ingested on the host in front of
 *     the device path: its history from the ChangeStore section (replayed like updates, fast_snapshot.rs:326-344), the set
 *     of root containers from the keys of its state section (an empty document initialises its state store from that
 *     section, fast_snapshot.rs:168-258, so a root in which nothing is visible is still part of the value).  The first
 *     snapshot among a document's blobs plays that role (import_batch imports snapshots first).  A document that is ONE
 *     snapshot rendered at its latest version is staged from the state section's VALUES instead (lm_state_documents below):
 *     no history is uploaded, decoded or replayed.  Shallow snapshots (history trimmed below a shallow root) are rendered
 *     that way at their latest version and at their shallow root; at any other version, next to other blobs, under
 *     lm_import or lm_richtext they are LM_UNSUPPORTED;
 *   - per document: <= 255 peers, <= 256 containers of which <= 64 roots, container nesting <= 16, counters < 2^24 per
 *     peer (element ids are packed peer:8 | counter:24), < 2^24 Map / MovableList move+set op rows, <= 18,000 tracker leaves per sequence
 *     replay (~190k op runs; the 1M-op documents of BASELINE configs[4] use ~1,200), JSON < 4 GiB, a blob < 4 GiB;
 *   - two root containers with the same name but different kinds; a StyleEnd op that does not directly follow its
 *     StyleStart (every writer emits them as a pair).  */
typedef struct lm_ctx lm_ctx;

typedef struct lm_doc_in {
  const uint8_t* const* blobs; /* n_blobs update blobs (EncodeMode::FastUpdates), imported in order */
  const size_t* blob_lens;
  size_t n_blobs;
  /* Optional checkout (LoroDoc::checkout, crates/loro-internal/src/loro.rs:1625-1760): render the state at
   * these frontiers instead of the latest version.  Bytes = Frontiers::encode() (version/frontiers.rs:219-223:
   * postcard Vec<ID>, sorted).  NULL = latest.  The one-byte encoding 00 is the empty version.  With a checkout
   * the vv output is the version of the rendered state (state_vv), not oplog_vv.  The batch entry replays the version's causal
   * closure only (same bytes as import + checkout on every healthy document); LM_CHECKOUT_FULL=1 in the environment makes it import
   * the whole history first, as the reference does — a document damaged OUTSIDE the rendered version then fails like there. */
  const uint8_t* checkout_frontiers;
  size_t checkout_len;
} lm_doc_in;

typedef struct lm_doc_out {
  int32_t status;       /* LM_* */
  const uint8_t* json;  /* canonical deep-value JSON (not NUL terminated) */
  size_t json_len;
  const uint8_t* vv;    /* postcard VersionVector, entries sorted by peer */
  size_t vv_len;
  uint64_t pending_ops; /* atoms parked because a dependency is missing */
} lm_doc_out;

/* Create a context on HIP device `device` (>= 0).  Returns NULL when no usable MI355X-class device or the
 * HIP runtime is missing — there is no CPU fallback. */
lm_ctx* lm_create(int device);
void lm_destroy(lm_ctx* ctx);
const char* lm_last_error(lm_ctx* ctx);

/* One-shot: stage + run + fetch.  Returns 0 on success (per-document results in outs[i].status). */
int lm_merge_batch(lm_ctx* ctx, const lm_doc_in* docs, size_t n_docs, lm_doc_out* outs);

/* Split form, for callers that keep batches resident in HBM (and for benchmarking the device path):
 * lm_stage packs the blobs and uploads them, lm_run executes the device pipeline on the staged batch
 * (may be repeated), lm_fetch copies the rendered states back and fills outs[0..n_docs). */
int lm_stage(lm_ctx* ctx, const lm_doc_in* docs, size_t n_docs);
int lm_run(lm_ctx* ctx);

/* Direct staging (round 6).  lm_stage copies the blobs host -> pinned staging buffer -> HBM; the first hop (a gather by a few host
 * threads) is what bounds a server that feeds the engine from pageable memory (~38 GB/s on the development box).  A host that
 * receives its blobs INTO memory obtained from lm_host_alloc (pinned; release with lm_host_free) skips it: when every blob of a
 * batch lies inside one such region — each at a 16-byte aligned address, in the order they are staged (document by document, blob
 * by blob), without overlap, and packed reasonably densely (the span from the first blob to the end of the last is copied whole:
 * it must not exceed twice the blobs' bytes + 1 MiB) — lm_stage hands that span to the copy engine as it is and touches no byte
 * of it.  The bytes between blobs are never interpreted.  The region must stay valid and unchanged until the NEXT lm_stage of the
 * context (documents that are replayed — lm_redo_documents, lm_import after a state-staged batch — are read from it again).
 * Anything else (a blob outside the region, unaligned, out of order; a snapshot, which is reframed on the host) takes the gather
 * as before.  lm_staged_direct: 1 when the batch staged last took this path.  LM_STAGE_DIRECT=0 switches it off. */
void* lm_host_alloc(size_t bytes);
void lm_host_free(void* p);
int lm_staged_direct(lm_ctx* ctx);
int lm_fetch(lm_ctx* ctx, lm_doc_out* outs);
/* Asynchronous lm_run: lm_run_async returns at once, lm_wait blocks until that run is finished (0 = ok).  Between the
 * two calls only other contexts may be used.  Two contexts in flight overlap one batch's decode stages with the other's
 * integrate kernels (double buffering). */
int lm_run_async(lm_ctx* ctx);
int lm_wait(lm_ctx* ctx);

/* Shared replay (a live LoroDoc serves every checkout from ONE imported history and its persistent DiffCalculator,
 * crates/loro-internal/src/loro.rs:1625-1760, diff_calc.rs:62-68): entries of a staged batch that name the same blobs — the same
 * pointers and lengths, in the same order — and differ only in checkout_frontiers are ONE document rendered at several versions.
 * lm_stage uploads such a document once; every lm_run imports it once (decode, causal graph, replay from the empty version) and
 * renders each entry by moving the document's trackers to the entry's version, instead of replaying the history once per entry.
 * Results are per entry and are those of import_batch + checkout on a document of its own.  lm_import on such a batch unfolds it
 * first — every entry becomes a resident document of its own over the bytes already in HBM — and then behaves as always; so does
 * lm_richtext (the trackers have to stand at every entry's version).  Since round 6 EVERY entry with checkout_frontiers takes this
 * path, alone in its group too: the whole history is imported before the version is rendered, as LoroDoc::import + LoroDoc::checkout
 * do — damage outside the rendered version fails the entry like the reference (LM_CHECKOUT_FULL=0: the closure replay of rounds
 * 1-5).  LM_SHARE_REPLAY=0 in the environment switches the folding off.  lm_shared_documents: the
 * number of documents the batch staged last was folded into (0 = every entry is its own document, also after an lm_import). */
int lm_shared_documents(lm_ctx* ctx);

/* Diagnostics of the last lm_run (round 6).  lm_fused_documents: LWW Map documents whose blocks hold scalar writes only and were
 * resolved WITHOUT op rows — op columns folded straight into the document's LWW table (diff_calc.rs:488-616 MapDiffCalculator over
 * block_encode.rs:417-428 columns; loro_amd/csrc/lm_k_map_fused.h).  lm_redo_documents: documents (entries) the run's configuration
 * had no path for and that were replayed once more, from the staged bytes, through the span-granular batch pipeline — a delete row
 * that does not match its position under the element-granular or the resident kernels (applied by position like the reference,
 * crdt_rope.rs:256-335), a Map document the fused kernel gave up on: the verdict of a document depends neither on its neighbours
 * in the batch nor on whether the host reused blob pointers. */
int lm_fused_documents(lm_ctx* ctx);
int lm_redo_documents(lm_ctx* ctx);
/* lm_state_documents: documents of the batch staged last that were staged from their snapshot's STATE section instead of its history
 * (encoding/fast_snapshot.rs:168-258: an empty document that imports a snapshot takes its state store from that
 * section and replays nothing).  A document qualifies when it is ONE FastSnapshot blob rendered at its latest version — or ONE
 * shallow snapshot rendered at exactly its shallow root (checkout_frontiers == shallow_since_frontiers, loro_js_interop.rs:141-147),
 * or ONE snapshot + update blobs whose changes continue its history (every dependency below the snapshot's version is its
 * frontiers; loro_amd/csrc/lm_snapshot_base.h) — and every state value is of a kind this engine renders; the staged bytes are then
 * proportional to the STATE (+ the updates), the snapshot's history is neither uploaded nor decoded nor replayed, and `vv` is the
 * snapshot's own merged with what the updates add.  Updates concurrent with part of the snapshot's history need that history: such
 * a document is replayed from the snapshot's ChangeStore (same bytes out).  A later lm_import into such a batch stages the
 * snapshots once more through their ChangeStore (a history is what an import builds on).  Snapshot + updates is taken where it is the
 * cheaper replay (lm_snapshot_base.h `pays`): always when the updates are ONE chain behind the snapshot's frontiers (replayed by the
 * linear prefix: 2.2 x the rate of the history path on configs[1]-shaped documents), with concurrent branches only while the updates
 * hold at most 4,096 ops or a fifth of the history's (their deletes of base content go through the tracker's by-position path).  A
 * document staged on a state that fails in ANY way is replayed from its snapshot's history (lm_redo_documents counts it): the verdict
 * is always the history's.  LM_SNAPSHOT_STATE=0 switches the state path off, =2 takes it wherever it is possible. */
int lm_state_documents(lm_ctx* ctx);

/* Resident documents: import MORE blobs into the documents the context already holds, and / or render them
 * at other versions — what a Rust host does with
 *     doc.import(bytes)           crates/loro/src/lib.rs:710 → loro-internal/src/loro.rs:568-649,720-851 (a document with history)
 *     doc.checkout(&frontiers)    loro.rs:1625-1760;  doc.checkout_to_latest()  loro.rs:1396-1417
 * on documents it keeps alive, instead of building each one again from all its blobs.  `docs` has one entry per document of
 * the batch staged last (same count, same order): its n_blobs (possibly 0) blobs are appended to that document's history, its
 * checkout_frontiers (NULL = the latest version) is the version the next lm_run renders.  The next lm_run then
 *   - decodes the blobs and rebuilds the causal graph (the tables are per batch),
 *   - continues every sequence container's tracker from where the previous run left it (the reference keeps its trackers the
 *     same way: DiffCalculatorRetainMode::Persist, diff_calc.rs:62-68, start_tracking reuse :1371-1376): only the changes the
 *     tracker has not applied are integrated — each after the tracker moved to the change's dependencies (Tracker::checkout /
 *     forward, container/richtext/tracker.rs:350-546) — and the tracker finally moves to the version being rendered,
 *   - renders JSON + VersionVector as always.  A run that only changes the versions (no new blob anywhere) skips the decode.
 * Semantics = the sequence of calls above, NOT one import_batch of all blobs: a root Text / List whose content was visible
 * after some earlier lm_run stays part of the value when it is empty later ("text":"" — the state store never drops a
 * container state, state.rs:621-849), where one batch of the same blobs omits it.  Each lm_run is: checkout_to_latest,
 * import of the step's blobs (all of a document's blobs of one lm_import, or none: a document whose import fails — status
 * LM_DECODE_ERROR, LM_CHECKSUM_MISMATCH, LM_DATA_CORRUPTION ... — keeps exactly what it held before, loro.rs:780-838),
 * then the optional checkout; a refused checkout (LM_FRONTIERS_NOT_FOUND) does not undo the import in front of it.
 * lm_stage starts a new batch and drops everything resident.  Limits: the span-granular integrate kernel (the default);
 * a document's blobs must lie within 4 GiB of the context's blob arena; a document that holds a MovableList is replayed from
 * the empty version whenever its element layout shifts (its results are the same, its import is not incremental). */
int lm_import(lm_ctx* ctx, const lm_doc_in* docs, size_t n_docs);
/* What the reference computes before it diffs an import — the common ancestors of the version a document was at and the version
 * it reaches, and the DiffMode they imply (dag.rs:318-332,487-765 `find_common_ancestor`; oplog.rs:591-615; diff_calc.rs:72-103) —
 * computed on the device for every resident document by the last lm_run (one step = one import of all the step's blobs).
 * modes[i]: 0 Checkout, 1 Import, 2 ImportGreaterUpdates, 3 Linear, -1 unknown (no resident run, failed document, more than 16
 * common-ancestor ids).  lm_import_lca writes Frontiers::encode() of document `doc`'s common ancestors into buf (returns the
 * length, -1 when unknown or cap is too small).  The engine itself does not branch on the mode — its trackers hold the whole
 * history, so the replay base is wherever the tracker stands — it reports what a host that mirrors the reference needs. */
int lm_import_modes(lm_ctx* ctx, int32_t* modes);
long lm_import_lca(lm_ctx* ctx, size_t doc, uint8_t* buf, size_t cap);
/* Diagnostics: how many documents of the last lm_run could not continue from a resident tracker and were replayed from the
 * empty version (first run, capacity grown, a failed run before, MovableList layout shift). */
int lm_resident_fresh(lm_ctx* ctx);

/* Per-document metadata of the last lm_run (arrays of n_docs entries, any may be NULL) without copying the
 * rendered bytes back: what a sharded deployment all-gathers as the merged-state summary. */
int lm_result_meta(lm_ctx* ctx, int32_t* status, uint64_t* json_len, uint64_t* vv_len, uint64_t* pending_ops);
/* xxh64 (seed 0) of every document's JSON (n_docs entries; 0 for a failed document), computed on the device: the content
 * word of the merged-state summary — ranks compare merged states without moving the JSON. */
int lm_result_hashes(lm_ctx* ctx, uint64_t* json_xxh64);

/* ---- The exchange step of a sharded deployment for a host without Python: documents shard across GPUs with no
 * data-path collective; after lm_run every rank contributes the summary of its own documents — 6 int64 words per document:
 * document id, status, pending ops, JSON length, VV length, xxh64(JSON) computed on the device — and receives the table of all
 * documents of all ranks, sorted by document id, through ONE RCCL all-gather over xGMI (preceded by a one-word all-gather of the
 * shard sizes).  One process per GPU: rank 0 calls lm_comm_unique_id and hands the 128 bytes to the others out of band; every
 * rank calls lm_comm_init(ctx, rank, world, id) once (world == 1 needs neither an id nor RCCL), then lm_summary_allgather after
 * each lm_run.  RCCL is dlopen'ed at lm_comm_init; the library does not link it.  Returns rows written, -1 on error. */
int lm_comm_unique_id(uint8_t out128[128]);
int lm_comm_init(lm_ctx* ctx, int rank, int world, const uint8_t id128[128]);
long lm_summary_allgather(lm_ctx* ctx, const int64_t* doc_ids, int64_t* table, size_t cap_rows);
/* The same exchange as ONE collective on device memory.  lm_summary_layout (after lm_stage): document i of this context has the
 * global id first_id + i * stride and every rank contributes rows_padded rows (>= its document count; with documents dealt
 * `doc % world` that is ceil(total / world): computed, never exchanged).  Every lm_run then writes the rows ON THE DEVICE (rows
 * beyond the context's documents are -1): lm_summary_rows_device is that buffer — a host that drives its own collective sends it
 * as it is (bench.py: torch.distributed over RCCL) — and lm_summary_allgather_device issues the single ncclAllGather of
 * rows_padded x 6 int64 per rank and returns the gathered table (world x rows_padded rows, rank-major), which stays in device
 * memory until it is read.  lm_stage drops the layout. */
int lm_summary_layout(lm_ctx* ctx, int64_t first_id, int64_t stride, size_t rows_padded);
const int64_t* lm_summary_rows_device(lm_ctx* ctx);
long lm_summary_allgather_device(lm_ctx* ctx, const int64_t** table_dev);

/* ---- Export / encode side (host only, no device needed): the inverse of the decode stage.
 * lm_block_tables = the tables of ONE change block — the changes of one peer, counter-contiguous — exactly what the decode
 * stage extracts from a block (crates/loro-internal/src/oplog/change_store/block_encode.rs:535-706); lm_encode_block writes
 * the bytes the reference's encode_block (block_encode.rs:137-278, block_meta_encode.rs:13-88) writes for them, and
 * lm_encode_updates frames encoded blocks into a FastUpdates blob (encoding.rs:440-473, fast_snapshot.rs:346-360).  Value
 * payloads and the position arena are opaque sections: their bytes are carried as given.  Outputs are malloc'ed; release
 * them with lm_free_bytes.  Returns 0, or -1 on malformed tables. */
#include "loro_block_tables.h"   /* lm_block_tables */
int lm_encode_block(const lm_block_tables* tables, uint8_t** out, size_t* out_len);
int lm_encode_updates(const uint8_t* const* blocks, const size_t* block_lens, size_t n_blocks, uint8_t** out, size_t* out_len);
/* LoroDoc::export(ExportMode::Updates{from}) (crates/loro/src/lib.rs:1306 → encoding.rs:399-405 export_fast_updates →
 * oplog/change_store.rs:718-752 export_blocks_from) for document `doc` of the batch the context holds, after lm_run: the changes
 * its oplog holds beyond `from_vv` (VersionVector::encode() bytes, version.rs:962-968; NULL / 0 = from the empty version) as one
 * FastUpdates blob — blocks ordered by (peer, counter), blocks the version covers dropped, a block it cuts sliced at the cut
 * (the first change then depends on its peer's previous op; a run cut inside an insert / delete is sliced like the reference
 * slices it, list_op.rs:251-277,426-433).  Changes still pending are not part of the oplog and are not exported.  Host work on
 * the blobs read back from the context's arena (lm_export.h); the block boundaries are those of the imported blobs, so what was
 * staged from one writer's export comes back byte for byte from the empty version.  malloc'ed: release with lm_free_bytes. */
int lm_export(lm_ctx* ctx, size_t doc, const uint8_t* from_vv, size_t from_vv_len, uint8_t** out, size_t* out_len);
void lm_free_bytes(uint8_t* p);

/* ---- Richtext values: what TextHandler::get_richtext_value (crates/loro/src/lib.rs:2774 →
 * crates/loro-internal/src/handler.rs:1502 → container/richtext/richtext_state.rs:2546-2584) returns for every Text container of
 * the documents of the last lm_run, at the version that run rendered: a list of spans {"insert": text, "attributes": {key: value}}
 * — a scalar carries, per style key, the value of the mark (StyleOp, container/richtext.rs:31-57) with the greatest (lamport, peer)
 * among those whose StyleStart anchor stands in front of it and whose StyleEnd anchor stands behind it (style_range_map.rs;
 * state/richtext_state.rs:730-812), keys whose value is null are dropped (unmark), neighbouring spans with equal attributes are one.
 * lm_richtext renders them ON THE DEVICE (k_richtext: one wave per document over the trackers the integrate stage left) and copies
 * them back; lm_richtext_result returns document `doc`'s bytes: one JSON object {"<container id>": [span, ...], ...} over the Text
 * containers in which something (a scalar, an anchor) is visible at that version (ContainerID Display: cid:root-<name>:Text / cid:<counter>@<peer>:Text; members in the
 * bytewise order of their JSON-encoded keys), every span the canonical JSON of the LoroValue map it is ({"attributes":{...},"insert":"..."},
 * keys bytewise sorted, no "attributes" member when there are none).  *status: LM_OK, the document's import error, or
 * LM_UNSUPPORTED (more than 64 marks open at one scalar).  The bytes stay valid until the next lm_richtext / lm_stage / lm_destroy.
 * Not available on a batch folded by shared replay.  get_deep_value (lm_fetch) never shows styles: this is a call of its own. */
int lm_richtext(lm_ctx* ctx);
int lm_richtext_result(lm_ctx* ctx, size_t doc, int32_t* status, const uint8_t** json, size_t* json_len);

/* Wave-primitive self test on the device (DPP scan, ballot ranks); returns the number of mismatches. */
int lm_selftest(lm_ctx* ctx);

/* Introspection for bench.py: byte counts of the last run and per-kernel HIP-event timings. */
typedef struct lm_run_stats {
  uint64_t n_docs, n_blobs;
  uint64_t in_bytes;   /* Σ blob lengths */
  uint64_t out_bytes;  /* Σ json_len + Σ vv_len */
  uint64_t device_bytes_allocated;
  uint32_t n_kernels;
} lm_run_stats;
int lm_get_stats(lm_ctx* ctx, lm_run_stats* out);
int lm_set_profiling(lm_ctx* ctx, int mode);                  /* hipEvents around every stage of lm_run (recorded without
                                                                * host syncs): 0 off | 1 and the context's streams run one after
                                                                * the other (a stage's own duration) | 2 streams overlapped as
                                                                * usual (durations as they occur in production) */
int lm_kernel_time(lm_ctx* ctx, uint32_t i, const char** name, double* ms); /* i < n_kernels, after lm_run */
/* A context splits a staged batch into contiguous document ranges, one engine on its own HIP stream each
 * (env LM_STREAMS, default 2; batches under 128 documents per stream stay whole) and lm_run drives them from
 * separate host threads so the stages of different ranges overlap on the device.  Returns the split of the
 * batch staged last.  lm_kernel_time lists the stages of every stream (same names, one entry per stream). */
int lm_n_streams(lm_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
