// Device-side tables of the batched merge engine (HBM layout).  See DESIGN.md §"Data layout in HBM".
#pragma once
#include <cstdint>

namespace lm {

static constexpr uint32_t NONE = 0xFFFFFFFFu;

// per-doc status codes == LM_* in include/loro_merge.h
enum : int32_t {
  ST_OK = 0,
  ST_DECODE_ERROR = 1,
  ST_CHECKSUM_MISMATCH = 2,
  ST_DATA_CORRUPTION = 3,
  ST_UNSUPPORTED = 4,
  ST_INTERNAL = 5,
  ST_FRONTIERS_NOT_FOUND = 6,
  ST_RETRY = 100,   // internal: the optimistic LDS directory overflowed; the document is re-run with the worst-case size
  ST_MF_BAIL = 102, // internal (a BLOCK's status): k_block_head met something in a fused Map document's block — the verdict is the row decoders': the document is flagged DF_REDO (k_doc_tables)
};

enum : uint32_t { SEC_HEADER = 0, SEC_META, SEC_CIDS, SEC_KEYS, SEC_POS, SEC_OPS, SEC_DEL, SEC_VALUES, SEC_N };

// container kinds on the wire (loro-common/src/lib.rs:748-793)
enum : uint32_t { CK_MAP = 0, CK_LIST = 1, CK_TEXT = 2, CK_TREE = 3, CK_MOVABLE = 4, CK_COUNTER = 5 };

// decoded op kinds (outdated_encode_reordered.rs:215-476 mapping)
enum : uint32_t {
  OK_OTHER = 0, OK_TEXT_INS = 1, OK_DEL = 2, OK_STYLE_START = 3, OK_STYLE_END = 4, OK_LIST_INS = 5, OK_MAP_SET = 6, OK_MAP_DEL = 7,
  OK_LIST_MOVE = 8, OK_LIST_SET = 9   // MovableList (docs/encoding.md §10.5): a0 = element peer idx, a1 = element lamport, move: prop = to, a2 = from
};

// limits of the packed element id (peer_idx:8 | counter:24)
static constexpr uint32_t MAX_PEERS = 256;
static constexpr uint32_t MAX_COUNTER = 1u << 24;
static constexpr uint32_t MAX_CONTAINERS = 256;  // containers (roots + children) per document
static constexpr uint32_t MAX_ROOTS = 64;        // root containers per document

struct BlockDesc {           // one per change block; owned by one lane during decode
  uint64_t base;             // absolute byte offset of the block in `data`
  uint32_t blob, doc;
  uint32_t sec_rel[SEC_N], sec_len[SEC_N];
  uint32_t counter_start, counter_len, lamport_start, lamport_len, n_changes;
  int32_t status;
  uint32_t flags;            // DF_* bits raised while decoding this block
  uint32_t pad;
};
// per-block row counts; scanned component-wise to get table offsets
struct BlockCounts { uint32_t n_chg, n_dep, n_op, n_key, n_cid, n_peer; };
static constexpr int BC_N = 6;

struct OpRow {               // 32 B, read with scalar loads by the integrate kernel
  uint32_t cidx_kind;        // container idx (low 16) | op kind (bits 16-23) | flags (bits 24-31)
  int32_t prop;              // position / key idx
  uint32_t len;              // atom length
  uint32_t ctr;              // absolute counter of the first atom
  uint32_t a0, a1;           // delete: target peer idx, target counter | style start: mark len | list insert: #items | move/set: element peer idx, lamport
  int32_t a2;                // delete: signed len | move: source position
  uint32_t chg;              // global change row
};

// OpRow.cidx_kind flag bits (k_fuse_rows): the row continues the run of the row in front of it / heads a run whose extent is in Dev::fuse
// OPF_NESTED (the decoders → k_remap): the row's value is a list / map (or a mark / list-set payload that may hold one): k_remap walks
// it once more to check nested map key indices against the block's key table, then drops the bit
enum : uint32_t { OPF_CONT = 1u << 24, OPF_HEAD = 1u << 25, OPF_NESTED = 1u << 26 };

struct ChangeRow {           // 32 B
  uint32_t peer;             // block-local 0 → doc peer idx after remap
  uint32_t ctr, len;
  uint32_t dep0, n_dep;      // global dep rows
  uint32_t op0, n_op;        // global op rows
  uint32_t blk;
};

struct ContRow {             // doc-level container
  uint64_t name_off;         // root: absolute offset of the name bytes
  uint32_t name_len;
  uint32_t kind_root;        // kind | is_root<<8
  uint32_t peer, counter;    // normal containers
  uint32_t touched;          // received at least one applied op
  uint32_t pad;
};

struct DocMeta {             // per document, filled progressively
  int32_t status;
  uint32_t blk0, n_blk;      // block range
  uint32_t chg0, n_chg;      // change rows
  uint32_t op0, n_op;
  uint32_t dep0, n_dep;
  uint32_t key0, n_key;
  uint32_t cid0, n_cid;      // raw (per-block) cid rows
  uint32_t praw0, n_praw;    // raw (per-block) peer rows
  uint32_t n_peers;          // unique peers (sorted ascending by PeerID)
  uint32_t n_cont;           // unique containers
  uint32_t atoms;            // Σ counter_len over blocks (upper bound on elements)
  uint32_t n_valid_chg;      // applied changes (sorted order array length)
  uint32_t n_nodes;
  uint32_t pending_lo, pending_hi;  // pending atoms (u64 split)
  uint32_t elem0_lo, elem0_hi;      // first element slot of this doc (u64 split)
  uint32_t leaf0, leaf_cap;  // leaf pool (also the capacity of the flushed leaf directory)
  uint32_t n_elems, n_mapop; // Σ len of insert-type op rows (element upper bound); number of Map op rows
  uint32_t node0;            // node rows
  uint32_t vvh0_lo, vvh0_hi; // vv_head rows (n_nodes × n_peers)
  uint32_t out_len;          // JSON bytes
  uint32_t vv_len;           // VV bytes
  uint32_t pad0;             // directory entries used by the integrate stage (sizing diagnostics)
  uint32_t flags;            // DF_* bits
  int32_t front_err;         // resident documents: why the requested checkout was refused (DF_FRONT_ERR); the import itself went through
};

// DocMeta.flags / BlockDesc.flags
enum : uint32_t {
  DF_SOFT_UNSUPPORTED = 1u,  // the document holds containers outside the device scope (Tree / Counter): they render as
                             // null, everything else is rendered, and the document is reported LM_UNSUPPORTED *with* its JSON
  DF_REEMIT = 2u,            // the rendered JSON did not fit the optimistic output slab: re-rendered at its exact size
  DF_MOVABLE = 4u,           // the document holds a MovableList container: k_mlist_post runs for it after the integrate stage
  DF_FRONT_ERR = 16u,        // resident documents (lm_import): the checkout frontiers were refused (DocMeta.front_err) — LoroDoc::checkout fails
                             // after LoroDoc::import succeeded: the blobs stay imported, the document is rendered at the latest version for
                             // the state store's sake and reported with that error
  DF_LAYOUT_SAME = 32u,      // resident documents (k_res_layout): the element slots of everything the stored tracker holds are where they were —
                             // loc[] is still right (only the new slots are cleared) and k_elem_fill skips the blocks of earlier runs
  DF_FILL_KEPT = 64u,        // … and that run left nothing pending: every payload slot of its blocks was filled then (k_elem_fill fills applied rows only)
  DF_LWW_RETRY = 128u,       // k_map_lww: the document's optimistic LWW table (sized for a few thousand keys) filled up: its Map rows are resolved
                             // again in a table sized for as many keys as it has Map rows (lm_pipeline.h)
  DF_FUSED = 256u,           // a plain document whose changes hold few rows each (one change per keystroke): k_fuse_rows chained its rows into runs
                             // across change boundaries, k_integrate_span_plain_fuse replays the runs (lm_k_fuse.h)
  DF_CUT = 512u,             // k_dag_a: the document is large enough for the node cut + descending-peer replay order to pay (k_dag_a / k_dag_b)
  DF_REDO = 1024u,           // the kernel that met this document has no path for it and another configuration of the pipeline has: an element-granular
                             // / resident replay that met a delete row which does not match its position (the span-granular batch kernels finish such
                             // rows by position, k_integrate_span_pos), a Map document the fused decode→LWW kernel (lm_k_map_fused.h) bailed out of.
                             // The context replays such documents once more in a side engine with that configuration (lm_capi_impl.h redo)
  DF_PLAIN = 8u,             // no sliced change, no style anchor, no MovableList (k_dag_a); the host clears it for checked-out documents
                             // and under LM_PLAIN=0: such a document is replayed by k_integrate_span_plain_sweep (lm_pipeline.h)
};

}  // namespace lm
