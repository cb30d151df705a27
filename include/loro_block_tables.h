/* loro_block_tables.h — the tables of ONE change block (part of the C ABI declared in loro_merge.h; a separate header so
 * that the host-side encoder, loro_amd/csrc/lm_encode.h, and test infrastructure can share the layout).
 * One block = the changes of one peer, counter-contiguous: exactly what the decode stage extracts from a block
 * (crates/loro-internal/src/oplog/change_store/block_encode.rs:535-706) and what encode_block consumes (:137-278). */
#ifndef LORO_BLOCK_TABLES_H
#define LORO_BLOCK_TABLES_H
#include <stddef.h>
#include <stdint.h>
typedef struct lm_block_tables {
  uint32_t counter_start, counter_len, lamport_start, lamport_len, n_changes;
  const uint64_t* peers; size_t n_peers;           /* peer table; peers[0] is the block's peer */
  const uint32_t* change_len;                       /* [n_changes] atoms per change */
  const uint8_t* dep_on_self;                       /* [n_changes] depends on its peer's previous op */
  const uint32_t* dep_count;                        /* [n_changes] number of other dependencies */
  const uint32_t* dep_peer_idx; const int32_t* dep_counter; size_t n_deps;   /* those dependencies, change by change */
  const uint32_t* lamport;                          /* [n_changes] */
  const int64_t* timestamp;                         /* [n_changes] */
  const uint32_t* msg_len; const uint8_t* msgs; size_t msgs_len;              /* [n_changes] commit message lengths + bytes */
  const uint8_t* cid_is_root; const uint8_t* cid_kind; const uint32_t* cid_peer_idx; const int32_t* cid_key_or_counter; size_t n_cids;
  const uint8_t* const* keys; const size_t* key_lens; size_t n_keys;
  const uint8_t* positions; size_t positions_len;   /* position arena section, as bytes */
  const uint32_t* op_container; const int32_t* op_prop; const uint8_t* op_value_type; const uint32_t* op_len; size_t n_ops;
  const uint32_t* del_peer_idx; const int32_t* del_counter; const int64_t* del_len; size_t n_dels;   /* delete-start ids */
  const uint8_t* values; size_t values_len;         /* value section, as bytes */
} lm_block_tables;
#endif
