// K10 (default for batches): Map LWW with the document's table in LDS — one workgroup per document.
//
// k_map_lww (lm_k_emit.h) resolves one op row per lane against a per-document open-addressing table in HBM: every row pays a
// chain of ≈10 dependent global gathers (block → document → key row → key bytes → slot → the claimer's key bytes → best) and the
// rows of one key contend for one HBM word — 36 ms per 2,048 configs[2] documents, 0.005 of the HBM roofline for work that is
// embarrassingly parallel.  An LWW history writes FEW keys MANY times (configs[2]: 1,024 keys, 160,000 rows per document), so
// the table of a document fits LDS: here a workgroup of LWW_WG lanes owns one document, walks its op rows LWW_WG at a time
// (coalesced 32-byte rows) and keeps (container, key) → running maximum of (lamport, peer, row) in LDS — claiming, probing
// and the maximum are LDS atomics; global memory is read for the row, its change and its key, never written until the end,
// when the claimed slots are copied slot for slot into the document's global table (same capacity, same slot numbers), which
// is what the emit stage reads.  A key of up to eight bytes is identified by its LDS entry alone (container, length, the eight
// prefix bytes): no second trip to the claimer's key.  A document with more distinct keys than half its table takes the
// existing second pass (DF_LWW_RETRY → k_map_lww on a table sized for its rows).
// Reference semantics: winner per key = max (lamport, peer) — diff_calc.rs:515-538, delta/map_delta.rs:20-46; a delete is a
// write of None and competes like any other (map_state.rs:438-449); writes beyond a checked-out version do not compete
// (history_cache.rs:630-703).
#pragma once

namespace lm {

static constexpr uint32_t LWW_LDS_CAP = 2048;   // slots of the largest table kept in LDS (24 bytes each: 48 KB — three workgroups per CU by LDS, two by waves)
static constexpr uint32_t LWW_WG = 1024;  // lanes per document: 16 waves — with two workgroups per CU (LDS) every SIMD holds its 8 waves; the rows' gathers are what the kernel waits for
static constexpr unsigned long long LWW_PFX_UNSET = ~0ull;

// LDS key word: cidx (8) | key length, saturated (16) | key row relative to the document's first (24); ~0 = empty
LM_DEV unsigned long long lww_word(uint32_t cidx, uint32_t kl, uint32_t krel) {
  return ((unsigned long long)cidx << 40) | ((unsigned long long)(kl > 0xffffu ? 0xffffu : kl) << 24) | krel;
}

LM_KERNEL LM_WAVES_PER_SIMD(8) void k_map_lww_doc(Dev d, uint32_t* retry_count) {
  uint32_t doc = d.doc_order[(uint32_t)lmw::bid()];
  const uint32_t tid = (uint32_t)lmw::tid();
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status) || m.n_mapop == 0) return;
  if (d.doc_fused && d.doc_fused[doc]) return;                          // k_map_fused's document
  const uint32_t cap = d.ht_cap[doc];
  if (cap == 0 || cap > LWW_LDS_CAP || (m.flags & DF_MOVABLE)) return;   // k_map_lww's document
  LM_DYN_SHARED(unsigned long long, s_mem64);
  unsigned long long* s_key = s_mem64;                 // [cap]
  unsigned long long* s_pfx = s_key + LWW_LDS_CAP;     // [cap] first eight key bytes, big endian, zero padded
  unsigned long long* s_best = s_pfx + LWW_LDS_CAP;    // [cap] (lamport, peer, row) + 1 of the best write so far, 0 = none
  uint32_t* s_end = (uint32_t*)(s_best + LWW_LDS_CAP); // [MAX_PEERS] version being rendered per peer
  uint32_t* s_touch = s_end + MAX_PEERS;               // [MAX_CONTAINERS / 32] containers that received an applied op
  uint32_t* s_misc = s_touch + MAX_CONTAINERS / 32;    // [0] claimed slots, [1] overflow, [2] soft-unsupported, [3] flush cursor, [4] error
  for (uint32_t i = tid; i < cap; i += LWW_WG) { s_key[i] = HT_EMPTY; s_pfx[i] = LWW_PFX_UNSET; s_best[i] = 0; }
  for (uint32_t i = tid; i < m.n_peers && i < MAX_PEERS; i += LWW_WG) s_end[i] = d.peer_end[m.praw0 + i];
  for (uint32_t i = tid; i < MAX_CONTAINERS / 32 + 8; i += LWW_WG) s_touch[i] = 0;   // (+ s_misc)
  lmw::block_sync();
  const uint64_t seed = 0xcbf29ce484222325ull;
  for (uint32_t i0 = 0; i0 < m.n_op; i0 += LWW_WG) {
    uint32_t i = i0 + tid;
    if (i >= m.n_op) continue;
    uint32_t t = m.op0 + i;
    OpRow r = d.op[t];
    uint32_t kind = (r.cidx_kind >> 16) & 0xff;
    if (kind != OK_MAP_SET && kind != OK_MAP_DEL && kind != OK_OTHER) continue;
    if (!d.chg_flag[r.chg]) continue;
    const ChangeRow ch = d.chg[r.chg];
    const uint32_t skip = d.chg_skip[r.chg];
    uint32_t cidx = r.cidx_kind & 0xffff;
    if (kind == OK_OTHER) {
      // an applied op of a container outside the device scope (Tree / Counter): known to the state store (it renders as null),
      // and the document is reported LM_UNSUPPORTED together with its JSON
      if (r.ctr + r.len <= ch.ctr + skip) continue;
      uint32_t ock = d.cont[m.cid0 + cidx].kind_root & 0xff;
      if (ock > CK_TEXT && ock != CK_MOVABLE) { lmw::atomic_or(&s_touch[cidx >> 5], 1u << (cidx & 31)); s_misc[2] = 1; }
      continue;
    }
    if (r.ctr < ch.ctr + skip) continue;                     // already-known prefix of a sliced change
    if (cidx >= MAX_CONTAINERS || ch.peer >= MAX_PEERS) { s_misc[4] = 1; continue; }
    lmw::atomic_or(&s_touch[cidx >> 5], 1u << (cidx & 31));
    if (r.ctr >= s_end[ch.peer]) continue;                   // a write beyond the rendered version does not compete
    if (i >= (1u << 24)) { s_misc[4] = 2; continue; }
    uint32_t krow = r.a0;                                    // (k_remap: the block's first key row + prop)
    uint32_t krel = krow - m.key0;
    if (krel >= (1u << 24)) { s_misc[4] = 2; continue; }
    const uint8_t* ks = d.data + d.key_off[krow];
    uint32_t kl = d.key_len[krow];
    unsigned long long pf = 0;
    for (uint32_t q = 0; q < 8; q++) pf = (pf << 8) | (q < kl ? ks[q] : 0u);
    uint64_t h = (seed ^ cidx ^ ((uint64_t)kl << 32)) * 0x100000001b3ull;
    h = (h ^ pf) * 0x9E3779B97F4A7C15ull;
    if (kl > 8) h = fnv1a(ks + 8, kl - 8, h);
    h ^= h >> 29;
    const unsigned long long mine = lww_word(cidx, kl, krel);
    const unsigned long long v = ((unsigned long long)(d.chg_lamport[r.chg] + (r.ctr - ch.ctr)) << 32) | ((unsigned long long)ch.peer << 24) | i;
    uint32_t slot = (uint32_t)h & (cap - 1);
    bool placed = false;
    for (uint32_t probe = 0; probe < cap; probe++, slot = (slot + 1) & (cap - 1)) {
      unsigned long long cur = s_key[slot];
      if (cur == HT_EMPTY) {
        cur = lmw::atomic_cas64(&s_key[slot], HT_EMPTY, mine);
        if (cur == HT_EMPTY) {   // this lane claimed the slot
          s_pfx[slot] = pf;
          uint32_t at = lmw::atomic_add(&s_misc[0], 1u);
          if (at >= cap / 2) s_misc[1] = 1;
          cur = mine;
        }
      }
      bool same = cur == mine;
      if (!same && (cur >> 24) == (mine >> 24)) {   // same container, same length: the bytes decide
        unsigned long long op = s_pfx[slot];
        if (op != LWW_PFX_UNSET && kl <= 8) same = op == pf;
        else if (op == LWW_PFX_UNSET || op == pf) {
          // (the claimer has not published its prefix yet, or the key is longer than the prefix: the claimer's key itself)
          uint32_t orow = m.key0 + (uint32_t)(cur & 0xffffffu);
          same = d.key_len[orow] == kl && bytes_eq(d.data + d.key_off[orow], ks, kl);
        }
      }
      if (same) {
        if (s_best[slot] < v + 1) lmw::atomic_max64(&s_best[slot], v + 1);   // (+1 so that 0 stays "no write")
        placed = true;
        break;
      }
    }
    if (!placed) s_misc[1] = 1;
  }
  lmw::block_sync();
  // ---- results: containers, flags, and the claimed slots copied slot for slot into the document's global table
  for (uint32_t c = tid; c < m.n_cont && c < MAX_CONTAINERS; c += LWW_WG)
    if ((s_touch[c >> 5] >> (c & 31)) & 1) d.cont[m.cid0 + c].touched = 1;
  if (tid == 0) {
    if (s_misc[2]) lmw::atomic_or(&d.doc[doc].flags, DF_SOFT_UNSUPPORTED);
    if (s_misc[4]) LM_SETERR(d.doc[doc].status, s_misc[4] == 2 ? ST_UNSUPPORTED : ST_INTERNAL);
  }
  if (s_misc[1]) {   // more distinct keys than half the table: the second pass resolves the document in a table sized for its rows
    if (tid == 0) { lmw::atomic_or(&d.doc[doc].flags, DF_LWW_RETRY); lmw::atomic_add(retry_count + 2, 1u); d.ht_cnt[doc] = 0; }
    return;
  }
  unsigned long long* keys = d.ht_key + d.ht0[doc];
  unsigned long long* best = d.ht_best + d.ht0[doc];
  uint32_t* list = d.ht_list + 2 * d.ht0[doc];
  for (uint32_t s = tid; s < cap; s += LWW_WG) {
    unsigned long long k = s_key[s];
    if (k == HT_EMPTY) continue;
    keys[s] = ((unsigned long long)(uint32_t)(k >> 40) << 32) | (m.key0 + (uint32_t)(k & 0xffffffu));
    best[s] = s_best[s];
    list[lmw::atomic_add(&s_misc[3], 1u)] = s;
  }
  lmw::block_sync();
  if (tid == 0) d.ht_cnt[doc] = s_misc[3];
}

}  // namespace lm
