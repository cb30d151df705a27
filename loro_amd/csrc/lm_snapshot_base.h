// SURVEY §8f N3, second half (round 6): UPDATES on top of a snapshot's state — `import_batch([snapshot, updates…])` where the updates
// continue the snapshot's history (fast_snapshot.rs:168-258: the empty document takes its state store from the snapshot, then
// loro.rs:568-649 imports the updates against it; the snapshot's own history is never replayed).
//
// lm_snapshot.h stages a snapshot as ONE synthetic change that writes its visible state.  The updates' changes are real: they keep
// their peers, counters and ops, and the device lets them start where the snapshot's version vector ends (Dev::vvo is the base
// version: lm_k_dag.h coverage walk).  What has to be said differently is how they hang on the base:
//   * a dependency on an op BELOW the snapshot's version cannot be looked up (that history is not staged): when a change's
//     below-base dependencies are exactly the snapshot's frontiers, it saw the whole base — they are replaced by one dependency on
//     the synthetic change's last op (its lamport then follows all of the base; lamports of changes that hang on the same frontiers
//     keep their differences, which is all that concurrent resolution compares);
//   * a child container of the base has a synthetic id (the op that writes it): the container ids of the updates' blocks that
//     name base children are rewritten to it.
// Everything else makes this reader DECLINE, and the document is replayed from the snapshot's ChangeStore as before: a change that
// hangs on an older version than the frontiers (concurrent with part of the base: the tracker needs that history), a change below
// or across the base version, a dependency that is nowhere (pending), MovableList move / set rows when the base holds list items
// (they name element ids the state does not carry over), a blob that does not decode or fails its checksum (the row decoders own
// the verdict), the synthetic peer in use.  Deletes of base content name real ids the synthetic elements do not have: the kernels
// finish them by position (crdt_rope.rs:256-335 — the reference's own path for content it knows only as a placeholder).
#pragma once
#include <set>
#include "lm_snapshot.h"
#include "lm_export.h"

namespace lmsnap {

// `pays` (optional): false when the replay on the state is expected to cost MORE than the replay of the snapshot's history — measured on
// configs[1]-shaped documents (tests/tools/gpu_snapbase.py): updates that are ONE chain behind the frontiers are replayed by
// the linear prefix, by position, at twice the speed of the history path; updates with concurrent branches go through the tracker, where
// every delete of base content takes the by-position path (ts_del_positional) and rows are not fused — about 1.1 ms per 1,000 update ops
// and 2,000 documents against 0.2 ms per 1,000 ops of history: the state pays while the updates hold less than a fifth of the history's ops
inline bool rebase_updates_on_state(const StateBase& sb, const std::vector<std::pair<const uint8_t*, size_t>>& U, std::vector<std::vector<uint8_t>>& out, bool* pays = nullptr) {
  if (sb.synth_len == 0) return false;
  struct Ch { uint64_t peer; uint32_t ctr, len, lamport; size_t blob, blk, idx; std::vector<std::pair<uint64_t, uint32_t>> deps; bool covers = false, done = false; };
  std::vector<std::vector<lmexp::Block>> blocks(U.size());
  std::vector<Ch> chs;
  try {
    for (size_t u = 0; u < U.size(); u++) {
      const uint8_t* p = U[u].first; const size_t n = U[u].second;
      if (n < 22 || memcmp(p, "loro", 4) != 0 || p[20] != 0 || p[21] != 4) return false;
      if (lmenc::xxh32(p + 20, n - 20, 0x4F524F4Cu) != rd32(p + 16)) return false;
      lmexp::blocks_of_blob(p, n, blocks[u]);
      for (size_t k = 0; k < blocks[u].size(); k++) {
        lmexp::Block& b = blocks[u][k];
        const uint64_t peer = b.peer();
        if (peer == sb.synth_peer) return false;
        for (uint64_t q : b.peers) if (q == sb.synth_peer) return false;
        auto bv = sb.vv.find(peer);
        const uint32_t base_end = bv == sb.vv.end() ? 0u : bv->second;
        if (b.counter_start < base_end) return false;                     // below or across the base version
        if (sb.has_movable) for (uint8_t vt : b.op_value_type) if ((vt & 0x7f) == 14 || (vt & 0x7f) == 15) return false;
        uint32_t ctr = b.counter_start;
        size_t dep_at = 0;
        for (size_t i = 0; i < b.change_len.size(); i++) {
          Ch c;
          c.peer = peer; c.ctr = ctr; c.len = b.change_len[i]; c.lamport = b.lamport[i]; c.blob = u; c.blk = k; c.idx = i;
          if (b.dep_on_self[i]) { if (ctr == 0) return false; c.deps.emplace_back(peer, ctr - 1); }
          for (uint32_t dd = 0; dd < b.dep_count[i]; dd++, dep_at++) {
            if (dep_at >= b.dep_peer_idx.size() || b.dep_peer_idx[dep_at] >= b.peers.size() || b.dep_counter[dep_at] < 0) return false;
            c.deps.emplace_back(b.peers[b.dep_peer_idx[dep_at]], (uint32_t)b.dep_counter[dep_at]);
          }
          chs.push_back(std::move(c));
          ctr += b.change_len[i];
        }
      }
    }
  } catch (const std::exception&) { return false; }
  // ---- which changes saw the whole base
  std::map<uint64_t, std::vector<size_t>> by_peer;       // peer -> its changes by counter
  for (size_t i = 0; i < chs.size(); i++) by_peer[chs[i].peer].push_back(i);
  for (auto& kv : by_peer) {
    std::sort(kv.second.begin(), kv.second.end(), [&](size_t a, size_t b) { return chs[a].ctr < chs[b].ctr; });
    for (size_t k = 1; k < kv.second.size(); k++) if (chs[kv.second[k]].ctr < chs[kv.second[k - 1]].ctr + chs[kv.second[k - 1]].len) return false;   // (duplicates / overlaps: the coverage walk's business)
  }
  auto below_base = [&](uint64_t peer, uint32_t c) { auto it = sb.vv.find(peer); return it != sb.vv.end() && c < it->second; };
  auto find = [&](uint64_t peer, uint32_t c) -> long {
    auto it = by_peer.find(peer);
    if (it == by_peer.end()) return -1;
    const std::vector<size_t>& v = it->second;
    size_t lo = 0, hi = v.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (chs[v[mid]].ctr + chs[v[mid]].len <= c) lo = mid + 1; else hi = mid; }
    if (lo < v.size() && chs[v[lo]].ctr <= c) return (long)v[lo];
    return -1;
  };
  std::set<std::pair<uint64_t, uint32_t>> fr(sb.frontiers.begin(), sb.frontiers.end());
  std::vector<size_t> order(chs.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return chs[a].lamport < chs[b].lamport; });   // (a dependency has the smaller lamport)
  bool one_chain = true;
  long prev = -1;
  uint64_t update_ops = 0;
  for (size_t oi : order) {
    Ch& c = chs[oi];
    update_ops += c.len;
    if (prev < 0) { if (c.deps.size() != fr.size()) one_chain = false; }   // (all of them below the base, or the walk below declines the document)
    else if (c.deps.size() != 1 || c.deps[0].first != chs[(size_t)prev].peer || c.deps[0].second != chs[(size_t)prev].ctr + chs[(size_t)prev].len - 1) one_chain = false;
    prev = (long)oi;
    std::set<std::pair<uint64_t, uint32_t>> below;
    bool via = false;
    for (auto& dp : c.deps) {
      if (below_base(dp.first, dp.second)) { below.insert(dp); continue; }
      long t = find(dp.first, dp.second);
      if (t < 0 || !chs[(size_t)t].done) return false;         // nowhere (pending), or not in lamport order
      via |= chs[(size_t)t].covers;
    }
    if (!below.empty()) { if (below != fr) return false; c.covers = true; }
    else c.covers = via;
    if (!c.covers) return false;                               // concurrent with (part of) the base
    c.done = true;
  }
  if (pays) {
    uint64_t base_ops = 0;
    for (auto& kv : sb.vv) base_ops += kv.second;
    *pays = one_chain || update_ops <= 4096 || update_ops * 5 <= base_ops;
  }
  // ---- the blocks again, hung on the synthetic change
  out.clear();
  try {
    for (size_t u = 0; u < U.size(); u++) {
      std::vector<lmenc::Bytes> enc;
      for (lmexp::Block& b : blocks[u]) {
        const uint64_t peer = b.peer();
        uint32_t s_idx = (uint32_t)b.peers.size();
        auto synth = [&]() { if (s_idx == b.peers.size()) b.peers.push_back(sb.synth_peer); return s_idx; };
        std::vector<uint8_t> n_self; std::vector<uint32_t> n_cnt, n_pi; std::vector<int32_t> n_ctr;
        uint32_t ctr = b.counter_start;
        size_t dep_at = 0;
        for (size_t i = 0; i < b.change_len.size(); i++) {
          bool hang = false;
          uint8_t self = b.dep_on_self[i];
          if (self && below_base(peer, ctr - 1)) { self = 0; hang = true; }
          uint32_t cnt = 0;
          for (uint32_t dd = 0; dd < b.dep_count[i]; dd++, dep_at++) {
            const uint64_t q = b.peers[b.dep_peer_idx[dep_at]];
            const uint32_t c = (uint32_t)b.dep_counter[dep_at];
            if (below_base(q, c)) { hang = true; continue; }
            n_pi.push_back(b.dep_peer_idx[dep_at]); n_ctr.push_back((int32_t)c); cnt++;
          }
          if (hang) { n_pi.push_back(synth()); n_ctr.push_back((int32_t)sb.synth_len - 1); cnt++; }
          n_self.push_back(self); n_cnt.push_back(cnt);
          ctr += b.change_len[i];
        }
        b.dep_on_self.swap(n_self); b.dep_count.swap(n_cnt); b.dep_peer_idx.swap(n_pi); b.dep_counter.swap(n_ctr);
        for (size_t k = 0; k < b.cid_kind.size(); k++) {
          if (b.cid_is_root[k]) continue;
          if (b.cid_peer_idx[k] >= b.peers.size()) return false;
          lmenc::Bytes id(13);
          id[0] = b.cid_kind[k];
          const uint64_t q = b.peers[b.cid_peer_idx[k]];
          for (int x = 0; x < 8; x++) id[1 + x] = (uint8_t)(q >> (8 * x));
          for (int x = 0; x < 4; x++) id[9 + x] = (uint8_t)((uint32_t)b.cid_key_or_counter[k] >> (8 * x));
          auto it = sb.child_ctr.find(id);
          if (it != sb.child_ctr.end()) { b.cid_peer_idx[k] = synth(); b.cid_key_or_counter[k] = (int32_t)it->second; }
        }
        lm_block_tables t = b.view();
        enc.push_back(lmenc::encode_block(t));
      }
      std::vector<const uint8_t*> ptrs; std::vector<size_t> lens;
      for (auto& e : enc) { ptrs.push_back(e.data()); lens.push_back(e.size()); }
      out.push_back(lmenc::encode_updates(ptrs.data(), lens.data(), enc.size()));
    }
  } catch (const std::exception&) { return false; }
  return true;
}

}  // namespace lmsnap
