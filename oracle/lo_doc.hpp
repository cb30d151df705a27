// ORACLE — TEST INFRASTRUCTURE ONLY (see lo_codec.hpp header).
// CPU restatement of LoroDoc::import → OpLog/DAG → diff_calc replay → state → canonical JSON + VV.
//
// Restates (reference file:line, relative to /root/reference/crates/loro-internal/src):
//   import filtering by receiver VV / slice         oplog/change_store.rs:329-352
//   sort by lamport, apply, pending                  encoding/fast_snapshot.rs:398, encoding/outdated_encode_reordered.rs:40-83,
//                                                    oplog/pending_changes.rs:12-36, encoding.rs:266-293
//   lamport from deps                                oplog/loro_dag.rs:1179-1187
//   trim known prefix                                oplog.rs:367-382
//   change/op slicing                                change.rs (Sliceable), container/list/list_op.rs:251-277,426-433
//   version vectors at a change's deps               oplog/loro_dag.rs:1083-1154,1192-1207
//   causal iteration (node-at-a-time Kahn order)     dag/iter.rs:199-386, oplog.rs:591-669
//   replay of a sequence container from empty        diff_calc.rs:445-485 (replay_container_ops_from_empty),
//                                                    :1301-1338 (build_full_crdt_tracker), :993-1137 (apply_crdt_op_to_tracker)
//   Map LWW                                          diff_calc.rs:515-538, delta/map_delta.rs:20-46, state/map_state.rs:240-292,438-449
//   deep value / JSON                                state.rs:1294-1329, loro-common/src/value.rs:719-738
//   VersionVector encode                             version.rs:962-964
//   MovableList                                      diff_calc.rs:1669-1993 (move = delete + insert on the list tracker),
//                                                    history_cache.rs:754-1003 (last_pos / last_value: LWW by (lamport, peer)),
//                                                    state/movable_list_state.rs:953-964 (value = items an element points at)
// The final state of a document is the same for every import order (CRDT convergence), so the oracle
// materialises it the way the reference itself rebuilds a container whenever an import is concurrent
// (diff_calc.rs:1615-1643): replay the container's ops from the empty version in a causal order and
// read the elements active at the final version.
#pragma once
#include <map>
#include <set>
#include <tuple>
#include <string>
#include <vector>
#include <cmath>
#include <charconv>
#include "lo_codec.hpp"
#include "lo_tracker.hpp"
#include "lo_dag.hpp"
#include "lo_snapshot.hpp"

namespace lo {

// ---------------------------------------------------------------- canonical JSON
inline void json_escape(const std::string& s, std::string& out) {
  static const char* hex = "0123456789abcdef";
  out.push_back('"');
  for (unsigned char c : s) {
    switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\b': out += "\\b"; break;
      case '\f': out += "\\f"; break;
      case '\n': out += "\\n"; break;
      case '\r': out += "\\r"; break;
      case '\t': out += "\\t"; break;
      default:
        if (c < 0x20) { out += "\\u00"; out.push_back(hex[c >> 4]); out.push_back(hex[c & 15]); }
        else out.push_back((char)c);
    }
  }
  out.push_back('"');
}
// serde_json / ryu "pretty" float formatting
inline void json_f64(double v, std::string& out) {
  if (!std::isfinite(v)) { out += "null"; return; }
  if (v == 0) { out += std::signbit(v) ? "-0.0" : "0.0"; return; }
  char buf[64];
  auto res = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
  std::string sci(buf, res.ptr);  // d[.ddd]e[+-]XX shortest round-trip
  bool neg = sci[0] == '-';
  size_t p = neg ? 1 : 0;
  size_t e = sci.find('e');
  std::string digits;
  for (size_t i = p; i < e; i++) if (sci[i] != '.') digits.push_back(sci[i]);
  int exp10 = atoi(sci.c_str() + e + 1);
  while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
  int len = (int)digits.size();
  int k = exp10 - (len - 1);  // value = digits * 10^k
  int kk = len + k;
  if (neg) out.push_back('-');
  if (0 <= k && kk <= 16) {
    out += digits; out.append((size_t)k, '0'); out += ".0";
  } else if (0 < kk && kk <= 16) {
    out.append(digits, 0, (size_t)kk); out.push_back('.'); out.append(digits, (size_t)kk, std::string::npos);
  } else if (-5 < kk && kk <= 0) {
    out += "0."; out.append((size_t)(-kk), '0'); out += digits;
  } else {
    out.push_back(digits[0]);
    if (len > 1) { out.push_back('.'); out.append(digits, 1, std::string::npos); }
    out.push_back('e');
    out += std::to_string(kk - 1);
  }
}

struct Doc;
inline void json_value(const Doc& d, const Value& v, std::string& out, int depth);

// ---------------------------------------------------------------- document
struct StyleRec { PeerID peer; Counter cnt; uint32_t end; };
// a style anchor as the richtext state keeps it (RichtextStateChunk::Style { style: Arc<StyleOp>, anchor_type },
// container/richtext.rs:31-57, richtext_state.rs:60-75): the StyleOp is identified by its StyleStart op
struct AnchorRec { bool is_end; PeerID peer; Counter start_cnt; Lamport lamport; std::string key; Value value; };

struct IdLpKey {   // IdLp (loro-common/src/lib.rs:524-528): ordered by lamport, then peer
  Lamport lamport; PeerID peer;
  bool operator<(const IdLpKey& o) const { return lamport != o.lamport ? lamport < o.lamport : peer < o.peer; }
  bool operator==(const IdLpKey& o) const { return lamport == o.lamport && peer == o.peer; }
};
struct SeqState {
  Tracker tr;
  std::vector<uint32_t> cps;      // text: unicode scalars; 0xFFFFFFFF = style anchor
  std::vector<Value> values;      // list
  std::vector<StyleRec> styles;
  std::map<uint32_t, AnchorRec> anchors;   // content index (cps) of an anchor → its StyleOp
  // MovableList: per list item (index = content index, parallel to `values`): its own IdLp and the element it positions
  struct ItemRec { IdLpKey item, elem; };
  std::vector<ItemRec> items;
  std::map<IdLpKey, uint32_t> elem_init;   // element → content index of its insert (initial value)
  std::map<IdLpKey, IdLpKey> pos_win;      // element → greatest move op inside the rendered version (last_pos)
  struct ValWin { IdLpKey id; Value v; };
  std::map<IdLpKey, ValWin> val_win;       // element → greatest set op inside the rendered version (last_value)
};
struct MapEntry { Lamport lamp; PeerID peer; bool has; Value v; };

struct Doc {
  std::map<PeerID, std::vector<Change>> changes;  // per peer: counter-contiguous from the first imported change
  VV vv;
  std::vector<Change> pending;
  std::vector<ContainerID> containers;
  std::map<ContainerID, uint32_t> container_idx;
  mutable bool unsupported = false;   // met a container kind outside Map/List/Text (registered, or as a child value)
  bool materialized = false;
  std::map<uint32_t, std::unique_ptr<SeqState>> seqs;
  std::map<uint32_t, std::map<std::string, MapEntry>> maps;
  std::set<uint32_t> touched;  // containers that received at least one applied op
  // Root containers the state store holds (diff_calc.rs:299 `if !diff.is_empty() || bring_back`, state.rs:621-849): a
  // container state is created by the first import / checkout step whose diff for it is not empty.  The batch is
  // imported like LoroDoc::import_batch (loro.rs:1432-1523): every blob goes into the OpLog while the document is
  // detached, then ONE diff from the empty version to the latest one is applied (one blob: LoroDoc::import, same
  // versions); a checkout is a second diff latest → version.  From the empty version a sequence diff (tracker.diff,
  // crdt_rope.rs:396-451; Linear mode: the composed DeltaRope, whose fully deleted inserts are dropped —
  // delta_item.rs:353-363) is empty exactly when nothing is visible at the target; a Map diff lists every key with an
  // op (deleted ones too, diff_calc.rs:553-605), so a Map exists once it was written.
  std::set<uint32_t> seq_exists;
  // Root containers an EMPTY document took over from the state section of a snapshot (fast_snapshot.rs:168-258): part of the
  // value even when nothing is visible in them
  std::set<uint32_t> state_roots;
  // Frontiers of the OpLog (version/frontiers.rs:233 update_frontiers_on_new_change) and the import steps taken
  std::vector<ID> oplog_frontiers;
  struct ImportStep { Frontiers from_f, to_f; VV from_vv, to_vv; };
  std::vector<ImportStep> steps;
  // optional checkout (LoroDoc::checkout, loro.rs:1625-1760): the state is rendered at `frontiers` instead of the
  // latest version; the OpLog (and so the set of root containers the state store knows) stays the imported one
  bool has_target = false;
  std::vector<ID> frontiers;
  VV target;

  // Frontiers::decode (version/frontiers.rs:226-231): postcard Vec<ID>, ID = { varint u64 peer, zigzag varint i32 counter }
  void set_checkout(const uint8_t* p, size_t n) {
    Reader r(p, n);
    uint64_t cnt = r.uleb();
    if (cnt > n) fail(ST_DECODE_ERROR, "frontiers length");
    frontiers.clear();
    for (uint64_t i = 0; i < cnt; i++) {
      ID id;
      id.peer = r.uleb();
      id.counter = (Counter)r.zigzag();
      frontiers.push_back(id);
    }
    if (!r.eof()) fail(ST_DECODE_ERROR, "trailing bytes after frontiers");
    has_target = true;
    materialized = false;
  }

  uint32_t reg(const ContainerID& c) {
    auto it = container_idx.find(c);
    if (it != container_idx.end()) return it->second;
    uint32_t i = (uint32_t)containers.size();
    containers.push_back(c);
    container_idx[c] = i;
    if (c.kind > CK_TEXT && c.kind != CK_MOVABLE) unsupported = true;
    return i;
  }

  // ---- op / change slicing (Sliceable impls)
  static Op slice_op(const Op& o, int32_t from, int32_t to) {
    Op r = o;
    r.counter = o.counter + from;
    r.len = to - from;
    switch (o.kind) {
      case OP_TEXT_INSERT:
        r.cps.assign(o.cps.begin() + from, o.cps.begin() + to);
        r.prop = o.prop + from;
        break;
      case OP_LIST_INSERT:
        r.values.assign(o.values.begin() + from, o.values.begin() + to);
        r.prop = o.prop + from;
        break;
      case OP_SEQ_DELETE: {
        // DeleteSpanWithId::slice (list_op.rs:251-277) + DeleteSpan::slice (:426-433)
        int64_t L = o.del_signed_len < 0 ? -o.del_signed_len : o.del_signed_len;
        if (o.del_signed_len > 0) {
          r.del_id_start.counter = o.del_id_start.counter + from;
          r.del_signed_len = to - from;
          // pos unchanged
        } else {
          r.del_id_start.counter = o.del_id_start.counter + (Counter)(L - to);
          r.prop = o.prop - from;
          r.del_signed_len = from - to;
        }
        break;
      }
      default: break;  // len-1 ops
    }
    return r;
  }
  static Change slice_change(const Change& c, int32_t from) {  // keep [from, len)
    Change r;
    r.id = ID{c.id.peer, c.id.counter + from};
    r.lamport = c.lamport + (Lamport)from;
    r.len = c.len - from;
    r.cids = c.cids;
    r.deps = from > 0 ? std::vector<ID>{ID{c.id.peer, c.id.counter + from - 1}} : c.deps;
    Counter cut = c.id.counter + from;
    for (const Op& o : c.ops) {
      if (o.counter + o.len <= cut) continue;
      if (o.counter >= cut) r.ops.push_back(o);
      else r.ops.push_back(slice_op(o, cut - o.counter, o.len));
    }
    return r;
  }

  const Change* find_change(ID id) const {
    auto it = changes.find(id.peer);
    if (it == changes.end()) return nullptr;
    const auto& v = it->second;
    size_t lo_ = 0, hi = v.size();
    while (lo_ < hi) {
      size_t mid = (lo_ + hi) / 2;
      if (v[mid].ctr_end() <= id.counter) lo_ = mid + 1; else hi = mid;
    }
    if (lo_ < v.size() && v[lo_].id.counter <= id.counter) return &v[lo_];
    return nullptr;
  }
  bool has_id(ID id) const {
    auto it = vv.find(id.peer);
    return it != vv.end() && id.counter < it->second && find_change(id) != nullptr;
  }
  // returns false if some dep is missing (loro_dag.rs:1179-1187)
  bool lamport_from_deps(const std::vector<ID>& deps, Lamport& out) const {
    Lamport l = 0;
    for (const ID& d : deps) {
      const Change* c = find_change(d);
      if (!c) return false;
      Lamport x = c->lamport + (Lamport)(d.counter - c->id.counter);
      l = std::max(l, x + 1);
    }
    out = l;
    return true;
  }
  Counter vv_get(PeerID p) const {
    auto it = vv.find(p);
    return it == vv.end() ? 0 : it->second;
  }
  // try to apply one decoded change (outdated_encode_reordered.rs:48-75); returns 0 applied/skipped, 1 pending
  int try_apply(Change& ch) {
    if (ch.ctr_end() <= vv_get(ch.id.peer)) return 0;  // skip included changes
    Lamport l;
    if (!lamport_from_deps(ch.deps, l)) return 1;
    Counter end = vv_get(ch.id.peer);
    if (ch.id.counter > end) return 1;  // counter gap: cannot be produced by a valid exporter; parked
    ch.lamport = l;
    Change c2 = ch.id.counter < end ? slice_change(ch, end - ch.id.counter) : std::move(ch);
    for (Op& o : c2.ops) o.container = reg(c2.cids[o.container]);
    c2.cids.clear();
    vv[c2.id.peer] = c2.ctr_end();
    {
      Frontiers nf;
      for (const ID& f : oplog_frontiers) {
        bool dep = f.peer == c2.id.peer;
        for (const ID& d : c2.deps) if (id_eq(d, f)) dep = true;
        if (!dep) nf.push_back(f);
      }
      nf.push_back(ID{c2.id.peer, c2.ctr_end() - 1});
      fr_norm(nf);
      oplog_frontiers.swap(nf);
    }
    changes[c2.id.peer].push_back(std::move(c2));
    return 0;
  }
  void import(const uint8_t* blob, size_t len) {
    std::vector<Change> decoded;
    if (blob_mode(blob, len) == 3) {
      // FastSnapshot: its ChangeStore arrives as changes (decode_oplog, fast_snapshot.rs:326-344); a document that holds
      // nothing yet also takes the state section's containers (loro.rs:582-638, fast_snapshot.rs:168-258)
      SnapshotParts sp;
      decode_snapshot_blob(blob, len, sp);
      if (changes.empty() && pending.empty())
        for (auto& r : sp.roots) { ContainerID c; c.root = true; c.kind = r.first; c.name = r.second; state_roots.insert(reg(c)); }
      decoded.swap(sp.changes);
    } else
      decode_updates_blob(blob, len, decoded);
    materialized = false;
    ImportStep step;
    step.from_f = oplog_frontiers; step.from_vv = vv;
    // receiver-VV filter happens per change in try_apply (drop / slice); application order = lamport order
    std::stable_sort(decoded.begin(), decoded.end(), [](const Change& a, const Change& b) { return a.lamport < b.lamport; });
    for (auto& c : decoded)
      if (try_apply(c)) pending.push_back(std::move(c));
    // pending retry until no progress (pending_changes.rs)
    bool progress = true;
    while (progress && !pending.empty()) {
      progress = false;
      std::vector<Change> still;
      for (auto& c : pending) {
        if (try_apply(c)) still.push_back(std::move(c)); else progress = true;
      }
      pending.swap(still);
    }
    step.to_f = oplog_frontiers; step.to_vv = vv;
    steps.push_back(std::move(step));
  }
  // ---- the DAG as the import path sees it: runs of one peer's changes linked only by self-dependency
  // (AppDagNode, loro_dag.rs:302-367,995-1019)
  std::vector<DagNodeT> dag_nodes() const {
    std::vector<DagNodeT> out;
    for (auto& kv : changes) {
      auto& v = kv.second;
      for (size_t i = 0; i < v.size(); i++) {
        bool cont = i > 0 && v[i].deps.size() == 1 && v[i].deps[0].peer == kv.first && v[i].deps[0].counter == v[i].id.counter - 1 &&
                    v[i].lamport == v[i - 1].lamport + (Lamport)v[i - 1].len;
        if (cont) out.back().len += v[i].len;
        else { DagNodeT n; n.id = v[i].id; n.len = v[i].len; n.lamport = v[i].lamport; n.deps = v[i].deps; out.push_back(n); }
      }
    }
    return out;
  }
  // DiffMode of every import() taken so far, as LoroDoc::import computes it for an attached document
  // (oplog.rs:591-615: find_common_ancestor(from, to); Checkout becomes Import when `to` is strictly greater)
  std::vector<std::pair<Frontiers, DiffMode>> import_modes() const {
    std::vector<DagNodeT> nodes = dag_nodes();
    DagGet get = [&nodes](ID id) -> const DagNodeT* {
      for (const DagNodeT& n : nodes) if (n.contains(id)) return &n;
      return nullptr;
    };
    std::vector<std::pair<Frontiers, DiffMode>> out;
    for (const ImportStep& st : steps) {
      if (st.from_vv == st.to_vv) { out.push_back({st.from_f, DM_LINEAR}); continue; }   // diff_calc.rs:150-152
      auto r = find_common_ancestor(get, st.from_f, st.to_f);
      if (r.second == DM_CHECKOUT) {
        bool ge = true, gt = false;
        for (auto& kv : st.from_vv) { auto it = st.to_vv.find(kv.first); Counter t = it == st.to_vv.end() ? 0 : it->second; if (t < kv.second) ge = false; }
        for (auto& kv : st.to_vv) { auto it = st.from_vv.find(kv.first); Counter f = it == st.from_vv.end() ? 0 : it->second; if (kv.second > f) gt = true; }
        if (ge && gt) r.second = DM_IMPORT;
      }
      out.push_back(r);
    }
    return out;
  }
  uint64_t pending_atoms() const {
    uint64_t n = 0;
    for (auto& c : pending) {
      Counter end = vv_get(c.id.peer);
      if (c.ctr_end() > end) n += (uint64_t)(c.ctr_end() - std::max(end, c.id.counter));
    }
    return n;
  }

  // ---- DAG nodes: runs of one peer's changes linked only by self-dependency (loro_dag.rs:995-1019)
  struct Node { PeerID peer; size_t first, last; std::vector<ID> deps; std::vector<size_t> succ; int indeg = 0; };

  void materialize() {
    if (materialized) return;
    seqs.clear();
    maps.clear();
    touched.clear();
    // nodes
    std::vector<Node> nodes;
    std::map<std::pair<PeerID, Counter>, size_t> node_of_change;  // change start → node
    for (auto& kv : changes) {
      auto& v = kv.second;
      for (size_t i = 0; i < v.size(); i++) {
        bool cont = i > 0 && v[i].deps.size() == 1 && v[i].deps[0].peer == kv.first &&
                    v[i].deps[0].counter == v[i].id.counter - 1;
        if (cont) nodes.back().last = i;
        else { Node n; n.peer = kv.first; n.first = n.last = i; n.deps = v[i].deps; nodes.push_back(n); }
        node_of_change[{kv.first, v[i].id.counter}] = nodes.size() - 1;
      }
    }
    for (size_t ni = 0; ni < nodes.size(); ni++) {
      std::set<size_t> dn;
      for (const ID& d : nodes[ni].deps) {
        const Change* dc = find_change(d);
        dn.insert(node_of_change[{dc->id.peer, dc->id.counter}]);
      }
      for (size_t x : dn) { nodes[x].succ.push_back(ni); nodes[ni].indeg++; }
    }
    std::vector<size_t> order, stack;
    for (size_t ni = nodes.size(); ni-- > 0;) if (nodes[ni].indeg == 0) stack.push_back(ni);
    while (!stack.empty()) {
      size_t n = stack.back();
      stack.pop_back();
      order.push_back(n);
      for (size_t s : nodes[n].succ) if (--nodes[s].indeg == 0) stack.push_back(s);
    }
    if (order.size() != nodes.size()) fail(ST_INTERNAL, "cycle in DAG");
    // vv_head[n] = version seen by the first op of node n (loro_dag.rs:1083-1154)
    std::vector<VV> vv_head(nodes.size());
    for (size_t ni : order) {
      Node& n = nodes[ni];
      VV& out = vv_head[ni];
      for (const ID& d : n.deps) {
        const Change* dc = find_change(d);
        const VV& sub = vv_head[node_of_change[{dc->id.peer, dc->id.counter}]];
        for (auto& kv : sub) Tracker::bump(out, kv.first, kv.second);
        Tracker::bump(out, d.peer, d.counter + 1);
      }
    }
    // checkout target: frontiers → version vector (AppDag::frontiers_to_vv, loro_dag.rs:1190-1207); an id the
    // OpLog does not hold is LoroError::FrontiersNotFound (loro.rs:1699-1701)
    target = vv;
    if (has_target) {
      target.clear();
      for (const ID& f : frontiers) {
        if (f.counter < 0 || !has_id(f)) fail(ST_FRONTIERS_NOT_FOUND, "frontiers not found in the OpLog");
        const Change* fc = find_change(f);
        for (auto& kv : vv_head[node_of_change[{fc->id.peer, fc->id.counter}]]) Tracker::bump(target, kv.first, kv.second);
        Tracker::bump(target, f.peer, f.counter + 1);
      }
    }
    auto in_target = [&](PeerID p, Counter c) { auto it = target.find(p); return it != target.end() && c < it->second; };
    // replay
    for (size_t ni : order) {
      Node& n = nodes[ni];
      auto& v = changes[n.peer];
      VV cur = vv_head[ni];
      for (size_t ci = n.first; ci <= n.last; ci++) {
        const Change& ch = v[ci];
        std::set<uint32_t> visited;
        for (const Op& op : ch.ops) {
          touched.insert(op.container);
          const ContainerID& cid = containers[op.container];
          if (cid.kind == CK_MAP) {
            if (op.kind != OP_MAP_SET && op.kind != OP_MAP_DELETE) continue;
            if (!in_target(ch.id.peer, op.counter)) continue;  // MapHistoryCache::get_container_latest_op_at_vv (history_cache.rs:630-703)
            Lamport lamp = ch.lamport + (Lamport)(op.counter - ch.id.counter);
            auto& m = maps[op.container];
            auto it = m.find(op.key);
            // keep old iff old > new by (lamport, peer) (diff_calc.rs:532-537, map_delta.rs:26-32)
            if (it != m.end() && (it->second.lamp > lamp || (it->second.lamp == lamp && it->second.peer > ch.id.peer))) continue;
            MapEntry e{lamp, ch.id.peer, op.kind == OP_MAP_SET, op.kind == OP_MAP_SET ? op.value : Value()};
            m[op.key] = std::move(e);
            continue;
          }
          if (cid.kind != CK_TEXT && cid.kind != CK_LIST && cid.kind != CK_MOVABLE) continue;
          auto& sp = seqs[op.container];
          if (!sp) sp.reset(new SeqState());
          SeqState& st = *sp;
          if (!visited.count(op.container)) {
            VV at = cur;
            Tracker::bump(at, ch.id.peer, op.counter);  // diff_calc.rs:480-481
            st.tr.checkout(at);
            visited.insert(op.container);
          }
          ID op_id{ch.id.peer, op.counter};
          switch (op.kind) {
            case OP_TEXT_INSERT: {
              uint32_t start = (uint32_t)st.cps.size();
              st.cps.insert(st.cps.end(), op.cps.begin(), op.cps.end());
              st.tr.insert(op_id, op.prop, (int32_t)op.cps.size(), start);
              break;
            }
            case OP_LIST_INSERT: {
              uint32_t start = (uint32_t)st.values.size();
              st.values.insert(st.values.end(), op.values.begin(), op.values.end());
              if (cid.kind == CK_MOVABLE)   // every inserted value is a new element positioned by its own list item (diff_calc.rs:1708-1725)
                for (size_t i = 0; i < op.values.size(); i++) {
                  IdLpKey k{ch.lamport + (Lamport)(op.counter - ch.id.counter) + (Lamport)i, ch.id.peer};
                  st.items.push_back(SeqState::ItemRec{k, k});
                  st.elem_init[k] = start + (uint32_t)i;
                }
              st.tr.insert(op_id, op.prop, (int32_t)op.values.size(), start);
              break;
            }
            case OP_LIST_MOVE: {
              if (cid.kind != CK_MOVABLE) break;
              IdLpKey me{ch.lamport + (Lamport)(op.counter - ch.id.counter), ch.id.peer}, el{op.elem_lamport, op.elem_peer};
              // the element must exist at the op's version ("moved element should have a visible source position",
              // diff_calc.rs:1927-1931): a blob naming an element no insert produced is damaged
              if (!st.elem_init.count(el)) fail(ST_DATA_CORRUPTION, "move of an unknown element");
              uint32_t start = (uint32_t)st.values.size();
              st.values.push_back(Value());
              st.items.push_back(SeqState::ItemRec{me, el});
              st.tr.move_item(op_id, op.move_from, op.prop, start);
              if (in_target(ch.id.peer, op.counter)) {   // last_pos (history_cache.rs:949-1003)
                auto it = st.pos_win.find(el);
                if (it == st.pos_win.end() || it->second < me) st.pos_win[el] = me;
              }
              break;
            }
            case OP_LIST_SET: {
              if (cid.kind != CK_MOVABLE) break;
              IdLpKey me{ch.lamport + (Lamport)(op.counter - ch.id.counter), ch.id.peer}, el{op.elem_lamport, op.elem_peer};
              if (!st.elem_init.count(el)) fail(ST_DATA_CORRUPTION, "set of an unknown element");
              if (in_target(ch.id.peer, op.counter)) {   // last_value (history_cache.rs:865-947)
                auto it = st.val_win.find(el);
                if (it == st.val_win.end() || it->second.id < me) st.val_win[el] = SeqState::ValWin{me, op.value};
              }
              break;
            }
            case OP_SEQ_DELETE: {
              int64_t sl = op.del_signed_len;
              int64_t L = sl < 0 ? -sl : sl;
              int64_t start = sl > 0 ? op.prop : (int64_t)op.prop + 1 + sl;  // DeleteSpan::start (list_op.rs:303-309)
              st.tr.del(op_id, op.del_id_start, start, (int32_t)L, sl < 0);
              break;
            }
            case OP_STYLE_START: {
              uint32_t start = (uint32_t)st.cps.size();
              st.cps.push_back(0xFFFFFFFFu);
              st.styles.push_back(StyleRec{ch.id.peer, op.counter, op.style_end});
              st.anchors[start] = AnchorRec{false, ch.id.peer, op.counter, ch.lamport + (Lamport)(op.counter - ch.id.counter), op.key, op.value};
              st.tr.insert(op_id, op.prop, 1, start);
              break;
            }
            case OP_STYLE_END: {
              // diff_calc.rs:1105-1132
              int64_t end_pos = -1;
              for (size_t k = st.styles.size(); k-- > 0;)
                if (st.styles[k].peer == ch.id.peer && st.styles[k].cnt == op.counter - 1) { end_pos = st.styles[k].end; break; }
              if (end_pos < 0) {
                const Change* sc = find_change(ID{ch.id.peer, op.counter - 1});
                if (sc)
                  for (const Op& so : sc->ops)
                    if (so.counter == op.counter - 1 && so.kind == OP_STYLE_START) end_pos = so.style_end;
                if (end_pos < 0) fail(ST_DATA_CORRUPTION, "style end without start");
              }
              uint32_t start = (uint32_t)st.cps.size();
              st.cps.push_back(0xFFFFFFFFu);
              st.anchors[start] = AnchorRec{true, ch.id.peer, op.counter - 1, 0, std::string(), Value()};
              int64_t pos = std::min<int64_t>(end_pos + 1, st.tr.active_len());
              st.tr.insert(op_id, pos, 1, start);
              break;
            }
            default: break;
          }
        }
        Tracker::bump(cur, ch.id.peer, ch.ctr_end());
      }
    }
    // the state store's containers: step 1 = diff(∅ → latest), step 2 (checkout) = diff(latest → version).
    // A container state, once created, stays (state.rs:621-849 never drops one): when a document is rendered in several
    // steps (import, render, import more, checkout, ... — lo_session_step), `seq_exists` carries over from step to step;
    // a batch is one step, for which it starts empty.
    for (auto& kv : seqs) {
      kv.second->tr.checkout(vv);
      if (kv.second->tr.active_len() > 0) seq_exists.insert(kv.first);
      // MovableList: the diff also lists every element the version knows (MovableListInnerDelta::is_empty,
      // delta/movable_list.rs:32-34; diff_calc.rs:1880-1924 keeps an element whose insert lies inside the version), so the
      // state exists once one element was inserted — even if it was deleted again
      if (containers[kv.first].kind == CK_MOVABLE && !kv.second->elem_init.empty()) seq_exists.insert(kv.first);
    }
    for (auto& kv : seqs) {
      kv.second->tr.checkout(target);   // tracker.rs:354-461: ops outside the version become future / un-deleted
      if (kv.second->tr.active_len() > 0) seq_exists.insert(kv.first);
    }
    materialized = true;
  }

  // ---- deep value as canonical JSON (state.rs:1294-1329; keys sorted bytewise)
  void container_json(uint32_t idx, std::string& out, int depth) const {
    if (depth > 200) { out += "null"; return; }
    const ContainerID& cid = containers[idx];
    if (cid.kind == CK_TEXT) {
      std::string s;
      auto it = seqs.find(idx);
      if (it != seqs.end())
        for (Span* sp = it->second->tr.head; sp; sp = sp->next)
          if (sp->active())
            for (int32_t k = 0; k < sp->len; k++) {
              // a visible span without content: a delete of ids no insert ever produced left its placeholder behind
              // (damaged input only; what the Rust state would hold for it is undefined)
              if ((uint64_t)sp->content + (uint32_t)k >= it->second->cps.size()) fail(ST_DATA_CORRUPTION, "visible span without content");
              uint32_t cp = it->second->cps[sp->content + (uint32_t)k];
              if (cp != 0xFFFFFFFFu) cp_to_utf8(cp, s);
            }
      json_escape(s, out);
    } else if (cid.kind == CK_LIST) {
      out.push_back('[');
      bool first = true;
      auto it = seqs.find(idx);
      if (it != seqs.end())
        for (Span* sp = it->second->tr.head; sp; sp = sp->next)
          if (sp->active())
            for (int32_t k = 0; k < sp->len; k++) {
              if (!first) out.push_back(',');
              first = false;
              if ((uint64_t)sp->content + (uint32_t)k >= it->second->values.size()) fail(ST_DATA_CORRUPTION, "visible span without content");
              json_value(*this, it->second->values[sp->content + (uint32_t)k], out, depth + 1);
            }
      out.push_back(']');
    } else if (cid.kind == CK_MOVABLE) {
      // movable_list_state.rs:953-964: the list items some element points at, in list order, each showing the element's value
      out.push_back('[');
      bool first = true;
      auto it = seqs.find(idx);
      if (it != seqs.end()) {
        const SeqState& st = *it->second;
        for (Span* sp = st.tr.head; sp; sp = sp->next)
          if (sp->active())
            for (int32_t k = 0; k < sp->len; k++) {
              size_t ci = (size_t)sp->content + (uint32_t)k;
              if (ci >= st.items.size()) fail(ST_DATA_CORRUPTION, "visible span without content");
              const SeqState::ItemRec& rec = st.items[ci];
              auto pw = st.pos_win.find(rec.elem);
              if (!((pw == st.pos_win.end() ? rec.elem : pw->second) == rec.item)) continue;   // not the element's current position
              auto vw = st.val_win.find(rec.elem);
              if (!first) out.push_back(',');
              first = false;
              json_value(*this, vw != st.val_win.end() ? vw->second.v : st.values[st.elem_init.at(rec.elem)], out, depth + 1);
            }
      }
      out.push_back(']');
    } else if (cid.kind == CK_MAP) {
      out.push_back('{');
      bool first = true;
      auto it = maps.find(idx);
      if (it != maps.end())
        for (auto& kv : it->second) {  // std::map<std::string,..> iterates bytewise-sorted
          if (!kv.second.has) continue;
          if (!first) out.push_back(',');
          first = false;
          json_escape(kv.first, out);
          out.push_back(':');
          json_value(*this, kv.second.v, out, depth + 1);
        }
      out.push_back('}');
    } else {
      out += "null";
    }
  }
  std::string to_json() {
    materialize();
    std::map<std::string, uint32_t> roots;
    for (uint32_t i = 0; i < containers.size(); i++) {
      if (!containers[i].root || !(touched.count(i) || state_roots.count(i))) continue;
      if ((containers[i].kind == CK_TEXT || containers[i].kind == CK_LIST || containers[i].kind == CK_MOVABLE) && !seq_exists.count(i) && !state_roots.count(i)) continue;
      if (roots.count(containers[i].name)) fail(ST_UNSUPPORTED, "two root containers share a name");
      roots[containers[i].name] = i;
    }
    std::string out = "{";
    bool first = true;
    for (auto& kv : roots) {
      if (!first) out.push_back(',');
      first = false;
      json_escape(kv.first, out);
      out.push_back(':');
      container_json(kv.second, out, 0);
    }
    out.push_back('}');
    return out;
  }
  // ---- richtext values (TextHandler::get_richtext_value, handler.rs:1502; richtext_state.rs:2500-2584)
  // A scalar carries the StyleOps whose Start anchor stands in front of it and whose End anchor stands behind it
  // (state/richtext_state.rs:730-812: the End anchor's insertion annotates start..=end; style_range_map.rs insert():
  // an element inserted inside a range inherits it, at a boundary the intersection of both sides; an anchor that is
  // deleted takes its range along, richtext_state.rs:2275-2300) — for every key the op with the greatest (lamport, peer)
  // decides (StyleValue::get = BTreeSet::last under StyleOp::cmp, container/richtext.rs:120-126), a null value removes the
  // key (StyleMeta::to_value, delta/text.rs:125-140); neighbouring spans with equal attributes are one span
  // (richtext_state.rs:2546-2584).  A span is rendered as the canonical JSON of the LoroValue map it is:
  // {"attributes":{…},"insert":"…"} (keys bytewise sorted; no attributes entry when the map is empty).
  // Anchors without their partner at the rendered version annotate nothing (more than 64 StyleOps open at one scalar:
  // the device path's limit, LM_UNSUPPORTED there).
  void richtext_json(uint32_t idx, std::string& out) const {
    out.push_back('[');
    auto it = seqs.find(idx);
    if (it == seqs.end()) { out.push_back(']'); return; }
    const SeqState& st = *it->second;
    std::map<std::tuple<PeerID, Counter, bool>, uint64_t> alive;   // visible anchors → position in the sequence
    uint64_t seq_pos = 0;
    for (Span* sp = st.tr.head; sp; sp = sp->next)
      if (sp->active())
        for (int32_t k = 0; k < sp->len; k++, seq_pos++) {
          auto a = st.anchors.find(sp->content + (uint32_t)k);
          if (a != st.anchors.end()) alive[std::make_tuple(a->second.peer, a->second.start_cnt, a->second.is_end)] = seq_pos;
        }
    seq_pos = 0;
    std::vector<const AnchorRec*> active;
    std::string open_attr, cur_text;
    bool span_open = false, first = true;
    auto attrs_now = [&]() {
      std::map<std::string, const AnchorRec*> win;
      for (const AnchorRec* a : active) {
        auto w = win.find(a->key);
        if (w == win.end() || w->second->lamport < a->lamport || (w->second->lamport == a->lamport && w->second->peer < a->peer)) win[a->key] = a;
      }
      std::string o;
      for (auto& kv : win) {
        if (kv.second->value.kind == V_NULL) continue;
        o.push_back(o.empty() ? '{' : ',');
        json_escape(kv.first, o);
        o.push_back(':');
        json_value(*this, kv.second->value, o, 1);
      }
      if (!o.empty()) o.push_back('}');
      return o;
    };
    auto close_span = [&]() {
      if (!span_open) return;
      if (!first) out.push_back(',');
      first = false;
      out.push_back('{');
      if (!open_attr.empty()) { out += "\"attributes\":"; out += open_attr; out.push_back(','); }
      out += "\"insert\":";
      json_escape(cur_text, out);
      out.push_back('}');
      span_open = false;
      cur_text.clear();
    };
    for (Span* sp = st.tr.head; sp; sp = sp->next)
      if (sp->active())
        for (int32_t k = 0; k < sp->len; k++, seq_pos++) {
          uint32_t ci = sp->content + (uint32_t)k;
          if ((uint64_t)ci >= st.cps.size()) fail(ST_DATA_CORRUPTION, "visible span without content");
          uint32_t cp = st.cps[ci];
          if (cp != 0xFFFFFFFFu) {
            std::string a = attrs_now();
            if (span_open && a != open_attr) close_span();
            if (!span_open) { span_open = true; open_attr = a; }
            cp_to_utf8(cp, cur_text);
            continue;
          }
          auto a = st.anchors.find(ci);
          if (a == st.anchors.end()) continue;
          const AnchorRec& r = a->second;
          if (!r.is_end) {   // (an End in FRONT of its Start — damaged input only — annotates nothing)
            auto e = alive.find(std::make_tuple(r.peer, r.start_cnt, true));
            if (e != alive.end() && e->second > seq_pos) active.push_back(&r);
          }
          else
            for (size_t j = 0; j < active.size(); j++)
              if (active[j]->peer == r.peer && active[j]->start_cnt == r.start_cnt) { active.erase(active.begin() + j); break; }
        }
    close_span();
    out.push_back(']');
  }
  static std::string cid_string(const ContainerID& c) {   // ContainerID's Display (loro-common/src/lib.rs: "cid:root-{name}:{Type}" / "cid:{counter}@{peer}:{Type}")
    return c.root ? "cid:root-" + c.name + ":Text" : "cid:" + std::to_string(c.counter) + "@" + std::to_string(c.peer) + ":Text";
  }
  // every Text container in which something (a scalar, an anchor) is visible at the rendered version:
  // {"<container id>": <richtext value>, …}, members in the bytewise order of their JSON-encoded keys
  std::string to_richtext() {
    materialize();
    std::map<std::string, uint32_t> texts;
    for (uint32_t i = 0; i < containers.size(); i++) {
      auto it = seqs.find(i);
      if (!(containers[i].kind == CK_TEXT && it != seqs.end() && it->second->tr.active_len() > 0)) continue;
      std::string esc;
      json_escape(cid_string(containers[i]), esc);   // members in the bytewise order of their JSON-encoded keys (the quotes are not compared)
      texts[esc.substr(1, esc.size() - 2)] = i;
    }
    std::string out = "{";
    bool first = true;
    for (auto& kv : texts) {
      if (!first) out.push_back(',');
      first = false;
      out.push_back('"'); out += kv.first; out.push_back('"');
      out.push_back(':');
      richtext_json(kv.second, out);
    }
    out.push_back('}');
    return out;
  }
  // postcard map, entries sorted by peer: the version of the rendered state (= oplog_vv() unless checked out)
  std::string vv_bytes() {
    materialize();
    const VV& vv = target;
    std::string out;
    auto uleb = [&](uint64_t v) {
      do { uint8_t b = v & 0x7f; v >>= 7; if (v) b |= 0x80; out.push_back((char)b); } while (v);
    };
    uleb(vv.size());
    for (auto& kv : vv) {
      uleb(kv.first);
      int64_t c = kv.second;
      uleb((uint64_t)((c << 1) ^ (c >> 63)));
    }
    return out;
  }
  // visible element ids of a sequence container (test helper for local-edit generators)
  std::vector<ID> visible_ids(const ContainerID& cid) {
    materialize();
    std::vector<ID> out;
    auto ci = container_idx.find(cid);
    if (ci == container_idx.end()) return out;
    auto it = seqs.find(ci->second);
    if (it == seqs.end()) return out;
    for (Span* sp = it->second->tr.head; sp; sp = sp->next)
      if (sp->active())
        for (int32_t k = 0; k < sp->len; k++) out.push_back(ID{sp->id.peer, sp->id.counter + k});
    return out;
  }
};

inline void json_value(const Doc& d, const Value& v, std::string& out, int depth) {
  switch (v.kind) {
    case V_NULL: out += "null"; break;
    case V_BOOL: out += v.b ? "true" : "false"; break;
    case V_I64: out += std::to_string(v.i); break;
    case V_F64: json_f64(v.f, out); break;
    case V_STR: json_escape(v.s, out); break;
    case V_BIN: {
      out.push_back('[');
      for (size_t i = 0; i < v.s.size(); i++) { if (i) out.push_back(','); out += std::to_string((unsigned)(uint8_t)v.s[i]); }
      out.push_back(']');
      break;
    }
    case V_LIST: {
      out.push_back('[');
      for (size_t i = 0; i < v.list.size(); i++) { if (i) out.push_back(','); json_value(d, v.list[i], out, depth + 1); }
      out.push_back(']');
      break;
    }
    case V_MAP: {
      std::vector<const std::pair<std::string, Value>*> es;
      for (auto& e : v.map) es.push_back(&e);
      std::stable_sort(es.begin(), es.end(), [](auto a, auto b) { return a->first < b->first; });
      out.push_back('{');
      bool first = true;
      for (size_t i = 0; i < es.size(); i++) {
        if (i + 1 < es.size() && es[i + 1]->first == es[i]->first) continue;  // last write wins on duplicate keys
        if (!first) out.push_back(',');
        first = false;
        json_escape(es[i]->first, out);
        out.push_back(':');
        json_value(d, es[i]->second, out, depth + 1);
      }
      out.push_back('}');
      break;
    }
    case V_CONTAINER: {
      auto it = d.container_idx.find(v.cid);
      if (it == d.container_idx.end()) {
        // child container that never received an op: empty value of its kind (state.rs:1550-1616)
        if (v.cid.kind == CK_TEXT) out += "\"\"";
        else if (v.cid.kind == CK_MAP) out += "{}";
        else if (v.cid.kind == CK_LIST || v.cid.kind == CK_MOVABLE) out += "[]";
        else { out += "null"; d.unsupported = true; }   // Tree / Counter child: outside the scope
      } else d.container_json(it->second, out, depth + 1);
      break;
    }
  }
}

}  // namespace lo
