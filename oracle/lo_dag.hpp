// ORACLE — TEST INFRASTRUCTURE ONLY (see lo_codec.hpp header).
// CPU restatement of the DAG queries the import path asks before it diffs (reference file:line, relative to
// /root/reference/crates/loro-internal/src):
//   _find_common_ancestor_new → (LCA frontiers, DiffMode)      dag.rs:487-765 (OrdIdSpan order :269-280, NodeType :282-287)
//   DiffMode                                                   diff_calc.rs:70-110 (Checkout / Import / ImportGreaterUpdates / Linear)
//   the Checkout → Import promotion when `to > from`           oplog.rs:610-615
// Pinned by tests/test_oracle_golden.py against the known answers of dag.rs:1108-1340 and the brute-force property of
// dag.rs:955-1009 (ancestor sets by exhaustive walk) on random DAGs.
#pragma once
#include <algorithm>
#include <functional>
#include <queue>
#include <set>
#include <vector>
#include "lo_codec.hpp"

namespace lo {

enum DiffMode : int32_t { DM_CHECKOUT = 0, DM_IMPORT = 1, DM_IMPORT_GREATER = 2, DM_LINEAR = 3 };

struct DagNodeT {   // a run of one peer's ops: [id, id+len), lamport of the first op, deps of the first op
  ID id;
  int32_t len = 0;
  Lamport lamport = 0;
  std::vector<ID> deps;
  bool contains(ID x) const { return x.peer == id.peer && x.counter >= id.counter && x.counter < id.counter + len; }
};
using Frontiers = std::vector<ID>;   // a set; kept sorted by (peer, counter)
using DagGet = std::function<const DagNodeT*(ID)>;

inline bool id_less(const ID& a, const ID& b) { return a.peer != b.peer ? a.peer < b.peer : a.counter < b.counter; }
inline bool id_eq(const ID& a, const ID& b) { return a.peer == b.peer && a.counter == b.counter; }
inline void fr_norm(Frontiers& f) {
  std::sort(f.begin(), f.end(), id_less);
  f.erase(std::unique(f.begin(), f.end(), id_eq), f.end());
}

namespace dagimpl {
struct OrdSpan {   // dag.rs:238-243
  ID id;
  Lamport lamport = 0;
  size_t len = 0;
  std::vector<ID> deps;
  Lamport lamport_last() const { return lamport + (Lamport)len - 1; }
  ID id_last() const { return ID{id.peer, id.counter + (Counter)len - 1}; }
  bool contains_id(ID x) const { return x.peer == id.peer && x.counter >= id.counter && x.counter < id.counter + (Counter)len; }
  bool same(const OrdSpan& o) const {   // derived PartialEq: every field
    if (!id_eq(id, o.id) || lamport != o.lamport || len != o.len || deps.size() != o.deps.size()) return false;
    Frontiers a = deps, b = o.deps;
    fr_norm(a); fr_norm(b);
    for (size_t i = 0; i < a.size(); i++) if (!id_eq(a[i], b[i])) return false;
    return true;
  }
};
// Ord (dag.rs:269-280): lamport_last, then peer, then the SHORTER span is the greater one
inline int span_cmp(const OrdSpan& a, const OrdSpan& b) {
  if (a.lamport_last() != b.lamport_last()) return a.lamport_last() < b.lamport_last() ? -1 : 1;
  if (a.id.peer != b.id.peer) return a.id.peer < b.id.peer ? -1 : 1;
  if (a.len != b.len) return a.len > b.len ? -1 : 1;
  return 0;
}
enum NodeType { NT_A = 0, NT_B = 1, NT_SHARED = 2 };
struct HeapItem { OrdSpan span; int type; };
struct HeapLess {
  bool operator()(const HeapItem& a, const HeapItem& b) const {
    int c = span_cmp(a.span, b.span);
    return c != 0 ? c < 0 : a.type < b.type;
  }
};
inline bool from_dag_node(ID id, const DagGet& get, OrdSpan& out) {   // dag.rs:291-304
  const DagNodeT* n = get(id);
  if (!n) return false;
  out.id = n->id; out.lamport = n->lamport; out.deps = n->deps; out.len = (size_t)(id.counter - n->id.counter) + 1;
  return true;
}
inline bool ids_to_spans(const Frontiers& ids, const DagGet& get, std::vector<OrdSpan>& out) {
  out.clear();
  for (const ID& id : ids) { OrdSpan s; if (!from_dag_node(id, get, s)) return false; out.push_back(s); }
  return true;
}
inline bool deps_to_spans(const OrdSpan& node, const DagGet& get, std::vector<OrdSpan>& deps) {   // dag.rs:592-608
  if (!ids_to_spans(node.deps, get, deps)) return false;
  if (node.id.counter > 0) {
    OrdSpan prev;
    if (from_dag_node(ID{node.id.peer, node.id.counter - 1}, get, prev)) {
      bool covered = false;
      for (auto& d : deps) if (d.contains_id(prev.id_last())) covered = true;
      if (!covered) deps.push_back(prev);
    }
  }
  return true;
}
inline bool contains_in_ancestors(const DagGet& get, ID frontier, const OrdSpan& target) {   // dag.rs:647-679
  std::set<std::pair<PeerID, Counter>> visited;
  std::vector<OrdSpan> pending;
  OrdSpan n;
  if (!from_dag_node(frontier, get, n)) return false;
  pending.push_back(n);
  while (!pending.empty()) {
    OrdSpan node = pending.back();
    pending.pop_back();
    if (node.contains_id(target.id_last())) return true;
    if (node.lamport_last() < target.lamport_last()) continue;
    if (!visited.insert({node.id.peer, node.id.counter}).second) continue;
    std::vector<OrdSpan> deps;
    if (deps_to_spans(node, get, deps)) for (auto& d : deps) pending.push_back(d);
  }
  return false;
}
inline Frontiers shrink_ancestor_frontiers(const Frontiers& ids, const DagGet& get) {   // dag.rs:610-634
  if (ids.size() <= 1) return ids;
  std::vector<OrdSpan> spans;
  if (!ids_to_spans(ids, get, spans)) fail(ST_INTERNAL, "common ancestors should be in dag");
  std::sort(spans.begin(), spans.end(), [](const OrdSpan& a, const OrdSpan& b) { return span_cmp(a, b) < 0; });
  Frontiers fr;
  for (size_t i = spans.size(); i-- > 0;) {
    bool ins = true;
    for (size_t k = fr.size(); k-- > 0;) if (contains_in_ancestors(get, fr[k], spans[i])) { ins = false; break; }
    if (ins) fr.push_back(spans[i].id_last());
  }
  fr_norm(fr);
  return fr;
}
inline bool has_trimmed_history_deps(const Frontiers& ids, const DagGet& get) {   // dag.rs:636-645
  for (const ID& id : ids) {
    OrdSpan n;
    if (!from_dag_node(id, get, n)) return true;
    std::vector<OrdSpan> d;
    if (!ids_to_spans(n.deps, get, d)) return true;
  }
  return false;
}
}  // namespace dagimpl

// dag.rs:318-332 + 487-765.  `left` = the version the state is at, `right` = the version to reach.
inline std::pair<Frontiers, DiffMode> find_common_ancestor(const DagGet& get, Frontiers left, Frontiers right) {
  using namespace dagimpl;
  fr_norm(left); fr_norm(right);
  if (right.empty()) return {Frontiers{}, DM_CHECKOUT};
  if (left.empty()) {
    if (right.size() == 1) {
      const DagNodeT* node = get(right[0]);
      if (!node) fail(ST_INTERNAL, "frontier not in dag");
      bool broke = false;
      while (node->deps.size() == 1) {
        const DagNodeT* next = get(node->deps[0]);
        if (!next) { broke = true; break; }
        node = next;
      }
      if (broke) return {Frontiers{}, DM_IMPORT_GREATER};
      if (node->deps.empty()) return {Frontiers{}, DM_LINEAR};
    }
    return {Frontiers{}, DM_IMPORT_GREATER};
  }
  if (left.size() == 1 && right.size() == 1) {
    ID l = left[0], r = right[0];
    if (l.peer == r.peer) {
      const DagNodeT* ls = get(l);
      const DagNodeT* rs = get(r);
      if (!ls || !rs) fail(ST_INTERNAL, "frontier not in dag");
      if (id_eq(ls->id, rs->id)) {
        if (l.counter < r.counter) return {Frontiers{l}, DM_LINEAR};
        return {Frontiers{r}, DM_CHECKOUT};
      }
      if (ls->deps.size() == 1 && rs->contains(ls->deps[0])) return {Frontiers{r}, DM_CHECKOUT};
      if (rs->deps.size() == 1 && ls->contains(rs->deps[0])) return {Frontiers{l}, DM_LINEAR};
    }
  }
  bool is_linear = left.size() <= 1 && right.size() == 1;
  bool is_right_greater = true, has_unmatched_branch = false;
  Frontiers ans;
  std::priority_queue<HeapItem, std::vector<HeapItem>, HeapLess> queue;
  std::vector<OrdSpan> spans;
  if (!ids_to_spans(left, get, spans)) fail(ST_INTERNAL, "frontier not in dag");
  for (auto& s : spans) queue.push(HeapItem{s, NT_A});
  if (!ids_to_spans(right, get, spans)) fail(ST_INTERNAL, "frontier not in dag");
  for (auto& s : spans) queue.push(HeapItem{s, NT_B});
  while (!queue.empty()) {
    HeapItem top = queue.top();
    queue.pop();
    OrdSpan node = top.span;
    int node_type = top.type;
    while (!queue.empty()) {
      const HeapItem& o = queue.top();
      if (node.same(o.span) || id_eq(node.id_last(), o.span.id_last())) {
        if (node_type != o.type) node_type = NT_SHARED;
        queue.pop();
      } else break;
    }
    if (node_type == NT_SHARED) { ans.push_back(node.id_last()); continue; }
    if (queue.empty()) { has_unmatched_branch = true; is_right_greater = false; break; }
    if (node_type == NT_A) is_right_greater = false;
    {
      const HeapItem& other = queue.top();
      if (node.contains_id(other.span.id_last()) && node_type != other.type) {
        node.len = (size_t)(other.span.id_last().counter - node.id.counter + 1);
        queue.push(HeapItem{node, node_type});
        continue;
      }
      if (node.len > 1) {
        if (other.span.lamport_last() >= node.lamport) {
          size_t a = (size_t)(other.span.lamport_last() - node.lamport + 1), b = node.len - 1;
          node.len = a < b ? a : b;
        } else node.len = 1;
        queue.push(HeapItem{node, node_type});
        continue;
      }
    }
    std::vector<OrdSpan> deps;
    if (deps_to_spans(node, get, deps)) {
      if (!deps.empty()) {
        for (auto& d : deps) queue.push(HeapItem{d, node_type});
        is_linear = false;
        continue;
      }
    } else { has_unmatched_branch = true; is_right_greater = false; continue; }
    // a root reached on one side only: an earlier common ancestor is the conservative base (dag.rs:727-735)
    has_unmatched_branch = true;
    is_right_greater = false;
  }
  fr_norm(ans);
  ans = shrink_ancestor_frontiers(ans, get);
  if (has_unmatched_branch && !has_trimmed_history_deps(ans, get)) ans.clear();
  DiffMode mode = is_right_greater ? (is_linear ? DM_LINEAR : DM_IMPORT_GREATER) : DM_CHECKOUT;
  return {ans, mode};
}

// ---- brute force used by the property test (dag.rs:955-1009): every id that is an ancestor of (or equal to) an id
inline void collect_ancestors(const std::vector<DagNodeT>& nodes, ID id, std::set<std::pair<PeerID, Counter>>& ans) {
  std::vector<ID> stack{id};
  std::set<std::pair<PeerID, Counter>> seen;
  while (!stack.empty()) {
    ID x = stack.back();
    stack.pop_back();
    if (!seen.insert({x.peer, x.counter}).second) continue;
    for (const DagNodeT& n : nodes) {
      if (n.id.peer != x.peer || n.id.counter > x.counter) continue;
      Counter end = std::min<Counter>(n.id.counter + n.len - 1, x.counter);
      for (Counter c = n.id.counter; c <= end; c++) ans.insert({n.id.peer, c});
      for (const ID& d : n.deps) stack.push_back(d);
    }
  }
}

}  // namespace lo
