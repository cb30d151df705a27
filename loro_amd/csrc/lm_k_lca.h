// The DAG query the reference asks before it diffs an import: the common ancestors of the version the document was at and the
// version it reaches, and the DiffMode they imply.  One wave per resident document whose history grew; the walk
// is a priority queue over spans of DAG nodes and is serial by nature — lane 0 runs it, on the tables k_dag_a / k_dag_b built.
// Reference (paths relative to /root/reference/crates/loro-internal/src):
//   find_common_ancestor / _find_common_ancestor_new        dag.rs:318-332, 487-765   (OrdIdSpan order :269-280, NodeType :282-287)
//   deps of a span incl. the peer's previous op              dag.rs:592-608
//   shrink_ancestor_frontiers / contains_in_ancestors        dag.rs:610-679
//   vv → frontiers                                           oplog/loro_dag.rs:1269-1298 (shrink_frontiers)
//   Checkout → Import when the target is greater             oplog.rs:610-615;  unchanged version → Linear  diff_calc.rs:150-152
//   DiffMode                                                 diff_calc.rs:72-103
#pragma once
#include "lm_k_integrate_span.h"

namespace lm {

enum : uint32_t { DM_CHECKOUT = 0, DM_IMPORT = 1, DM_IMPORT_GREATER = 2, DM_LINEAR = 3, DM_UNKNOWN = 0xFFFFFFFFu };
static constexpr uint32_t LCA_MAXF = 16;               // common-ancestor ids reported per document (more: DM_UNKNOWN)
static constexpr uint32_t LCA_OUT = 2 + 3 * LCA_MAXF;  // words per document: mode, n, n × (PeerID lo, PeerID hi, counter)

struct DevLca {
  uint32_t* out;                 // [doc * LCA_OUT]
  uint32_t* scratch;             // per document: heap, DFS stack, visited marks
  const uint64_t* scratch_off;   // [n_docs + 1] word offsets into scratch
  // the previous run's version of every document (copies taken before this run's kernels overwrote the tables)
  const DocMeta* prev_doc;       // nullptr: there was no previous run
  const uint64_t* prev_uniq;
  const uint32_t* prev_end;
};

struct LcaNode { uint32_t peer, ctr0, len, lam, dep0, n_dep; };
struct LcaCtx {
  const Dev* d; const DevDag* g; const DocMeta* m; uint32_t P; uint64_t vvh0;
};
LM_DEV LcaNode lca_node(const LcaCtx& x, uint32_t n) {
  const Dev& d = *x.d; const DocMeta& m = *x.m;
  uint32_t first = d.node_first[m.chg0 + n], last = d.node_last[m.chg0 + n];
  const ChangeRow fc = d.chg[d.chg_sorted[m.chg0 + first]], lc = d.chg[d.chg_sorted[m.chg0 + last]];
  LcaNode r;
  r.peer = fc.peer; r.ctr0 = fc.ctr; r.len = lc.ctr + lc.len - fc.ctr; r.lam = x.g->node_lam[m.chg0 + n]; r.dep0 = fc.dep0; r.n_dep = fc.n_dep;
  return r;
}
LM_DEV uint32_t lca_node_of(const LcaCtx& x, uint32_t peer, uint32_t ctr) {   // the node holding op (peer, ctr), NONE if the history does not
  if (peer >= x.P) return NONE;
  uint32_t ci = find_change(*x.d, *x.m, peer, ctr);
  return ci == NONE ? NONE : x.g->chg_node[x.m->chg0 + ci];
}
// a span = the first `len` ops of node `n`; heap order: lamport of its last op, then peer, then the SHORTER span first, then type
struct LcaSpan { uint32_t n, len, type; };
LM_DEV bool lca_less(const LcaCtx& x, const LcaSpan& a, const LcaSpan& b) {
  LcaNode na = lca_node(x, a.n), nb = lca_node(x, b.n);
  uint32_t la = na.lam + a.len - 1, lb = nb.lam + b.len - 1;
  if (la != lb) return la < lb;
  if (na.peer != nb.peer) return na.peer < nb.peer;
  if (a.len != b.len) return a.len > b.len;
  return a.type < b.type;
}
struct LcaHeap { LcaSpan* a; uint32_t n, cap; bool overflow; };
LM_DEV void lca_push(const LcaCtx& x, LcaHeap& h, LcaSpan s) {
  if (h.n >= h.cap) { h.overflow = true; return; }
  uint32_t i = h.n++;
  h.a[i] = s;
  while (i > 0) {
    uint32_t p = (i - 1) / 2;
    if (!lca_less(x, h.a[p], h.a[i])) break;
    LcaSpan t = h.a[p]; h.a[p] = h.a[i]; h.a[i] = t;
    i = p;
  }
}
LM_DEV LcaSpan lca_pop(const LcaCtx& x, LcaHeap& h) {
  LcaSpan top = h.a[0];
  h.a[0] = h.a[--h.n];
  uint32_t i = 0;
  for (;;) {
    uint32_t l = 2 * i + 1, r = l + 1, b = i;
    if (l < h.n && lca_less(x, h.a[b], h.a[l])) b = l;
    if (r < h.n && lca_less(x, h.a[b], h.a[r])) b = r;
    if (b == i) break;
    LcaSpan t = h.a[b]; h.a[b] = h.a[i]; h.a[i] = t;
    i = b;
  }
  return top;
}
// the spans a span depends on (dag.rs:592-608): the node's dependencies, each up to the op depended on, and the peer's own
// previous op unless one of them already covers it.  Returns false when a dependency is not in the history.
LM_DEV bool lca_deps(const LcaCtx& x, const LcaNode& nd, LcaSpan* out, uint32_t& n_out, uint32_t cap, uint32_t type) {
  n_out = 0;
  for (uint32_t k = nd.dep0; k < nd.dep0 + nd.n_dep; k++) {
    uint32_t q = x.d->dep_peer[k], c = x.d->dep_ctr[k];
    uint32_t dn = lca_node_of(x, q, c);
    if (dn == NONE || n_out >= cap) return false;
    out[n_out++] = LcaSpan{dn, c - lca_node(x, dn).ctr0 + 1, type};
  }
  if (nd.ctr0 > 0) {
    uint32_t pn = lca_node_of(x, nd.peer, nd.ctr0 - 1);
    if (pn != NONE) {
      bool covered = false;
      for (uint32_t i = 0; i < n_out; i++) {
        LcaNode o = lca_node(x, out[i].n);
        covered |= o.peer == nd.peer && o.ctr0 <= nd.ctr0 - 1 && nd.ctr0 - 1 < o.ctr0 + out[i].len;
      }
      if (!covered) { if (n_out >= cap) return false; out[n_out++] = LcaSpan{pn, nd.ctr0 - lca_node(x, pn).ctr0, type}; }
    }
  }
  return true;
}
// frontiers of a version vector: the last op of every peer that no other peer's last op has in its causal past
LM_DEV uint32_t lca_vv_frontiers(const LcaCtx& x, const uint32_t* vv, uint32_t* f_peer, uint32_t* f_ctr, uint32_t cap) {
  uint32_t n = 0;
  for (uint32_t p = 0; p < x.P; p++) {
    uint32_t e = vv[p];
    if (e == 0) continue;
    bool dominated = false;
    for (uint32_t q = 0; q < x.P && !dominated; q++) {
      if (q == p || vv[q] == 0) continue;
      uint32_t nq = lca_node_of(x, q, vv[q] - 1);
      if (nq != NONE && x.d->vvh[x.vvh0 + (uint64_t)nq * x.P + p] >= e) dominated = true;
    }
    if (!dominated) { if (n < cap) { f_peer[n] = p; f_ctr[n] = e - 1; } n++; }
  }
  return n;
}

static constexpr uint32_t LCA_FMAX = 64;   // heads a version may have here (a version with more: DM_UNKNOWN)
LM_DEV uint32_t lca_scratch_words(uint32_t n_nodes) { return 3 * (8 * n_nodes + 64) + 3 * (4 * n_nodes + 64) + n_nodes + 16; }

LM_KERNEL void k_import_lca(Dev d, DevDag g, DevLca lc) {
  uint32_t doc = (uint32_t)lmw::bid();
  if (lmw::lane() != 0) return;
  uint32_t* out = lc.out + (uint64_t)doc * LCA_OUT;
  out[0] = DM_UNKNOWN; out[1] = 0;
  const DocMeta m = d.doc[doc];
  if (status_fatal(m.status)) return;
  LcaCtx x;
  x.d = &d; x.g = &g; x.m = &m; x.P = m.n_peers; x.vvh0 = ((uint64_t)m.vvh0_hi << 32) | m.vvh0_lo;
  const uint32_t P = m.n_peers, N = m.n_nodes;
  if (P > MAX_PEERS) return;
  uint32_t from[MAX_PEERS], to[MAX_PEERS];
  bool same = true, to_ge = true, to_gt = false;
  for (uint32_t p = 0; p < P; p++) { from[p] = 0; to[p] = d.peer_end_all[m.praw0 + p]; }
  if (lc.prev_doc) {
    const DocMeta pm = lc.prev_doc[doc];
    if (!status_fatal(pm.status))
      for (uint32_t q = 0; q < pm.n_peers; q++) {
        uint64_t id = lc.prev_uniq[pm.praw0 + q];
        uint32_t lo = 0, hi = P;
        while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.peer_uniq[m.praw0 + mid] < id) lo = mid + 1; else hi = mid; }
        if (lo < P && d.peer_uniq[m.praw0 + lo] == id) from[lo] = lc.prev_end[pm.praw0 + q];
      }
  }
  for (uint32_t p = 0; p < P; p++) { same &= from[p] == to[p]; to_ge &= to[p] >= from[p]; to_gt |= to[p] > from[p]; if (from[p] > to[p]) from[p] = to[p]; }
  uint32_t lf_p[LCA_FMAX], lf_c[LCA_FMAX], rf_p[LCA_FMAX], rf_c[LCA_FMAX];
  uint32_t nl = lca_vv_frontiers(x, from, lf_p, lf_c, LCA_FMAX), nr = lca_vv_frontiers(x, to, rf_p, rf_c, LCA_FMAX);
  if (nl > LCA_FMAX || nr > LCA_FMAX) return;
  uint32_t ans_p[LCA_FMAX], ans_c[LCA_FMAX], n_ans = 0;
  uint32_t mode = DM_UNKNOWN;
  auto finish = [&]() {
    if (mode == DM_CHECKOUT && to_ge && to_gt) mode = DM_IMPORT;   // oplog.rs:610-615
    if (n_ans > LCA_MAXF) { out[0] = DM_UNKNOWN; return; }
    // sorted by (peer, counter): peers ascend with their index
    for (uint32_t i = 1; i < n_ans; i++)
      for (uint32_t j = i; j > 0 && (ans_p[j - 1] > ans_p[j] || (ans_p[j - 1] == ans_p[j] && ans_c[j - 1] > ans_c[j])); j--) {
        uint32_t t = ans_p[j]; ans_p[j] = ans_p[j - 1]; ans_p[j - 1] = t; t = ans_c[j]; ans_c[j] = ans_c[j - 1]; ans_c[j - 1] = t;
      }
    out[1] = n_ans;
    for (uint32_t i = 0; i < n_ans; i++) {
      uint64_t id = d.peer_uniq[m.praw0 + ans_p[i]];
      out[2 + 3 * i] = (uint32_t)id; out[3 + 3 * i] = (uint32_t)(id >> 32); out[4 + 3 * i] = ans_c[i];
    }
    out[0] = mode;
  };
  if (same) { mode = DM_LINEAR; for (uint32_t i = 0; i < nl; i++) { ans_p[i] = lf_p[i]; ans_c[i] = lf_c[i]; } n_ans = nl; finish(); return; }   // diff_calc.rs:150-152
  // ---- the fast exits of _find_common_ancestor_new (dag.rs:494-546)
  if (nr == 0) { mode = DM_CHECKOUT; finish(); return; }
  if (nl == 0) {
    mode = DM_IMPORT_GREATER;
    if (nr == 1) {
      uint32_t n = lca_node_of(x, rf_p[0], rf_c[0]);
      if (n == NONE) { out[0] = DM_UNKNOWN; return; }
      LcaNode nd = lca_node(x, n);
      bool broke = false;
      for (uint32_t guard = 0; nd.n_dep == 1 && guard <= N; guard++) {
        uint32_t nx = lca_node_of(x, d.dep_peer[nd.dep0], d.dep_ctr[nd.dep0]);
        if (nx == NONE) { broke = true; break; }
        nd = lca_node(x, nx);
      }
      if (!broke && nd.n_dep == 0) mode = DM_LINEAR;
    }
    finish();
    return;
  }
  if (nl == 1 && nr == 1 && lf_p[0] == rf_p[0]) {
    uint32_t ln = lca_node_of(x, lf_p[0], lf_c[0]), rn = lca_node_of(x, rf_p[0], rf_c[0]);
    if (ln == NONE || rn == NONE) { out[0] = DM_UNKNOWN; return; }
    LcaNode L = lca_node(x, ln), R = lca_node(x, rn);
    if (ln == rn) {
      if (lf_c[0] < rf_c[0]) { mode = DM_LINEAR; ans_p[0] = lf_p[0]; ans_c[0] = lf_c[0]; } else { mode = DM_CHECKOUT; ans_p[0] = rf_p[0]; ans_c[0] = rf_c[0]; }
      n_ans = 1; finish(); return;
    }
    if (L.n_dep == 1 && d.dep_peer[L.dep0] == R.peer && d.dep_ctr[L.dep0] >= R.ctr0 && d.dep_ctr[L.dep0] < R.ctr0 + R.len) {
      mode = DM_CHECKOUT; ans_p[0] = rf_p[0]; ans_c[0] = rf_c[0]; n_ans = 1; finish(); return;
    }
    if (R.n_dep == 1 && d.dep_peer[R.dep0] == L.peer && d.dep_ctr[R.dep0] >= L.ctr0 && d.dep_ctr[R.dep0] < L.ctr0 + L.len) {
      mode = DM_LINEAR; ans_p[0] = lf_p[0]; ans_c[0] = lf_c[0]; n_ans = 1; finish(); return;
    }
  }
  // ---- the general walk (dag.rs:548-765)
  uint32_t* sc = lc.scratch + lc.scratch_off[doc];
  LcaHeap h;
  h.a = (LcaSpan*)sc; h.n = 0; h.cap = 8 * N + 64; h.overflow = false;
  LcaSpan* stack = (LcaSpan*)(sc + 3 * h.cap);
  const uint32_t stack_cap = 4 * N + 64;
  uint32_t* visited = sc + 3 * h.cap + 3 * stack_cap;
  enum { NT_A = 0, NT_B = 1, NT_SHARED = 2 };
  bool is_linear = nl <= 1 && nr == 1, is_right_greater = true, unmatched = false, bad = false;
  for (uint32_t i = 0; i < nl; i++) { uint32_t n = lca_node_of(x, lf_p[i], lf_c[i]); if (n == NONE) bad = true; else lca_push(x, h, LcaSpan{n, lf_c[i] - lca_node(x, n).ctr0 + 1, NT_A}); }
  for (uint32_t i = 0; i < nr; i++) { uint32_t n = lca_node_of(x, rf_p[i], rf_c[i]); if (n == NONE) bad = true; else lca_push(x, h, LcaSpan{n, rf_c[i] - lca_node(x, n).ctr0 + 1, NT_B}); }
  LcaSpan deps[MAX_PEERS + 1];
  for (uint32_t guard = 0; h.n > 0 && !bad && !h.overflow && guard < (1u << 24); guard++) {
    LcaSpan node = lca_pop(x, h);
    LcaNode nd = lca_node(x, node.n);
    uint32_t node_last = nd.ctr0 + node.len - 1;
    while (h.n > 0) {
      const LcaSpan o = h.a[0];
      LcaNode on = lca_node(x, o.n);
      bool same_span = o.n == node.n && o.len == node.len;
      bool same_last = on.peer == nd.peer && on.ctr0 + o.len - 1 == node_last;
      if (!(same_span || same_last)) break;
      if (node.type != o.type) node.type = NT_SHARED;
      (void)lca_pop(x, h);
    }
    if (node.type == NT_SHARED) { if (n_ans < LCA_FMAX) { ans_p[n_ans] = nd.peer; ans_c[n_ans] = node_last; } n_ans++; continue; }
    if (h.n == 0) { unmatched = true; is_right_greater = false; break; }
    if (node.type == NT_A) is_right_greater = false;
    {
      const LcaSpan o = h.a[0];
      LcaNode on = lca_node(x, o.n);
      uint32_t o_last = on.ctr0 + o.len - 1, o_lam_last = on.lam + o.len - 1;
      if (on.peer == nd.peer && o_last >= nd.ctr0 && o_last <= node_last && node.type != o.type) {
        node.len = o_last - nd.ctr0 + 1;
        lca_push(x, h, node);
        continue;
      }
      if (node.len > 1) {
        if (o_lam_last >= nd.lam) { uint32_t a = o_lam_last - nd.lam + 1, b = node.len - 1; node.len = a < b ? a : b; }
        else node.len = 1;
        lca_push(x, h, node);
        continue;
      }
    }
    uint32_t n_deps = 0;
    if (lca_deps(x, nd, deps, n_deps, MAX_PEERS + 1, node.type)) {
      if (n_deps) { for (uint32_t i = 0; i < n_deps; i++) lca_push(x, h, deps[i]); is_linear = false; continue; }
    } else { unmatched = true; is_right_greater = false; continue; }
    unmatched = true;   // a root reached on one side only (dag.rs:727-735)
    is_right_greater = false;
  }
  if (bad || h.overflow || n_ans > LCA_FMAX) { out[0] = DM_UNKNOWN; return; }
  // ---- shrink_ancestor_frontiers (dag.rs:610-634): drop the ids that lie in the past of another one
  if (n_ans > 1) {
    // descending by the span order (lamport of the id, peer; a span up to an id of the same node: the shorter the greater)
    uint32_t ord[LCA_FMAX];
    for (uint32_t i = 0; i < n_ans; i++) ord[i] = i;
    auto key_less = [&](uint32_t a, uint32_t b) {
      uint32_t na = lca_node_of(x, ans_p[a], ans_c[a]), nb = lca_node_of(x, ans_p[b], ans_c[b]);
      LcaNode A = lca_node(x, na), B = lca_node(x, nb);
      return lca_less(x, LcaSpan{na, ans_c[a] - A.ctr0 + 1, 0}, LcaSpan{nb, ans_c[b] - B.ctr0 + 1, 0});
    };
    for (uint32_t i = 1; i < n_ans; i++) for (uint32_t j = i; j > 0 && key_less(ord[j - 1], ord[j]); j--) { uint32_t t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t; }   // greatest first
    uint32_t kp[LCA_FMAX], kc[LCA_FMAX], nk = 0;
    for (uint32_t oi = 0; oi < n_ans; oi++) {
      uint32_t tp = ans_p[ord[oi]], tc = ans_c[ord[oi]];
      uint32_t tn = lca_node_of(x, tp, tc);
      uint32_t t_lam = lca_node(x, tn).lam + (tc - lca_node(x, tn).ctr0);
      bool inside = false;
      for (uint32_t k = 0; k < nk && !inside; k++) {
        // contains_in_ancestors (dag.rs:647-679): is (tp, tc) in the causal past of the kept id k?
        for (uint32_t i = 0; i < N; i++) visited[i] = 0;
        uint32_t sp = 0;
        uint32_t fn = lca_node_of(x, kp[k], kc[k]);
        stack[sp++] = LcaSpan{fn, kc[k] - lca_node(x, fn).ctr0 + 1, 0};
        while (sp > 0 && !inside) {
          LcaSpan s = stack[--sp];
          LcaNode sn = lca_node(x, s.n);
          if (sn.peer == tp && sn.ctr0 <= tc && tc < sn.ctr0 + s.len) { inside = true; break; }
          if (sn.lam + s.len - 1 < t_lam) continue;
          if (visited[s.n]) continue;
          visited[s.n] = 1;
          uint32_t nd2 = 0;
          if (lca_deps(x, sn, deps, nd2, MAX_PEERS + 1, 0))
            for (uint32_t i = 0; i < nd2; i++) { if (sp < stack_cap) stack[sp++] = deps[i]; else bad = true; }
        }
      }
      if (!inside) { kp[nk] = tp; kc[nk] = tc; nk++; }
    }
    if (bad) { out[0] = DM_UNKNOWN; return; }
    for (uint32_t i = 0; i < nk; i++) { ans_p[i] = kp[i]; ans_c[i] = kc[i]; }
    n_ans = nk;
  }
  if (unmatched) n_ans = 0;   // (no trimmed history here: the replay base is the empty version, dag.rs:733-747)
  mode = is_right_greater ? (is_linear ? DM_LINEAR : DM_IMPORT_GREATER) : DM_CHECKOUT;
  finish();
}

}  // namespace lm
