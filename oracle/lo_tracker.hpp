// ORACLE — TEST INFRASTRUCTURE ONLY (see lo_codec.hpp header).
// CPU restatement of the Event-Graph-Walker tracker used for Text/List containers.
//
// Restates (reference file:line, relative to /root/reference/crates/loro-internal/src):
//   Tracker{applied_vv,current_vv,rope,id_to_cursor}   container/richtext/tracker.rs:25-30
//   insert / _insert_inner                              tracker.rs:88-160
//   delete / _delete                                    tracker.rs:193-252
//   checkout / forward                                  tracker.rs:354-546
//   CrdtRope::insert (Fugue integrate)                  container/richtext/tracker/crdt_rope.rs:63-247
//   CrdtRope::delete                                    crdt_rope.rs:256-335
//   CrdtRope::update (status toggles with span splits)  crdt_rope.rs:345-381
//   ActiveLenQueryPreferLeft/Right                      crdt_rope.rs:564-672
//   FugueSpan slice rule / Status                       container/richtext/fugue_span.rs:257-279,374-386
//   IdToCursor                                          container/richtext/tracker/id_to_cursor.rs
// The reference keeps spans in generic-btree leaves; the B-tree shape is not observable, so this
// restatement keeps spans in a doubly linked list indexed by blocks with cached active lengths
// (same queries, same split rules).  Spans are never merged (merging is unobservable).
#pragma once
#include <map>
#include <unordered_map>
#include <vector>
#include <cstdint>
#include "lo_codec.hpp"

namespace lo {

struct OptID {
  bool some = false;
  ID id{0, 0};
  bool operator==(const OptID& o) const { return some == o.some && (!some || id == o.id); }
  bool operator!=(const OptID& o) const { return !(*this == o); }
};

struct Block;
struct Span {
  ID id;
  int32_t len;
  bool future = false;
  int16_t del = 0;
  OptID ol, orr;
  uint32_t content;  // index of first element in the tracker's content arena
  Span *prev = nullptr, *next = nullptr;
  Block* blk = nullptr;
  bool active() const { return !future && del == 0; }
  int32_t alen() const { return active() ? len : 0; }
  bool contains(ID x) const { return x.peer == id.peer && x.counter >= id.counter && x.counter < id.counter + len; }
};
struct Block {
  Span* first = nullptr;
  int32_t count = 0;
  int64_t active = 0;
  int32_t idx = 0;
};

struct Cursor {
  Span* s = nullptr;  // nullptr → empty rope
  int32_t off = 0;
};

typedef std::map<PeerID, Counter> VV;  // exclusive end counters (version.rs)

struct Tracker {
  static const int BLK = 64;
  std::vector<Block*> blocks;
  std::vector<Span*> all_spans;
  Span *head = nullptr, *tail = nullptr;
  int64_t total_active = 0;
  VV current_vv, applied_vv;

  struct Entry {
    bool is_del;
    int32_t len;
    Span* span;       // insert: the span whose ids start at the key
    ID target;        // delete: leftmost target id of this piece
    bool reversed;    // delete: op offset j deletes target + (len-1-j)
    bool is_move = false;   // MovableList move (Cursor::new_move, id_to_cursor.rs): an insert entry whose op ALSO deleted `target`
  };
  std::unordered_map<PeerID, std::map<Counter, Entry>> index;  // id_to_cursor

  ~Tracker() {
    for (auto s : all_spans) delete s;
    for (auto b : blocks) delete b;
  }

  // ------------------------------------------------------------ rope plumbing
  void reindex_blocks(size_t from) {
    for (size_t i = from; i < blocks.size(); i++) blocks[i]->idx = (int32_t)i;
  }
  void add_active(Span* s, int64_t d) {
    s->blk->active += d;
    total_active += d;
  }
  void split_block_if_needed(Block* b) {
    if (b->count <= 2 * BLK) return;
    Block* nb = new Block();
    Span* s = b->first;
    for (int i = 0; i < BLK; i++) s = s->next;
    nb->first = s;
    int moved = 0;
    int64_t act = 0;
    for (Span* t = s; t && t->blk == b; t = t->next) { t->blk = nb; moved++; act += t->alen(); }
    nb->count = moved;
    nb->active = act;
    b->count -= moved;
    b->active -= act;
    blocks.insert(blocks.begin() + b->idx + 1, nb);
    reindex_blocks((size_t)b->idx + 1);
  }
  // link `n` right after `after` (nullptr → at the very beginning)
  void link_after(Span* after, Span* n) {
    all_spans.push_back(n);
    if (!head) {
      head = tail = n;
      Block* b = new Block();
      b->first = n;
      b->count = 1;
      b->idx = 0;
      blocks.push_back(b);
      n->blk = b;
      b->active = 0;
      add_active(n, n->alen());
      return;
    }
    if (!after) {
      n->next = head;
      head->prev = n;
      head = n;
      n->blk = n->next->blk;
      n->blk->first = n;
    } else {
      n->prev = after;
      n->next = after->next;
      if (after->next) after->next->prev = n; else tail = n;
      after->next = n;
      n->blk = after->blk;
    }
    n->blk->count++;
    add_active(n, n->alen());
    split_block_if_needed(n->blk);
  }
  // split `s` at offset k (0<k<len); returns the right part (fugue_span.rs:257-279)
  Span* split(Span* s, int32_t k) {
    Span* r = new Span(*s);
    r->prev = r->next = nullptr;
    r->id.counter = s->id.counter + k;
    r->len = s->len - k;
    r->content = s->content + (uint32_t)k;
    r->ol.some = true;
    r->ol.id = ID{s->id.peer, s->id.counter + k - 1};
    // r->orr unchanged
    int64_t a_before = s->alen();
    s->len = k;
    add_active(s, (int64_t)s->alen() - a_before);
    link_after(s, r);
    index[r->id.peer][r->id.counter] = Entry{false, r->len, r, ID{0, 0}, false};
    auto it = index[s->id.peer].find(s->id.counter);
    if (it != index[s->id.peer].end()) it->second.len = k;
    return r;
  }
  int cmp_pos(Span* a, Span* b) {
    if (a == b) return 0;
    if (a->blk != b->blk) return a->blk->idx < b->blk->idx ? -1 : 1;
    for (Span* t = a->blk->first; t && t->blk == a->blk; t = t->next) {
      if (t == a) return -1;
      if (t == b) return 1;
    }
    return 0;
  }
  // ActiveLenQueryPreferLeft (crdt_rope.rs:564-615)
  Cursor find_prefer_left(int64_t pos) {
    if (!head) return Cursor{};
    int64_t left = pos;
    for (Block* b : blocks) {
      if (left <= b->active) {
        for (Span* s = b->first; s && s->blk == b; s = s->next) {
          int32_t a = s->alen();
          if (left <= a) return s->active() ? Cursor{s, (int32_t)left} : Cursor{s, 0};
          left -= a;
        }
      }
      left -= b->active;
    }
    return Cursor{tail, tail->len};  // beyond the end: missing → clamp to the end
  }
  // ActiveLenQueryPreferRight (crdt_rope.rs:622-672)
  Cursor find_prefer_right(int64_t pos) {
    if (!head) return Cursor{};
    int64_t left = pos;
    for (Block* b : blocks) {
      if (left < b->active) {
        for (Span* s = b->first; s && s->blk == b; s = s->next) {
          int32_t a = s->alen();
          if (left < a) return Cursor{s, (int32_t)left};
          left -= a;
        }
      }
      left -= b->active;
    }
    return Cursor{nullptr, 0};
  }
  Span* lookup_insert(ID id) {
    auto pit = index.find(id.peer);
    if (pit == index.end()) return nullptr;
    auto it = pit->second.upper_bound(id.counter);
    if (it == pit->second.begin()) return nullptr;
    --it;
    if (it->second.is_del) return nullptr;
    Span* s = it->second.span;
    return s->contains(id) ? s : nullptr;
  }

  // ------------------------------------------------------------ insert (crdt_rope.rs:63-247)
  void insert(ID op_id, int64_t pos, int32_t len, uint32_t content) {
    Span* n = new Span();
    n->id = op_id;
    n->len = len;
    n->content = content;
    if (!head) {
      link_after(nullptr, n);
    } else {
      Cursor start = find_prefer_left(pos);
      // origin_left: the active element at pos-1 (crdt_rope.rs:89-108)
      OptID origin_left;
      if (start.off == 0) {
        Span* left = start.s->prev;
        if (left) { origin_left.some = true; origin_left.id = ID{left->id.peer, left->id.counter + left->len - 1}; }
      } else {
        origin_left.some = true;
        origin_left.id = ID{start.s->id.peer, start.s->id.counter + start.off - 1};
      }
      // origin_right: first non-future element at/after the cursor (crdt_rope.rs:110-149)
      OptID origin_right;
      Span* parent_right = nullptr;
      std::vector<Span*> in_between;
      {
        bool first = true;
        for (Span* it = start.s; it; it = it->next, first = false) {
          int32_t off = first ? start.off : 0;
          if (first && off >= it->len) continue;
          if (!it->future) {
            origin_right.some = true;
            origin_right.id = ID{it->id.peer, it->id.counter + off};
            if (off > 0) parent_right = it;
            else if (it->ol == origin_left) parent_right = it;
            break;
          }
          in_between.push_back(it);
        }
      }
      n->ol = origin_left;
      n->orr = origin_right;
      // insert position: cursor, moved right past concurrent siblings per Fugue (crdt_rope.rs:156-237)
      Span* ins_after;   // insert right after this span (at ins_off inside it), nullptr → very beginning
      int32_t ins_off;
      if (start.off == 0) { ins_after = start.s->prev; ins_off = ins_after ? ins_after->len : 0; }
      else { ins_after = start.s; ins_off = start.off; }
      if (!in_between.empty()) {
        bool scanning = false;
        std::vector<std::pair<ID, int32_t>> visited;
        for (Span* other : in_between) {
          if (other->ol != origin_left) {
            bool in_visited = false;
            if (other->ol.some)
              for (auto& v : visited)
                if (v.first.peer == other->ol.id.peer && other->ol.id.counter >= v.first.counter &&
                    other->ol.id.counter < v.first.counter + v.second) { in_visited = true; break; }
            if (!in_visited) break;
          }
          visited.emplace_back(other->id, other->len);
          if (other->ol == origin_left) {
            if (other->orr == origin_right) {
              if (other->id.peer > op_id.peer) break;
              scanning = false;
            } else {
              Span* other_pr = nullptr;
              if (other->orr.some) {
                Span* e = lookup_insert(other->orr.id);
                if (!e) fail(ST_INTERNAL, "origin_right not found");
                if (e->id != other->orr.id) e = split(e, other->orr.id.counter - e->id.counter);
                if (e->ol == origin_left) other_pr = e;
              }
              // cmp_pos(other_parent_right, parent_right): None sorts after Some (crdt_rope.rs:453-466)
              int c;
              if (other_pr && parent_right) c = cmp_pos(other_pr, parent_right);
              else if (other_pr) c = -1;
              else if (parent_right) c = 1;
              else c = 0;
              if (c < 0) scanning = true;
              else if (c == 0 && other->id.peer > op_id.peer) break;
              else scanning = false;
            }
          }
          if (!scanning) { ins_after = other; ins_off = other->len; }
        }
      }
      if (ins_after && ins_off < ins_after->len) split(ins_after, ins_off);
      link_after(ins_after, n);
    }
    index[op_id.peer][op_id.counter] = Entry{false, len, n, ID{0, 0}, false};
    bump(current_vv, op_id.peer, op_id.counter + len);
    bump(applied_vv, op_id.peer, op_id.counter + len);
  }
  static void bump(VV& vv, PeerID p, Counter end) {
    auto it = vv.find(p);
    if (it == vv.end()) vv[p] = end;
    else if (it->second < end) it->second = end;
  }
  int64_t active_len() const { return total_active; }

  // ------------------------------------------------------------ delete (tracker.rs:193-252, crdt_rope.rs:256-335)
  void del(ID op_id, ID /*target_start*/, int64_t pos, int32_t len, bool reversed) {
    std::vector<std::pair<ID, int32_t>> pieces;  // left-to-right deleted id spans
    int32_t remaining = len;
    Cursor c = find_prefer_right(pos);
    Span* s = c.s;
    int32_t off = c.off;
    while (remaining > 0) {
      if (!s) fail(ST_DATA_CORRUPTION, "delete beyond the end");
      if (!s->active()) { s = s->next; off = 0; continue; }
      if (off > 0) { s = split(s, off); off = 0; }
      if (s->len > remaining) split(s, remaining);
      add_active(s, -(int64_t)s->len);
      s->del += 1;
      pieces.emplace_back(s->id, s->len);
      remaining -= s->len;
      s = s->next;
    }
    if (reversed) std::reverse(pieces.begin(), pieces.end());
    Counter cur = op_id.counter;
    for (auto& p : pieces) {
      index[op_id.peer][cur] = Entry{true, p.second, nullptr, p.first, reversed};
      cur += p.second;
    }
    bump(current_vv, op_id.peer, op_id.counter + len);
    bump(applied_vv, op_id.peer, op_id.counter + len);
  }

  // ------------------------------------------------------------ move (tracker.rs:289-347: MovableList)
  // The element's current list item — the active item at `from` — is deleted and a new item with the op's id is inserted
  // at `to` (evaluated after the deletion); retreating / forwarding the op undoes / redoes both halves.
  void move_item(ID op_id, int64_t from, int64_t to, uint32_t content) {
    if (from < 0 || from >= total_active) fail(ST_DATA_CORRUPTION, "move source beyond the end");
    Cursor c = find_prefer_right(from);
    Span* s = c.s;
    int32_t off = c.off;
    while (s && (!s->active() || off >= s->len)) { s = s->next; off = 0; }
    if (!s) fail(ST_DATA_CORRUPTION, "move source beyond the end");
    if (off > 0) s = split(s, off);
    if (s->len > 1) split(s, 1);
    add_active(s, -1);
    s->del += 1;
    ID target = s->id;
    if (to < 0 || to > total_active) fail(ST_DATA_CORRUPTION, "move destination beyond the end");
    insert(op_id, to, 1, content);
    Entry& e = index[op_id.peer][op_id.counter];
    e.is_move = true;
    e.target = target;
  }

  // ------------------------------------------------------------ checkout (tracker.rs:354-546)
  // apply a status change to the inserted ids [c0,c1) of `peer` (crdt_rope.rs:345-381)
  void update_ids(PeerID peer, Counter c0, Counter c1, int set_future /* -1 none, 0, 1 */, int del_diff) {
    auto pit = index.find(peer);
    if (pit == index.end()) return;
    Counter c = c0;
    while (c < c1) {
      auto& m = pit->second;
      auto it = m.upper_bound(c);
      if (it == m.begin()) {
        if (it == m.end()) return;
        c = it->first;
        continue;
      }
      auto pv = std::prev(it);
      if (pv->second.is_del || pv->first + pv->second.len <= c) {
        if (it == m.end()) return;
        c = it->first;
        continue;
      }
      Span* s = pv->second.span;
      if (s->id.counter < c) s = split(s, c - s->id.counter);
      if (s->id.counter + s->len > c1) split(s, c1 - s->id.counter);
      int64_t before = s->alen();
      if (set_future >= 0) s->future = set_future != 0;
      s->del = (int16_t)(s->del + del_diff);
      add_active(s, (int64_t)s->alen() - before);
      c = s->id.counter + s->len;
    }
  }
  // retreat (dir=-1) or forward (dir=+1) the ops with ids [c0,c1) of `peer`
  void move_ops(PeerID peer, Counter c0, Counter c1, int dir) {
    auto pit = index.find(peer);
    if (pit == index.end()) return;
    // collect first: update_ids may split spans and insert index entries for the same peer
    struct Item { bool is_del; Counter a, b; ID target; bool reversed; int32_t len; };
    std::vector<Item> items;
    auto& m = pit->second;
    auto it = m.upper_bound(c0);
    if (it != m.begin()) --it;
    for (; it != m.end() && it->first < c1; ++it) {
      Counter k = it->first, e = k + it->second.len;
      Counter a = std::max(k, c0), b = std::min(e, c1);
      if (a >= b) continue;
      items.push_back(Item{it->second.is_del, (Counter)(a - k), (Counter)(b - k), it->second.target, it->second.reversed,
                           it->second.len});
      if (!it->second.is_del) {
        items.back().target = ID{peer, k};
        if (it->second.is_move) items.push_back(Item{true, 0, 1, it->second.target, false, 1});   // the move's deleted source item
      }
    }
    for (auto& x : items) {
      if (!x.is_del) {
        update_ids(peer, x.target.counter + x.a, x.target.counter + x.b, dir < 0 ? 1 : 0, 0);
      } else {
        Counter t0, t1;
        if (!x.reversed) { t0 = x.target.counter + x.a; t1 = x.target.counter + x.b; }
        else { t0 = x.target.counter + (x.len - x.b); t1 = x.target.counter + (x.len - x.a); }
        update_ids(x.target.peer, t0, t1, -1, dir);
      }
    }
  }
  void checkout(const VV& vv) {
    std::vector<std::pair<PeerID, std::pair<Counter, Counter>>> retreat, forward;
    for (auto& kv : current_vv) {
      auto it = vv.find(kv.first);
      Counter tgt = it == vv.end() ? 0 : it->second;
      if (kv.second > tgt) retreat.push_back({kv.first, {tgt, kv.second}});
    }
    for (auto& kv : vv) {
      auto it = current_vv.find(kv.first);
      Counter cur = it == current_vv.end() ? 0 : it->second;
      if (kv.second > cur) forward.push_back({kv.first, {cur, kv.second}});
    }
    for (auto& r : retreat) move_ops(r.first, r.second.first, r.second.second, -1);
    for (auto& f : forward) move_ops(f.first, f.second.first, f.second.second, +1);
    current_vv = vv;
  }
};

}  // namespace lo
