// Symbolic analysis of the pose-graph SPA solver: nested dissection, supernodes, assembly tree (see spa_symbolic.hpp).
//
// Ordering = nested dissection on the graph itself.  A subset is cut at a level of the breadth-first level structure
// rooted at a pseudo-peripheral vertex (George): every edge of the subset joins equal or adjacent levels, so the edges
// between level l - 1 and level l are an edge separator, and a MINIMUM VERTEX COVER of that bipartite edge set (Koenig:
// from a maximum matching) is a vertex separator -- never larger than either boundary, on a pose graph typically 30-40 %
// smaller than the whole level the first version of this solver removed.  The pose graph of a mapper that drives the same
// aisles lap after lap is a thick chain (every cut has about the same width), so the separator size enters the
// factorisation with its third power at every level of the tree.  Purely topological: independent of how far the pose
// estimates have drifted.  Leaves and separators become the supernodes (= fronts) of the multifrontal factorisation.
#include "spa_symbolic.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <string>
#include <thread>

#include "../../include/karto_hip.h"

namespace kh
{
void set_error(const std::string & s);

namespace
{

struct NdContext
{
  const std::vector<int32_t> * adj_ptr = nullptr;
  const std::vector<int32_t> * adj_idx = nullptr;
  struct Vertex {int32_t tag, dist;}; // subset membership stamp, BFS level: one cache line brings both
  std::vector<Vertex> vs;
  std::vector<int32_t> loc;           // scratch: local index of a vertex inside the bipartite cut graph
  std::atomic<int32_t> stamp{0};      // sibling subsets are dissected concurrently on disjoint vertices
  SymbolicOptions opt;
};
using SupernodeList = std::vector<std::vector<int32_t>>;

// What a subset inherits from the split that made it: a vertex at one of its ends, and whether the subset's vertices already
// ARE the breadth-first order from that vertex with their levels in ctx.dist (the near side of a cut: see nd_split).
struct NdHint {int32_t vertex = -1; bool levels_ready = false;};

// Work vectors of one dissecting thread, kept across the ~800 splits of an analysis: allocated afresh per split they were
// 26 000 vector reallocations per analysis of the 10 000-node graph -- half of its time.
struct NdScratch
{
  struct Cand {int32_t est, l;};
  std::vector<Cand> cands;
  std::vector<int32_t> order, lstart, cover;
  std::vector<int32_t> left, right, eptr, eidx, match_l, match_r, seen, stack_v, stack_e, work;
  std::vector<uint8_t> zl, zr;
};

// BFS inside the subset stamped `from`; returns the visit order (sorted by level; levels in the vertices' dist).  The vertices it
// reaches are re-stamped `to`: the subset's stamp from here on (no pass to un-mark them).
void nd_bfs(NdContext & ctx, int32_t start, int32_t from, int32_t to, std::vector<int32_t> & order)
{
  const std::vector<int32_t> & ap = *ctx.adj_ptr, & ai = *ctx.adj_idx;
  NdContext::Vertex * const vs = ctx.vs.data();
  order.clear();
  order.push_back(start);
  vs[start].dist = 0;
  vs[start].tag = to;
  for (size_t h = 0; h < order.size(); ++h) {
    const int32_t v = order[h];
    const int32_t dv = vs[v].dist + 1;
    for (int32_t k = ap[v]; k < ap[v + 1]; ++k) {
      const int32_t w = ai[k];
      if (vs[w].tag == from) {vs[w].tag = to; vs[w].dist = dv; order.push_back(w);}
    }
  }
}

// Minimum vertex cover of the edges between level l - 1 (left) and level l (right) of the level structure in `order`
// (level q = order[lstart[q] .. lstart[q + 1])).  Kuhn's augmenting paths (the sides hold tens of vertices), then Koenig's
// construction: with Z = the vertices reachable from the unmatched left vertices by alternating paths, the cover is
// (left \ Z) + (right & Z).
void cut_vertex_cover(NdContext & ctx, NdScratch & sc, const std::vector<int32_t> & order, const std::vector<int32_t> & lstart, int32_t l,
  int32_t st, std::vector<int32_t> & cover)
{
  const std::vector<int32_t> & ap = *ctx.adj_ptr, & ai = *ctx.adj_idx;
  cover.clear();
  std::vector<int32_t> & left = sc.left, & right = sc.right;     // vertex ids
  std::vector<int32_t> & eptr = sc.eptr, & eidx = sc.eidx;       // left local -> right locals
  left.clear(); right.clear(); eidx.clear(); eptr.assign(1, 0);
  for (int32_t q = lstart[l]; q < lstart[l + 1]; ++q) {ctx.loc[order[q]] = -1;}
  for (int32_t q = lstart[l - 1]; q < lstart[l]; ++q) {
    const int32_t v = order[q];
    const size_t before = eidx.size();
    for (int32_t k = ap[v]; k < ap[v + 1]; ++k) {
      const int32_t w = ai[k];
      if (ctx.vs[w].tag != st || ctx.vs[w].dist != l) {continue;}
      if (ctx.loc[w] < 0) {ctx.loc[w] = static_cast<int32_t>(right.size()); right.push_back(w);}
      eidx.push_back(ctx.loc[w]);
    }
    if (eidx.size() > before) {left.push_back(v); eptr.push_back(static_cast<int32_t>(eidx.size()));}
  }
  const int32_t nl = static_cast<int32_t>(left.size()), nr = static_cast<int32_t>(right.size());
  std::vector<int32_t> & match_l = sc.match_l, & match_r = sc.match_r, & seen = sc.seen;
  match_l.assign(nl, -1); match_r.assign(nr, -1); seen.assign(nr, -1);
  // iterative augmenting-path search from left vertex `root`
  std::vector<int32_t> & stack_v = sc.stack_v, & stack_e = sc.stack_e;
  for (int32_t root = 0; root < nl; ++root) {
    stack_v.assign(1, root); stack_e.assign(1, eptr[root]);
    bool found = false;
    while (!stack_v.empty() && !found) {
      const int32_t u = stack_v.back();
      int32_t & e = stack_e.back();
      if (e >= eptr[u + 1]) {stack_v.pop_back(); stack_e.pop_back(); continue;}
      const int32_t r = eidx[e++];
      if (seen[r] == root) {continue;}
      seen[r] = root;
      if (match_r[r] < 0) {
        // flip the path: the stack holds the left vertices, each one's last taken edge is stack_e - 1
        int32_t rr = r;
        for (int32_t d = static_cast<int32_t>(stack_v.size()) - 1; d >= 0; --d) {
          const int32_t lu = stack_v[d];
          const int32_t prev = match_l[lu];
          match_l[lu] = rr; match_r[rr] = lu;
          rr = prev;
        }
        found = true;
      } else {
        stack_v.push_back(match_r[r]); stack_e.push_back(eptr[match_r[r]]);
      }
    }
  }
  std::vector<uint8_t> & zl = sc.zl, & zr = sc.zr;
  zl.assign(nl, 0); zr.assign(nr, 0);
  std::vector<int32_t> & work = sc.work;
  work.clear();
  for (int32_t u = 0; u < nl; ++u) {if (match_l[u] < 0) {zl[u] = 1; work.push_back(u);}}
  while (!work.empty()) {
    const int32_t u = work.back(); work.pop_back();
    for (int32_t e = eptr[u]; e < eptr[u + 1]; ++e) {
      const int32_t r = eidx[e];
      if (zr[r]) {continue;}
      zr[r] = 1;
      const int32_t u2 = match_r[r];
      if (u2 >= 0 && !zl[u2]) {zl[u2] = 1; work.push_back(u2);}
    }
  }
  for (int32_t u = 0; u < nl; ++u) {if (!zl[u]) {cover.push_back(left[u]);}}
  for (int32_t r = 0; r < nr; ++r) {if (zr[r]) {cover.push_back(right[r]);}}
}

// One dissection step: the subset `nodes` is either a leaf (returns false; `nodes` sorted) or split into A, B and the
// separator `cover` (returns true).  `hint`: a vertex of the subset known to lie at one of its ends (the root or the farthest
// vertex of the parent's level structure, whichever side the subset came from), or -1: with a hint ONE breadth-first search
// gives connectivity and the level structure.  A disconnected subset is split into one component and the rest, no separator.
// Sibling subsets are dissected concurrently: they touch disjoint entries of the context's per-vertex arrays.
bool nd_split(NdContext & ctx, NdScratch & sc, std::vector<int32_t> & nodes, NdHint hint, std::vector<int32_t> & A, std::vector<int32_t> & B,
  std::vector<int32_t> & best_cover, NdHint & hint_a, NdHint & hint_b)
{
  A.clear(); B.clear(); best_cover.clear(); hint_a = NdHint(); hint_b = NdHint();
  auto as_leaf = [&]() {std::sort(nodes.begin(), nodes.end()); return false;};
  if (static_cast<int32_t>(nodes.size()) <= ctx.opt.leaf_nodes) {return as_leaf();}
  int32_t st = ++ctx.stamp;             // the subset's current stamp: a search re-stamps what it reaches
  for (int32_t v : nodes) {ctx.vs[v].tag = st;}
  std::vector<int32_t> & order = sc.order;
  order.clear();
  order.reserve(nodes.size());
  const bool hinted = hint.vertex >= 0 && ctx.vs[hint.vertex].tag == st;
  if (hinted && hint.levels_ready && nodes[0] == hint.vertex) {
    // the near side of the parent's cut: its vertices came out of the parent's breadth-first order (from this very root) in
    // that order, and a search from the root inside the subset would find them in the same order at the same levels -- every
    // vertex below the cut is discovered from the level before it, all of which stayed in the subset.  No search: half of the
    // dissection's breadth-first work.
    order.assign(nodes.begin(), nodes.end());
  } else {
    const int32_t unreached = st;
    st = ++ctx.stamp;
    nd_bfs(ctx, hinted ? hint.vertex : nodes[0], unreached, st, order);
    if (order.size() < nodes.size()) {
      // disconnected subset: split off this component (independent subtrees)
      B.reserve(nodes.size() - order.size());
      for (int32_t v : nodes) {if (ctx.vs[v].tag == unreached) {B.push_back(v);}}
      A.assign(order.begin(), order.end());
      return true;
    }
  }
  // pseudo-peripheral start: restart the BFS from the farthest vertex (a hinted start already is such a vertex)
  if (!hinted) {
    const int32_t far = order.back();
    const int32_t before = st;
    st = ++ctx.stamp;
    nd_bfs(ctx, far, before, st, order);
  }
  const int32_t depth_bfs = ctx.vs[order.back()].dist;
  if (depth_bfs < 2) {return as_leaf();}      // clique-like: no level can separate anything
  std::vector<int32_t> & lstart = sc.lstart;
  lstart.assign(depth_bfs + 2, 0);
  for (int32_t v : order) {lstart[ctx.vs[v].dist + 1]++;}
  for (int32_t l = 0; l <= depth_bfs; ++l) {lstart[l + 1] += lstart[l];}
  // candidate cuts "between level l - 1 and level l": those that leave balance_lo .. balance_hi of the vertices on the near
  // side, cheapest boundary estimate first; if the level structure is too coarse for that, the cut closest to the median
  const double total = static_cast<double>(order.size());
  using Cand = NdScratch::Cand;
  std::vector<Cand> & cands = sc.cands;
  cands.clear();
  int32_t fallback = 1; double fallback_dist = 1e300;
  for (int32_t l = 1; l <= depth_bfs; ++l) {
    const double frac = lstart[l] / total;         // vertices at distance < l
    const int32_t below = lstart[l] - lstart[l - 1], at = lstart[l + 1] - lstart[l];
    if (std::fabs(frac - 0.5) < fallback_dist) {fallback_dist = std::fabs(frac - 0.5); fallback = l;}
    if (frac >= ctx.opt.balance_lo && frac <= ctx.opt.balance_hi) {cands.push_back({std::min(below, at), l});}
  }
  if (cands.empty()) {cands.push_back({0, fallback});}
  // (by estimate, ties in level order: the candidates were pushed in level order, so this is the stable sort by estimate)
  std::sort(cands.begin(), cands.end(), [](const Cand & a, const Cand & b) {return a.est != b.est ? a.est < b.est : a.l < b.l;});
  if (static_cast<int32_t>(cands.size()) > ctx.opt.separator_candidates) {cands.resize(ctx.opt.separator_candidates);}
  std::vector<int32_t> & cover = sc.cover;
  int32_t best_l = -1;
  for (const Cand & c : cands) {
    if (best_l >= 0 && c.est >= static_cast<int32_t>(best_cover.size())) {
      // the cover of a cut is at most its smaller boundary, but equal to it only in the worst case: a boundary that is
      // already no smaller than the best cover cannot win by much, and the candidates are sorted by that estimate
      if (c.est > static_cast<int32_t>(best_cover.size()) + static_cast<int32_t>(best_cover.size()) / 2) {break;}
    }
    cut_vertex_cover(ctx, sc, order, lstart, c.l, st, cover);
    if (cover.empty()) {continue;}
    if (best_l < 0 || cover.size() < best_cover.size()) {best_cover.swap(cover); best_l = c.l;}
  }
  if (best_l < 0) {best_cover.clear(); return as_leaf();}
  for (int32_t v : best_cover) {ctx.vs[v].tag = 0;}              // out of the subset
  A.reserve(lstart[best_l]); B.reserve(order.size() - lstart[best_l]);
  for (int32_t v : order) {
    if (ctx.vs[v].tag != st) {continue;}
    if (ctx.vs[v].dist < best_l) {A.push_back(v);} else {B.push_back(v);}
  }
  if (A.empty() || B.empty()) {
    for (int32_t v : best_cover) {ctx.vs[v].tag = st;}
    A.clear(); B.clear(); best_cover.clear();
    return as_leaf();
  }
  // the near side holds the root of this level structure (and is listed in its order), the far side its last vertex
  hint_a.vertex = order.front(); hint_a.levels_ready = true; hint_b.vertex = order.back();
  std::sort(best_cover.begin(), best_cover.end());
  return true;
}

// A whole subtree, serially: the supernodes of `nodes` in elimination order (A's, B's, then the separator).  The recursion is
// as deep as the dissection tree: a few dozen frames.
void dissect_subtree(NdContext & ctx, NdScratch & sc, std::vector<int32_t> & nodes, NdHint hint, std::vector<std::vector<int32_t>> & out)
{
  std::vector<int32_t> A, B, cover;
  NdHint ha, hb;
  if (!nd_split(ctx, sc, nodes, hint, A, B, cover, ha, hb)) {
    if (!nodes.empty()) {out.push_back(std::move(nodes));}
    return;
  }
  std::vector<int32_t>().swap(nodes);
  dissect_subtree(ctx, sc, A, ha, out);
  dissect_subtree(ctx, sc, B, hb, out);
  if (!cover.empty()) {out.push_back(std::move(cover));}
}

}  // namespace

int nested_dissection(int32_t n_free, const std::vector<int32_t> & adj_ptr, const std::vector<int32_t> & adj_idx,
  const SymbolicOptions & opt, std::vector<std::vector<int32_t>> & supernodes)
{
  NdContext ctx;
  ctx.adj_ptr = &adj_ptr;
  ctx.adj_idx = &adj_idx;
  ctx.opt = opt;
  ctx.vs.assign(n_free, NdContext::Vertex{0, 0});
  ctx.loc.assign(n_free, -1);
  // The top of the dissection tree is grown level by level, serially, until a level holds kWholeSubtrees subsets; each of
  // those is then dissected to its leaves by one task of ONE parallel loop when the caller supplies one (opt.parallel_for:
  // the library's persistent host pool; the subsets are disjoint vertex sets, so the tasks share nothing but the read-only
  // adjacency).  One loop, not one per level: waking the pool's sleeping threads costs about 0.2 ms a time, and a loop per
  // level (fifteen per analysis, ever smaller subsets) made the mapper's analyses slower than serial ones (+2.7 ms per loop
  // closure of the 50 000-scan replay); a thread per split, as through round 3, paid thread creation instead.  The three
  // serial levels are a fifth of the work.  The elimination order is read off the finished tree in post-order: A's
  // supernodes, B's, then the separator.
  constexpr size_t kWholeSubtrees = 8;
  struct Task {std::vector<int32_t> nodes, cover; NdHint hint; int32_t a = -1, b = -1; bool split = false, whole = false;
    std::vector<std::vector<int32_t>> done;};
  std::vector<Task> tree(1);
  tree[0].nodes.resize(n_free);
  for (int32_t i = 0; i < n_free; ++i) {tree[0].nodes[i] = i;}
  size_t level_begin = 0;
  NdScratch top_scratch;
  while (level_begin < tree.size()) {
    const size_t level_end = tree.size();
    const size_t count = level_end - level_begin;
    if (!opt.parallel_for || count >= kWholeSubtrees) {
      auto whole = [&](size_t i) {
        Task & t = tree[level_begin + i];
        t.whole = true;
        NdScratch sc;
        dissect_subtree(ctx, sc, t.nodes, t.hint, t.done);
      };
      if (opt.parallel_for && count > 1) {opt.parallel_for(count, whole);} else {for (size_t i = 0; i < count; ++i) {whole(i);}}
      break;
    }
    std::vector<std::vector<int32_t>> As(count), Bs(count);
    std::vector<NdHint> ha(count), hb(count);
    auto work = [&](size_t i) {
      Task & t = tree[level_begin + i];
      t.split = nd_split(ctx, top_scratch, t.nodes, t.hint, As[i], Bs[i], t.cover, ha[i], hb[i]);
    };
    for (size_t i = 0; i < count; ++i) {work(i);}
    for (size_t i = 0; i < count; ++i) {
      if (!tree[level_begin + i].split) {continue;}
      Task ta, tb;
      ta.nodes.swap(As[i]); ta.hint = ha[i];
      tb.nodes.swap(Bs[i]); tb.hint = hb[i];
      tree[level_begin + i].a = static_cast<int32_t>(tree.size()); tree.push_back(std::move(ta));
      tree[level_begin + i].b = static_cast<int32_t>(tree.size()); tree.push_back(std::move(tb));
      std::vector<int32_t>().swap(tree[level_begin + i].nodes);
    }
    level_begin = level_end;
  }
  supernodes.clear();
  // post-order without recursion: (task, state) stack
  std::vector<std::pair<int32_t, int>> stack(1, {0, 0});
  while (!stack.empty()) {
    const int32_t k = stack.back().first;
    int & state = stack.back().second;
    Task & t = tree[k];
    if (t.whole) {
      for (auto & sn : t.done) {supernodes.push_back(std::move(sn));}
      stack.pop_back();
    } else if (!t.split) {
      if (!t.nodes.empty()) {supernodes.push_back(std::move(t.nodes));}
      stack.pop_back();
    } else if (state == 0) {
      state = 1; stack.push_back({t.a, 0});
    } else if (state == 1) {
      state = 2; stack.push_back({t.b, 0});
    } else {
      if (!t.cover.empty()) {supernodes.push_back(std::move(t.cover));}
      stack.pop_back();
    }
  }
  return KH_OK;
}

int build_symbolic(Symbolic & sym, int32_t n_free, const std::vector<int32_t> & adj_ptr, const std::vector<int32_t> & adj_idx,
  const SymbolicOptions & opt)
{
  SupernodeList dissected;
  const auto t_nd0 = std::chrono::steady_clock::now();
  int rc = nested_dissection(n_free, adj_ptr, adj_idx, opt, dissected);
  if (rc) {return rc;}
  if (std::getenv("KH_SPA_DEBUG")) {
    std::fprintf(stderr, "[kh_spa] symbolic: nested dissection %.2f ms\n",
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_nd0).count());
  }
  return build_structure(sym, n_free, adj_ptr, adj_idx, opt, dissected);
}

int build_structure(Symbolic & sym, int32_t n_free, const std::vector<int32_t> & adj_ptr, const std::vector<int32_t> & adj_idx,
  const SymbolicOptions & opt, std::vector<std::vector<int32_t>> & dissected)
{
  sym = Symbolic();
  sym.n_free = n_free;
  const auto t_nd1 = std::chrono::steady_clock::now();
  // a front's pivot block is factored inside one workgroup's LDS: larger supernodes become a chain of fronts (each part
  // the only child of the next; same columns, same fill)
  SupernodeList supernodes;
  supernodes.reserve(dissected.size());
  const int32_t cap = std::max(1, opt.max_pivot_nodes);
  for (auto & sn : dissected) {
    const int32_t n = static_cast<int32_t>(sn.size());
    if (n <= cap) {supernodes.push_back(std::move(sn)); continue;}
    const int32_t parts = (n + cap - 1) / cap, step = (n + parts - 1) / parts;
    for (int32_t b = 0; b < n; b += step) {supernodes.emplace_back(sn.begin() + b, sn.begin() + std::min(n, b + step));}
  }
  const int32_t K = static_cast<int32_t>(supernodes.size());
  sym.n_fronts = K;
  sym.elim_of_free.assign(n_free, -1);
  sym.free_of_elim.assign(n_free, -1);
  sym.sn_of_elim.assign(n_free, -1);
  // `o` = index of a supernode in elimination order, `k` = its front id: fronts are numbered level by level (largest first
  // inside a level), so that a level is a contiguous id range and its descriptors are contiguous in memory
  std::vector<int32_t> first_o(K + 1, 0);
  int32_t pos = 0;
  for (int32_t o = 0; o < K; ++o) {
    first_o[o] = pos;
    for (int32_t v : supernodes[o]) {
      if (v < 0 || v >= n_free || sym.elim_of_free[v] >= 0) {set_error("nested dissection produced an invalid ordering"); return KH_ERR_SOLVER;}
      sym.elim_of_free[v] = pos; sym.free_of_elim[pos] = v; sym.sn_of_elim[pos] = o; ++pos;     // o for now, k below
    }
  }
  first_o[K] = pos;
  if (pos != n_free) {set_error("nested dissection lost nodes"); return KH_ERR_SOLVER;}

  // struct rows, parents, children (in elimination order: children before parents)
  std::vector<std::vector<int32_t>> rows(K), children(K);
  std::vector<int32_t> stamp(n_free, -1), parent_o(K, -1), level_o(K, 0), m_o(K, 0);
  int32_t max_level = 0;
  std::vector<int32_t> r;                      // collected here, stored at its final size (one allocation per front)
  r.reserve(4096);
  for (int32_t o = 0; o < K; ++o) {
    const int32_t end = first_o[o + 1];
    r.clear();
    for (int32_t e = first_o[o]; e < end; ++e) {
      const int32_t v = sym.free_of_elim[e];
      for (int32_t q = adj_ptr[v]; q < adj_ptr[v + 1]; ++q) {
        const int32_t ew = sym.elim_of_free[adj_idx[q]];
        if (ew >= end && stamp[ew] != o) {stamp[ew] = o; r.push_back(ew);}
      }
    }
    for (int32_t c : children[o]) {
      for (int32_t ew : rows[c]) {
        if (ew >= end && stamp[ew] != o) {stamp[ew] = o; r.push_back(ew);}
      }
      level_o[o] = std::max(level_o[o], level_o[c] + 1);
    }
    std::sort(r.begin(), r.end());
    rows[o].assign(r.begin(), r.end());
    if (!r.empty()) {
      parent_o[o] = sym.sn_of_elim[r[0]];
      children[parent_o[o]].push_back(o);
    }
    m_o[o] = 3 * (end - first_o[o] + static_cast<int32_t>(r.size()));
    max_level = std::max(max_level, level_o[o]);
    if (m_o[o] > 8000) {set_error("front too large for the triangular-solve kernels (m > 8000)"); return KH_ERR_SOLVER;}
  }
  std::vector<int32_t> o_of_k(K), k_of_o(K);
  for (int32_t o = 0; o < K; ++o) {o_of_k[o] = o;}
  std::stable_sort(o_of_k.begin(), o_of_k.end(), [&](int32_t a, int32_t b) {
    if (level_o[a] != level_o[b]) {return level_o[a] < level_o[b];}
    return m_o[a] > m_o[b];                  // largest fronts first: they are the critical path of their level
  });
  for (int32_t k = 0; k < K; ++k) {k_of_o[o_of_k[k]] = k;}
  for (int32_t e = 0; e < n_free; ++e) {sym.sn_of_elim[e] = k_of_o[sym.sn_of_elim[e]];}

  sym.parent.assign(K, -1);
  sym.rows_ptr.assign(K + 1, 0);
  sym.child_ptr.assign(K + 1, 0);
  sym.relpos_ptr.assign(K + 1, 0);
  sym.level.assign(K, 0);
  sym.front_off.assign(K, 0); sym.front_m.assign(K, 0); sym.front_ns.assign(K, 0); sym.front_first.assign(K, 0);
  sym.winv_off.assign(K, 0);
  sym.levels.assign(max_level + 1, {});
  int64_t off = 0, woff = 0;
  for (int32_t k = 0; k < K; ++k) {
    const int32_t o = o_of_k[k];
    sym.rows_ptr[k + 1] = sym.rows_ptr[k] + static_cast<int32_t>(rows[o].size());
    sym.child_ptr[k + 1] = sym.child_ptr[k] + static_cast<int32_t>(children[o].size());
    sym.relpos_ptr[k + 1] = sym.relpos_ptr[k] + static_cast<int32_t>(rows[o].size());
    sym.parent[k] = parent_o[o] < 0 ? -1 : k_of_o[parent_o[o]];
    sym.level[k] = level_o[o];
    sym.levels[level_o[o]].push_back(k);
    const int32_t ncols = first_o[o + 1] - first_o[o];
    sym.front_ns[k] = 3 * ncols;
    sym.front_m[k] = m_o[o];
    sym.front_first[k] = first_o[o];
    sym.front_off[k] = off;
    off += static_cast<int64_t>(sym.front_m[k]) * sym.front_m[k];
    const int64_t nsp = (sym.front_ns[k] + 15) & ~15;
    sym.winv_off[k] = woff;
    woff += nsp * nsp;
    sym.max_m = std::max(sym.max_m, sym.front_m[k]); sym.max_ns = std::max(sym.max_ns, sym.front_ns[k]);
    sym.nnz_factor += static_cast<int64_t>(sym.front_ns[k]) * (sym.front_ns[k] + 1) / 2 +
      static_cast<int64_t>(sym.front_ns[k]) * (sym.front_m[k] - sym.front_ns[k]);
    for (int32_t j = 0; j < sym.front_ns[k]; ++j) {
      sym.factor_flops += static_cast<int64_t>(sym.front_m[k] - j) * (sym.front_m[k] - j);
    }
  }
  sym.fronts_size = off;
  sym.winv_size = woff;
  sym.rows.reserve(sym.rows_ptr[K]); sym.child_list.reserve(sym.child_ptr[K]); sym.relpos.assign(sym.relpos_ptr[K], 0);
  sym.scatter_mode.assign(K, 0); sym.n_deferred.assign(K, 0); sym.has_b.assign(K, 0);
  for (int32_t k = 0; k < K; ++k) {
    const int32_t o = o_of_k[k];
    sym.rows.insert(sym.rows.end(), rows[o].begin(), rows[o].end());
    // destinations of the children's update matrices (see Symbolic::scatter_mode): level by level, lowest front number first
    std::vector<int32_t> kids;
    for (int32_t c : children[o]) {kids.push_back(k_of_o[c]);}
    std::sort(kids.begin(), kids.end());           // fronts are numbered level by level
    bool b_used = false;
    for (size_t g0 = 0; g0 < kids.size();) {
      size_t g1 = g0;
      while (g1 < kids.size() && sym.level[kids[g1]] == sym.level[kids[g0]]) {++g1;}
      for (size_t q = g0; q < g1; ++q) {
        const size_t idx = q - g0;
        sym.scatter_mode[kids[q]] = idx == 0 ? 1 : (idx == 1 || (idx == 2 && !b_used)) ? 2 : 0;
      }
      if (g1 - g0 >= 2) {b_used = true;}
      g0 = g1;
    }
    for (int32_t c : kids) {if (sym.scatter_mode[c] == 0) {sym.child_list.push_back(c); ++sym.n_deferred[k];}}
    for (int32_t c : kids) {if (sym.scatter_mode[c] != 0) {sym.child_list.push_back(c);}}
    for (int32_t c : kids) {if (sym.scatter_mode[c] == 2) {sym.has_b[k] = 1;}}
  }
  // where a child's rows sit in its parent's front: the parent's rows are numbered once (`stamp` doubles as the map from a row
  // to its position, tagged with the parent so that stale entries are recognised), then every child looks its rows up
  std::vector<int32_t> & where = stamp;
  std::vector<int32_t> where_of(n_free, -1);
  for (int32_t po = 0; po < K; ++po) {
    if (children[po].empty()) {continue;}
    const int32_t pfirst = first_o[po], pend = first_o[po + 1], pcols = pend - pfirst;
    const int32_t mark = K + po;                // (the first pass left values < K in `stamp`)
    for (size_t q = 0; q < rows[po].size(); ++q) {where[rows[po][q]] = mark; where_of[rows[po][q]] = pcols + static_cast<int32_t>(q);}
    for (int32_t c : children[po]) {
      const int32_t kc = k_of_o[c];
      for (size_t q = 0; q < rows[c].size(); ++q) {
        const int32_t rr = rows[c][q];
        int32_t at;
        if (rr < pend) {
          at = rr - pfirst;
        } else {
          if (where[rr] != mark) {set_error("symbolic: child row missing in parent front"); return KH_ERR_SOLVER;}
          at = where_of[rr];
        }
        sym.relpos[sym.relpos_ptr[kc] + q] = at;
      }
    }
  }
  // gather maps: the inverse of relpos, per (front, child)
  sym.cinv_ptr.assign(K + 1, 0);
  for (int32_t k = 0; k < K; ++k) {
    sym.cinv_ptr[k + 1] = sym.cinv_ptr[k] + (sym.child_ptr[k + 1] - sym.child_ptr[k]) * (sym.front_m[k] / 3);
  }
  sym.cinv.assign(sym.cinv_ptr[K], -1);
  for (int32_t k = 0; k < K; ++k) {
    const int32_t mp = sym.front_m[k] / 3;
    for (int32_t ci = sym.child_ptr[k]; ci < sym.child_ptr[k + 1]; ++ci) {
      const int32_t c = sym.child_list[ci];
      int32_t * inv = sym.cinv.data() + sym.cinv_ptr[k] + (ci - sym.child_ptr[k]) * mp;
      for (int32_t q = sym.relpos_ptr[c]; q < sym.relpos_ptr[c + 1]; ++q) {inv[sym.relpos[q]] = q - sym.relpos_ptr[c];}
    }
  }
  if (std::getenv("KH_SPA_DEBUG")) {
    std::fprintf(stderr, "[kh_spa] symbolic: structure %.2f ms\n",
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_nd1).count());
  }
  return KH_OK;
}

}  // namespace kh
