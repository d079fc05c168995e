// ba_front_plan.h -- host-side plan of the multifrontal ("front tree") factorisation of the reduced camera system.
//
// Replaces, for camera graphs with small vertex separators, what Eigen's LLT does behind ceres::Solve(DENSE_SCHUR)
// (reference src/BundleAdjustment.cpp:116,123): S z = g with S = the Schur complement on cameras + focal.  S is
// dense by storage only; its block pattern is the camera co-visibility graph.  The graph is dissected recursively
// (separator, components, separators of the components, ...); every tree node ("front") owns the columns of its
// separator (a leaf: of its component) and carries as its border ("struct") the columns of ancestors that its
// subtree's elimination fills.  A front is small enough (own <= FP_NO_MAX tiles, own + struct <= FP_T_MAX tiles
// of 32 columns) to be factored by ONE workgroup with its panels in LDS and its tiles in registers (ba_front.h):
//      [ D_vv  .    ]   L_vv = chol(D_vv), L_bv = D_bv L_vv^-T, contribution U_bb = -L_bv L_bv^T goes to the parent
//      [ D_bv  U_bb ]
// The dependency chain is one root-to-leaf path of the tree, not all columns (cfg4's ring of 200 cameras: 3 + 2 + 2
// + 4 tile steps instead of 16 with one level of dissection, 38 dense).
//
// Pure host C++ (no HIP): tests/test_front_plan.py compiles it with g++ and checks the plan's index maps by running a
// numpy multifrontal factorisation over them against a dense solve.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <map>
#include <numeric>
#include <vector>

namespace fplan {

constexpr int FP_TILE = 32;
constexpr int FP_NO_MAX = 4;      // own tiles of a front (the chain wave walks them one after the other)
constexpr int FP_T_MAX = 7;       // own + struct tiles (two panel generations of T - 1 tiles each live in LDS)
constexpr int FP_WAVES = 12;      // waves of a front's workgroup
constexpr int FP_SLOTS = 3;       // register tiles per tile wave
// wave roles: 0 the factorisation chain, 4 the right-hand side, 8 polls the children's flags (SIMD 0 keeps its
// matrix pipe free for the chain wave); the other nine hold tiles
static inline bool fp_is_tile_wave(int w) { return (w & 3) != 0 || w == 8; }

struct Front {
  int parent = -1, level = 0;
  std::vector<int> cams;      // own cameras, ascending
  bool has_focal = false;     // the root owns the focal column (last own column)
  std::vector<int> own;       // own parameter indices (6 c + j, focal = 6 nc) in front order
  std::vector<int> strct;     // the TRUE border parameters (what the elimination fills), in elimination order
  std::vector<std::pair<int, int>> btiles;  // the border as whole tiles of ancestors' own layouts: (front, tile), in elimination order
  std::vector<int> children;
  int no = 0, ns = 0, T = 0, nb_last = 8;
  std::vector<int> inv;       // 32 T: front index -> parameter index, -1 = padding
  std::vector<int> ptile;     // ns: border tile -> tile of the parent's front (the same 32 parameters in the same order)
  unsigned live = 0;          // bit 2 t + h: row half h of front tile t holds an own or a true border parameter
  std::vector<uint8_t> sched; // FP_WAVES x FP_SLOTS x 2: (r, c) of the tile a wave slot holds, 0xFF = empty
  std::vector<uint8_t> turn;  // FP_WAVES x FP_SLOTS: a border tile's place in its SIMD's queue when the tiles are folded at the end
  // helper workgroups (round 5): a front of one or two own tiles folds its border x border tiles when its factorisation is
  // over, and that fold -- 15 to 21 tiles x 6 to 14 blocks x up to 4 f64 MFMAs on ONE compute unit's four matrix pipes -- is
  // what a tree level costs beyond its POTRFs.  With helpers the tiles are dealt over 1 + nhelp workgroups: the front keeps
  // every (nhelp + 1)-th tile of the ancestors' order, helper h the others; a helper reads the front's finished panels from
  // memory (the rows of L it stores for the down-sweep anyway), folds, and sends its tiles to the parent like the front does.
  int nhelp = 0;
  std::vector<std::vector<uint8_t>> hsched;  // per helper: FP_WAVES x FP_SLOTS x 2, (r, c) of the tile a wave slot holds, 0xFF = empty
};

struct Plan {
  bool ok = false;
  int nc = 0;
  std::vector<Front> fronts;      // children before parents (post-order)
  std::vector<int> up_order;      // front indices, deepest level first
  int levels = 0;
  int chain_blocks = 0;           // 4-column block steps on the longest leaf-to-root path
  int chain_tiles = 0;            // tile steps on that path
  int max_T = 0;
  const char* why = "";
};

struct Builder {
  int nc;
  const std::vector<std::vector<int>>& nb;
  int leaf_cols;
  Plan plan;
  std::vector<int> node_of_cam;
  Builder(int nc_, const std::vector<std::vector<int>>& nb_, int leaf_cols_) : nc(nc_), nb(nb_), leaf_cols(leaf_cols_), node_of_cam(nc_, -1) {}

  // components of the subgraph induced by `in` (mask) restricted to `verts`
  std::vector<std::vector<int>> components(const std::vector<int>& verts, const std::vector<char>& in) {
    std::vector<std::vector<int>> out;
    std::vector<char> seen(nc, 0);
    for (int s : verts) {
      if (!in[s] || seen[s]) continue;
      out.emplace_back();
      std::vector<int>& comp = out.back();
      comp.push_back(s);
      seen[s] = 1;
      for (size_t h = 0; h < comp.size(); ++h)
        for (int v : nb[comp[h]])
          if (in[v] && !seen[v]) {
            seen[v] = 1;
            comp.push_back(v);
          }
      std::sort(comp.begin(), comp.end());
    }
    return out;
  }

  // separator of a CONNECTED vertex set: reverse Cuthill-McKee positions from a pseudo-peripheral start, a cut at
  // position p puts every vertex at or behind p that sees a vertex before p into the separator
  bool find_cut(const std::vector<int>& V, bool root, std::vector<int>& sep, std::vector<std::vector<int>>& comps) {
    std::vector<char> in(nc, 0);
    for (int v : V) in[v] = 1;
    std::vector<int> lvl(nc, -1), order, deg(nc, 0);
    for (int v : V)
      for (int u : nb[v]) deg[v] += in[u];  // degree inside V: the end of a band has the fewest neighbours
    auto bfs = [&](int start) {
      order.clear();
      for (int v : V) lvl[v] = -1;
      order.push_back(start);
      lvl[start] = 0;
      for (size_t h = 0; h < order.size(); ++h) {
        const int u = order[h];
        std::vector<int> nx;
        for (int v : nb[u])
          if (in[v] && lvl[v] < 0) {
            lvl[v] = lvl[u] + 1;
            nx.push_back(v);
          }
        // ties in the degree: the vertex that shares more neighbours with u is the closer one (a ring's start sees
        // both directions: u+1, u-1, u+2, ... keeps every prefix of the order one contiguous arc)
        std::vector<int> common(nx.size(), 0);
        for (size_t i = 0; i < nx.size(); ++i)
          for (int w : nb[nx[i]])
            if (in[w] && std::binary_search(nb[u].begin(), nb[u].end(), w)) ++common[i];
        std::vector<int> idx(nx.size());
        std::iota(idx.begin(), idx.end(), 0);
        std::sort(idx.begin(), idx.end(), [&](int a, int c) {
          if (deg[nx[a]] != deg[nx[c]]) return deg[nx[a]] < deg[nx[c]];
          if (common[a] != common[c]) return common[a] > common[c];
          return nx[a] < nx[c];
        });
        for (int i : idx) order.push_back(nx[i]);
      }
    };
    int start = V[0];
    for (int rep = 0; rep < 2; ++rep) {
      bfs(start);
      start = order.back();
    }
    bfs(start);
    if (order.size() != V.size()) return false;  // (not connected: the caller splits into components first)
    std::vector<int> pos(nc, -1), minpos(nc, 0);
    for (size_t i = 0; i < order.size(); ++i) pos[order[i]] = (int)i;
    for (int v : V) {
      int m = pos[v];
      for (int u : nb[v])
        if (in[u]) m = std::min(m, pos[u]);
      minpos[v] = m;
    }
    const int n = (int)V.size();
    long best = -1;
    int best_p = -1;
    const int G = std::min(n - 1, 384);  // (every position up to 385 vertices: one camera off balance can cost a level)
    for (int k = 1; k <= G; ++k) {
      const int p = (int)((long long)n * k / (G + 1));
      if (p < 1 || p >= n) continue;
      std::vector<char> keep(nc, 0);
      int nsep = 0;
      for (int v : V) {
        if (pos[v] >= p && minpos[v] < p) ++nsep;
        else keep[v] = 1;
      }
      const int sep_cols = 6 * nsep + (root ? 1 : 0);
      if (sep_cols > FP_TILE * FP_NO_MAX) continue;
      auto cs = components(V, keep);
      if (cs.size() < 2) continue;
      size_t mx = 0;
      for (auto& c : cs) mx = std::max(mx, c.size());
      const long score = 6 * (long)mx + 2 * (long)sep_cols;
      if (best < 0 || score < best) best = score, best_p = p;
    }
    if (best_p < 0) return false;
    std::vector<char> keep(nc, 0);
    sep.clear();
    for (int v : V) {
      if (pos[v] >= best_p && minpos[v] < best_p) sep.push_back(v);
      else keep[v] = 1;
    }
    std::sort(sep.begin(), sep.end());
    comps = components(V, keep);
    return true;
  }

  // returns the node index, -1 on failure
  int build(const std::vector<int>& V, bool root, int level) {
    const int cols = 6 * (int)V.size() + (root ? 1 : 0);
    std::vector<int> sep;
    std::vector<std::vector<int>> comps;
    bool split = false;
    if (root) {
      std::vector<char> in(nc, 0);
      for (int v : V) in[v] = 1;
      auto cs = components(V, in);
      if (cs.size() > 1) {  // cameras that share no point: only the focal couples them
        comps = cs;
        split = true;
      }
    }
    if (!split && cols > leaf_cols) {
      split = find_cut(V, root, sep, comps);
      if (!split && cols > FP_TILE * FP_NO_MAX) return -1;
    }
    std::vector<int> kids;
    if (split)
      for (auto& c : comps) {
        const int k = build(c, false, level + 1);
        if (k < 0) return -1;
        kids.push_back(k);
      }
    const int id = (int)plan.fronts.size();
    plan.fronts.emplace_back();
    Front& f = plan.fronts.back();
    f.level = level;
    f.cams = split ? sep : V;
    f.has_focal = root;
    f.children = kids;
    for (int k : kids) plan.fronts[k].parent = id;
    for (int c : f.cams) node_of_cam[c] = id;
    return id;
  }
};

// adjacency as nc x wpr bit rows (the layout sfmhip_ba keeps); leaf_cols: components up to this many columns become leaves
static inline Plan build_plan(int nc, const unsigned long long* adj_bits, int wpr, int leaf_cols, int n_helpers = 0, int keep = 8) {
  std::vector<std::vector<int>> nb(nc);
  for (int i = 0; i < nc; ++i)
    for (int j = 0; j < nc; ++j)
      if (j != i && (((adj_bits[(size_t)i * wpr + (j >> 6)] >> (j & 63)) & 1ull) || ((adj_bits[(size_t)j * wpr + (i >> 6)] >> (i & 63)) & 1ull)))
        nb[i].push_back(j);
  Builder B(nc, nb, leaf_cols);
  std::vector<int> all(nc);
  std::iota(all.begin(), all.end(), 0);
  const int root = B.build(all, true, 0);
  Plan& P = B.plan;
  P.nc = nc;
  if (root < 0) {
    P.why = "a component without a small separator";
    return P;
  }
  const int F = (int)P.fronts.size();
  // elimination position of every camera: fronts are in post-order, cameras ascending inside a front
  std::vector<int> epos(nc + 1, -1);
  {
    int k = 0;
    for (int f = 0; f < F; ++f)
      for (int c : P.fronts[f].cams) epos[c] = k++;
    epos[nc] = k;  // the focal, last
  }
  // border cameras, bottom-up: (neighbours of the own cameras + the children's borders) minus the subtree
  std::vector<std::vector<int>> bcams(F);
  for (int f = 0; f < F; ++f) {
    Front& fr = P.fronts[f];
    std::vector<int> acc;
    for (int c : fr.cams)
      for (int u : nb[c])
        if (epos[u] > epos[c] && B.node_of_cam[u] != f) acc.push_back(u);
    for (int k : fr.children)
      for (int u : bcams[k])
        if (B.node_of_cam[u] != f) acc.push_back(u);
    std::sort(acc.begin(), acc.end(), [&](int a, int c) { return epos[a] < epos[c]; });
    acc.erase(std::unique(acc.begin(), acc.end()), acc.end());
    // every border camera must belong to an ancestor (a proper dissection guarantees it)
    for (int u : acc) {
      int a = fr.parent;
      while (a >= 0 && B.node_of_cam[u] != a) a = P.fronts[a].parent;
      if (a < 0) {
        P.why = "border outside the ancestors";
        return P;
      }
    }
    bcams[f] = acc;
  }
  // ---- own layouts: a front's own parameters in tiles of 32 (the last one padded)
  std::vector<int> owner_of(6 * (size_t)nc + 1, -1), pos_in_owner(6 * (size_t)nc + 1, -1);
  for (int f = 0; f < F; ++f) {
    Front& fr = P.fronts[f];
    if (fr.has_focal) {
      // the root: every front's border holds the focal, so it goes where its tile is in those borders anyway -- behind the
      // first connected group of the root's cameras (a ring's root is two arcs: [arc | focal | arc])
      std::vector<char> in(nc, 0);
      for (int c : fr.cams) in[c] = 1;
      auto groups = B.components(fr.cams, in);
      for (size_t gi = 0; gi < groups.size(); ++gi) {
        for (int c : groups[gi])
          for (int j = 0; j < 6; ++j) fr.own.push_back(6 * c + j);
        if (gi == 0) fr.own.push_back(6 * nc);
      }
      if (groups.empty()) fr.own.push_back(6 * nc);
    } else {
      for (int c : fr.cams)
        for (int j = 0; j < 6; ++j) fr.own.push_back(6 * c + j);
    }
    if (fr.own.empty()) {
      P.why = "empty front";
      return P;
    }
    for (size_t i = 0; i < fr.own.size(); ++i) owner_of[fr.own[i]] = f, pos_in_owner[fr.own[i]] = (int)i;
    fr.no = ((int)fr.own.size() + FP_TILE - 1) / FP_TILE;
    fr.nb_last = ((int)fr.own.size() - FP_TILE * (fr.no - 1) + 3) / 4;
  }
  // ---- borders as WHOLE TILES of the ancestors' own layouts: every tile that holds a true border parameter, in
  // elimination order.  A border tile of a child is then, parameter for parameter, a tile of its parent's front, and the
  // child's contribution block adds onto the parent tile by tile with no index map (the parameters of such a tile that the
  // elimination does not reach are zero rows of the front: work, not error).
  auto tile_params = [&](int fo, int t, int k) -> int {
    const Front& o = P.fronts[fo];
    const size_t i = (size_t)FP_TILE * t + k;
    return i < o.own.size() ? o.own[i] : -1;
  };
  for (int f = 0; f < F; ++f) {
    Front& fr = P.fronts[f];
    for (int c : bcams[f])
      for (int j = 0; j < 6; ++j) fr.strct.push_back(6 * c + j);
    if (!fr.has_focal) fr.strct.push_back(6 * nc);
    for (int p : fr.strct) fr.btiles.emplace_back(owner_of[p], pos_in_owner[p] / FP_TILE);
    std::sort(fr.btiles.begin(), fr.btiles.end());  // (fronts are in post-order: ancestors have the larger indices)
    fr.btiles.erase(std::unique(fr.btiles.begin(), fr.btiles.end()), fr.btiles.end());
    fr.ns = (int)fr.btiles.size();
    fr.T = fr.no + fr.ns;
    if (fr.no > FP_NO_MAX || fr.T > FP_T_MAX) {
      P.why = "a front exceeds the tile limits";
      return P;
    }
    P.max_T = std::max(P.max_T, fr.T);
    fr.inv.assign((size_t)FP_TILE * fr.T, -1);
    for (size_t i = 0; i < fr.own.size(); ++i) fr.inv[i] = fr.own[i];
    std::vector<char> truly(6 * (size_t)nc + 1, 0);
    for (int p : fr.strct) truly[p] = 1;
    fr.live = 0;
    for (int t = 0; t < fr.no; ++t)
      for (int h = 0; h < 2; ++h)
        if ((size_t)FP_TILE * t + 16 * h < fr.own.size()) fr.live |= 1u << (2 * t + h);
    for (int i = 0; i < fr.ns; ++i)
      for (int k = 0; k < FP_TILE; ++k) {
        const int p = tile_params(fr.btiles[i].first, fr.btiles[i].second, k);
        fr.inv[(size_t)FP_TILE * (fr.no + i) + k] = p;
        if (p >= 0 && truly[p]) fr.live |= 1u << (2 * (fr.no + i) + k / 16);
      }
  }
  // ---- border tile -> tile of the parent's front
  for (int f = 0; f < F; ++f) {
    Front& fr = P.fronts[f];
    fr.ptile.assign(fr.ns, -1);
    if (fr.parent < 0) continue;
    const Front& pa = P.fronts[fr.parent];
    int last = -1;
    for (int i = 0; i < fr.ns; ++i) {
      int at = -1;
      if (fr.btiles[i].first == fr.parent) {
        at = fr.btiles[i].second;
      } else {
        for (int k = 0; k < pa.ns; ++k)
          if (pa.btiles[k] == fr.btiles[i]) at = pa.no + k;
      }
      if (at <= last) {
        P.why = "a border tile is missing from the parent's front, or out of order";
        return P;
      }
      fr.ptile[i] = last = at;
    }
  }
  // tile -> (wave, slot): every tile (r, c), c <= r < T, but (0, 0) (the chain wave assembles it).  A triangular solve
  // is a chain of dependent f64 vector operations, and a wave that streams f64 MFMAs on the same SIMD lets it issue one
  // of them per MFMA (MI355X: the f64 matrix instructions run on the vector lanes) -- so the tiles below the diagonal of
  // the own columns (one solve each) go to the waves of SIMD 1 first, then SIMD 2, and the border x border tiles (all
  // updates, no solve) to SIMD 3 first.  A wave holds at most one tile of an own column below the diagonal (one solve
  // per step and wave).
  const int simd_waves[3][3] = {{1, 5, 9}, {2, 6, 10}, {3, 7, 11}};
  for (int f = 0; f < F; ++f) {
    Front& fr = P.fronts[f];
    fr.sched.assign((size_t)FP_WAVES * FP_SLOTS * 2, 0xFF);
    fr.turn.assign((size_t)FP_WAVES * FP_SLOTS, 0);
    int used[FP_WAVES] = {0}, load[FP_WAVES] = {0};
    std::vector<std::vector<int>> cols_of(FP_WAVES);
    // pref: the three SIMD classes in order of preference; the least loaded wave among the first nbal classes, a later
    // class only when those have no slot left
    auto place = [&](int r, int c, const int* pref, int nbal) -> bool {
      const bool solve = c < fr.no && r > c;
      const int w_tile = std::min(c, fr.no) * 4 + (solve ? 3 : 0) + 1;
      int best = -1, best_simd_load = 0;
      for (int pi = 0; pi < 3 && (best < 0 || pi < nbal); ++pi) {
        const int cls = pref[pi];
        const int sl = load[simd_waves[cls][0]] + load[simd_waves[cls][1]] + load[simd_waves[cls][2]];  // the SIMD's load first
        for (int k = 0; k < 3; ++k) {
          const int w = simd_waves[cls][k];
          if (used[w] >= FP_SLOTS) continue;
          if (solve && std::find(cols_of[w].begin(), cols_of[w].end(), c) != cols_of[w].end()) continue;
          if (best < 0 || sl < best_simd_load || (sl == best_simd_load && load[w] < load[best])) best = w, best_simd_load = sl;
        }
      }
      if (best < 0) return false;
      fr.sched[((size_t)best * FP_SLOTS + used[best]) * 2] = (uint8_t)r;
      fr.sched[((size_t)best * FP_SLOTS + used[best]) * 2 + 1] = (uint8_t)c;
      ++used[best];
      load[best] += w_tile;
      if (solve) cols_of[best].push_back(c);
      return true;
    };
    const int pref_solve[3] = {1, 2, 0}, pref_crit[3] = {0, 1, 2}, pref_upd[3] = {2, 1, 0};
    bool ok = true;
    // the own columns, nearest the diagonal first (the next step's diagonal tile waits for them)
    for (int c = 0; c < fr.no && ok; ++c)
      for (int r = c; r < fr.T && ok; ++r) {
        if (r == 0 && c == 0) continue;
        // the tile next to the diagonal gates the next step's factorisation: SIMD 1; the other solves two per SIMD on 2 and 3
        // (the LAST own column has no next diagonal tile to protect: its solves spread over SIMDs 1 to 3 -- the border
        // tiles cannot be folded before they are through)
        ok = r == c + 1 ? place(r, c, pref_crit, 1) : place(r, c, pref_solve, c == fr.no - 1 && r > c ? 3 : 2);
      }
    // border x border tiles.  A front of one or two own tiles keeps both panel generations to the end, and front_up folds
    // the panels into these tiles only when the factorisation is over ("deferred"): nothing streams MFMAs beside the
    // solves, and the tiles can sit on any SIMD -- wave 8 included -- in column order (the parent wants the low columns
    // first), dealt round-robin over the SIMDs.  A front of more own tiles folds as it goes: SIMDs 3 and 2, then 1.
    if (fr.no <= 2) {
      // in column order (the order in which the parent wants them), dealt round-robin over the four SIMDs; on a SIMD the
      // tiles are folded one after the other ("turn": one wave streaming MFMAs has the matrix pipe to itself, three waves
      // at once would all finish late), so tile p is done about p / 4 + 1 folds after the factorisation's end
      // A wave sends a tile and waits for the stores to leave before it raises the tile's flag, so the first tiles go to
      // ten different waves (rounds: every wave gets its t-th border tile before any gets its (t+1)-th).
      const int simd_of[4][3] = {{8, -1, -1}, {1, 5, 9}, {2, 6, 10}, {3, 7, 11}};
      int turn[4] = {0, 0, 0, 0}, ndef[FP_WAVES] = {0};
      struct Item { int d, cls, a, b, r, c; };
      std::vector<Item> items;
      for (int c = fr.no; c < fr.T; ++c)
        for (int r = c; r < fr.T; ++r) {
          Item it{1, 2, c, r, r, c};
          int cur = f, R = r, C = c, d = 1;
          while (P.fronts[cur].parent >= 0) {
            const Front& cf = P.fronts[cur];
            const Front& pf = P.fronts[cf.parent];
            const int Rp = cf.ptile[R - cf.no], Cp = cf.ptile[C - cf.no];
            if (Cp < pf.no) {
              it = Rp < pf.no ? Item{d, 0, Rp, Cp, r, c} : Item{d, 1, Cp, Rp, r, c};
              break;
            }
            cur = cf.parent, R = Rp, C = Cp, ++d;
          }
          items.push_back(it);
        }
      std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) {
        if (x.d != y.d) return x.d < y.d;
        if (x.cls != y.cls) return x.cls < y.cls;
        if (x.a != y.a) return x.a < y.a;
        return x.b < y.b;
      });
      std::vector<std::pair<int, int>> todo;
      {
        // the front's own share, and the helpers': the first `keep` tiles of the ancestors' order are the front's (a helper's tile
        // reaches the parent two hand-offs later than one the front folds in its first rounds: what the parent's factorisation
        // waits for first must not take that way), the others are dealt over the helpers and the front in turn
        const int H = fr.parent >= 0 ? std::max(0, std::min(n_helpers, (int)items.size() - keep)) : 0;
        fr.nhelp = H;
        fr.hsched.assign(H, std::vector<uint8_t>((size_t)FP_WAVES * FP_SLOTS * 2, 0xFF));
        std::vector<int> hk(H, 0);
        for (size_t q = 0; q < items.size(); ++q) {
          const int owner = H && (int)q >= keep ? (int)((q - (size_t)keep) % (size_t)(H + 1)) + 1 > H ? 0 : (int)((q - (size_t)keep) % (size_t)(H + 1)) + 1 : 0;
          if (owner == 0) {
            todo.emplace_back(items[q].r, items[q].c);
            continue;
          }
          const int k = hk[owner - 1]++;  // the helper's k-th tile: wave k (SIMD k mod 4); one tile per wave (ba_front.h, fr_helper)
          if (k >= FP_WAVES) {
            P.why = "no helper slot left for a tile";
            return P;
          }
          std::vector<uint8_t>& hs = fr.hsched[owner - 1];
          hs[((size_t)(k % FP_WAVES) * FP_SLOTS + k / FP_WAVES) * 2] = (uint8_t)items[q].r;
          hs[((size_t)(k % FP_WAVES) * FP_SLOTS + k / FP_WAVES) * 2 + 1] = (uint8_t)items[q].c;
        }
      }
      size_t p = 0;
      for (int round = 0; round < FP_SLOTS && p < todo.size(); ++round)
        for (int k = 0; k < 3 && p < todo.size(); ++k)
          for (int q = 0; q < 4 && p < todo.size(); ++q) {
            const int w = simd_of[q][k];
            if (w < 0 || used[w] >= FP_SLOTS || ndef[w] != round) continue;
            fr.sched[((size_t)w * FP_SLOTS + used[w]) * 2] = (uint8_t)todo[p].first;
            fr.sched[((size_t)w * FP_SLOTS + used[w]) * 2 + 1] = (uint8_t)todo[p].second;
            fr.turn[(size_t)w * FP_SLOTS + used[w]] = (uint8_t)turn[q]++;
            ++used[w];
            ++ndef[w];
            ++p;
          }
      if (p < todo.size()) ok = false;
    } else {
      for (int c = fr.no; c < fr.T && ok; ++c)
        for (int r = c; r < fr.T && ok; ++r) ok = place(r, c, pref_upd, 2);
    }
    if (!ok) {
      P.why = "no wave slot left for a tile";
      return P;
    }
  }
  // levels, the up-sweep order (deepest first) and the chain length
  int maxl = 0;
  for (auto& fr : P.fronts) maxl = std::max(maxl, fr.level);
  P.levels = maxl + 1;
  for (int l = maxl; l >= 0; --l)
    for (int f = 0; f < F; ++f)
      if (P.fronts[f].level == l) P.up_order.push_back(f);
  std::vector<int> cb(F, 0), ct(F, 0);
  for (int f = 0; f < F; ++f) {  // post-order: children first
    const Front& fr = P.fronts[f];
    int mb = 0, mt = 0;
    for (int k : fr.children) mb = std::max(mb, cb[k]), mt = std::max(mt, ct[k]);
    cb[f] = mb + 8 * (fr.no - 1) + fr.nb_last;
    ct[f] = mt + fr.no;
  }
  P.chain_blocks = cb[root];
  P.chain_tiles = ct[root];
  P.ok = true;
  return P;
}

}  // namespace fplan

// ---- the flat form the device reads (and the test checks): one int pool + offsets into one pool of doubles
namespace fplan {

constexpr int FD_INTS = 28;  // ints per front descriptor
enum {
  FD_NO = 0, FD_NS, FD_T, FD_NB_LAST, FD_PARENT, FD_LEVEL, FD_NCHILD, FD_CHILD_OFF, FD_INV_OFF, FD_PTINV_OFF, FD_SCHED_OFF,
  FD_NCAM, FD_CAM_OFF, FD_HAS_FOCAL, FD_OFF_L, FD_OFF_Y, FD_OWN_COLS, FD_OFF_PBUF, FD_PTILE_OFF, FD_LIVE, FD_FOCAL_POS, FD_TFLAG_OFF,
  FD_NHELP, FD_HELP_OFF, FD_PFLAG_OFF, FD_USED
};
static_assert(FD_USED <= FD_INTS, "descriptor size");

struct Flat {
  int n_fronts = 0, levels = 0, max_T = 0;
  std::vector<int> ints;       // [FD_INTS x fronts | pools]
  std::vector<int> up_order;   // deepest level first
  std::vector<int> up_roles;   // the up-sweep's workgroups in grid order: front | helper << 16 (0: the front itself, h: its h-th
                               // helper), level by level from the deepest, a level's fronts before their helpers -- whatever a
                               // workgroup waits for comes before it in the grid
  std::vector<int> down_order; // root first
  int n_tflags = 0;            // per front: a flag per tile of its contribution block + one for the rhs (FD_TFLAG_OFF)
  size_t n_doubles = 0;        // per front: L (32 T x 32 no, row-major), y (32 T), and the buffer in which it hands its
                               // contribution block to its parent: the lower-triangle tiles of its border x border block,
                               // tile (i, j) at (i (i + 1) / 2 + j) * 1024, each in the accumulator layout by register pairs
};

static inline Flat flatten(const Plan& P) {
  Flat fl;
  const int F = (int)P.fronts.size();
  fl.n_fronts = F;
  fl.levels = P.levels;
  fl.max_T = P.max_T;
  fl.up_order = P.up_order;
  fl.down_order.assign(P.up_order.rbegin(), P.up_order.rend());
  fl.ints.assign((size_t)FD_INTS * F, 0);
  size_t nd = 0;
  for (int f = 0; f < F; ++f) {
    const Front& fr = P.fronts[f];
    auto put = [&](int k, int v) { fl.ints[(size_t)FD_INTS * f + k] = v; };
    put(FD_NO, fr.no);
    put(FD_NS, fr.ns);
    put(FD_T, fr.T);
    put(FD_NB_LAST, fr.nb_last);
    put(FD_PARENT, fr.parent);
    put(FD_LEVEL, fr.level);
    put(FD_NCHILD, (int)fr.children.size());
    put(FD_HAS_FOCAL, fr.has_focal ? 1 : 0);
    put(FD_OWN_COLS, (int)fr.own.size());
    put(FD_LIVE, (int)fr.live);
    put(FD_CHILD_OFF, (int)fl.ints.size());
    for (int k : fr.children) fl.ints.push_back(k);
    put(FD_INV_OFF, (int)fl.ints.size());
    fl.ints.insert(fl.ints.end(), fr.inv.begin(), fr.inv.end());
    // per child: tile of this front -> border tile of the child (-1: the child's border does not reach it)
    put(FD_PTINV_OFF, (int)fl.ints.size());
    for (int k : fr.children) {
      const Front& ch = P.fronts[k];
      std::vector<int> ptinv(FP_T_MAX + 1, -1);
      for (int i = 0; i < ch.ns; ++i) ptinv[ch.ptile[i]] = i;
      fl.ints.insert(fl.ints.end(), ptinv.begin(), ptinv.end());
    }
    put(FD_PTILE_OFF, (int)fl.ints.size());
    fl.ints.insert(fl.ints.end(), fr.ptile.begin(), fr.ptile.end());
    put(FD_SCHED_OFF, (int)fl.ints.size());
    for (int i = 0; i < FP_WAVES * FP_SLOTS; ++i) {
      const int r = fr.sched[2 * i], c = fr.sched[2 * i + 1];
      fl.ints.push_back(r == 0xFF ? -1 : (r | (c << 8) | (fr.turn[i] << 16)));
    }
    put(FD_NCAM, (int)fr.cams.size());
    put(FD_CAM_OFF, (int)fl.ints.size());  // the own cameras, then where each one's six columns start in the front
    fl.ints.insert(fl.ints.end(), fr.cams.begin(), fr.cams.end());
    int focal_pos = -1;
    for (int c : fr.cams)
      for (size_t i = 0; i < fr.own.size(); ++i)
        if (fr.own[i] == 6 * c) fl.ints.push_back((int)i);
    for (size_t i = 0; i < fr.own.size(); ++i)
      if (fr.own[i] == 6 * P.nc) focal_pos = (int)i;
    put(FD_FOCAL_POS, focal_pos);
    put(FD_OFF_L, (int)nd);
    nd += (size_t)FP_TILE * fr.T * FP_TILE * fr.no;
    put(FD_OFF_Y, (int)nd);
    nd += (size_t)FP_TILE * fr.T;
    nd = (nd + 15) & ~(size_t)15;
    put(FD_OFF_PBUF, (int)nd);
    nd += (size_t)(fr.ns * (fr.ns + 1) / 2) * FP_TILE * FP_TILE;
    put(FD_TFLAG_OFF, fl.n_tflags);
    fl.n_tflags += fr.ns * (fr.ns + 1) / 2 + 1;
    // helpers: their tiles, and a flag per solved panel tile (step j, border row i: no x ns) that tells them its rows of L are in memory
    put(FD_NHELP, fr.nhelp);
    put(FD_HELP_OFF, (int)fl.ints.size());
    for (int h = 0; h < fr.nhelp; ++h)
      for (int i = 0; i < FP_WAVES * FP_SLOTS; ++i) {
        const int r = fr.hsched[h][2 * i], c = fr.hsched[h][2 * i + 1];
        fl.ints.push_back(r == 0xFF ? -1 : (r | (c << 8)));
      }
    put(FD_PFLAG_OFF, fl.n_tflags);
    if (fr.nhelp) fl.n_tflags += fr.no * fr.ns;
  }
  for (int l = P.levels - 1; l >= 0; --l) {
    for (int f = 0; f < F; ++f)
      if (P.fronts[f].level == l) fl.up_roles.push_back(f);
    for (int f = 0; f < F; ++f)
      if (P.fronts[f].level == l)
        for (int h = 1; h <= P.fronts[f].nhelp; ++h) fl.up_roles.push_back(f | (h << 16));
  }
  fl.n_doubles = nd;
  return fl;
}

}  // namespace fplan
