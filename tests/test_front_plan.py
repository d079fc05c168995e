"""The front-tree plan of the reduced camera solve (csrc/ba_front_plan.h), on the CPU: the plan's index maps drive a numpy
multifrontal factorisation (assemble from S, extend-add of the children's contribution blocks, partial Cholesky, forward
and backward substitution along the tree) whose solution must equal a dense solve.  Replaces Eigen's LLT behind
ceres::Solve(DENSE_SCHUR), reference src/BundleAdjustment.cpp:116,123; the device kernels (csrc/ba_front.h) walk the same
maps."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FD_INTS = 28
(FD_NO, FD_NS, FD_T, FD_NB_LAST, FD_PARENT, FD_LEVEL, FD_NCHILD, FD_CHILD_OFF, FD_INV_OFF, FD_PTINV_OFF, FD_SCHED_OFF,
 FD_NCAM, FD_CAM_OFF, FD_HAS_FOCAL, FD_OFF_L, FD_OFF_Y, FD_OWN_COLS, FD_OFF_PBUF, FD_PTILE_OFF, FD_LIVE) = range(20)
T_MAX = 7


@pytest.fixture(scope="module")
def fp(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fplan") / "libfplan.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tests", "stub", "front_plan_capi.cpp")])
    lib = C.CDLL(so)
    lib.fplan_build_flat.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return lib


def ring_adj(nc, k):
    """cfg3 / cfg4's co-visibility: every point is seen by k consecutive cameras of a ring (synth.ba_problem)."""
    a = np.zeros((nc, nc), bool)
    for i in range(nc):
        for d in range(1, k):
            a[i, (i + d) % nc] = a[(i + d) % nc, i] = True
    return a


def band_adj(nc, k):
    a = np.zeros((nc, nc), bool)
    for i in range(nc):
        for d in range(1, k):
            if i + d < nc:
                a[i, i + d] = a[i + d, i] = True
    return a


def build(fp, adj, leaf_cols=None):
    """leaf_cols None: the solver's own rule (sfmhip_ba's set-up) -- the largest of 96, 64, 32 whose fronts fit."""
    if leaf_cols is None:
        for lc in (96, 64, 32):
            pl = build(fp, adj, lc)
            if pl is not None:
                return pl
        return None
    nc = len(adj)
    wpr = (nc + 63) // 64
    bits = np.zeros((nc, wpr), np.uint64)
    for i in range(nc):
        for j in np.nonzero(adj[i])[0]:
            bits[i, j >> 6] |= np.uint64(1) << np.uint64(j & 63)
    header = np.zeros(8, np.int32)
    ints = np.zeros(4 << 20, np.int32)
    order = np.zeros(4096, np.int32)
    rc = fp.fplan_build_flat(nc, bits.ctypes.data, wpr, leaf_cols, header.ctypes.data, ints.ctypes.data, len(ints),
                             order.ctypes.data, len(order))
    assert rc == 0
    if not header[0]:
        return None
    F = int(header[1])
    return dict(F=F, levels=int(header[2]), max_T=int(header[3]), ints=ints[:header[4]].copy(), n_doubles=int(header[5]),
                chain_blocks=int(header[6]), chain_tiles=int(header[7]), up=order[:F].copy())


def desc(pl, f):
    return pl["ints"][FD_INTS * f: FD_INTS * (f + 1)]


def random_system(adj, seed):
    nc = len(adj)
    dim = 6 * nc + 1
    rng = np.random.default_rng(seed)
    S = np.zeros((dim, dim))
    for i in range(nc):
        for j in range(i + 1, nc):
            if adj[i, j]:
                b = rng.normal(size=(6, 6))
                S[6 * i:6 * i + 6, 6 * j:6 * j + 6] = b
                S[6 * j:6 * j + 6, 6 * i:6 * i + 6] = b.T
    S[-1, :-1] = S[:-1, -1] = rng.normal(size=dim - 1)
    S += np.diag(np.abs(S).sum(axis=1) + 1.0)   # diagonally dominant: positive definite
    return S, rng.normal(size=dim)


def multifrontal_solve(pl, S, g):
    ints, F = pl["ints"], pl["F"]
    L, U, Y = {}, {}, {}
    for f in pl["up"]:                        # deepest level first: children before parents
        d = desc(pl, f)
        no, ns, T = int(d[FD_NO]), int(d[FD_NS]), int(d[FD_T])
        n, o = 32 * T, 32 * no
        inv = ints[d[FD_INV_OFF]: d[FD_INV_OFF] + n]
        Fm = np.zeros((n, n))
        y = np.zeros(n)
        for b in range(o):                    # own columns: entries of S, the identity on the padding
            if inv[b] < 0:
                Fm[b, b] = 1.0
                continue
            y[b] = g[inv[b]]
            for a in range(b, n):
                if inv[a] >= 0:
                    Fm[a, b] = S[inv[a], inv[b]]
        for k in range(int(d[FD_NCHILD])):    # extend-add, tile by tile: a child's border tile IS a tile of this front
            ch = int(ints[d[FD_CHILD_OFF] + k])
            ptinv = ints[d[FD_PTINV_OFF] + k * (T_MAX + 1): d[FD_PTINV_OFF] + (k + 1) * (T_MAX + 1)]
            dc = desc(pl, ch)
            assert int(dc[FD_PARENT]) == f
            cno, cns = int(dc[FD_NO]), int(dc[FD_NS])
            ptile = ints[dc[FD_PTILE_OFF]: dc[FD_PTILE_OFF] + cns]
            assert np.all(np.diff(ptile) > 0)          # monotone: a lower triangle maps onto a lower triangle
            cinvd = ints[dc[FD_INV_OFF]: dc[FD_INV_OFF] + 32 * int(dc[FD_T])]
            Uc, yc = U[ch], Y[ch]
            for i in range(cns):
                R = int(ptile[i])
                assert ptinv[R] == i
                # the same 32 parameters, padding included, in the same order
                assert np.array_equal(cinvd[32 * (cno + i): 32 * (cno + i + 1)], inv[32 * R: 32 * (R + 1)])
                y[32 * R: 32 * R + 32] += yc[32 * (cno + i): 32 * (cno + i + 1)]
                for j in range(i + 1):
                    C_ = int(ptile[j])
                    blk = Uc[32 * i: 32 * i + 32, 32 * j: 32 * j + 32]
                    Fm[32 * R: 32 * R + 32, 32 * C_: 32 * C_ + 32] += np.tril(blk) if i == j else blk
            assert sum(1 for t in range(T) if ptinv[t] >= 0) == cns
        Lvv = np.linalg.cholesky(Fm[:o, :o] + np.tril(Fm[:o, :o], -1).T)
        Lbv = np.linalg.solve(Lvv, Fm[o:, :o].T).T
        L[f] = np.vstack([Lvv, Lbv])
        low = np.tril(Fm[o:, o:])
        U[f] = low + np.tril(low, -1).T - Lbv @ Lbv.T
        y[:o] = np.linalg.solve(Lvv, y[:o])
        y[o:] -= Lbv @ y[:o]
        Y[f] = y
    z = np.zeros(len(g))
    seen = np.zeros(len(g), int)
    for f in pl["up"][::-1]:                  # root first
        d = desc(pl, f)
        no, T = int(d[FD_NO]), int(d[FD_T])
        n, o = 32 * T, 32 * no
        inv = ints[d[FD_INV_OFF]: d[FD_INV_OFF] + n]
        zb = np.array([z[i] if i >= 0 else 0.0 for i in inv[o:]])
        w = Y[f][:o] - L[f][o:].T @ zb
        zv = np.linalg.solve(L[f][:o].T, w)
        for b in range(o):
            if inv[b] >= 0:
                z[inv[b]] = zv[b]
                seen[inv[b]] += 1
    assert np.all(seen == 1)                  # every parameter is owned by exactly one front
    return z


def check_schedule(pl):
    ints = pl["ints"]
    for f in range(pl["F"]):
        d = desc(pl, f)
        no, T = int(d[FD_NO]), int(d[FD_T])
        inv = ints[d[FD_INV_OFF]: d[FD_INV_OFF] + 32 * T]
        for t in range(T):                        # a row half that is not live holds no own parameter
            for h in range(2):
                if not (int(d[FD_LIVE]) >> (2 * t + h)) & 1 and t < no:
                    assert np.all(inv[32 * t + 16 * h: 32 * t + 16 * h + 16] < 0)
        assert 1 <= no <= 4 and T <= 7 and 1 <= d[FD_NB_LAST] <= 8
        sched = ints[d[FD_SCHED_OFF]: d[FD_SCHED_OFF] + 36].reshape(12, 3)
        tiles = []
        for w in range(12):
            cols = []
            for s in range(3):
                if sched[w, s] < 0:
                    continue
                assert w % 4 != 0 or w == 8       # waves 0 and 4 hold no tiles
                if w == 8:
                    assert no <= 2 and ((int(sched[w, s]) >> 8) & 255) >= no   # wave 8: border tiles of a front that folds them at the end
                r, c = int(sched[w, s]) & 255, (int(sched[w, s]) >> 8) & 255
                tiles.append((r, c))
                if c < no and r > c:
                    cols.append(c)
            assert len(cols) == len(set(cols))    # one triangular solve per step and wave
        want = [(r, c) for c in range(T) for r in range(c, T) if (r, c) != (0, 0)]
        assert sorted(tiles) == sorted(want)


@pytest.mark.parametrize("nc,k", [(200, 10), (50, 10), (96, 6), (560, 8), (30, 4)])
def test_ring_plans_solve_the_system(fp, nc, k):
    adj = ring_adj(nc, k)
    pl = build(fp, adj)
    assert pl is not None
    check_schedule(pl)
    S, g = random_system(adj, nc)
    z = multifrontal_solve(pl, S, g)
    zr = np.linalg.solve(S, g)
    assert np.abs(z - zr).max() <= 1e-10 * np.abs(zr).max()
    if (nc, k) == (200, 10):                  # cfg4: 16 leaves of one tile, three levels of 2-tile separators, a 4-tile root
        assert pl["levels"] == 5 and pl["F"] == 31 and pl["chain_tiles"] == 11 and pl["max_T"] == 7


def test_band_and_disconnected_graphs(fp):
    adj = band_adj(120, 7)
    pl = build(fp, adj)
    assert pl is not None
    check_schedule(pl)
    S, g = random_system(adj, 1)
    assert np.abs(multifrontal_solve(pl, S, g) - np.linalg.solve(S, g)).max() <= 1e-10
    # two groups of cameras that share no point: only the focal couples them
    adj2 = np.zeros((40, 40), bool)
    adj2[:20, :20] = band_adj(20, 5)
    adj2[20:, 20:] = band_adj(20, 5)
    pl2 = build(fp, adj2)
    assert pl2 is not None
    S, g = random_system(adj2, 2)
    assert np.abs(multifrontal_solve(pl2, S, g) - np.linalg.solve(S, g)).max() <= 1e-10


def test_small_and_dense_graphs(fp):
    # a handful of cameras: one front holds everything
    adj = np.ones((8, 8), bool) & ~np.eye(8, dtype=bool)
    pl = build(fp, adj)
    assert pl is not None and pl["F"] == 1
    S, g = random_system(adj, 3)
    assert np.abs(multifrontal_solve(pl, S, g) - np.linalg.solve(S, g)).max() <= 1e-10
    # random visibility: no small separator exists, the plan is refused (the dense factorisation stays)
    rng = np.random.default_rng(0)
    a = rng.random((200, 200)) < 0.4
    assert build(fp, a | a.T) is None


def test_helper_workgroups_share_the_deferred_tiles_exactly(tmp_path):
    """Front::nhelp (round 5, off by default): with helpers a front of one or two own tiles keeps the first `keep` border x border
    tiles of the ancestors' order and deals the others over itself and its helpers -- every tile exactly once, a helper's at most
    one per wave; the up-sweep's roles list every front before its helpers and every level before the one above."""
    src = tmp_path / "helpers.cpp"
    src.write_text(r'''
#include "%s/sfm_danpipeline_amd/csrc/ba_front_plan.h"
#include <cstdio>
#include <set>
int main() {
  const int nc = 200, k = 10, wpr = (nc + 63) / 64;
  std::vector<unsigned long long> adj((size_t)nc * wpr, 0);
  for (int i = 0; i < nc; ++i)
    for (int d = 1; d < k; ++d) {
      const int j = (i + d) %% nc;
      adj[(size_t)i * wpr + (j >> 6)] |= 1ull << (j & 63);
      adj[(size_t)j * wpr + (i >> 6)] |= 1ull << (i & 63);
    }
  for (int H : {0, 1, 3}) {
    fplan::Plan P;
    for (int leaf : {96, 64, 32}) {
      P = fplan::build_plan(nc, adj.data(), wpr, leaf, H, 4);
      if (P.ok) break;
    }
    if (!P.ok) return 1;
    fplan::Flat fl = fplan::flatten(P);
    int helpers = 0;
    for (size_t f = 0; f < P.fronts.size(); ++f) {
      const fplan::Front& fr = P.fronts[f];
      std::multiset<std::pair<int, int>> seen;
      for (int w = 0; w < fplan::FP_WAVES; ++w)
        for (int s = 0; s < fplan::FP_SLOTS; ++s) {
          const int r = fr.sched[((size_t)w * fplan::FP_SLOTS + s) * 2], c = fr.sched[((size_t)w * fplan::FP_SLOTS + s) * 2 + 1];
          if (r != 0xFF && c >= fr.no) seen.insert({r, c});
        }
      for (int h = 0; h < fr.nhelp; ++h)
        for (int w = 0; w < fplan::FP_WAVES; ++w)
          for (int s = 0; s < fplan::FP_SLOTS; ++s) {
            const int r = fr.hsched[h][((size_t)w * fplan::FP_SLOTS + s) * 2], c = fr.hsched[h][((size_t)w * fplan::FP_SLOTS + s) * 2 + 1];
            if (r == 0xFF) continue;
            if (s != 0 || c < fr.no) return 2;       // one tile per helper wave, border x border only
            seen.insert({r, c});
          }
      helpers += fr.nhelp;
      if (fr.nhelp > H || (fr.no > 2 && fr.nhelp)) return 3;
      for (int c = fr.no; c < fr.T; ++c)
        for (int r = c; r < fr.T; ++r)
          if (seen.count({r, c}) != 1) return 4;     // every border x border tile exactly once
      if ((int)seen.size() != fr.ns * (fr.ns + 1) / 2) return 5;
    }
    if ((int)fl.up_roles.size() != (int)P.fronts.size() + helpers) return 6;
    if (H == 0 && helpers) return 7;
    if (H == 3 && !helpers) return 8;
    // grid order: a front before its helpers, a deeper level before a shallower one
    std::vector<int> at(P.fronts.size(), -1);
    int last_level = 1 << 30;
    for (size_t i = 0; i < fl.up_roles.size(); ++i) {
      const int f = fl.up_roles[i] & 0xFFFF, h = fl.up_roles[i] >> 16;
      if (h == 0) at[f] = (int)i;
      else if (at[f] < 0) return 9;
      if (P.fronts[f].level > last_level) return 10;
      last_level = P.fronts[f].level;
    }
  }
  std::puts("ok");
  return 0;
}
''' % ROOT)
    exe = str(tmp_path / "helpers")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, str(src)])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)
