"""Diagnostic (run through gpurun, -DSFM_FRONT_STAMPS build): repeat the front tree's solve on one system and, when an answer
differs from the first, say which fronts' data (L, y, contribution tiles) differ from the first run's -- deepest level first."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import synth, bundle, _lib

FD_INTS = 24
(FD_NO, FD_NS, FD_T, FD_NB_LAST, FD_PARENT, FD_LEVEL, FD_NCHILD, FD_CHILD_OFF, FD_INV_OFF, FD_PTINV_OFF, FD_SCHED_OFF,
 FD_NCAM, FD_CAM_OFF, FD_HAS_FOCAL, FD_OFF_L, FD_OFF_Y, FD_OWN_COLS, FD_OFF_PBUF, FD_PTILE_OFF, FD_LIVE) = range(20)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = _lib.default_context()
L = _lib.lib()
L.sfmhip_debug_tree_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p]
nc, npt, k = 200, 20000, 10
pb = synth.ba_problem(nc, npt, k, seed=5)
prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
S, g, _ = prob.reduced_system(1e4)
zr = np.linalg.solve(S, g)
sizes = np.zeros(2, np.int64)
prob.reduced_step(1e4)
assert L.sfmhip_debug_tree_dump(prob.h, None, 0, None, 0, sizes.ctypes.data) == 0
ints = np.zeros(sizes[0], np.int32)
F = prob.reduced_tree()["fronts"]


def dump():
    pool = np.zeros(sizes[1], np.float64)
    assert L.sfmhip_debug_tree_dump(prob.h, ints.ctypes.data, len(ints), pool.ctypes.data, len(pool), sizes.ctypes.data) == 0
    return pool


pool0, found = None, 0
for rep in range(reps):
    z, failed = prob.reduced_step(1e4)
    good = not failed and np.abs(z - zr).max() <= 1e-10 * np.abs(zr).max()
    if good:
        if pool0 is None:
            pool0 = dump()
        continue
    if pool0 is None:
        continue
    pool = dump()
    print(f"rep {rep}: wrong answer (max |dz| / max |z| {np.abs(z - zr).max() / np.abs(zr).max():.3e})")
    rows = []
    for f in range(F):
        d = ints[FD_INTS * f: FD_INTS * (f + 1)]
        no, ns, T = int(d[FD_NO]), int(d[FD_NS]), int(d[FD_T])
        oy, op = int(d[FD_OFF_Y]), int(d[FD_OFF_PBUF])
        yy = pool[oy: oy + 32 * T] != pool0[oy: oy + 32 * T]
        oL = int(d[FD_OFF_L])
        La = pool[oL: oL + 32 * T * 32 * no].reshape(32 * T, 32 * no)
        Lb = pool0[oL: oL + 32 * T * 32 * no].reshape(32 * T, 32 * no)
        dl = np.tril(La != Lb)
        if dl.any():
            rr, cc = np.nonzero(dl)
            print(f"  front {f}: L (lower triangle) differs at {int(dl.sum())} places, rows {rr.min()}..{rr.max()}, columns {cc.min()}..{cc.max()}; tiles {sorted(set(zip((rr // 32).tolist(), (cc // 32).tolist())))}")
        tiles = []
        for i in range(ns):
            for j in range(i + 1):
                a = op + (i * (i + 1) // 2 + j) * 1024
                nd = int((pool[a: a + 1024] != pool0[a: a + 1024]).sum())
                if nd:
                    tiles.append((no + i, no + j, nd))
        if yy.any() or tiles:
            rows.append((int(d[FD_LEVEL]), f, np.nonzero(yy)[0][:6].tolist(), int(yy.sum()), tiles, no, T))
    for lvl, f, ywhere, ny, tiles, no, T in sorted(rows, reverse=True)[:6]:
        print(f"  level {lvl} front {f} (no {no}, T {T}): y differs at {ny} places (first {ywhere}); tiles (r, c, entries) {tiles}")
    found += 1
    if found >= 4:
        break
print(f"{found} wrong answers looked at in {rep + 1} solves")
