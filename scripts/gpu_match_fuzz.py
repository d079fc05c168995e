"""Ad-hoc GPU check (run through gpurun): k-NN lists and match lists of the device matcher against the oracle's, bit for bit, on
random pairs of many shapes -- clustered rows with frequent exact distance ties, copies at lane / tile / epoch distances, both
descriptor types, widths 32 ... 128, query / train sizes from 1 to a few thousand.  usage: gpu_match_fuzz.py [seed0] [cases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import _lib, matcher
from oracle import orc

orc.build()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = _lib.default_context()
bad = 0
nq_tot = nm_tot = ties_tot = 0
t00 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    dim = int(rng.choice([32, 64, 96, 128, 128, 128]))
    nt = int(rng.choice([1, 2, 31, 33, 64, 200, 513, 900, 2047, 2500, 8200, 9100]))
    nq = int(rng.choice([1, 31, 32, 65, 130, 257, 700]))
    kind = int(rng.integers(0, 4))
    if kind == 0:      # uniform random rows: no ties to speak of
        t = rng.integers(0, 256, (nt, dim))
    else:              # clusters with small steps in a few dimensions: ties are frequent
        centres = rng.integers(0, 256, (max(nt // int(rng.integers(3, 12)), 2), dim))
        t = centres[rng.integers(0, len(centres), nt)].copy()
        for _ in range(int(rng.integers(1, 5))):
            d_ = rng.integers(0, dim, nt)
            t[np.arange(nt), d_] = np.clip(t[np.arange(nt), d_] + rng.integers(-kind, kind + 1, nt), 0, 255)
    if nt > 40:
        for off in (1, 2, 31, 32, 33, 64, 97, 160, 4000, 8192):
            if off < nt:
                src = rng.integers(0, nt, max(nt // 12, 1))
                t[np.clip(src + rng.choice([-off, off], len(src)), 0, nt - 1)] = t[src]
    q = t[rng.integers(0, nt, nq)].copy()
    dq = rng.integers(0, dim, nq)
    q[np.arange(nq), dq] = np.clip(q[np.arange(nq), dq] + rng.integers(-1, 2, nq), 0, 255)
    cast = np.float32 if rng.random() < 0.6 else np.uint8
    q, t = np.ascontiguousarray(q.astype(cast)), np.ascontiguousarray(t.astype(cast))
    s = matcher.ImageSet([q, t], ctx=ctx)
    s.prepare_async()
    pl = matcher.MatchPlan(s, np.array([[0, 1]], np.int32))
    pl.run_async(0.8)
    ki, kd = pl.fetch_knn(0)
    a = pl.fetch_pair(0)
    r = orc.match_knn2(q, t, norm=orc.NORM_L2, want_knn=True, threads=8)
    ok = (np.array_equal(ki, r[3]) and np.array_equal(kd.view(np.uint32), r[4].view(np.uint32)) and np.array_equal(a[0], r[0])
          and np.array_equal(a[1], r[1]) and np.array_equal(a[2].view(np.uint32), r[2].view(np.uint32)))
    nq_tot += nq
    nm_tot += len(a[0])
    ties_tot += int(np.sum(r[4][:, 0] == r[4][:, 1])) if nt > 1 else 0
    if not ok:
        bad += 1
        w = np.nonzero(np.any(ki != r[3], axis=1))[0]
        print(f"case {case} (seed {seed0 + case}): dim {dim} nq {nq} nt {nt} kind {kind} {cast.__name__}: DIFFERS, first queries {w[:5]}", flush=True)
    pl.close()
    s.close()
print(f"{bad} of {n_cases} differ ({time.time() - t00:.0f} s); {nq_tot} queries compared, {nm_tot} matches, {ties_tot} queries whose best two distances are equal")
