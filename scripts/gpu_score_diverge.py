"""Where do the device's and the checker's five-point models part?  For bench.py's scoring scenes: every RANSAC sample of a
pair through sfmhip_score_five_point and through the C restatement, the largest model difference per sample, and the first
sample whose inlier counts differ.  usage: python scripts/gpu_score_diverge.py [pair index ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import orc, sfm_oracle_score as SC
from sfm_danpipeline_amd import _lib, scoring, synth

K = np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1]])
ctx = _lib.default_context()
srng = np.random.default_rng(4321)
scenes = []
for p_ in range(1225):
    sc_ = synth.two_view_scene(m=int(srng.integers(150, 900)), seed=5000 + p_, K=K, noise_px=0.4, outlier_frac=float(srng.uniform(0.1, 0.5)))
    scenes.append((sc_["xy1"], sc_["xy2"]))
which = [int(a) for a in sys.argv[1:]] or [44]
for p in which:
    a, b = scenes[p]
    inl, _, its = scoring.score_essential([(a, b)], K, ctx=ctx)
    cnt, mask, E, it = SC.find_essential_mat_ransac(a, b, K)
    print("pair", p, "matches", len(a), "device", int(inl[0]), int(its[0]), "oracle", cnt, it)
    n1, n2 = orc.em_normalize(a, K), orc.em_normalize(b, K)
    tab = SC.subset_table(len(a), max(int(its[0]), it))
    dm, dn, df = scoring.five_point(n1[tab], n2[tab], ctx=ctx)
    thr = 1.0 / ((K[0, 0] + K[1, 1]) / 2)
    t = np.float32(thr * thr)
    def count(E):
        x1 = np.concatenate([n1, np.ones((len(n1), 1))], 1); x2 = np.concatenate([n2, np.ones((len(n2), 1))], 1)
        Ex1 = x1 @ E.T; Etx2 = x2 @ E
        num = np.sum(x2 * Ex1, 1)
        den = Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2
        with np.errstate(all="ignore"):
            return int(((num * num / den).astype(np.float32) <= t).sum())
    for k in range(len(tab)):
        om, ofl = orc.five_point(n1[tab[k]], n2[tab[k]])
        if len(om) != dn[k]:
            print("  sample", k, "model count device", dn[k], "oracle", len(om), "flags", df[k], ofl)
            print("    device counts", [count(dm[k, m]) for m in range(dn[k])], "oracle counts", [count(E_) for E_ in om])
            continue
        dmax = max([np.abs(dm[k, m] - om[m]).max() for m in range(len(om))], default=0.0)
        cd, co = [count(dm[k, m]) for m in range(dn[k])], [count(E_) for E_ in om]
        if cd != co or dmax > 1e-9:
            print("  sample", k, "max |dE|", dmax, "counts", cd, co)
