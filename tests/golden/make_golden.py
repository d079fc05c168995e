"""Generates tests/golden/*.npz: inputs + expected outputs for the hot path.

The reference ships no fixtures, so these vectors come from the
independent numpy / scipy restatements in oracle/np_check.py (brute-force sort k-NN,
numpy.linalg.svd DLT, complex-step Jacobians, dense normal equations, scipy least_squares) --
NOT from the C oracle and NOT from the HIP path, both of which are tested against them.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_check as nc  # noqa: E402
from sfm_danpipeline_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def f32sqrt(s):
    return np.sqrt(np.float32(s)).astype(np.float32)


def match_cases():
    rng = np.random.default_rng(2024)
    cases = {}
    # 1. SIFT-like integer rows
    imgs = synth.sift_image_set(2, 90, 128, bank=120, seed=11)
    cases["sift"] = (imgs[0].astype(np.uint8), imgs[1][:70].astype(np.uint8))
    # 2. duplicate train rows -> equal distances must keep the lower train index first
    t = np.repeat(imgs[1][:20], 3, axis=0).astype(np.uint8)
    cases["ties"] = (imgs[0][:40].astype(np.uint8), t)
    # 3. distinct squared distances whose float sqrt collide (s >= 2^22), larger s at the lower index
    q = np.zeros((1, 128), np.uint8)
    found = None
    for level in range(255, 200, -1):  # rows (level,...,level,0) and (level,...,level,1): s and s+1
        lo = np.full(128, level, np.int64)
        lo[-1] = 0
        hi = lo.copy()
        hi[-1] = 1
        s1, s2 = int((lo ** 2).sum()), int((hi ** 2).sum())
        if s1 >= (1 << 22) and f32sqrt(s1) == f32sqrt(s2):
            found = (lo, hi)
            break
    assert found is not None
    lo, hi = found
    far = np.full(128, 255, np.int64)
    cases["sqrt_collision"] = (q, np.stack([far, hi, lo]).astype(np.uint8))  # hi (larger s) at index 1 < lo at 2
    # 4. ratio exactly at the boundary d0 == 0.8f*d1 : d0=4 (s=16), d1=5 (s=25)
    q = np.zeros((1, 128), np.uint8)
    t = np.zeros((3, 128), np.uint8)
    t[0, 0] = 5
    t[1, 0] = 4
    t[2, 0] = 200
    cases["ratio_boundary"] = (q, t)
    # 5. degenerate train sizes
    cases["nt0"] = (imgs[0][:5].astype(np.uint8), np.zeros((0, 128), np.uint8))
    cases["nt1"] = (imgs[0][:5].astype(np.uint8), imgs[1][:1].astype(np.uint8))
    cases["nt2"] = (imgs[0][:5].astype(np.uint8), imgs[1][:2].astype(np.uint8))
    cases["nq0"] = (np.zeros((0, 128), np.uint8), imgs[1][:9].astype(np.uint8))
    out = {}
    for name, (q, t) in cases.items():
        idx, dist = nc.knn2_bruteforce(q, t, "l2")
        mq, mt, md = nc.ratio_filter(idx, dist, 0.8)
        out.update({f"{name}_q": q, f"{name}_t": t, f"{name}_idx": idx, f"{name}_dist": dist,
                    f"{name}_mq": mq, f"{name}_mt": mt, f"{name}_md": md})
    out["names"] = np.array(sorted(cases))
    np.savez_compressed(os.path.join(OUT, "match_l2.npz"), **out)

    orbs = synth.orb_image_set(2, 80, bank=100, seed=5)
    q, t = orbs[0], orbs[1][:64]
    t2 = np.repeat(orbs[1][:10], 2, axis=0)
    out = {}
    for name, (a, b) in {"orb": (q, t), "orb_ties": (q[:30], t2)}.items():
        idx, dist = nc.knn2_bruteforce(a, b, "hamming")
        mq, mt, md = nc.ratio_filter(idx, dist, 0.8)
        out.update({f"{name}_q": a, f"{name}_t": b, f"{name}_idx": idx, f"{name}_dist": dist,
                    f"{name}_mq": mq, f"{name}_mt": mt, f"{name}_md": md})
        # the reference-literal behaviour: L2 on the raw bytes (src/Sfm.cpp:593)
        idx, dist = nc.knn2_bruteforce(a, b, "l2")
        out.update({f"{name}_l2idx": idx, f"{name}_l2dist": dist})
    out["names"] = np.array(["orb", "orb_ties"])
    np.savez_compressed(os.path.join(OUT, "match_hamming.npz"), **out)


def tri_case():
    sc = synth.two_view_scene(96, seed=7)
    X, err, keep = nc.triangulate_svd(sc["P1"], sc["P2"], sc["K"], sc["xy1"], sc["xy2"])
    margin = np.minimum(np.abs(err[:, 0] - 6.0), np.abs(err[:, 1] - 6.0)) > 0.5  # away from the 6 px edge
    sel = np.nonzero(margin)[0]
    np.savez_compressed(os.path.join(OUT, "triangulate.npz"), P1=sc["P1"], P2=sc["P2"], K=sc["K"], dist=sc["dist"],
                        xy1=sc["xy1"][sel], xy2=sc["xy2"][sel], X=X[sel], err=err[sel], keep=keep[sel])


def ba_case():
    pb = synth.ba_problem(5, 40, 4, seed=21)
    S, g, scale, step, cost = nc.reduced_system_dense(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"],
                                                      pb["obs_pt"], pb["obs_xy"], radius=1e4)
    c, p, f, cost_opt = nc.solve_scipy(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"])
    # Jacobian known answers in both angle-axis branches
    cams = np.array([[0.3, -0.2, 0.5, 0.1, 0.2, 5.0], [0, 0, 0, 0.1, 0.2, 5.0], [1e-9, -2e-9, 0, 0.0, 0.0, 4.0]])
    Xp = np.array([0.3, -0.4, 0.8])
    obs = np.array([10.0, -20.0])
    J = np.stack([nc.jacobian_complex_step(cam, Xp, 1500.0, obs) for cam in cams])
    r = np.stack([nc.residual(cam, Xp, 1500.0, obs) for cam in cams])
    np.savez_compressed(os.path.join(OUT, "ba.npz"), cams0=pb["cams0"], pts0=pb["pts0"], focal0=pb["focal0"],
                        obs_cam=pb["obs_cam"], obs_pt=pb["obs_pt"], obs_xy=pb["obs_xy"], S=S, g=g, scale=scale,
                        cost0=cost, cost_opt=cost_opt, focal_opt=f, jac_cams=cams, jac_X=Xp, jac_obs=obs, jac_J=J,
                        jac_r=r)


if __name__ == "__main__":
    match_cases()
    tri_case()
    ba_case()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
