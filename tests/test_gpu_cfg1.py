"""GPU: BASELINE.json configs[0] -- the reference's own data/temple sequence (ten 640 x 480 PNG frames +
camera_calibration_template.xml, committed as DATA under tests/golden/temple/) through the C++ host mirror in the
reference's call order (src/Sfm.cpp:38-63, 408-492):

    imagesLOAD -> getCameraMatrix -> extractFeature (SIFT) -> findBestPair's all-pairs getMatching (matchAllPairs) ->
    findBestPair (120-match cut, E-matrix RANSAC score, homography inliers) -> [pose] -> triangulateViews ->
    adjustCurrentBundle

with every stage compared with the oracle on the same inputs.  The pose step between findBestPair and triangulateViews
(getCameraPose: cv::recoverPose) is out of scope; the test derives a pose from the oracle's E of
the pair the reference would try first and hands both sides the same P.  Parity is UNPINNED by the reference (no
OpenCV / Ceres here): the oracle restates the libraries (oracle/*.c, *.py headers)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import sfm_oracle_score as SC
from sfm_danpipeline_amd import build, bundle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TEMPLE = os.path.join(HERE, "golden", "temple")
XML = os.path.join(TEMPLE, "camera_calibration_template.xml")


class _Reader:
    def __init__(self, raw):
        self.raw, self.pos = raw, 0

    def i32(self, n=1):
        v = struct.unpack_from("<%di" % n, self.raw, self.pos)
        self.pos += 4 * n
        return v[0] if n == 1 else v

    def arr(self, dtype, count):
        dt = np.dtype(dtype)
        a = np.frombuffer(self.raw, dt, count, self.pos)
        self.pos += dt.itemsize * count
        return a


MATCH = np.dtype([("q", "<i4"), ("t", "<i4"), ("d", "<f4")])
CLOUD = np.dtype([("X", "<f8", 3), ("q", "<i4"), ("t", "<i4")])


def _run(tmp_path, pose=None):
    exe = build.build_io_demo()
    cmd = [exe, "--cfg1", TEMPLE, XML, str(tmp_path / "out.bin")]
    if pose is not None:
        q, t, Pq, Pt = pose
        with open(tmp_path / "pose.bin", "wb") as f:
            f.write(struct.pack("<ii", q, t) + np.asarray(Pq, "<f8").tobytes() + np.asarray(Pt, "<f8").tobytes())
        cmd.append(str(tmp_path / "pose.bin"))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    rd = _Reader(open(tmp_path / "out.bin", "rb").read())
    out = {"stdout": r.stdout}
    n = rd.i32()
    feats = []
    for _ in range(n):
        nk = rd.i32()
        feats.append((rd.arr("<f4", 6 * nk).reshape(nk, 6), rd.arr("<f4", 128 * nk).reshape(nk, 128)))
    out["feats"] = feats
    out["K"] = rd.arr("<f8", 9).reshape(3, 3)
    out["dist"] = rd.arr("<f8", 5)
    matches = {}
    for _ in range(rd.i32()):
        q, t, nm = rd.i32(3)
        matches[(q, t)] = rd.arr(MATCH, nm)
    out["matches"] = matches
    nmap = rd.i32()
    out["map"] = [(np.float32(k), (int(q), int(t))) for k, q, t in rd.arr(np.dtype([("k", "<f4"), ("q", "<i4"), ("t", "<i4")]), nmap)]
    scored = {}
    for _ in range(rd.i32()):
        q, t, nm, e_inl, e_it, h_inl, h_it = rd.i32(7)
        scored[(q, t)] = dict(n=nm, e_inl=e_inl, e_it=e_it, h_inl=h_inl, h_it=h_it, e_mask=rd.arr(np.uint8, nm), h_mask=rd.arr(np.uint8, nm))
    out["scored"] = scored
    out["flags"] = rd.i32()
    if pose is not None:
        nc = rd.i32()
        out["cloud"] = rd.arr(CLOUD, nc)
        out["K_ba"] = rd.arr("<f8", 9).reshape(3, 3)
        out["Pq_ba"] = rd.arr("<f8", 12).reshape(3, 4)
        out["Pt_ba"] = rd.arr("<f8", 12).reshape(3, 4)
        out["X_ba"] = rd.arr("<f8", 3 * nc).reshape(nc, 3)
    assert rd.pos == len(rd.raw)
    return out


def _pose_from_E(E, K, a, b):
    """the four (R, t) of an essential matrix, the one with most points in front of both cameras (test scaffolding for
    the out-of-scope getCameraPose; plain numpy)"""
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    W = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    Ki = np.linalg.inv(K)
    x1 = np.concatenate([a, np.ones((len(a), 1))], 1) @ Ki.T
    x2 = np.concatenate([b, np.ones((len(b), 1))], 1) @ Ki.T
    best = None
    for R in (U @ W @ Vt, U @ W.T @ Vt):
        for t in (U[:, 2], -U[:, 2]):
            P2 = np.hstack([R, t[:, None]])
            front = 0
            for p, q in zip(x1[:200], x2[:200]):
                A = np.array([p[0] * np.array([0, 0, 1.0, 0]) - np.array([1.0, 0, 0, 0]), p[1] * np.array([0, 0, 1.0, 0]) - np.array([0, 1.0, 0, 0]),
                              q[0] * P2[2] - P2[0], q[1] * P2[2] - P2[1]])
                X = np.linalg.svd(A)[2][3]
                X = X[:3] / X[3]
                front += (X[2] > 0) and ((R @ X + t)[2] > 0)
            if best is None or front > best[0]:
                best = (front, P2)
    return np.hstack([np.eye(3), np.zeros((3, 1))]), best[1]


def test_cfg1_temple_sequence_end_to_end(tmp_path, orc):
    g = np.load(os.path.join(HERE, "golden", "temple_sift.npz"))
    run = _run(tmp_path)
    # ---- imagesLOAD + getCameraMatrix: ten frames, the calibration file's numbers
    assert len(run["feats"]) == 10
    assert np.array_equal(run["K"], np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1.0]])) and not run["dist"].any()
    # ---- extractFeature against the SIFT restatement's fixture: bit-equal (as in tests/test_gpu_sift.py)
    for i, (kp, desc) in enumerate(run["feats"]):
        kpo, do = g[f"kp{i}"], g[f"desc{i}"].astype(np.float32)
        assert len(kp) > 500 and kp.shape == kpo.shape, (i, kp.shape, kpo.shape)
        assert np.array_equal(kp.view(np.int32), kpo.view(np.int32)), i       # all six fields, bit for bit
        assert np.array_equal(desc, do), i
    # ---- all 45 pairs of findBestPair's loop: bit-exact match lists against the oracle on the same descriptors
    pairs = [(q, t) for q in range(9) for t in range(q + 1, 10)]
    assert list(run["matches"]) == pairs
    for (q, t), m in run["matches"].items():
        rq, rt, rdist = orc.match_knn2(run["feats"][q][1], run["feats"][t][1])
        assert np.array_equal(m["q"], rq) and np.array_equal(m["t"], rt) and np.array_equal(m["d"].view(np.int32), rdist.view(np.int32)), (q, t)
    # ---- the scores of every pair that passes the reference's 120-match cut (src/Sfm.cpp:533)
    pts = [kp[:, :2].astype(np.float64) for kp, _ in run["feats"]]                       # keypointstoPoints
    big = [p for p in pairs if len(run["matches"][p]) >= 120]
    assert len(big) >= 9 and list(run["scored"]) == big, [len(run["matches"][p]) for p in pairs]
    assert run["flags"] == 0
    disagree = []
    for p in big:
        m, s = run["matches"][p], run["scored"][p]
        a, b = pts[p[0]][m["q"]], pts[p[1]][m["t"]]
        cnt, mask, E, it = SC.find_essential_mat_ransac(a, b, run["K"])
        hc, hmask, hit = SC.find_homography_ransac(a, b, 0.004 * float(a.max()))
        if (s["e_inl"], s["e_it"]) != (cnt, it) or not np.array_equal(s["e_mask"], mask):
            disagree.append(("E", p, (s["e_inl"], s["e_it"]), (cnt, it)))
        if (s["h_inl"], s["h_it"]) != (hc, hit) or not np.array_equal(s["h_mask"], hmask):
            disagree.append(("H", p, (s["h_inl"], s["h_it"]), (hc, hit)))
    assert not disagree, disagree          # every pair, one oracle route: a disagreement is reported, not routed around
    # ---- findBestPair's std::map<float, pair>
    want = SC.find_best_pair_scores([(p, pts[p[0]][run["matches"][p]["q"]], pts[p[1]][run["matches"][p]["t"]]) for p in pairs], run["K"])
    assert [(float(k), v) for k, v in run["map"]] == [(float(k), v) for k, v in want]
    # (what the reference prints per pair, src/Sfm.cpp:567)
    assert run["stdout"].count("pose inliers ratio.") == len(big)
    # ---- baseReconstruction takes the map's FIRST entry (ascending keys: SURVEY.md appendix B.6)
    q, t = run["map"][0][1]
    m = run["matches"][(q, t)]
    a, b = pts[q][m["q"]], pts[t][m["t"]]
    E = SC.find_essential_mat_ransac(a, b, run["K"])[2]
    Pq, Pt = _pose_from_E(E, run["K"], a, b)
    run2 = _run(tmp_path, pose=(q, t, Pq, Pt))
    assert [(float(k), v) for k, v in run2["map"]] == [(float(k), v) for k, v in run["map"]]     # the run repeats itself
    # ---- triangulateViews: keep-mask (= the tracks that survive) and points
    Xo, _, keepo = orc.triangulate(Pq, Pt, run["K"], np.zeros(5), a, b)
    kept = np.nonzero(keepo)[0]
    c = run2["cloud"]
    assert len(c) == len(kept) >= 50, (len(c), len(kept))
    assert np.array_equal(c["q"], m["q"][kept]) and np.array_equal(c["t"], m["t"][kept])
    assert np.array_equal(c["X"].view(np.uint64), np.ascontiguousarray(Xo[kept]).view(np.uint64))   # bit for bit
    # ---- adjustCurrentBundle: the Python mirror of adjustBundle with the oracle as solver, same containers
    from tests.test_host_logic import _orc_solver
    cloud = [dict(pt=tuple(X), idxImage={q: int(fq), t: int(ft)}) for X, fq, ft in zip(c["X"], c["q"], c["t"])]
    poses = [np.zeros((3, 4)) for _ in range(10)]
    poses[q], poses[t] = Pq.copy(), Pt.copy()
    Kpy = run["K"].copy()
    summ = bundle.adjust_bundle(cloud, poses, Kpy, [[tuple(xy) for xy in p] for p in pts], solver=_orc_solver(orc))
    assert np.allclose(run2["K_ba"], Kpy, rtol=1e-6, atol=1e-9), (summ.termination, run2["K_ba"], Kpy)
    assert np.allclose(run2["Pq_ba"], poses[q], rtol=1e-6, atol=1e-8) and np.allclose(run2["Pt_ba"], poses[t], rtol=1e-6, atol=1e-8)
    assert np.allclose(run2["X_ba"], np.array([p["pt"] for p in cloud]), rtol=1e-6, atol=1e-8)
    assert (summ.termination == orc.CONVERGENCE) == (not np.array_equal(run2["X_ba"], c["X"]))   # write-back only on CONVERGENCE
