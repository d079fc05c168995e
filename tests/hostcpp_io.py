"""Shared by the GPU run of the C++ host self-test (tests/test_gpu_host_cpp.py) and its sanitizer run on the CPU
(tests/test_sanitizers_cpu.py): the self-test's input file and the check of its output against the oracle."""
import struct

import numpy as np

from sfm_danpipeline_amd import bundle, synth


def write_input(tmp_path):
    rng = np.random.default_rng(5)
    # BA block (reference-shaped containers) -- its K is the pipeline's single cameraMatrix
    pb, cloud, poses, K, feats = __import__("tests.test_host_logic", fromlist=["_scene"])._scene(6, 120, 4, 41)
    # two views of one scene: descriptors of matching features are noisy copies
    sc = synth.two_view_scene(400, seed=8, K=K.copy())
    base = rng.integers(0, 256, (400, 128)).astype(np.int16)
    d0 = np.clip(base + np.rint(rng.normal(0, 5, base.shape)).astype(np.int16), 0, 255).astype(np.float32)
    perm = rng.permutation(400)
    d1 = np.clip(base + np.rint(rng.normal(0, 5, base.shape)).astype(np.int16), 0, 255).astype(np.float32)[perm]
    xy0, xy1 = sc["xy1"], sc["xy2"][perm]
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("<i", 2))
        for d in (d0, d1):
            f.write(struct.pack("<iii", d.shape[0], d.shape[1], 5))
            f.write(d.tobytes())
        f.write(struct.pack("<i", len(feats)))
        # views 0/1 carry the two-view scene's points for triangulateViews; the BA block indexes
        # its own feature lists, so give every view max(len) entries with the BA features first
        views = []
        for v in range(len(feats)):
            pts = list(feats[v])
            views.append(pts)
        for v, pts in enumerate(views):
            extra = xy0 if v == 0 else xy1 if v == 1 else np.zeros((0, 2))
            allp = np.array(pts, np.float64).reshape(-1, 2)
            f.write(struct.pack("<i", len(allp) + len(extra)))
            f.write(np.concatenate([extra, allp]).astype("<f8").tobytes())
        f.write(sc["K"].astype("<f8").tobytes())
        f.write(np.zeros(5, "<f8").tobytes())
        f.write(sc["P1"].astype("<f8").tobytes())
        f.write(sc["P2"].astype("<f8").tobytes())
        f.write(struct.pack("<i", len(poses)))
        for P in poses:
            f.write(np.asarray(P, "<f8").tobytes())
        f.write(struct.pack("<i", len(cloud)))
        for p in cloud:
            f.write(np.asarray(p["pt"], "<f8").tobytes())
            f.write(struct.pack("<i", len(p["idxImage"])))
            for view in sorted(p["idxImage"]):
                off = 400 if view in (0, 1) else 0          # BA features sit after the scene points
                f.write(struct.pack("<ii", view, p["idxImage"][view] + off))
    return dict(d0=d0, d1=d1, perm=perm, xy0=xy0, xy1=xy1, sc=sc, cloud=cloud, poses=poses, K=K, feats=feats)


def check_output(tmp_path, orc, w, scored_by_stub=False):
    d0, d1, perm, xy0, xy1, sc = w["d0"], w["d1"], w["perm"], w["xy0"], w["xy1"], w["sc"]
    cloud, poses, K, feats = w["cloud"], w["poses"], w["K"], w["feats"]
    raw = open(tmp_path / "out.bin", "rb").read()
    pos = 0
    n = struct.unpack_from("<i", raw, pos)[0]; pos += 4
    m = np.frombuffer(raw, dtype=np.dtype([("q", "<i4"), ("t", "<i4"), ("d", "<f4")]), count=n, offset=pos); pos += 12 * n
    rq, rt, rd = orc.match_knn2(d0, d1)
    assert np.array_equal(m["q"], rq) and np.array_equal(m["t"], rt) and np.array_equal(m["d"], rd)
    assert n > 300 and np.array_equal(perm[m["t"]], m["q"])           # the planted correspondences
    n2 = struct.unpack_from("<i", raw, pos)[0]; pos += 4
    c = np.frombuffer(raw, dtype=np.dtype([("X", "<f8", 3), ("q", "<i4"), ("t", "<i4")]), count=n2, offset=pos); pos += 32 * n2
    Xo, erro, keepo = orc.triangulate(sc["P1"], sc["P2"], sc["K"], np.zeros(5), xy0[rq], xy1[rt])
    kept = np.nonzero(keepo)[0]
    assert n2 == len(kept) and np.array_equal(c["q"], rq[kept]) and np.array_equal(c["t"], rt[kept])  # tracks
    assert np.abs(c["X"] - Xo[kept]).max() < 1e-10
    # pair cache: the batched all-pairs launch serves getMatching with identical results
    n3 = struct.unpack_from("<i", raw, pos)[0]; pos += 4
    mc = np.frombuffer(raw, dtype=m.dtype, count=n3, offset=pos); pos += 12 * n3
    assert n3 == n and np.array_equal(mc, m)
    # find2D3DMatches(NEW_VIEW=1, done {0}) against the oracle's literal loops
    done_view, n4 = struct.unpack_from("<ii", raw, pos); pos += 8
    f23 = np.frombuffer(raw, dtype=np.dtype([("X", "<f8", 3), ("xy", "<f8", 2)]), count=n4, offset=pos); pos += 40 * n4
    assert done_view == 0
    trk_ptr = np.arange(0, 2 * n2 + 1, 2, dtype=np.int32)
    trk_views = np.tile(np.array([0, 1], np.int32), n2)
    trk_feats = np.stack([c["q"], c["t"]], 1).reshape(-1).astype(np.int32)
    oc, of = orc.find_2d3d(trk_ptr, trk_views, trk_feats, 0, 1, rq, rt)
    assert n4 == len(oc) and np.array_equal(f23["X"], c["X"][oc]) and np.array_equal(f23["xy"], xy1[of])
    # mergeNewPoints
    before, after = struct.unpack_from("<ii", raw, pos); pos += 8
    added = np.frombuffer(raw, "<f8", 3 * (after - before), pos).reshape(-1, 3); pos += 24 * (after - before)
    fresh = np.concatenate([c["X"] + [0, 0, 0.004], c["X"] + [5.0, 0, 0], c["X"] + [5.0, 0, 0]])
    acc, nacc = orc.merge_new_points(c["X"], fresh)
    assert before == n2 and after - before == nacc and np.array_equal(added, fresh[acc])
    assert not acc[:n2].any() and not acc[2 * n2:].any()         # too close / duplicates of appended points
    # findBestPair: one pair (two images), >= 120 matches, keyed by the pose-inlier ratio of the E-matrix RANSAC
    nbp = struct.unpack_from("<i", raw, pos)[0]; pos += 4
    bp = np.frombuffer(raw, dtype=np.dtype([("ratio", "<f4"), ("q", "<i4"), ("t", "<i4")]), count=nbp, offset=pos); pos += 12 * nbp
    assert nbp == 1 and (int(bp["q"][0]), int(bp["t"][0])) == (0, 1)
    hom = struct.unpack_from("<i", raw, pos)[0]; pos += 4
    from oracle import sfm_oracle_score as score
    cnt = score.find_essential_mat_ransac(xy0[rq], xy1[rt], sc["K"])[0]
    assert bp["ratio"][0] == np.float32(np.float32(cnt) / np.float32(n)) and cnt > 0.8 * n
    if scored_by_stub:
        assert hom == n  # (the C-ABI stand-in of the sanitizer run counts every match for the homography)
    else:
        assert hom == score.find_homography_inliers(xy0[rq], xy1[rt])
    Kout = np.frombuffer(raw, "<f8", 9, pos).reshape(3, 3); pos += 72
    poses_out = np.frombuffer(raw, "<f8", 12 * len(poses), pos).reshape(-1, 3, 4); pos += 96 * len(poses)
    pts_out = np.frombuffer(raw, "<f8", 3 * len(cloud), pos).reshape(-1, 3)
    # the same adjustBundle through the Python mirror with the oracle as solver
    from tests.test_host_logic import _orc_solver
    bundle.adjust_bundle(cloud, poses, K, feats, solver=_orc_solver(orc))
    assert abs(Kout[0, 0] - K[0, 0]) < 1e-6 * K[0, 0] and Kout[0, 0] == Kout[1, 1]
    assert np.allclose(poses_out, np.array(poses), rtol=1e-6, atol=1e-9)
    assert np.allclose(pts_out, np.array([p["pt"] for p in cloud]), rtol=1e-6, atol=1e-9)


