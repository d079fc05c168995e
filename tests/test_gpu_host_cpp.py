"""GPU: the C++ host mirror (StructFromMotion::getMatching / triangulateViews /
adjustCurrentBundle -> BundleAdjustment::adjustBundle with the reference's signatures) driven end
to end through the C ABI and compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sfm_danpipeline_amd import _lib, build, bundle, synth

pytestmark = pytest.mark.gpu


def test_cpp_host_classes_end_to_end(tmp_path, orc):
    from tests import hostcpp_io
    exe = build.build_host_demo()
    assert exe and os.path.exists(exe)
    w = hostcpp_io.write_input(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    hostcpp_io.check_output(tmp_path, orc, w)


def test_native_rccl_allreduce_in_a_cpp_program(tmp_path, ctx):
    """libsfmhip_rccl.so: the reduced-system exchange as ncclAllReduce(sum, ncclDouble) issued from C++
    (no Python callback).  A GPU box has one device, and RCCL refuses two ranks on one device, so the
    program runs a 1-rank communicator declared as rank 0 of a 2-rank job whose other rank holds no
    points: the pack / all-reduce / unpack path executes and must walk the single-process iterates."""
    build.build_rccl()
    exe = os.path.join(os.path.dirname(build.SO), "sfm_rccl_selftest")
    assert os.path.exists(exe)
    pb = synth.ba_problem(12, 3000, 6, seed=31)
    iters = 5
    with open(tmp_path / "pb.bin", "wb") as f:
        f.write(struct.pack("<iiii", 12, 3000, len(pb["obs_cam"]), iters))
        f.write(pb["cams0"].astype("<f8").tobytes())
        f.write(pb["pts0"].astype("<f8").tobytes())
        f.write(struct.pack("<d", float(pb["focal0"])))
        f.write(pb["obs_cam"].astype("<i4").tobytes())
        f.write(pb["obs_pt"].astype("<i4").tobytes())
        f.write(pb["obs_xy"].astype("<f8").tobytes())
    # (one node, no network here: RCCL's bootstrap listens on the loopback interface instead of probing for another.  The
    # communicator's start-up hung ONCE in some fifty runs of this test on the pool -- a fresh box, 300 s, nothing of this
    # library on the stack yet; scripts/gpu_rccl_loop.sh: 30 of 30 on another box -- so a start that takes two minutes is
    # given one more try before the test fails.)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    for attempt in (0, 1):
        try:
            r = subprocess.run([exe, str(tmp_path / "pb.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True,
                               timeout=120, env=env)
            break
        except subprocess.TimeoutExpired:
            if attempt:
                raise
    assert r.returncode == 0, r.stderr[-2000:]
    raw = open(tmp_path / "out.bin", "rb").read()
    cams = np.frombuffer(raw[:12 * 6 * 8], "<f8").reshape(12, 6)
    focal, cost = struct.unpack("<dd", raw[12 * 6 * 8:12 * 6 * 8 + 16])
    it = struct.unpack("<i", raw[12 * 6 * 8 + 16:12 * 6 * 8 + 20])[0]
    one = bundle.BaProblem(12, 3000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    one.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    s1 = one.iterate(iters)
    c1, p1, f1 = one.get_params()
    assert it == s1.iterations == iters
    assert abs(cost - s1.final_cost) <= 1e-9 * s1.final_cost
    assert np.allclose(cams, c1, rtol=1e-9, atol=1e-12) and abs(focal - f1) <= 1e-9 * f1


def _aa_to_R(aa):
    return synth._aa_to_rotation(aa)


def _write_containers(path, pb, cx, cy):
    return synth.write_ba_containers(path, pb, cx, cy)


def _read_containers(path, nc, npt):
    raw = np.frombuffer(open(path, "rb").read(), "<f8")
    assert raw.size == 9 + 12 * nc + 3 * npt
    return raw[:9].reshape(3, 3), raw[9:9 + 12 * nc].reshape(nc, 3, 4), raw[9 + 12 * nc:].reshape(npt, 3)


def test_adjust_bundle_writes_back_only_on_convergence_at_cfg4_size(tmp_path, ctx):
    """BundleAdjustment::adjustBundle in the reference's containers at BASELINE cfg4's size (200 views, 100 000 points, 10^6
    observations): the policy of src/BundleAdjustment.cpp:118-129.  (1) as shipped -- 500 iterations / 10 s -- the solve converges
    and K, the poses and the cloud come back adjusted, equal to the C-ABI solve of the same problem; (2) the wall-clock limit
    reached first => "Bundle adjustment failed." and every container byte for byte as it went in; (3) the iteration limit
    reached first => the same."""
    exe = build.build_ba_demo()
    assert exe and os.path.exists(exe)
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    cx, cy = 960.0, 540.0
    poses0, K0 = _write_containers(tmp_path / "in.bin", pb, cx, cy)

    def run(**env):
        e = dict(os.environ, **env)
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode == 0, r.stderr[-2000:]
        return r, _read_containers(tmp_path / "out.bin", 200, 100000)

    r, (K, poses, pts) = run()
    assert "Bundle adjustment: iterations" in r.stdout and "failed" not in r.stderr
    c, p, f, s = bundle.ba_solve(pb["cams0"], pb["pts0"], pb["focal0"], pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx,
                                 opts=bundle.default_opts())
    assert s.termination == _lib.BA_CONVERGENCE
    assert int(r.stdout.split("iterations")[1].split(",")[0]) == s.iterations
    assert K[0, 0] == K[1, 1] and abs(K[0, 0] - f) <= 1e-6 * f and K[0, 0] != K0[0, 0]
    assert (K[0, 2], K[1, 2], K[2, 2], K[0, 1]) == (cx, cy, 1.0, 0.0)       # only the focal length is a parameter
    want = np.stack([np.concatenate([_aa_to_R(c[i, :3]), c[i, 3:, None]], axis=1) for i in range(200)])
    assert np.allclose(poses, want, rtol=1e-6, atol=1e-8) and np.allclose(pts, p, rtol=1e-6, atol=1e-8)
    c0, c1 = (float(v.split(",")[0]) for v in r.stdout.split("cost")[1].split("->"))
    assert c1 < 0.1 * c0 and abs(c1 - s.final_cost) <= 1e-5 * c1 and not np.array_equal(pts, pb["pts0"])

    for env in (dict(SFM_BA_TEST_MAX_TIME_S="1e-9"), dict(SFM_BA_TEST_MAX_ITERATIONS="3")):
        r, (K, poses, pts) = run(**env)
        assert "Bundle adjustment failed." in r.stderr, (env, r.stdout, r.stderr)
        assert K.tobytes() == K0.tobytes() and poses.tobytes() == poses0.tobytes() and pts.tobytes() == pb["pts0"].tobytes(), env
