"""GPU: the C++ host mirror (StructFromMotion::getMatching / triangulateViews /
adjustCurrentBundle -> BundleAdjustment::adjustBundle with the reference's signatures) driven end
to end through the C ABI and compared with the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sfm_danpipeline_amd import build, bundle, synth

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
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, str(tmp_path / "pb.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True,
                       timeout=300, env=env)
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
