"""Diagnostic (run through gpurun): phase times inside one chol_step panel workgroup from s_memtime
stamps.  Builds a SEPARATE library with -DSFM_CHOL_STAMPS; the product library has no stamps."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "sfm_danpipeline_amd")
SO = os.path.join(PKG, "libsfmhip_diag.so")


def build():
    srcs = ["context.hip", "match.hip", "triangulate.hip", "incremental.hip", "ba.hip"]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-Wno-unused-value", "-ffp-contract=fast", "-DSFM_CHOL_STAMPS", f"-I{ROOT}/include", "-o", SO] + \
          [os.path.join(PKG, "csrc", s) for s in srcs]
    subprocess.check_call(cmd)


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
        sys.exit(0)
    import torch  # noqa: F401  (load torch's ROCm runtime first)
    from sfm_danpipeline_amd import _lib
    _lib.SO = SO
    from sfm_danpipeline_amd import bundle, synth
    ctx = _lib.default_context()
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(3)
    ctx.synchronize()
    out = (C.c_ulonglong * 16)()
    L = _lib.lib()
    assert L.sfmhip_debug_chol_stamps(out) == 0
    t = list(out)
    names0 = ["start", "loads landed", "tile in LDS", "POTRF done"]
    names1 = {8: "start", 9: "loads landed", 10: "tile in LDS", 11: "TRSM done", 12: "stored"}
    base = min(t[0], t[8])
    print("s_memtime ticks (100 MHz constant clock? -> shown raw and as deltas)")
    for i, n in enumerate(names0):
        print(f" wave0 {n:14s} {t[i] - base:8d}")
    for i, n in names1.items():
        print(f" wave1 {n:14s} {t[i] - base:8d}")
