"""Diagnostic (run through gpurun): phase times inside one chol_step2 panel workgroup from s_memtime
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
    srcs = ["context.hip", "match.hip", "triangulate.hip", "incremental.hip", "score.hip", "sift.hip", "ba.hip"]
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
    ctx.set_timing(True)
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(3)
    ctx.synchronize()
    print(prob.last_timing())
    out = (C.c_ulonglong * 32)()
    L = _lib.lib()
    assert L.sfmhip_debug_chol_stamps(out) == 0
    t = list(out)
    # slot -> phase boundary (cycles of the shader clock from the workgroup's start; the stamped
    # workgroup is tile row 1 of launch k2 = 2)
    names = {0: "w0 start", 1: "w0 D_aa s0 in LDS", 2: "w0 POTRF_a done", 3: "w3 D_bb final (U2 done)",
             4: "w3 POTRF_b done", 6: "w6 T_a in registers", 7: "w6 X_a done", 11: "w2 D_ba in registers",
             12: "w2 Y done", 9: "w5 T_b final (U2 done)", 10: "w5 X_b done", 16: "w6 D_ba s0 in LDS",
             17: "w10 D_ba s1 in LDS", 18: "w7 D_ba s2 in LDS", 19: "w9 D_ba s3 in LDS", 20: "w7 T_a s0 s1 in LDS",
             22: "w1 T_a s2 s3 in LDS", 13: "w0 T_b right half in LDS", 14: "w0 D_bb (1,1) in LDS",
             15: "w3 starts its fold", 23: "w3 fold done"}
    for i, n in sorted(names.items(), key=lambda kv: t[kv[0]]):
        print(f" {n:26s} {t[i] - t[0]:8d}")
