"""Diagnostic (run through gpurun): phase times inside one ba_eliminate_mfma workgroup (workgroup 0, wave 0) from
s_memtime stamps.  Builds a SEPARATE library with -DSFM_ELIM_STAMPS; the product library has no stamps."""
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
           "-Wno-unused-value", "-ffp-contract=fast", "-DSFM_ELIM_STAMPS", f"-I{ROOT}/include", "-o", SO] + \
          [os.path.join(PKG, "csrc", s) for s in srcs]
    subprocess.check_call(cmd)


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
        sys.exit(0)
    os.environ["SFMHIP_SO"] = SO
    from sfm_danpipeline_amd import _lib, bundle, synth
    ctx = _lib.default_context()
    pb = synth.ba_problem(200, 100000, 10, seed=777)
    prob = bundle.BaProblem(200, 100000, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
    prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
    prob.iterate(3)
    ctx.synchronize()
    out = (C.c_ulonglong * 32)()
    L = C.CDLL(SO)
    assert L.sfmhip_debug_elim_stamps(out) == 0
    t = list(out)
    names = {0: "kernel start", 1: "prologue done (tables in LDS)", 2: "first point data requested", 8: "  it2: top",
             9: "  it2: linearised", 10: "  it2: F^T F ds_adds issued", 11: "  it2: row sums done", 12: "  it2: 3x3 inverse done",
             13: "  it2: panel written", 14: "  it2: 30 MFMAs issued", 3: "loop done", 15: "all waves done (barrier A)", 16: "wave 0 staged (barrier B)", 4: "waves added, F^T F folded (C)",
             5: "scalars folded (D)", 6: "scatter issued (end)", 20: "wave 0 loop done", 21: "wave 1 loop done", 22: "wave 2 loop done", 23: "wave 3 loop done",
             24: "wave 0 loop start", 25: "wave 1 loop start", 26: "wave 2 loop start", 27: "wave 3 loop start"}
    for i, n in sorted(names.items(), key=lambda kv: t[kv[0]]):
        print(f" {n:34s} {t[i] - t[0]:8d}" if i == 0 else f" {n:34s} {t[i] - t[0]:8d}")
