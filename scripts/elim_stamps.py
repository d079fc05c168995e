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
    from sfm_danpipeline_amd import build as B
    B.build_ba_variant("diag", ["-DSFM_ELIM_STAMPS"] + [a for a in os.environ.get("ELIM_DEFINES", "").split() if a], force=True)


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
    prob.iterate(int(os.environ.get("ELIM_STAMPS_ITERS", "3")))   # (3: the solve still moves; 150: long past convergence, radius 0)
    ctx.synchronize()
    out = (C.c_ulonglong * 32)()
    L = C.CDLL(SO)
    assert L.sfmhip_debug_elim_stamps(out) == 0
    t = list(out)
    names = {0: "kernel start", 1: "prologue done (tables in LDS)", 2: "first point data requested", 8: "  it2: top",
             9: "  it2: linearised", 10: "  it2: F^T F ds_adds issued", 11: "  it2: row sums done", 12: "  it2: 3x3 inverse done",
             13: "  it2: panel written", 14: "  it2: 30 MFMAs issued", 3: "loop done", 15: "all waves done (barrier A)", 16: "wave 0 staged (barrier B)", 4: "waves added, F^T F folded (C)",
             5: "scalars folded (D)", 6: "slab stored / scatter issued (end)", 20: "wave 0 loop done", 21: "wave 1 loop done", 22: "wave 2 loop done", 23: "wave 3 loop done",
             24: "wave 0 loop start", 25: "wave 1 loop start", 26: "wave 2 loop start", 27: "wave 3 loop start"}
    for i, n in sorted(names.items(), key=lambda kv: t[kv[0]]):
        print(f" {n:34s} {t[i] - t[0]:8d}" if i == 0 else f" {n:34s} {t[i] - t[0]:8d}")

    # every workgroup of the last launch: start / end in core cycles (s_memtime) and in 10 ns ticks (s_memrealtime)
    import numpy as np
    wg = (C.c_ulonglong * (2048 * 5))()
    assert L.sfmhip_debug_elim_wg(wg) == 0
    w = np.array(list(wg), dtype=np.uint64).reshape(2048, 5)
    w = w[w[:, 0] != 0]
    cyc = (w[:, 1] - w[:, 0]).astype(np.int64)
    r0, r1 = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64)
    span = (r1.max() - r0.min()) * 10e-3
    print(f" workgroups {len(w)}; launch span {span:.1f} us (first start to last end, 100 MHz clock); starts spread over {(r0.max() - r0.min()) * 10e-3:.1f} us")
    dur = (r1 - r0) * 10e-3
    print(f" workgroup durations us: min {dur.min():.1f} median {np.median(dur):.1f} max {dur.max():.1f};  cycles: min {cyc.min()} median {int(np.median(cyc))} max {cyc.max()}")
    ghz = cyc / np.maximum(dur, 1e-9) * 1e-3
    print(f" core clock seen by the workgroups (cycles / duration): min {ghz.min():.2f} median {np.median(ghz):.2f} max {ghz.max():.2f} GHz")
    order = np.argsort(-dur)[:8]
    print(" longest:", [(int(i), round(float(dur[i]), 1), round(float((r0[i] - r0.min()) * 10e-3), 1)) for i in order], "(workgroup, duration us, start offset us)")
    cnt = (w[:, 4] >> np.uint64(48)).astype(np.int64)
    xcc = (w[:, 4] >> np.uint64(32)).astype(np.int64) & 0xF
    hw = w[:, 4].astype(np.int64) & 0xFFFFFFFF
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 10 + cu
    u, cnts = np.unique(key, return_counts=True)
    print(f" distinct (xcc, se, sh, cu): {len(u)}; workgroups per CU: min {cnts.min()} max {cnts.max()}  histogram {np.bincount(cnts).tolist()}")
    print(" duration by piece size (points: workgroups, median us, max us):")
    for lo in range(0, 400, 20):
        m = (cnt >= lo) & (cnt < lo + 20)
        if m.any():
            print(f"   {lo:3d}-{lo + 19:3d}: {int(m.sum()):4d}  {np.median(dur[m]):6.1f} {dur[m].max():6.1f}   cycles/iteration (median) {np.median((cyc[m] - 15000) / np.ceil(np.ceil(cnt[m] / 4) / (4 if len(w) <= 600 else 2))):7.0f}")
    print(" per CU: (points, duration us) of its workgroups, the eight CUs that finish last and the four that finish first:")
    ends = {}
    for k in u:
        m = key == k
        ends[k] = (r1[m].max() - r0.min()) * 10e-3
    ks = sorted(ends, key=lambda k: -ends[k])
    for k in ks[:8] + ks[-4:]:
        m = np.nonzero(key == k)[0]
        print(f"   xcc {k // 1000} se {k // 100 % 10} cu {k % 100:2d}: end {ends[k]:5.1f}  " + "  ".join(f"wg {int(i)} {int(cnt[i])} pts {dur[i]:.1f}" for i in m))
    per_xcc = [round(float(max(ends[k] for k in ends if k // 1000 == x)), 1) for x in range(8)]
    print(" last end per XCC:", per_xcc)
