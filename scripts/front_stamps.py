"""Diagnostic (run through gpurun): where a front's workgroup spends its time, from s_memtime stamps of the front tree's
up-sweep (csrc/ba_front.h).  Needs the -DSFM_FRONT_STAMPS build: python scripts/build_ba_variant.py stamps -DSFM_FRONT_STAMPS
-ffp-contract=fast, then SFMHIP_SO=sfm_danpipeline_amd/libsfmhip_stamps.so python scripts/front_stamps.py.  Not product."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import _lib, bundle, synth

nc, npt, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (200, 100000, 10)
ctx = _lib.default_context()
pb = synth.ba_problem(nc, npt, k, seed=777)
prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
prob.iterate(6)
ctx.synchronize()
tree = prob.reduced_tree()
F = tree["fronts"]
out = np.zeros((min(F, 128), 128), np.uint64)
L = _lib.lib()
L.sfmhip_debug_front_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.sfmhip_debug_front_stamps(out.ctypes.data, F) == 0
t = out.astype(np.int64)
print(tree)
t0 = t[:, 0].min()
names = {0: "start", 1: "assembled", 2: "children done", 3: "extend-add done", 4: "potrf0>", 5: "potrf0<", 6: "upd0<", 7: "potrf1>", 8: "potrf1<",
         9: "upd1<", 10: "potrf2>", 11: "potrf2<", 12: "upd2<", 13: "potrf3>", 14: "potrf3<", 15: "upd3<", 16: "w1 assembled",
         17: "w1 loop end", 18: "w11 loop end", 19: "w1 stores issued", 20: "rhs loop end", 21: "w1 stores drained", 22: "flag stored", 23: "w2 loop end", 24: "w3 loop end", 25: "w6 loop end",
         26: "w7 loop end", 27: "w8 tile0 folded", 28: "w1 deferred0 folded", 29: "w8 tile0 flag"}
for f in range(len(t)):
    real = (t[f, 31] - t[f, 30]) / 100.0  # us (100 MHz)
    cyc = t[f, 22] - t[f, 0]
    line = [f"front {f}: {real:.2f} us, {cyc} clk ({cyc / max(real, 1e-9) / 1e3:.2f} GHz), start +{(t[f, 0] - t0)}"]
    for sl in sorted(names):
        if t[f, sl] > 0:
            line.append(f"{names[sl]} {t[f, sl] - t[f, 0]}")
    print("  ".join(line))

# ---- the deferred phase of the tile waves: per wave, loop end | ahead-fetch done | (turn start, fold end) per slot
for f in (int(a) for a in os.environ.get("STAMP_FRONTS", "0,2,6,14").split(",")):
    if f >= len(t):
        continue
    print(f"front {f}: deferred phase (clk since the front's start)")
    for w in (8, 0, 1, 5, 9, 2, 6, 10, 3, 7, 11):
        b = 32 + 8 * w
        v = [int(t[f, b + k] - t[f, 0]) if t[f, b + k] > 0 else -1 for k in range(8)]
        print(f"  wave {w:2d} (SIMD {w & 3}): fetch> {v[6]:7d} fetch< {v[7]:7d}  slots " + "  ".join(f"[{v[2 * k]:7d} {v[2 * k + 1]:7d}]" for k in range(3)))

# ---- the down-sweep (s_memrealtime, 100 MHz): per front, us from the earliest start
dn = np.zeros((min(F, 128), 8), np.uint64)
L.sfmhip_debug_down_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.sfmhip_debug_down_stamps(dn.ctypes.data, F) == 0
dn = dn.astype(np.int64)
d0 = dn[:, 0][dn[:, 0] > 0].min()
print("down-sweep (us since the first workgroup's start): start | L prefetched + inverted | z_b arrived | w ready | z_v solved | flag stored | cameras done")
for f in range(len(dn)):
    print(f"  front {f}: " + " ".join(f"{(dn[f, k] - d0) / 100.0:7.2f}" for k in (0, 1, 2, 3, 4, 5, 6)))
