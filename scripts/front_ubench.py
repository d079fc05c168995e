"""Diagnostic (run through gpurun): the update loop of the front kernels (fr_update, csrc/ba_front.h) on one wave, alone and
beside other work on its SIMD / CU -- shader clocks per four-column block (4 f64 MFMAs 16x16x4: 256 cycles of matrix pipe).
Needs the -DSFM_FRONT_STAMPS build (scripts/build_ba_variant.py stamps -DSFM_FRONT_STAMPS -ffp-contract=fast).  Not product."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sfm_danpipeline_amd import _lib

ctx = _lib.default_context()
L = _lib.lib()
L.sfmhip_debug_front_ubench.argtypes = [C.c_int, C.c_void_p]
names = {0: "alone", 1: "+ f64 vector wave on the SIMD", 2: "+ two waves polling LDS (s_sleep 1) on the SIMD", 16: "+ two waves polling (s_sleep 8)",
         4: "+ folds on the other three SIMDs", 8: "+ a second fold on the SIMD", 5: "+ vector wave + other SIMDs", 6: "+ pollers + other SIMDs",
         12: "+ second fold + other SIMDs", 40: "+ second and third fold on the SIMD", 7: "+ vector wave, poller, other SIMDs"}
for mode in (0, 1, 2, 16, 4, 8, 40, 5, 6, 12, 7):
    out = np.zeros(24, np.uint64)
    assert L.sfmhip_debug_front_ubench(mode, out.ctypes.data) == 0
    clk, real = int(out[16]), int(out[17])
    print(f"mode {mode:2d} {names[mode]:48s}: {clk / 512:7.1f} clk per block ({clk / max(real, 1) / 10:.2f} GHz); others " +
          " ".join(f"w{w}:{int(out[2 * w]) / 512:.0f}" for w in (0, 1, 2, 3, 4) if out[2 * w]))

out = np.zeros(24, np.uint64)
assert L.sfmhip_debug_front_ubench(64, out.ctypes.data) == 0
print(f"bare MFMAs from registers, one wave: 4 independent per iteration {int(out[18]) / 2048:.1f} clk each; 8 per iteration {int(out[19]) / 2048:.1f}; "
      f"two dependent per iteration {int(out[20]) / 1024:.1f}")
print(f"4 LDS reads + 4 MFMAs that do not use them, per iteration: {int(out[21]) / 512:.1f} clk; the reads feeding the next iteration: {int(out[22]) / 512:.1f} clk")
print(f"update loop candidates, clk per block: one block ahead, branch-free {int(out[0]) / 512:.1f}; four blocks' operands then 16 MFMAs {int(out[1]) / 512:.1f}; "
      f"eight blocks' operands then 32 MFMAs {int(out[2]) / 512:.1f}; all eight read up front, MFMAs in two groups {int(out[3]) / 512:.1f}")
