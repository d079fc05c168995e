"""Diagnostic build of libsfmhip.so with extra defines for csrc/ba.hip (A/B of kernel experiments): libsfmhip_<name>.so.
Usage: python scripts/build_ba_variant.py <name> -DFLAG [-DFLAG ...].  Load with SFMHIP_SO=<path>.  Not product."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_danpipeline_amd import build

build.build()
objdir = os.path.join(build.HERE, "build")
name, defs = sys.argv[1], sys.argv[2:]
obj = os.path.join(objdir, f"ba_{name}.o")
subprocess.check_call([build._hipcc()] + build.FLAGS + defs + ["-c", os.path.join(build.CSRC, "ba.hip"), "-o", obj])
objs = [os.path.join(objdir, n.replace(".hip", ".o")) for n in build.SOURCES if n != "ba.hip"] + [obj]
so = os.path.join(build.HERE, f"libsfmhip_{name}.so")
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
print(so)
