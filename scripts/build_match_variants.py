"""Diagnostic builds of libsfmhip.so with parts of the k-NN kernel switched off (SFM_DBG in
csrc/match.hip): wrong results, timing only.  Load one with SFMHIP_SO=<path>.  Not product."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_danpipeline_amd import build

build.build()
objdir = os.path.join(build.HERE, "build")
for v in (int(a) for a in sys.argv[1:]):
    obj = os.path.join(objdir, f"match_dbg{v}.o")
    subprocess.check_call([build._hipcc()] + build.FLAGS + ["-ffp-contract=off", f"-DSFM_DBG={v}", "-c",
                                                            os.path.join(build.CSRC, "match.hip"), "-o", obj])
    objs = [os.path.join(objdir, n.replace(".hip", ".o")) for n in build.SOURCES if n != "match.hip"] + [obj]
    so = os.path.join(build.HERE, f"libsfmhip_dbg{v}.so")
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print(so)
