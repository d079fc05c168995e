"""Diagnostic builds of libsfmhip.so with parts of the k-NN kernel switched off (SFM_DBG in
csrc/match.hip): wrong results, timing only.  Load one with SFMHIP_SO=<path>.  Not product."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sfm_danpipeline_amd import build

build.build()
objdir = os.path.join(build.HERE, "build")
for a in sys.argv[1:]:
    # "<n>": SFM_DBG=<n> -> libsfmhip_dbg<n>.so;  "<name>:<-Dflag>[,<-Dflag>...]": those defines -> libsfmhip_<name>.so
    if ":" in a:
        v, defs = a.split(":", 1)[0], a.split(":", 1)[1].split(",")
    else:
        v, defs = f"dbg{int(a)}", [f"-DSFM_DBG={int(a)}"]
    obj = os.path.join(objdir, f"match_{v}.o")
    subprocess.check_call([build._hipcc()] + build.FLAGS + ["-ffp-contract=off"] + defs + ["-c", os.path.join(build.CSRC, "match.hip"), "-o", obj])
    objs = [os.path.join(objdir, n.replace(".hip", ".o")) for n in build.SOURCES if n != "match.hip"] + [obj]
    so = os.path.join(build.HERE, f"libsfmhip_{v}.so")
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print(so)
