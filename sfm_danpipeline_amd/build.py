"""Builds libsfmhip.so (HIP kernels + C ABI, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the dev container; the .so travels to the
GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libsfmhip.so")
# per-source floating-point contraction: the matcher's exact kernel and the triangulation restate
# OpenCV's operation order (no compiler-chosen FMAs); the BA kernels are tolerance-level f64
# and want v_fma_f64 -- "fast-honor-pragmas", not "fast": the one function that must NOT contract, the trust-region decision
# lm_decide (the same bits on the host and on the device), says so with a pragma, which plain "fast" ignores
SOURCES = {"context.hip": "off", "match.hip": "off", "triangulate.hip": "off", "incremental.hip": "off",
           "score.hip": "off", "sift.hip": "off", "ba.hip": "fast-honor-pragmas", "probe.hip": "off"}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sfmhip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for name, contract in SOURCES.items():
        src = os.path.join(CSRC, name)
        obj = os.path.join(objdir, name.replace(".hip", ".o"))
        objs.append(obj)
        hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]  # (common.h, hypot_glibc.h, ba_front*.h)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(
                [os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "include", "sfmhip.h"))] +
                [os.path.getmtime(h) for h in hdrs]):
            continue
        cmd = [_hipcc()] + FLAGS + [f"-ffp-contract={contract}", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


def build_ba_variant(name, defines, force=False):
    """libsfmhip_<name>.so: the library with ba.hip compiled under extra -D switches, every other object shared with the product
    build (diagnostic and A/B builds: loaded through SFMHIP_SO, never by the product)."""
    build()
    objdir = os.path.join(HERE, "build")
    so = os.path.join(HERE, f"libsfmhip_{name}.so")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    if not force and os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(d) for d in deps):
        return so
    obj = os.path.join(objdir, f"ba_{name}.o")
    subprocess.check_call([_hipcc()] + FLAGS + [f"-ffp-contract={SOURCES['ba.hip']}"] + list(defines) +
                          ["-c", os.path.join(CSRC, "ba.hip"), "-o", obj])
    objs = [os.path.join(objdir, n.replace(".hip", ".o")) for n in SOURCES if n != "ba.hip"] + [obj]
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    return so


def build_timeout_diag(force=False):
    """Two diagnostic builds of the library in which one hand-off never arrives (test infrastructure: tests/test_gpu_geometry.py
    checks what the solver reports then; loaded through SFMHIP_SO, never by the product): libsfmhip_dbg_breakfront.so -- a child
    front of the tree never raises its flag in the one-launch form (-DSFM_FRONT_BREAK_HANDOFF) -- and libsfmhip_dbg_breakchol.so --
    an LDS hand-off inside chol_step2 never arrives (-DSFM_CHOL_BREAK_HANDOFF)."""
    return [build_ba_variant("dbg_breakfront", ["-DSFM_FRONT_BREAK_HANDOFF"], force),
            build_ba_variant("dbg_breakchol", ["-DSFM_CHOL_BREAK_HANDOFF"], force)]


def build_host_demo(force=False):
    """C++ host mirror (Sfm.h / BundleAdjustment.h call surface) + its self-test executable."""
    host = os.path.join(CSRC, "host")
    exe = os.path.join(HERE, "sfm_host_selftest")
    return _build_host_exe(exe, ("Sfm.cpp", "SfmIO.cpp", "BundleAdjustment.cpp", "selftest.cpp"), force)


def build_ba_demo(force=False):
    """BundleAdjustment::adjustBundle on a problem of any size in the reference's containers (tests/test_gpu_host_cpp.py: the
    write-back-only-on-CONVERGENCE policy at cfg4 size)."""
    return _build_host_exe(os.path.join(HERE, "sfm_ba_selftest"), ("Sfm.cpp", "SfmIO.cpp", "BundleAdjustment.cpp", "ba_selftest.cpp"), force)


def build_io_demo(force=False):
    """The I/O self-test of the host mirror (imagesLOAD / getCameraMatrix / PMVS2).
    Runs without a GPU: the host classes open the device only when a matching / BA call needs it."""
    return _build_host_exe(os.path.join(HERE, "sfm_io_selftest"), ("Sfm.cpp", "SfmIO.cpp", "BundleAdjustment.cpp", "io_selftest.cpp"), force)


def _build_host_exe(exe, files, force):
    host = os.path.join(CSRC, "host")
    srcs = [os.path.join(host, f) for f in files]
    deps = srcs + [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".h")]
    if not force and os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(s) for s in deps):
        return exe
    build()
    cmd = ["g++", "-O2", "-std=c++14", "-pthread", "-I", os.path.join(HERE, "..", "include"), "-I", host, "-o", exe] + srcs + \
          ["-L", HERE, "-lsfmhip", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    return exe


def build_rccl(force=False):
    """libsfmhip_rccl.so: the native RCCL binding of the sharded BA (include/sfmhip_rccl.h)."""
    src = os.path.join(CSRC, "rccl", "sfmhip_rccl.cpp")
    so = os.path.join(HERE, "libsfmhip_rccl.so")
    if not force and os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(src), os.path.getmtime(SO)):
        return so
    build()
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
           "-o", so, src, "-L", HERE, "-lsfmhip", "-L", os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.check_call(cmd)
    exe = os.path.join(HERE, "sfm_rccl_selftest")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-o", exe, os.path.join(CSRC, "rccl", "rccl_selftest.cpp"), "-L", HERE,
                           "-lsfmhip_rccl", "-lsfmhip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    return so


if __name__ == "__main__":
    print(build(force=True, verbose=True))
