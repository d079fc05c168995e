"""CPU: the C-ABI library builds, loads and exports every symbol include/sfmhip.h declares; the
product path fails loudly without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import pytest

from sfm_danpipeline_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    return build.build()


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sfmhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sfmhip_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(so):
    L = ctypes.CDLL(so)
    missing = [s for s in _declared_symbols() if not hasattr(L, s)]
    assert not missing, missing


def test_version_and_error_strings(so):
    L = _lib.lib()
    assert L.sfmhip_version() == 1
    for code in (0, -1, -2, -3, -4, -5, -6, -7, -99):
        assert L.sfmhip_error_string(code)


def test_default_ba_options_match_the_reference(so):
    o = _lib.BaOpts()
    _lib.lib().sfmhip_ba_default_opts(ctypes.byref(o))
    assert o.max_iterations == 500 and o.max_time_s == 10.0  # src/BundleAdjustment.cpp:118,120
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)
    assert (o.initial_radius, o.min_relative_decrease, o.min_lm_diagonal, o.max_lm_diagonal) == (1e4, 1e-3, 1e-6, 1e32)


def test_no_gpu_means_loud_failure_not_fallback(so):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.SfmHipError):
        _lib.Context(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sfm_danpipeline_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in text and "import oracle" not in text and "sfm_oracle" not in text, f


def test_rccl_library_exports_what_its_header_declares():
    """libsfmhip_rccl.so (native RCCL binding of the sharded BA) builds, exports every symbol of
    include/sfmhip_rccl.h and really calls into RCCL (no compute: symbol tables only)."""
    import subprocess
    so = build.build_rccl()
    hdr = open(os.path.join(ROOT, "include", "sfmhip_rccl.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(sfmhip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sfmhip_[a-z0-9_]+)", syms))
    assert not [s for s in declared if s not in exported]
    assert re.search(r" U ncclAllReduce", syms) and re.search(r" U ncclCommInitRank", syms)
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True, check=True).stdout
    assert "librccl.so" in needed and "libsfmhip.so" in needed


def test_device_hypot_restatement_equals_the_hosts_libm(tmp_path):
    """csrc/hypot_glibc.h (what the device's Jacobi SVDs call instead of ocml's hypot) compiled for the host against
    this image's libm: bit-equal on random arguments, the scaling branches and the non-finite corners -- the five-point
    solver must follow its CPU checker to the last bit (an ill-conditioned sample amplifies one ulp to 1e-3 of E)"""
    import subprocess
    src = tmp_path / "h.cpp"
    src.write_text(r"""
#include "hypot_glibc.h"
#include <cstdio>
#include <cstring>
#include <random>
int main() {
  std::mt19937_64 g(11);
  long bad = 0;
  for (int i = 0; i < 2000000; ++i) {
    double a = std::ldexp(std::generate_canonical<double, 53>(g) + 0.5, (int)(g() % 60) - 30) * ((g() & 1) ? 1 : -1);
    double b = std::ldexp(std::generate_canonical<double, 53>(g) + 0.5, (int)(g() % 60) - 30);
    switch (i % 1000) {
      case 0: a = std::ldexp(a, 520); break;
      case 1: b = std::ldexp(b, -500); break;
      case 2: a = b; break;
      case 3: a = 0; break;
      case 4: b = NAN; break;
      case 5: a = INFINITY; b = NAN; break;
      case 6: a = std::ldexp(a, -480); b = std::ldexp(b, -490); break;
      case 7: a = std::ldexp(a, 530); b = std::ldexp(b, 525); break;
    }
    const double h = std::hypot(a, b), k = sfm_hypot(a, b);
    if (std::memcmp(&h, &k, 8) != 0 && !(std::isnan(h) && std::isnan(k))) ++bad;
  }
  std::printf("%ld\n", bad);
  return 0;
}
""")
    exe = tmp_path / "h"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "sfm_danpipeline_amd", "csrc"), "-o", str(exe), str(src)],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    assert out.strip() == "0", out
