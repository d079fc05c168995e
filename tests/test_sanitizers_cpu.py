"""CPU: AddressSanitizer + UBSan runs (`make -C oracle asan`) of the oracle's entry points and of the C++ host
mirror of the reference's call surface (csrc/host/*.cpp + selftest.cpp).  The host mirror runs against
tests/stub/sfmhip_stub.c, a stand-in for the C ABI answered by the oracle: sanitizers are not available on the
GPU pool, so this is where the host-side container handling gets its memory-safety check."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)


def _clean(r):
    bad = [m for m in ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:") if m in r.stderr]
    assert r.returncode == 0 and not bad, r.stderr[-3000:]


def test_oracle_entry_points_under_asan_ubsan():
    _build()
    r = subprocess.run([os.path.join(ROOT, "oracle", "_asan", "oracle_asan_driver")], capture_output=True, text=True,
                       timeout=600, env=ENV)
    _clean(r)
    assert "oracle asan driver ok" in r.stdout


def test_cpp_host_mirror_under_asan_ubsan(tmp_path, orc):
    from tests import hostcpp_io
    _build()
    w = hostcpp_io.write_input(tmp_path)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_asan", "host_selftest_asan"), str(tmp_path / "in.bin"),
                        str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=600, env=ENV)
    _clean(r)
    hostcpp_io.check_output(tmp_path, orc, w, scored_by_stub=True)     # and the containers it filled are the oracle's results


def test_match_all_pairs_falls_back_to_the_copying_fetch(tmp_path, orc):
    """matchAllPairs (csrc/host/Sfm.cpp) when the pinned buffers of the pipelined fetch are refused, and when a batch holds
    more matches than they do: the pass goes on through sfmhip_matchplan_fetch and fills the same containers (round-3
    advisor: a refused hipHostMalloc aborted the whole pass).  Under ASan / UBSan, against the stub."""
    from tests import hostcpp_io
    _build()
    w = hostcpp_io.write_input(tmp_path)
    exe = os.path.join(ROOT, "oracle", "_asan", "host_selftest_asan")
    outs = {}
    for mode in ("", "refuse", "small"):
        out = tmp_path / f"out_{mode or 'piped'}.bin"
        r = subprocess.run([exe, str(tmp_path / "in.bin"), str(out)], capture_output=True, text=True, timeout=600,
                           env=dict(ENV, SFMHIP_STUB_PIPELINE=mode))
        _clean(r)
        assert ("copying fetch" in r.stderr) == (mode == "refuse"), r.stderr[-500:]
        outs[mode] = out.read_bytes()
    assert outs["refuse"] == outs[""] and outs["small"] == outs[""]


def test_host_io_under_asan_ubsan(tmp_path):
    """imagesLOAD / getCameraMatrix / PMVS2 (csrc/host/SfmIO.cpp: PNG inflate + unfilter, XML, file export) on
    good, truncated and corrupt inputs."""
    import numpy as np
    from tests import test_host_io as hio
    _build()
    exe = os.path.join(ROOT, "oracle", "_asan", "io_selftest_asan")
    d, _ = hio._make_dir(tmp_path, np.random.default_rng(8))
    (tmp_path / "cam.xml").write_text(hio.XML)
    r = subprocess.run([exe, str(d), str(tmp_path / "cam.xml"), str(tmp_path / "o.bin")], capture_output=True, text=True,
                       timeout=600, env=ENV, cwd=tmp_path)
    _clean(r)
    assert os.path.exists(tmp_path / "denseCloud" / "txt" / "0000.txt")
    rng = np.random.default_rng(9)
    goods = [(d / "b_rgb.PNG").read_bytes(),
             hio._png_interlaced(rng.integers(0, 256, (23, 31, 3)), 2, 8),           # Adam7: ragged passes
             hio._png_interlaced(rng.integers(0, 4, (9, 14, 1)), 0, 2)]               # ... below a byte per pixel
    for k in range(36):                      # corrupt / truncated streams must fail cleanly, never read out of bounds
        good = goods[k % 3]
        bad = tmp_path / ("bad%d" % k)
        bad.mkdir()
        (bad / "a.png").write_bytes(good)
        b = bytearray(good)
        if k % 3 == 0:
            b = b[:int(rng.integers(9, len(b)))]
        else:
            for _ in range(1 + k % 5):
                b[int(rng.integers(8, len(b)))] ^= int(rng.integers(1, 256))
        (bad / "b.png").write_bytes(bytes(b))
        r = subprocess.run([exe, str(bad), str(tmp_path / "cam.xml"), str(tmp_path / "o.bin")], capture_output=True, text=True,
                           timeout=600, env=ENV, cwd=tmp_path)
        _clean(r)


def test_jpeg_decoder_under_asan_ubsan(tmp_path):
    """the JPEG decoder of imagesLOAD on good, truncated and bit-flipped streams (sequential and progressive; headers, tables
    and entropy-coded data alike): it fails cleanly or decodes garbage pixels, never reads or writes out of bounds"""
    import io
    import numpy as np
    from PIL import Image
    from tests import test_host_io as hio
    _build()
    exe = os.path.join(ROOT, "oracle", "_asan", "io_selftest_asan")
    (tmp_path / "cam.xml").write_text(hio.XML)
    rng = np.random.default_rng(10)
    arr = np.clip(rng.normal(128, 60, (70, 90, 3)), 0, 255).astype(np.uint8)
    goods = []
    for kw in (dict(quality=80, subsampling=2), dict(quality=90, subsampling=0, restart_marker_blocks=2), dict(quality=70, subsampling=1),
               dict(quality=85, subsampling=2, progressive=True), dict(quality=60, subsampling=0, progressive=True, restart_marker_blocks=3)):
        b = io.BytesIO()
        Image.fromarray(arr).save(b, "JPEG", **kw)
        goods.append(b.getvalue())
    ex = Image.Exif()                      # an EXIF directory in front (orientation 6): the tag reader's bounds
    ex[0x0112] = 6
    ex[0x010F] = "a camera maker's name, so that the directory points past its entries"
    b = io.BytesIO()
    Image.fromarray(arr).save(b, "JPEG", quality=75, exif=ex.tobytes())
    goods.append(b.getvalue())
    for k in range(72):
        good = goods[k % 6]
        bad = tmp_path / ("jbad%d" % k)
        bad.mkdir()
        (bad / "a.jpg").write_bytes(good)
        b = bytearray(good)
        if k % 4 == 0:
            b = b[:int(rng.integers(4, len(b)))]
        else:
            lo = 2 if k % 4 == 1 or k % 6 == 5 else len(b) // 2            # headers / tables (the EXIF file: always), or the scan
            for _ in range(1 + k % 7):
                b[int(rng.integers(lo, len(b)))] ^= int(rng.integers(1, 256))
        (bad / "b.jpg").write_bytes(bytes(b))
        r = subprocess.run([exe, str(bad), str(tmp_path / "cam.xml"), str(tmp_path / "o.bin")], capture_output=True, text=True,
                           timeout=600, env=ENV, cwd=tmp_path)
        bad_msgs = [m for m in ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:") if m in r.stderr]
        assert not bad_msgs and r.returncode == 0, r.stderr[-3000:]
