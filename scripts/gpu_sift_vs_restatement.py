"""Ad-hoc GPU check (run through gpurun): where the device's SIFT output differs from the numpy restatement's committed fixture
(tests/golden/temple_sift.npz) on the ten temple frames -- per field, how many values are the same bit pattern."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from sfm_danpipeline_amd import features

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "temple_sift.npz"))
names = sorted(n for n in os.listdir(os.path.join(ROOT, "tests", "golden", "temple")) if n.endswith(".png"))
tot = np.zeros(6, np.int64)
nk = nd = dd = 0
ulps = [[] for _ in range(5)]
for i, name in enumerate(names):
    rgb = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "temple", name)).convert("RGB")).astype(np.int64)
    b, gr, r = rgb[..., 2], rgb[..., 1], rgb[..., 0]
    gray = ((b * 1868 + gr * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)
    kp, desc = features.sift_detect_and_compute(gray)
    kpo, do = g[f"kp{i}"], g[f"desc{i}"].astype(np.float32)
    assert kp.shape == kpo.shape, (name, kp.shape, kpo.shape)
    same = kp.view(np.int32) == kpo.view(np.int32)
    tot += same.sum(0)
    nk += len(kp)
    for f in range(5):
        d = np.abs(kp[:, f].view(np.int32).astype(np.int64) - kpo[:, f].view(np.int32).astype(np.int64))
        ulps[f].append(d[d > 0])
    nd += desc.size
    dd += int((desc != do).sum())
print(f"{nk} keypoints on {len(names)} frames; bit-identical per field [x, y, size, angle, response, octave]: {(tot / nk).round(5).tolist()}")
for f, nm in enumerate(("x", "y", "size", "angle", "response")):
    u = np.concatenate(ulps[f]) if ulps[f] else np.zeros(0)
    print(f"  {nm}: {len(u)} differ; ulp distance median {np.median(u) if len(u) else 0:.0f}, max {u.max() if len(u) else 0}")
print(f"descriptor entries that differ: {dd} of {nd} ({dd / nd:.2e})")
