"""GPU: the SIFT front end (reference src/Sfm.cpp:300-330) -- sfmhip_sift_detect_and_compute
against the numpy restatement of OpenCV 3.4.1's SIFT (oracle/sfm_oracle_sift.py; PARITY UNPINNED: OpenCV is not in
the image, see that file's header).  Float pipeline: keypoints must agree to float rounding, descriptors (integers
0..255) exactly but for entries on a rounding boundary."""
import numpy as np
import pytest

from oracle import sfm_oracle_sift as S
from sfm_danpipeline_amd import features

pytestmark = pytest.mark.gpu


def _blobs(h, w, n, seed, noise=2.0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w))
    for _ in range(n):
        cx, cy, s, a = rng.uniform(8, w - 8), rng.uniform(8, h - 8), rng.uniform(1.2, 6), rng.uniform(40, 200)
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return np.clip(img + rng.normal(0, noise, (h, w)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape,seed", [((96, 128), 0), ((75, 101), 1), ((160, 120), 2)])
def test_keypoints_and_descriptors_match_the_restatement(ctx, shape, seed):
    img = _blobs(shape[0], shape[1], 30, seed)
    K, D = features.sift_detect_and_compute(img, ctx=ctx)
    Ko, Do = S.detect_and_compute(img)
    assert len(Ko) > 20
    assert K.shape == Ko.shape and D.shape == Do.shape, (K.shape, Ko.shape)
    assert np.array_equal(K[:, 5].view(np.int32), Ko[:, 5].view(np.int32))               # octave / layer / xi bits
    assert np.allclose(K[:, :5], Ko[:, :5], rtol=2e-5, atol=2e-4)
    diff = np.abs(D - Do)
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3                                  # integers; a rounding boundary at most
    n = np.linalg.norm(D, axis=1)
    assert np.all((n > 480) & (n < 540)) and D.min() >= 0 and D.max() <= 255 and np.array_equal(D, np.rint(D))


def test_flat_and_tiny_images(ctx):
    K, D = features.sift_detect_and_compute(np.full((64, 64), 90, np.uint8), ctx=ctx)
    assert len(K) == 0 and D.shape == (0, 128)
    img = _blobs(24, 20, 3, 5)
    K, D = features.sift_detect_and_compute(img, ctx=ctx)
    Ko, Do = S.detect_and_compute(img)
    assert K.shape == Ko.shape and (len(K) == 0 or np.allclose(K[:, :5], Ko[:, :5], rtol=2e-5, atol=2e-4))


def test_descriptors_feed_the_matcher(ctx):
    """two views of the same blobs (a shift): SIFT -> getMatching finds the shift"""
    from sfm_danpipeline_amd import matcher
    big = _blobs(140, 180, 60, 9, noise=1.0)
    a, b = big[10:130, 10:170], big[14:134, 18:178]                 # b(x, y) = a(x + 8, y + 4)
    Ka, Da = features.sift_detect_and_compute(a, ctx=ctx)
    Kb, Db = features.sift_detect_and_compute(b, ctx=ctx)
    q, t, d = matcher.get_matching(Da, Db, ratio=0.8, ctx=ctx)
    assert len(q) >= 15
    shift = Ka[q, :2] - Kb[t, :2]
    good = (np.abs(shift[:, 0] - 8) < 1.0) & (np.abs(shift[:, 1] - 4) < 1.0)
    assert good.mean() > 0.8
