"""CPU: the numpy restatement of SIFT::detectAndCompute (oracle/sfm_oracle_sift.py) against properties that do not
need OpenCV: kernel construction, the blur of an impulse, scale selection on Gaussian blobs, descriptor normalisation,
behaviour under a shift of the image content."""
import numpy as np

from oracle import sfm_oracle_sift as S


def _blob_image(h, w, blobs, noise=0.0, seed=0):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w))
    for cx, cy, s, a in blobs:
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    if noise:
        img += np.random.default_rng(seed).normal(0, noise, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_gaussian_kernel_and_blur():
    k = S.gaussian_kernel(1.6)
    assert len(k) == 15 and k.dtype == np.float32 and abs(k.sum() - 1) < 1e-6 and np.array_equal(k, k[::-1])   # cvRound(13.8) | 1
    assert len(S.gaussian_kernel(1.2262734984654078)) == 11
    img = np.zeros((31, 31), np.float32)
    img[15, 15] = 1
    out = S.gaussian_blur(img, 1.6)
    assert np.allclose(out[8:23, 8:23], np.outer(k, k), atol=1e-8) and abs(out.sum() - 1) < 1e-5
    tiny = np.arange(12, dtype=np.float32).reshape(3, 4)          # kernel radius > image: reflect-101 repeated
    assert S.gaussian_blur(tiny, 3.0).shape == (3, 4) and np.isfinite(S.gaussian_blur(tiny, 3.0)).all()


def test_pyramid_shapes():
    gp, dog, n_oct = S.build_pyramids(_blob_image(60, 80, [(40, 30, 3, 200)]))
    assert n_oct == int(np.rint(np.log2(120) - 2)) + 1 == 6 and len(gp) == 6 * n_oct and len(dog) == 5 * n_oct
    assert gp[0].shape == (120, 160) and gp[6].shape == (60, 80) and gp[12].shape == (30, 40)
    assert np.array_equal(gp[6], gp[3][::2, ::2])                # the next octave starts from layer nOctaveLayers, halved


def test_blobs_are_found_at_their_scale_and_place():
    blobs = [(30, 30, 2.0, 200), (80, 40, 4.0, 200), (45, 80, 6.0, 200)]
    K, D = S.detect_and_compute(_blob_image(110, 120, blobs))
    assert len(K) >= 3 and D.shape == (len(K), 128)
    for cx, cy, s, _ in blobs:
        d = np.hypot(K[:, 0] - cx, K[:, 1] - cy)
        i = int(np.argmin(d))
        assert d[i] < 1.0
        assert 0.8 < K[i, 2] / (2 * s) < 1.0                   # size = 2 x the detection scale ~ 0.89 x 2 x the blob's sigma, at every scale
    n = np.linalg.norm(D, axis=1)
    assert np.all((n > 480) & (n < 540)) and np.array_equal(D, np.rint(D)) and D.max() <= 255 and D.min() >= 0
    order = [tuple(k[:2]) for k in K]
    assert order == sorted(order)                                # removeDuplicatedSorted: by x, then y


def test_descriptors_survive_a_shift():
    rng = np.random.default_rng(3)
    blobs = [(rng.uniform(20, 140), rng.uniform(20, 100), rng.uniform(1.5, 5), rng.uniform(80, 200)) for _ in range(25)]
    big = _blob_image(120, 160, blobs, noise=1.0, seed=1)
    a, b = big[8:104, 8:136], big[12:108, 16:144]               # b(x, y) = a(x + 8, y + 4)
    Ka, Da = S.detect_and_compute(a)
    Kb, Db = S.detect_and_compute(b)
    d2 = ((Da[:, None, :] - Db[None, :, :]) ** 2).sum(2)
    nn = d2.argmin(1)
    srt = np.sort(d2, 1)
    keep = srt[:, 0] < 0.64 * srt[:, 1]                          # the ratio test of getMatching on squared distances
    shift = Ka[keep, :2] - Kb[nn[keep], :2]
    assert keep.sum() >= 10 and (np.abs(shift - [8, 4]).max(1) < 1.0).mean() > 0.8
