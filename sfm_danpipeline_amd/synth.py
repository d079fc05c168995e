"""Seeded synthetic workloads of the BASELINE.json configs.

Pure numpy, no oracle and no GPU use: the same arrays feed the HIP path, the oracle (in tests)
and the CPU-baseline leg of bench.py.
"""
import numpy as np


def sift_image_set(n_images=50, n_feat=2000, dim=128, bank=20000, sigma=6.0, seed=1234):
    """cfg2: integer-valued 0..255 float32 rows, as OpenCV SIFT emits (src/Sfm.cpp:315-327).
    Image i draws n_feat distinct rows of a shared bank, adds round(N(0,sigma)), clamps, shuffles."""
    rng = np.random.default_rng(seed)
    W = rng.integers(0, 256, size=(bank, dim), dtype=np.int16)
    out = []
    for i in range(n_images):
        r = np.random.default_rng(seed + 1 + i)
        rows = r.choice(bank, size=n_feat, replace=False)
        d = W[rows] + np.rint(r.normal(0.0, sigma, size=(n_feat, dim))).astype(np.int16)
        d = np.clip(d, 0, 255)
        r.shuffle(d, axis=0)
        out.append(np.ascontiguousarray(d, dtype=np.float32))
    return out


def orb_image_set(n_images=500, n_feat=5000, nbytes=32, bank=100000, p_flip=0.04, seed=4321):
    """cfg5: ORB-style 256-bit rows (n_feat x 32 uint8, src/Sfm.cpp:360-375 layout)."""
    rng = np.random.default_rng(seed)
    W = rng.integers(0, 256, size=(bank, nbytes), dtype=np.uint8)
    out = []
    for i in range(n_images):
        r = np.random.default_rng(seed + 1 + i)
        rows = r.choice(bank, size=n_feat, replace=False)
        flips = np.packbits(r.random((n_feat, nbytes * 8)) < p_flip, axis=1)
        out.append(np.ascontiguousarray(W[rows] ^ flips))
    return out


def all_pairs(n_images):
    """The q<t loop of findBestPair (src/Sfm.cpp:511-512), in its iteration order."""
    q, t = np.triu_indices(n_images, k=1)
    return np.stack([q, t], axis=1).astype(np.int32)


def _rodrigues_to_aa(R):
    """Angle-axis of a rotation matrix (numpy; generator-side only)."""
    c = (np.trace(R) - 1.0) / 2.0
    c = min(1.0, max(-1.0, c))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    n = np.linalg.norm(v)
    if n < 1e-12:
        return np.zeros(3)
    return v / n * th


def ba_problem(n_cam=200, n_pt=100000, obs_per_pt=10, seed=777, focal=1520.0, noise_px=0.5,
               pt_sigma=1e-2, cam_sigma=1e-3, focal_factor=1.01):
    """cfg3/cfg4: cameras on a ring of radius 10 looking at the origin (+-5 % radial/height
    jitter), points uniform in the unit ball, each observed by `obs_per_pt` consecutive cameras
    starting at a uniformly drawn index; principal point already subtracted
    (src/BundleAdjustment.cpp:94-97).  Returns a dict with the truth, the perturbed start and
    the observation triples in the order the reference adds residual blocks (point-major,
    ascending camera id inside a point -- std::map order, src/BundleAdjustment.cpp:87)."""
    rng = np.random.default_rng(seed)
    ang = 2 * np.pi * np.arange(n_cam) / n_cam
    rad = 10.0 * (1 + rng.uniform(-0.05, 0.05, n_cam))
    hgt = 10.0 * rng.uniform(-0.05, 0.05, n_cam)
    C = np.stack([rad * np.cos(ang), hgt, rad * np.sin(ang)], axis=1)
    cams = np.zeros((n_cam, 6))
    Rs = np.zeros((n_cam, 3, 3))
    for i in range(n_cam):
        z = -C[i] / np.linalg.norm(C[i])
        up = np.array([0.0, 1.0, 0.0])
        x = np.cross(up, z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        Rs[i] = R
        cams[i, :3] = _rodrigues_to_aa(R)
        cams[i, 3:] = -R @ C[i]
    # points uniform in the unit ball
    v = rng.normal(size=(n_pt, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    pts = v * rng.uniform(0, 1, (n_pt, 1)) ** (1.0 / 3.0)
    start = rng.integers(0, n_cam, n_pt)
    k = min(obs_per_pt, n_cam)
    cam_idx = (start[:, None] + np.arange(k)[None, :]) % n_cam
    cam_idx.sort(axis=1)  # std::map<const int,int> iteration order
    obs_pt = np.repeat(np.arange(n_pt, dtype=np.int32), k)
    obs_cam = cam_idx.reshape(-1).astype(np.int32)
    P = np.einsum("oij,oj->oi", Rs[obs_cam], pts[obs_pt]) + cams[obs_cam, 3:]
    xy = focal * P[:, :2] / P[:, 2:3] + rng.normal(0, noise_px, (obs_pt.size, 2))
    cams0 = cams + rng.normal(0, cam_sigma, cams.shape)
    pts0 = pts + rng.normal(0, pt_sigma, pts.shape)
    return dict(n_cam=n_cam, n_pt=n_pt, n_obs=int(obs_pt.size), cams_true=cams, pts_true=pts,
                focal_true=focal, cams0=cams0, pts0=pts0, focal0=focal * focal_factor,
                obs_cam=obs_cam, obs_pt=obs_pt, obs_xy=np.ascontiguousarray(xy))


def two_view_scene(m=500, seed=99, K=None, noise_px=0.3, outlier_frac=0.1):
    """A seeded two-view scene for triangulateViews (src/Sfm.cpp:804-878): P1=[I|0] like the
    reference's base pair (src/Sfm.cpp:432), P2 a small rotation + baseline."""
    rng = np.random.default_rng(seed)
    if K is None:
        K = np.array([[1520.0, 0, 302.2], [0, 1520.0, 246.87], [0, 0, 1]])  # calibration xml:8-10
    X = np.stack([rng.uniform(-1, 1, m), rng.uniform(-1, 1, m), rng.uniform(4, 8, m)], axis=1)
    P1 = np.hstack([np.eye(3), np.zeros((3, 1))])
    a = np.array([0.02, -0.15, 0.01])
    th = np.linalg.norm(a)
    k = a / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    P2 = np.hstack([R, np.array([[-1.0], [0.05], [0.1]])])

    def proj(P):
        p = X @ P[:, :3].T + P[:, 3]
        return (p[:, :2] / p[:, 2:3]) @ K[:2, :2].T + K[:2, 2]

    xy1 = proj(P1) + rng.normal(0, noise_px, (m, 2))
    xy2 = proj(P2) + rng.normal(0, noise_px, (m, 2))
    bad = rng.random(m) < outlier_frac
    xy2[bad] += rng.uniform(-60, 60, (int(bad.sum()), 2))
    return dict(P1=P1, P2=P2, K=K, dist=np.zeros(5), xy1=xy1, xy2=xy2, X_true=X)


def random_tracks_and_matches(n_cloud, n_views, n_matches, seed=0, done_view=0, n_feat=400):
    """Cloud tracks ({view: feature} per point, 1..4 views each) and a match list (q, t) between
    done_view and another view; features of done_view repeat on the train side so that the
    'first match wins' rule of find2D3DMatches (reference src/Sfm.cpp:1061-1084) is exercised."""
    rng = np.random.default_rng(seed)
    cloud = []
    for _ in range(n_cloud):
        k = int(rng.integers(1, 5))
        views = rng.choice(n_views, size=k, replace=False)
        cloud.append({int(v): int(rng.integers(0, n_feat)) for v in views})
    q = rng.permutation(n_feat)[:n_matches]                  # queryIdx: unique, ascending like getMatching
    q.sort()
    t = rng.integers(0, n_feat // 2, n_matches)              # trainIdx: repeats
    return cloud, [(int(a), int(b)) for a, b in zip(q, t)]


def tracks_to_csr(cloud):
    ptr = np.zeros(len(cloud) + 1, np.int32)
    views, feats = [], []
    for i, tr in enumerate(cloud):
        for v in sorted(tr):
            views.append(v)
            feats.append(tr[v])
        ptr[i + 1] = len(views)
    return ptr, np.asarray(views, np.int32), np.asarray(feats, np.int32)


def match_mix(q, t, dist):
    """A 64-bit mix of one match (queryIdx, trainIdx, bit pattern of the float distance), vectorised: the per-match value
    of the pair checksums that full-size runs compare instead of whole lists (the C checker computes the same function)."""
    with np.errstate(over="ignore"):
        x = q.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + t.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F) + \
            np.ascontiguousarray(dist, np.float32).view(np.uint32).astype(np.uint64) * np.uint64(0x165667B19E3779F9)
        x ^= x >> np.uint64(29)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(32)
    return x


def pair_checksums(counts, oq, ot, od):
    """(n_pairs, 2) uint64 [sum, xor] of match_mix over each pair's slice of concatenated match lists."""
    x = match_mix(np.asarray(oq), np.asarray(ot), np.asarray(od))
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    cs = np.zeros((len(counts), 2), np.uint64)
    with np.errstate(over="ignore"):
        for p in range(len(counts)):
            seg = x[off[p]:off[p + 1]]
            if len(seg):
                cs[p, 0] = np.sum(seg, dtype=np.uint64)
                cs[p, 1] = np.bitwise_xor.reduce(seg)
    return cs


def _aa_to_rotation(aa):
    """Rodrigues' formula (generator-side only: the containers below want [R|t])."""
    th = np.linalg.norm(aa)
    if th == 0:
        return np.eye(3)
    k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def write_ba_containers(path, pb, cx, cy):
    """A ba_problem in the binary layout csrc/host/ba_selftest.cpp reads into the REFERENCE's containers (cv::Matx34d poses,
    Point3D cloud with std::map tracks, Intrinsics, per-view feature lists -- the arguments of BundleAdjustment::adjustBundle,
    include/BundleAdjustment.h:19-20): i32 n_cam, n_pt, n_obs | poses [R|t] | points | K | (view, point) per observation |
    pixel coordinates with the principal point added back.  Returns (poses, K) as written."""
    import struct
    nc, npt, no = pb["n_cam"], pb["n_pt"], pb["n_obs"]
    poses = np.zeros((nc, 3, 4))
    for i in range(nc):
        poses[i, :, :3] = _aa_to_rotation(pb["cams0"][i, :3])
        poses[i, :, 3] = pb["cams0"][i, 3:]
    K = np.array([[pb["focal0"], 0, cx], [0, pb["focal0"], cy], [0, 0, 1.0]])
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", nc, npt, no))
        f.write(poses.astype("<f8").tobytes())
        f.write(pb["pts0"].astype("<f8").tobytes())
        f.write(K.astype("<f8").tobytes())
        f.write(np.stack([pb["obs_cam"], pb["obs_pt"]], axis=1).astype("<i4").tobytes())
        f.write((pb["obs_xy"] + np.array([cx, cy])).astype("<f8").tobytes())
    return poses, K
