"""Host-side mirror of BundleAdjustment::adjustBundle (reference src/BundleAdjustment.cpp:46-175,
decl include/BundleAdjustment.h:19-20) over the C ABI.

`adjust_bundle` keeps the reference's argument meaning and policies: [R|t] -> (angle-axis, t)
6-vectors with all-zero-diagonal poses treated as empty (:59-63, :142-145), one shared focal
taken from K(0,0) (:79), one residual block per (camera, feature) of every Point3D::idxImage
with the principal point subtracted (:83-110), write-back ONLY when the solver reports
CONVERGENCE (:126-129), K(0,0)=K(1,1)=focal (:133-134).

`BaProblem` is the persistent device object (one per rank) used by bench.py and the sharded
multi-GPU path; the rotation helpers restate ceres/rotation.h for the host-side conversions.
"""
import ctypes as C
import math
import sys

import numpy as np

from . import _lib
from ._lib import BA_CONVERGENCE, BaOpts, BaSummary, check, lib

_EPS = np.finfo(np.float64).eps


# ----------------------------------------------------------------------------- ceres/rotation.h
def rotation_matrix_to_angle_axis(R):
    """ceres::RotationMatrixToAngleAxis (via the quaternion, 4-branch trace test) of a
    mathematical rotation matrix R (numpy row-major)."""
    R = np.asarray(R, np.float64)
    q = [0.0, 0.0, 0.0, 0.0]
    trace = R[0, 0] + R[1, 1] + R[2, 2]
    if trace >= 0.0:
        t = math.sqrt(trace + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1] = (R[2, 1] - R[1, 2]) * t
        q[2] = (R[0, 2] - R[2, 0]) * t
        q[3] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i + 1] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[j + 1] = (R[j, i] + R[i, j]) * t
        q[k + 1] = (R[k, i] + R[i, k]) * t
    s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
    if s2 > 0.0:
        st = math.sqrt(s2)
        ct = q[0]
        two_theta = 2.0 * (math.atan2(-st, -ct) if ct < 0.0 else math.atan2(st, ct))
        k = two_theta / st
    else:
        k = 2.0
    return np.array([q[1] * k, q[2] * k, q[3] * k])


def angle_axis_to_rotation_matrix(aa):
    """ceres::AngleAxisToRotationMatrix; returns the mathematical R (numpy row-major)."""
    aa = np.asarray(aa, np.float64)
    theta2 = float(aa @ aa)
    R = np.empty((3, 3))
    if theta2 > _EPS:
        theta = math.sqrt(theta2)
        wx, wy, wz = aa / theta
        c, s = math.cos(theta), math.sin(theta)
        R[0, 0] = c + wx * wx * (1.0 - c)
        R[1, 0] = wz * s + wx * wy * (1.0 - c)
        R[2, 0] = -wy * s + wx * wz * (1.0 - c)
        R[0, 1] = wx * wy * (1.0 - c) - wz * s
        R[1, 1] = c + wy * wy * (1.0 - c)
        R[2, 1] = wx * s + wy * wz * (1.0 - c)
        R[0, 2] = wy * s + wx * wz * (1.0 - c)
        R[1, 2] = -wx * s + wy * wz * (1.0 - c)
        R[2, 2] = c + wz * wz * (1.0 - c)
    else:
        R[:] = [[1.0, -aa[2], aa[1]], [aa[2], 1.0, -aa[0]], [-aa[1], aa[0], 1.0]]
    return R


def default_opts(**kw):
    o = BaOpts()
    lib().sfmhip_ba_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


# ----------------------------------------------------------------------------- device problem
def linearize_obs(cams6, pts3, focal, obs_xy, ctx=None):
    """Test hook: residual r (n,2) and Jacobian blocks Jc (n,2,6), Jp (n,2,3), Jf (n,2) of n single
    observations, computed by the solver's own device linearisation (sfmhip_ba_linearize_obs)."""
    from ._lib import default_context
    ctx = ctx or default_context()
    cams = np.ascontiguousarray(cams6, np.float64).reshape(-1, 6)
    pts = np.ascontiguousarray(pts3, np.float64).reshape(-1, 3)
    xy = np.ascontiguousarray(obs_xy, np.float64).reshape(-1, 2)
    n = cams.shape[0]
    assert pts.shape[0] == n and xy.shape[0] == n
    r, Jc, Jp, Jf = np.empty((n, 2)), np.empty((n, 2, 6)), np.empty((n, 2, 3)), np.empty((n, 2))
    check(lib().sfmhip_ba_linearize_obs(ctx.h, n, cams.ctypes.data, pts.ctypes.data, float(focal), xy.ctypes.data,
                                        r.ctypes.data, Jc.ctypes.data, Jp.ctypes.data, Jf.ctypes.data),
          "sfmhip_ba_linearize_obs")
    return r, Jc, Jp, Jf


class BaProblem:
    """sfmhip_ba object: this rank's points + observations, all cameras and the shared focal."""

    def __init__(self, n_cam, n_pt, obs_cam, obs_pt, obs_xy, ctx=None):
        self.ctx = ctx or _lib.default_context()
        self.n_cam, self.n_pt = int(n_cam), int(n_pt)
        oc = np.ascontiguousarray(obs_cam, np.int32)
        op = np.ascontiguousarray(obs_pt, np.int32)
        xy = np.ascontiguousarray(obs_xy, np.float64).reshape(-1, 2)
        self.n_obs = int(oc.shape[0])
        self.h = C.c_void_p()
        check(lib().sfmhip_ba_create(self.ctx.h, self.n_cam, self.n_pt, self.n_obs, oc.ctypes.data, op.ctypes.data,
                                     xy.ctypes.data, C.byref(self.h)), "sfmhip_ba_create")
        self._cb = None

    def set_allreduce(self, fn, rank, world):
        """fn(device_ptr:int, count:int) must sum `count` float64 in place across ranks."""
        def _tramp(ptr, count, _user):
            try:
                fn(int(ptr), int(count))
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print(f"[sfm_danpipeline_amd] all-reduce callback failed: {e!r}", file=sys.stderr)
                return 1
        self._cb = _lib.ALLREDUCE_FN(_tramp)
        check(lib().sfmhip_ba_set_allreduce(self.h, self._cb, None, int(rank), int(world)), "sfmhip_ba_set_allreduce")

    def set_params(self, cams6, pts3, focal):
        c = np.ascontiguousarray(cams6, np.float64).reshape(self.n_cam, 6)
        p = np.ascontiguousarray(pts3, np.float64).reshape(self.n_pt, 3)
        check(lib().sfmhip_ba_set_params(self.h, c.ctypes.data, p.ctypes.data if self.n_pt else None, float(focal)),
              "sfmhip_ba_set_params")

    def get_params(self):
        c = np.empty((self.n_cam, 6))
        p = np.empty((max(self.n_pt, 1), 3))
        f = C.c_double(0.0)
        check(lib().sfmhip_ba_get_params(self.h, c.ctypes.data, p.ctypes.data, C.addressof(f)), "sfmhip_ba_get_params")
        return c, p[:self.n_pt], f.value

    def run(self, opts=None):
        s = BaSummary()
        opts = opts or default_opts()
        check(lib().sfmhip_ba_run(self.h, C.byref(opts), C.byref(s)), "sfmhip_ba_run")
        return s

    def iterate(self, iters):
        s = BaSummary()
        check(lib().sfmhip_ba_iterate(self.h, int(iters), C.byref(s)), "sfmhip_ba_iterate")
        return s

    def reduced_system(self, radius=1e4):
        dim = 6 * self.n_cam + 1
        S = np.empty((dim, dim))
        g = np.empty(dim)
        cost = C.c_double(0.0)
        check(lib().sfmhip_ba_reduced_system(self.h, float(radius), S.ctypes.data, g.ctypes.data, C.addressof(cost)),
              "sfmhip_ba_reduced_system")
        return S, g, cost.value

    def last_timing(self):
        t = np.zeros(4)
        n = C.c_int(0)
        check(lib().sfmhip_ba_last_timing(self.h, t.ctypes.data, C.addressof(n)), "sfmhip_ba_last_timing")
        return dict(eliminate_s=t[0], allreduce_s=t[1], solve_s=t[2], backsub_s=t[3], launches=n.value)

    def reduced_step(self, radius):
        """z with (S + D/radius) z = g from the solver's own factorisation (sfmhip_ba_reduced_step); returns
        (z, chol_failed)."""
        z = np.zeros(6 * self.n_cam + 1)
        info = C.c_int(0)
        check(lib().sfmhip_ba_reduced_step(self.h, float(radius), z.ctypes.data, C.addressof(info)), "sfmhip_ba_reduced_step")
        return z, info.value

    def reduced_layout(self):
        """How the reduced camera system is factored (sfmhip_ba_reduced_layout): chains = 0 is the dense
        factorisation; valid after the first run / iterate."""
        a = np.zeros(4, np.int32)
        check(lib().sfmhip_ba_reduced_layout(self.h, a.ctypes.data), "sfmhip_ba_reduced_layout")
        return dict(chains=int(a[0]), chain_tiles=int(a[1]), separator_tiles=int(a[2]), dense_tiles=int(a[3]))

    def reduced_tree(self):
        """The front tree of the reduced solve (sfmhip_ba_reduced_tree): fronts = 0 when the chains / dense plans run."""
        a = np.zeros(4, np.int32)
        check(lib().sfmhip_ba_reduced_tree(self.h, a.ctypes.data), "sfmhip_ba_reduced_tree")
        return dict(fronts=int(a[0]), levels=int(a[1]), chain_tiles=int(a[2]), max_front_tiles=int(a[3]))

    def close(self):
        if self.h:
            lib().sfmhip_ba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ba_solve(cams6, pts3, focal, obs_cam, obs_pt, obs_xy, opts=None, ctx=None):
    """One-shot sfmhip_ba_solve; returns (cams6, pts3, focal, summary), inputs untouched."""
    ctx = ctx or _lib.default_context()
    cams = np.array(cams6, np.float64, order="C").reshape(-1, 6)
    pts = np.array(pts3, np.float64, order="C").reshape(-1, 3)
    f = C.c_double(float(focal))
    oc = np.ascontiguousarray(obs_cam, np.int32)
    op = np.ascontiguousarray(obs_pt, np.int32)
    xy = np.ascontiguousarray(obs_xy, np.float64).reshape(-1, 2)
    opts = opts or default_opts()
    s = BaSummary()
    check(lib().sfmhip_ba_solve(ctx.h, cams.shape[0], pts.shape[0], oc.shape[0], cams.ctypes.data,
                                pts.ctypes.data if pts.shape[0] else None, C.addressof(f), oc.ctypes.data,
                                op.ctypes.data, xy.ctypes.data, C.byref(opts), C.byref(s)), "sfmhip_ba_solve")
    return cams, pts, f.value, s


def last_solve_profile(ctx=None):
    """Stage times (ms) of the last `ba_solve` on the context, and whether it reused the previous call's plan."""
    ctx = ctx or _lib.default_context()
    p = _lib.BaSolveProfile()
    check(lib().sfmhip_ba_last_solve_profile(ctx.h, C.byref(p)), "sfmhip_ba_last_solve_profile")
    return {k: getattr(p, k) for k, _ in p._fields_ if k != "pad"}


# ----------------------------------------------------------------------------- adjustBundle
def adjust_bundle(point_cloud, camera_poses, K, image2d_features, opts=None, ctx=None, solver=None, log=None):
    """BundleAdjustment::adjustBundle.  Mutates `point_cloud[i]['pt']`, `camera_poses[i]` (3x4
    numpy arrays) and `K` in place, and only on CONVERGENCE; returns the solver summary.

    point_cloud: list of {'pt': (x,y,z), 'idxImage': {view: featIdx}, ...} (Point3D,
    include/Utilities.h:37-43); camera_poses: list of 3x4 arrays (cv::Matx34d);
    image2d_features[view][feat] = (u, v).  `solver` lets tests substitute another backend
    with ba_solve's signature."""
    log = log or (lambda m: print(m, file=sys.stderr))
    n_cam = len(camera_poses)
    cams6 = np.zeros((n_cam, 6))
    empty = np.zeros(n_cam, bool)
    for i, pose in enumerate(camera_poses):
        pose = np.asarray(pose, np.float64)
        if pose[0, 0] == 0 and pose[1, 1] == 0 and pose[2, 2] == 0:
            empty[i] = True  # src/BundleAdjustment.cpp:59-63
            continue
        cams6[i, :3] = rotation_matrix_to_angle_axis(pose[:, :3])  # :64-67 (R.t().val is col-major R)
        cams6[i, 3:] = pose[:, 3]
    focal = float(K[0, 0])
    cx, cy = float(K[0, 2]), float(K[1, 2])
    pts3 = np.array([p["pt"] for p in point_cloud], np.float64).reshape(-1, 3)
    obs_cam, obs_pt, obs_xy = [], [], []
    for i, p in enumerate(point_cloud):
        for view in sorted(p["idxImage"]):  # std::map iteration order
            u, v = image2d_features[view][p["idxImage"][view]]
            obs_cam.append(view)
            obs_pt.append(i)
            obs_xy.append((u - cx, v - cy))
    solver = solver or (lambda *a: ba_solve(*a, opts=opts, ctx=ctx))
    cams_o, pts_o, focal_o, summary = solver(cams6, pts3, focal, np.array(obs_cam, np.int32),
                                             np.array(obs_pt, np.int32), np.array(obs_xy, np.float64).reshape(-1, 2))
    if summary.termination != BA_CONVERGENCE:
        log("Bundle adjustment failed.")  # src/BundleAdjustment.cpp:126-129: inputs stay untouched
        return summary
    K[0, 0] = focal_o
    K[1, 1] = focal_o
    for i, pose in enumerate(camera_poses):
        if empty[i]:
            continue  # :142-145
        R = angle_axis_to_rotation_matrix(cams_o[i, :3])
        pose[:, :3] = R
        pose[:, 3] = cams_o[i, 3:]
    for i, p in enumerate(point_cloud):
        p["pt"] = (float(pts_o[i, 0]), float(pts_o[i, 1]), float(pts_o[i, 2]))
    return summary
