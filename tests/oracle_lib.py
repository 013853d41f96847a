"""ctypes binding of oracle/libmad_oracle.so — the CPU restatement of the reference path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  Builds the library on first use if it is missing (plain `make`).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle")
_LIB = None

c_dp = C.POINTER(C.c_double)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.environ.get("MADICP_ORACLE_SO")  # another implementation of the same ABI (tests/test_reference_structure_pin.py:
    if so:                                   # the reference's own sources behind it)
        return so
    out = os.environ.get("MADICP_ORACLE_DIR")  # a second build with other defines (tests/test_redux_variant.py)
    so = os.path.join(out or _DIR, "libmad_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["OUT=" + out] if out else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_tree_build.restype = C.c_void_p
        L.orc_tree_build.argtypes = [c_dp, C.c_int64, C.c_double, C.c_double, C.c_int]
        L.orc_tree_free.argtypes = [C.c_void_p]
        L.orc_tree_num_nodes.restype = C.c_int64
        L.orc_tree_num_nodes.argtypes = [C.c_void_p]
        L.orc_tree_num_leaves.restype = C.c_int64
        L.orc_tree_num_leaves.argtypes = [C.c_void_p]
        L.orc_tree_export.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_i32p, c_i32p, c_i32p]
        L.orc_tree_leaves.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
        L.orc_tree_transform.argtypes = [C.c_void_p, c_dp, c_dp]
        L.orc_tree_search.argtypes = [C.c_void_p, c_dp, C.c_int64, c_u32p, c_i32p, c_dp]
        L.orc_icp_linearize.restype = C.c_int64
        L.orc_icp_linearize.argtypes = [C.c_void_p, C.c_void_p, c_dp, C.c_double, C.c_double, C.c_double,
                                        c_dp, c_dp, c_u32p, c_u8p, c_u8p]
        L.orc_icp_register.restype = C.c_double
        L.orc_icp_register.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, c_dp, C.c_int, C.c_double,
                                       C.c_double, C.c_double, C.c_int, c_dp, c_dp, c_u8p, c_dp,
                                       C.POINTER(C.c_int64)]
        L.orc_eig3.argtypes = [c_dp, c_dp, c_dp]
        L.orc_ldlt6_solve.argtypes = [c_dp, c_dp, c_dp]
        L.orc_det_inverse6.restype = C.c_double
        L.orc_det_inverse6.argtypes = [c_dp]
        L.orc_expmap_so3.argtypes = [c_dp, c_dp]
        L.orc_logmap_so3.argtypes = [c_dp, c_dp]
        L.orc_pipeline_create.restype = C.c_void_p
        L.orc_pipeline_create.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_int, C.c_int, C.c_int]
        L.orc_pipeline_free.argtypes = [C.c_void_p]
        L.orc_pipeline_compute.argtypes = [C.c_void_p, C.c_double, c_dp, C.c_int64]
        L.orc_pipeline_current_pose.argtypes = [C.c_void_p, c_dp]
        L.orc_pipeline_keyframe_pose.argtypes = [C.c_void_p, c_dp]
        for name in ("orc_pipeline_current_id", "orc_pipeline_keyframe_id", "orc_pipeline_num_keyframes"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc_pipeline_is_map_updated.argtypes = [C.c_void_p]
        if hasattr(L, "orc_pipeline_set_virtual_times"):  # (the oracle's own test seam: the reference stand-in has none)
            L.orc_pipeline_set_virtual_times.argtypes = [C.c_void_p, C.c_double, C.c_double]
            L.orc_pipeline_last_rounds.argtypes = [C.c_void_p]
        L.orc_pipeline_last_icp_ms.restype = C.c_double
        L.orc_pipeline_last_icp_ms.argtypes = [C.c_void_p]
        L.orc_pipeline_last_inliers_ratio.restype = C.c_double
        L.orc_pipeline_last_inliers_ratio.argtypes = [C.c_void_p]
        L.orc_pipeline_current_leaves.restype = C.c_int64
        L.orc_pipeline_current_leaves.argtypes = [C.c_void_p, c_dp, C.c_int64]
        L.orc_pipeline_model_leaves.restype = C.c_int64
        L.orc_pipeline_model_leaves.argtypes = [C.c_void_p, c_dp, C.c_int64]
        L.orc_deskew.argtypes = [c_dp, C.c_int64, c_dp, c_dp, C.c_double, c_dp]
        if hasattr(L, "orc_pipeline_last_guess"):  # (instrumentation of the oracle: not in the reference-stand-in library)
            L.orc_pipeline_last_guess.argtypes = [C.c_void_p, c_dp]
            L.orc_pipeline_predict.argtypes = [C.c_void_p, c_dp]
            L.orc_pipeline_keyframe_borrow.restype = C.c_void_p
            L.orc_pipeline_keyframe_borrow.argtypes = [C.c_void_p, C.c_int64]
            L.orc_pipeline_keyframe_num_nodes.restype = C.c_int64
            L.orc_pipeline_keyframe_num_nodes.argtypes = [C.c_void_p, C.c_int64]
            L.orc_pipeline_keyframe_export.argtypes = [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp, c_i32p, c_i32p, c_i32p]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def pose12(T):
    """4x4 (or 3x4) -> 12 doubles: R row-major then t."""
    T = np.asarray(T, dtype=np.float64)
    return np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]).copy()


def pose44(x):
    T = np.eye(4)
    T[:3, :3] = np.asarray(x[:9]).reshape(3, 3)
    T[:3, 3] = x[9:12]
    return T


class Tree:
    """oracle::MADtree built by the restated mad_tree.cpp:47-130."""

    def __init__(self, points, b_max, b_min, max_parallel_level=0):
        pts = _cloud(points)
        self._h = lib().orc_tree_build(_dp(pts), pts.shape[0], b_max, b_min, max_parallel_level)
        if not self._h:
            raise ValueError("empty cloud")
        self.num_nodes = lib().orc_tree_num_nodes(self._h)
        self.num_leaves = lib().orc_tree_num_leaves(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_tree_free(self._h)
            self._h = None

    def export(self):
        n = self.num_nodes
        out = dict(mean=np.empty((n, 3)), evecs=np.empty((n, 3, 3)), bbox=np.empty((n, 3)),
                   left=np.empty(n, np.int32), right=np.empty(n, np.int32), num_points=np.empty(n, np.int32))
        lib().orc_tree_export(self._h, _dp(out["mean"]), _dp(out["evecs"]), _dp(out["bbox"]),
                              out["left"].ctypes.data_as(c_i32p), out["right"].ctypes.data_as(c_i32p),
                              out["num_points"].ctypes.data_as(c_i32p))
        return out

    def leaves(self):
        L = self.num_leaves
        mean, normal, bbox0 = np.empty((L, 3)), np.empty((L, 3)), np.empty(L)
        lib().orc_tree_leaves(self._h, _dp(mean), _dp(normal), _dp(bbox0))
        return mean, normal, bbox0

    def transform(self, R, t):
        R = np.ascontiguousarray(R, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        lib().orc_tree_transform(self._h, _dp(R), _dp(t))

    def search(self, queries, want_dist=False):
        q = _cloud(queries)
        n = q.shape[0]
        leaf = np.empty(n, np.uint32)
        depth = np.empty(n, np.int32)
        dist = np.empty(n) if want_dist else None
        lib().orc_tree_search(self._h, _dp(q), n, leaf.ctypes.data_as(c_u32p), depth.ctypes.data_as(c_i32p),
                              _dp(dist) if want_dist else None)
        return (leaf, depth, dist) if want_dist else (leaf, depth)


def icp_linearize(moving, fixed, T, min_ball, rho_ker, b_ratio):
    """One MADicp::update for one tree: H(6,6), b(6), corr, rejected, matched, depth_sum."""
    L = moving.num_leaves
    X = pose12(T)
    H, b = np.empty((6, 6)), np.empty(6)
    corr, rej, mat = np.empty(L, np.uint32), np.empty(L, np.uint8), np.empty(L, np.uint8)
    depth = lib().orc_icp_linearize(moving._h, fixed._h, _dp(X), min_ball, rho_ker, b_ratio, _dp(H), _dp(b),
                                    corr.ctypes.data_as(c_u32p), rej.ctypes.data_as(c_u8p), mat.ctypes.data_as(c_u8p))
    return H, b, corr, rej, mat, depth


def icp_register(moving, fixed_list, T, n_iters, min_ball, rho_ker, b_ratio, num_threads=1):
    """GN driver loop; returns dict(T, H, b, matched, X_iters (poses before each round), depth_sum, ms)."""
    K = len(fixed_list)
    hs = (C.c_void_p * K)(*[f._h for f in fixed_list])
    X = pose12(T)
    L = moving.num_leaves
    H, b = np.empty((6, 6)), np.empty(6)
    matched = np.empty(L, np.uint8)
    X_iters = np.empty((n_iters, 12))
    depth = C.c_int64(0)
    ms = lib().orc_icp_register(moving._h, hs, K, _dp(X), n_iters, min_ball, rho_ker, b_ratio, num_threads, _dp(H),
                                _dp(b), matched.ctypes.data_as(c_u8p), _dp(X_iters), C.byref(depth))
    return dict(T=pose44(X), H=H, b=b, matched=matched, X_iters=X_iters, depth_sum=depth.value, ms=ms)


def export_to_nodes(ex):
    """Tree.export() / Pipeline.keyframeTree() -> the 64-byte DFS-preorder node array of include/madicp_hip.h (what
    madicp_tree_upload takes) and the leaf count: left child = i + 1, `right` = offset of the right child, a leaf carries its
    surface point, its normal (eigenvector 0) and its getLeafs() ordinal; an internal node its centroid and split normal
    (eigenvector 2) — the mapping tests/test_host_builder.py holds the product's host builder to."""
    from mad_icp_amd import capi

    n = ex["mean"].shape[0]
    leaf = ex["left"] < 0
    nodes = np.zeros(n, dtype=capi.NODE_DTYPE)
    nodes["mean"] = ex["mean"]
    nodes["dir"] = np.where(leaf[:, None], ex["evecs"][:, :, 0], ex["evecs"][:, :, 2])
    idx = np.arange(n, dtype=np.int32)
    nodes["right"] = np.where(leaf, 0, ex["right"] - idx)
    nodes["leaf_id"] = -1
    nodes["leaf_id"][leaf] = np.arange(int(leaf.sum()), dtype=np.int32)
    nodes["bbox0"] = ex["bbox"][:, 0]
    return nodes, int(leaf.sum())


def eig3(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    w, V = np.empty(3), np.empty((3, 3))
    lib().orc_eig3(_dp(A), _dp(w), _dp(V))
    return w, V


def ldlt6_solve(A, b):
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(6)
    lib().orc_ldlt6_solve(_dp(A), _dp(b), _dp(x))
    return x


def det_inverse6(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    return lib().orc_det_inverse6(_dp(A))


def expmap_so3(w):
    w = np.ascontiguousarray(w, dtype=np.float64)
    R = np.empty((3, 3))
    lib().orc_expmap_so3(_dp(w), _dp(R))
    return R


def logmap_so3(R):
    R = np.ascontiguousarray(R, dtype=np.float64)
    w = np.empty(3)
    lib().orc_logmap_so3(_dp(R), _dp(w))
    return w


def deskew(points, T_prev, T_now, sensor_hz):
    """Pipeline::deskew (pipeline.cpp:79-123): returns (compensated cloud in azimuth order, naive_vel (6,))."""
    pts = np.ascontiguousarray(points, dtype=np.float64).copy()
    vel = np.empty(6)
    lib().orc_deskew(_dp(pts), pts.shape[0], _dp(pose12(T_prev)), _dp(pose12(T_now)), float(sensor_hz), _dp(vel))
    return pts, vel


def ingest_f32(records, min_range, max_range, kitti):
    """apps/cpp_runners/bin_runner.cpp:126-166 restated in numpy (float32 norm, Eigen's AngleAxisd rotation), the checker
    of madicp_cloud_ingest_f32.  records: (n, >=3) float32."""
    r = np.asarray(records, dtype=np.float32)
    x, y, z = r[:, 0], r[:, 1], r[:, 2]
    nrm = np.sqrt(x * x + (y * y + z * z))          # Vector3f::norm(): float, unrolled scalar reduction x0 + (x1 + x2)
    with np.errstate(invalid="ignore"):
        drop = (nrm.astype(np.float64) < min_range) | (nrm.astype(np.float64) > max_range) | np.isnan(x) | np.isnan(y) | np.isnan(z)
    p = r[~drop, :3].astype(np.float64)
    if not kitti:
        return p
    X, Y, Z = p[:, 0], p[:, 1], p[:, 2]
    r0, r1, r2 = Y * 1.0 - Z * 0.0, Z * 0.0 - X * 1.0, X * 0.0 - Y * 0.0   # p.cross((0,0,1))
    sq = (r0 * r0 + r1 * r1) + r2 * r2                                        # squaredNorm of a contiguous Vector3d
    nn = np.sqrt(sq)
    pos = sq > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        a0, a1, a2 = np.where(pos, r0 / nn, r0), np.where(pos, r1 / nn, r1), np.where(pos, r2 / nn, r2)
    ang = (0.205 * np.pi) / 180.0                                             # VERTICAL_ANGLE_OFFSET, bin_runner.cpp:55
    s, c = np.sin(ang), np.cos(ang)
    s0, s1, s2 = s * a0, s * a1, s * a2
    c0, c1, c2 = (1.0 - c) * a0, (1.0 - c) * a1, (1.0 - c) * a2
    R = np.empty((p.shape[0], 3, 3))
    t = c0 * a1; R[:, 0, 1] = t - s2; R[:, 1, 0] = t + s2
    t = c0 * a2; R[:, 0, 2] = t + s1; R[:, 2, 0] = t - s1
    t = c1 * a2; R[:, 1, 2] = t - s0; R[:, 2, 1] = t + s0
    R[:, 0, 0] = c0 * a0 + c; R[:, 1, 1] = c1 * a1 + c; R[:, 2, 2] = c2 * a2 + c
    out = np.empty_like(p)
    for i in range(3):
        out[:, i] = R[:, i, 0] * X + (R[:, i, 1] * Y + R[:, i, 2] * Z)       # row . vector, strided: x0 + (x1 + x2)
    return out


class Pipeline:
    """oracle::Pipeline (pipeline.cpp:34-308)."""

    def __init__(self, sensor_hz, deskew, b_max, rho_ker, p_th, b_min, b_ratio, num_keyframes, num_threads, realtime):
        self._h = lib().orc_pipeline_create(sensor_hz, int(deskew), b_max, rho_ker, p_th, b_min, b_ratio,
                                            num_keyframes, num_threads, int(realtime))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_pipeline_free(self._h)
            self._h = None

    def compute(self, stamp, points):
        pts = _cloud(points)
        lib().orc_pipeline_compute(self._h, stamp, _dp(pts), pts.shape[0])

    def currentPose(self):
        x = np.empty(12)
        lib().orc_pipeline_current_pose(self._h, _dp(x))
        return pose44(x)

    def keyframePose(self):
        x = np.empty(12)
        lib().orc_pipeline_keyframe_pose(self._h, _dp(x))
        return pose44(x)

    def currentID(self):
        return lib().orc_pipeline_current_id(self._h)

    def keyframeID(self):
        return lib().orc_pipeline_keyframe_id(self._h)

    def isMapUpdated(self):
        return bool(lib().orc_pipeline_is_map_updated(self._h))

    def numKeyframes(self):
        return lib().orc_pipeline_num_keyframes(self._h)

    def lastIcpMs(self):
        return lib().orc_pipeline_last_icp_ms(self._h)

    def lastInliersRatio(self):
        return lib().orc_pipeline_last_inliers_ratio(self._h)

    def setVirtualTimes(self, pre_ms, round_ms):
        """test seam (realtime=True): the frame's preprocessing took pre_ms, every GN round round_ms (pre_ms < 0: wall clock)"""
        lib().orc_pipeline_set_virtual_times(self._h, float(pre_ms), float(round_ms))

    def lastRounds(self):
        return lib().orc_pipeline_last_rounds(self._h)

    def currentLeaves(self):
        n = lib().orc_pipeline_current_leaves(self._h, None, 0)
        out = np.empty((n, 3))
        lib().orc_pipeline_current_leaves(self._h, _dp(out), n)
        return out

    def lastGuess(self):
        """the constant-velocity prediction the last frame's GN loop started from (pipeline.cpp:146-152)"""
        x = np.empty(12)
        lib().orc_pipeline_last_guess(self._h, _dp(x))
        return pose44(x)

    def predict(self):
        """the prediction the NEXT frame's GN loop would start from, on the state as it stands (pipeline.cpp:146-152)"""
        x = np.empty(12)
        lib().orc_pipeline_predict(self._h, _dp(x))
        return pose44(x)

    def borrowKeyframe(self, k):
        """keyframe tree k as a Tree that does not own it: valid until this pipeline's next compute()"""
        t = Tree.__new__(Tree)
        t._h = lib().orc_pipeline_keyframe_borrow(self._h, k)
        if not t._h:
            raise IndexError(k)
        t.num_nodes = lib().orc_tree_num_nodes(t._h)
        t.num_leaves = lib().orc_tree_num_leaves(t._h)
        return t

    def keyframeTree(self, k):
        """keyframe tree k of the local map as it stands (map frame), in Tree.export()'s form"""
        n = lib().orc_pipeline_keyframe_num_nodes(self._h, k)
        out = dict(mean=np.empty((n, 3)), evecs=np.empty((n, 3, 3)), bbox=np.empty((n, 3)),
                   left=np.empty(n, np.int32), right=np.empty(n, np.int32), num_points=np.empty(n, np.int32))
        lib().orc_pipeline_keyframe_export(self._h, k, _dp(out["mean"]), _dp(out["evecs"]), _dp(out["bbox"]),
                                           out["left"].ctypes.data_as(c_i32p), out["right"].ctypes.data_as(c_i32p),
                                           out["num_points"].ctypes.data_as(c_i32p))
        return out

    def modelLeaves(self):
        n = lib().orc_pipeline_model_leaves(self._h, None, 0)
        out = np.empty((n, 3))
        lib().orc_pipeline_model_leaves(self._h, _dp(out), n)
        return out
