"""ctypes binding of the two C ABIs (include/madicp_hip.h, include/madicp_host.h).

This is plumbing for tests and bench.py; the product classes with the reference's names live in
mad_icp_amd.pybind.{pyvector,pymadtree,pymadicp,pypeline} (C++), which bind to the same ABI.
There is no CPU implementation behind any HIP entry point: without libmadicp_hip.so, or without a GPU,
the calls fail loudly.
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))

NODE_DTYPE = np.dtype([("mean", "<f8", (3,)), ("dir", "<f8", (3,)), ("right", "<i4"), ("leaf_id", "<i4"),
                       ("bbox0", "<f8")])
assert NODE_DTYPE.itemsize == 64

MAX_TREES = 128
MAX_BATCH = 64


class MadIcpError(RuntimeError):
    pass


class IcpParams(C.Structure):
    _fields_ = [("min_ball", C.c_double), ("rho_ker", C.c_double), ("b_ratio", C.c_double)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)

# madicp_host_allreduce_fn: int (*)(void* user, void* buf, int64_t count, int kind)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
REDUCE_SUM_F64, REDUCE_MAX_U8 = 0, 1

_hip = None
_host = None
_DIR_OVERRIDE = None  # set on the second instance of this module that measure_variant() makes
_measure_mod = None


def measure_variant():
    """This module a second time, bound to the MEASUREMENT build of the libraries (mad_icp_amd/_measure: the same sources with
    -DMADICP_MEASURE, built by _build.build_measure()): `capi.measure_variant().Context(0)` has include/madicp_hip_measure.h's
    timing / calibration / test aids, which the product library the default `capi` loads does not export.  Both can be used
    in one process (bench.py times the product and takes its roofline's launch times from the variant's identical kernels)."""
    global _measure_mod
    if _DIR_OVERRIDE is not None:
        return sys.modules[__name__]
    if os.environ.get("MADICP_NATIVE_DIR"):  # (a process already pointed at another build: that build is the variant)
        return sys.modules[__name__]
    if _measure_mod is None:
        import importlib.util

        from . import _build

        spec = importlib.util.spec_from_file_location(__name__ + "_measure", os.path.abspath(__file__),
                                                      submodule_search_locations=None)
        m = importlib.util.module_from_spec(spec)
        m.__package__ = __package__
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        m._DIR_OVERRIDE = _build.MEASURE_DIR
        _measure_mod = m
    return _measure_mod


def _load(name):
    path = os.path.join(_DIR_OVERRIDE or os.environ.get("MADICP_NATIVE_DIR") or _PKG, name)  # (same variable as _build.OUT)
    if name == "libmadicp_hip.so" and os.environ.get("MADICP_HIP_LIB"):
        path = os.environ["MADICP_HIP_LIB"]  # an instrumented build of the same library (tools/stamps.py)
    if not os.path.exists(path):
        raise MadIcpError(f"{name} is not built (run `python -m mad_icp_amd._build`); there is no fallback path")
    return C.CDLL(path)


def hip_lib():
    global _hip
    if _hip is None:
        L = _load("libmadicp_hip.so")
        L.madicp_last_error.restype = C.c_char_p
        L.madicp_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.madicp_ctx_destroy.argtypes = [C.c_void_p]
        L.madicp_ctx_synchronize.argtypes = [C.c_void_p]
        L.madicp_ctx_get_option.argtypes = [C.c_void_p, C.c_char_p, _i64p]
        L.madicp_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.madicp_ctx_get_option.argtypes = [C.c_void_p, C.c_char_p, _i64p]
        L.madicp_tree_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, _ip]
        L.madicp_tree_upload_trusted.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, _ip]
        L.madicp_tree_release.argtypes = [C.c_void_p, C.c_int]
        L.madicp_tree_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int32]
        L.madicp_tree_transform.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.madicp_nn_search.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64, _u32p, _u32p, _dp, _i32p]
        L.madicp_nn_search_device_enqueue.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.madicp_moving_upload.argtypes = [C.c_void_p, _dp, C.c_int32, _ip]
        L.madicp_moving_update.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int32]
        L.madicp_moving_release.argtypes = [C.c_void_p, C.c_int]
        if hasattr(L, "madicp_moving_update_async"):
            L.madicp_moving_update_async.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int32]
            L.madicp_icp_publish_enqueue.argtypes = [C.c_void_p, C.c_int, _ip]
            L.madicp_icp_publish_collect.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _i32p, _u64p]
        L.madicp_stream_submit.argtypes = [C.c_void_p, _dp, C.c_int32, _ip, C.c_int, _dp, C.POINTER(IcpParams), C.c_int, _ip]
        L.madicp_stream_submit_tree.argtypes = [C.c_void_p, C.c_int, _ip, C.c_int, _dp, C.POINTER(IcpParams), C.c_int, _ip]
        L.madicp_stream_collect.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _u8p, _i32p, _u64p]
        # include/madicp_hip_measure.h: only the measurement build exports these (measure_variant())
        if hasattr(L, "madicp_nn_time_descend"):
            L.madicp_nn_time_descend.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64, C.c_int, _dp, _u64p]
            L.madicp_debug_stream_copy.argtypes = [C.c_void_p, C.c_int64, C.c_int, _dp]
            L.madicp_icp_time_linearize.argtypes = [C.c_void_p, C.c_int, _ip, _ip, C.c_int, _dp, C.POINTER(IcpParams), C.c_int,
                                                    _dp, _u64p]
            L.madicp_icp_time_registration.argtypes = [C.c_void_p, C.c_int, _ip, _ip, C.c_int, _dp, C.POINTER(IcpParams), C.c_int,
                                                       C.c_int, _dp, _dp, _u64p, _u64p]
            L.madicp_debug_tree_build_points.argtypes = [C.c_void_p, _dp, C.c_int64]
        if hasattr(L, "madicp_debug_gather16"):  # (absent from an older build loaded through MADICP_HIP_LIB for an A/B)
            L.madicp_debug_gather16.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_int, _dp]
        L.madicp_icp_linearize.argtypes = [C.c_void_p, C.c_int, _ip, C.c_int, _dp, C.POINTER(IcpParams), _dp, _dp,
                                           _u32p, _u8p, _u64p]
        L.madicp_icp_register.argtypes = [C.c_void_p, C.c_int, _ip, C.c_int, _dp, C.POINTER(IcpParams), C.c_int, _dp,
                                          _dp, _u8p, _dp, _u64p]
        L.madicp_icp_register_batch.argtypes = [C.c_void_p, C.c_int, _ip, _ip, C.c_int, _dp, C.POINTER(IcpParams),
                                                C.c_int, _dp, _dp, _i32p, _u64p]
        L.madicp_icp_register_batch_enqueue.argtypes = [C.c_void_p, C.c_int, _ip, _ip, C.c_int, _dp,
                                                        C.POINTER(IcpParams), C.c_int]
        L.madicp_icp_fetch.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _i32p, _u64p]
        L.madicp_icp_fetch_matched.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int32]
        L.madicp_cloud_upload.argtypes = [C.c_void_p, _dp, C.c_int64, _ip]
        L.madicp_cloud_release.argtypes = [C.c_void_p, C.c_int]
        L.madicp_cloud_size.argtypes = [C.c_void_p, C.c_int, _i64p]
        L.madicp_cloud_download.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64]
        L.madicp_cloud_ingest_f32.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int64, C.c_int, C.c_double, C.c_double,
                                              C.c_int, _ip, _i64p]
        L.madicp_cloud_deskew.argtypes = [C.c_void_p, C.c_int, _dp, C.c_double, _i32p]
        L.madicp_tree_build.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, _ip, _i32p]
        L.madicp_tree_build_begin.argtypes = [C.c_void_p, _dp, C.c_int64, C.c_double, C.c_double]
        L.madicp_tree_build_end.argtypes = [C.c_void_p, _ip, _i32p]
        L.madicp_tree_build_cancel.argtypes = [C.c_void_p]
        L.madicp_tree_info.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p]
        L.madicp_tree_build_stats.argtypes = [C.c_void_p, _i32p]
        L.madicp_comm_unique_id.argtypes = [_u8p]
        L.madicp_comm_init.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int]
        L.madicp_comm_destroy.argtypes = [C.c_void_p]
        L.madicp_comm_init_host.argtypes = [C.c_void_p, C.c_int, C.c_int, HOST_ALLREDUCE_FN, C.c_void_p]
        if hasattr(L, "madicp_p2p_export"):
            L.madicp_p2p_export.argtypes = [C.c_void_p, _u8p]
            L.madicp_p2p_attach.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int]
            L.madicp_p2p_detach.argtypes = [C.c_void_p]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        L = _load("libmadicp_host.so")
        L.madicp_host_tree_build.restype = C.c_void_p
        L.madicp_host_tree_build.argtypes = [_dp, C.c_int64, C.c_double, C.c_double, C.c_int]
        L.madicp_host_tree_free.argtypes = [C.c_void_p]
        L.madicp_host_tree_num_nodes.restype = C.c_int32
        L.madicp_host_tree_num_nodes.argtypes = [C.c_void_p]
        L.madicp_host_tree_num_leaves.restype = C.c_int32
        L.madicp_host_tree_num_leaves.argtypes = [C.c_void_p]
        L.madicp_host_tree_nodes.restype = C.c_void_p
        L.madicp_host_tree_nodes.argtypes = [C.c_void_p]
        L.madicp_host_tree_leaf_nodes.restype = C.c_void_p
        L.madicp_host_tree_leaf_nodes.argtypes = [C.c_void_p]
        L.madicp_host_tree_leaf_means.argtypes = [C.c_void_p, _dp]
        L.madicp_host_tree_transform.argtypes = [C.c_void_p, _dp, _dp]
        L.madicp_host_gn_update.argtypes = [_dp, _dp, _dp]
        L.madicp_host_det_of_inverse6.restype = C.c_double
        L.madicp_host_det_of_inverse6.argtypes = [_dp]
        L.madicp_host_set_threads.argtypes = [C.c_int]
        L.madicp_host_tree_rho2.restype = C.c_double
        L.madicp_host_tree_rho2.argtypes = [C.c_void_p]
        L.madicp_host_debug_deskew.argtypes = [_dp, C.c_int64, _dp, _dp, C.c_double, C.c_int, _dp]
        L.madicp_host_debug_tree_points.restype = C.c_int64
        L.madicp_host_debug_tree_points.argtypes = [_dp, C.c_int64, C.c_double, C.c_double, C.c_int]
        L.madicp_host_debug_partition.restype = C.c_int64
        L.madicp_host_debug_partition.argtypes = [_dp, C.c_int64, _dp, _dp, C.c_int]
        _host = L
    return _host


def _aid(name):
    """an entry point of include/madicp_hip_measure.h: not in the product library"""
    f = getattr(hip_lib(), name, None)
    if f is None:
        raise MadIcpError(name + " is a measurement / test aid: the product library does not export it — use "
                          "mad_icp_amd.capi.measure_variant() (mad_icp_amd/_measure, built by _build.build_measure())")
    return f


def _check(rc):
    if rc != 0:
        raise MadIcpError(f"madicp error {rc}: {hip_lib().madicp_last_error().decode()}")


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def pose12(T):
    T = np.asarray(T, dtype=np.float64)
    if T.shape == (12,):
        return T.copy()
    return np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]).copy()


def pose44(x):
    T = np.eye(4)
    T[:3, :3] = np.asarray(x[:9]).reshape(3, 3)
    T[:3, 3] = x[9:12]
    return T


def gn_update(H, b, X12):
    """MADicp::updateState on joined adders (host): returns the updated pose (12,)."""
    H = _f64(H, (36,))
    b = _f64(b, (6,))
    X = _f64(X12, (12,)).copy()
    host_lib().madicp_host_gn_update(H.ctypes.data_as(_dp), b.ctypes.data_as(_dp), X.ctypes.data_as(_dp))
    return X


class HostTree:
    """Linear MAD-tree built on the host (include/madicp_host.h)."""

    def __init__(self, points, b_max, b_min, max_parallel_level=0):
        pts = _f64(points)
        if pts.ndim != 2 or pts.shape[1] != 3 or pts.shape[0] == 0:
            raise ValueError("points must be a non-empty (N,3) array")
        self._h = host_lib().madicp_host_tree_build(pts.ctypes.data_as(_dp), pts.shape[0], b_max, b_min,
                                                    max_parallel_level)
        if not self._h:
            raise MadIcpError("tree build failed")
        self.num_nodes = host_lib().madicp_host_tree_num_nodes(self._h)
        self.num_leaves = host_lib().madicp_host_tree_num_leaves(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            host_lib().madicp_host_tree_free(self._h)
            self._h = None

    @property
    def nodes(self):
        """numpy view (NODE_DTYPE) of the library-owned node array (the view keeps this object alive)."""
        ptr = host_lib().madicp_host_tree_nodes(self._h)
        buf = (C.c_char * (64 * self.num_nodes)).from_address(ptr)
        buf._owner = self  # the array's base keeps the tree (and so the memory) alive
        return np.frombuffer(buf, dtype=NODE_DTYPE)

    @property
    def leaf_nodes(self):
        ptr = host_lib().madicp_host_tree_leaf_nodes(self._h)
        buf = (C.c_char * (4 * self.num_leaves)).from_address(ptr)
        buf._owner = self
        return np.frombuffer(buf, dtype=np.int32)

    def leaf_means(self):
        out = np.empty((self.num_leaves, 3))
        host_lib().madicp_host_tree_leaf_means(self._h, out.ctypes.data_as(_dp))
        return out

    def transform(self, R, t):
        R = _f64(R, (9,))
        t = _f64(t, (3,))
        host_lib().madicp_host_tree_transform(self._h, R.ctypes.data_as(_dp), t.ctypes.data_as(_dp))

    @property
    def rho2(self):
        return host_lib().madicp_host_tree_rho2(self._h)


def host_tree_points(points, b_max, b_min, max_parallel_level=0):
    """The cloud as the host builder's construction leaves it (the reference's container after MADtree::build): every
    leaf's members in the order the splits produced, the leaf's first member overwritten by its representative."""
    pts = np.ascontiguousarray(points, dtype=np.float64).copy()
    nl = host_lib().madicp_host_debug_tree_points(pts.ctypes.data_as(_dp), pts.shape[0], float(b_max), float(b_min),
                                                  int(max_parallel_level))
    if nl < 0:
        raise MadIcpError("madicp_host_debug_tree_points: bad arguments")
    return pts, int(nl)


def host_deskew(points, T_prev, T_now, sensor_hz, route=0):
    """Pipeline::deskew on the host (csrc/host/deskew.h).  Returns (cloud in azimuth order, naive velocity (6,), used_parallel_order)."""
    pts = np.ascontiguousarray(points, dtype=np.float64).copy()
    vel = np.empty(6)
    a, b = pose12(T_prev), pose12(T_now)
    rc = host_lib().madicp_host_debug_deskew(pts.ctypes.data_as(_dp), pts.shape[0], a.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                             float(sensor_hz), int(route), vel.ctypes.data_as(_dp))
    if rc < 0:
        raise MadIcpError("madicp_host_debug_deskew: bad arguments")
    return pts, vel, bool(rc)


class Context:
    """One device + one stream (include/madicp_hip.h)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        _check(hip_lib().madicp_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            hip_lib().madicp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def synchronize(self):
        _check(hip_lib().madicp_ctx_synchronize(self._h))

    def set_option(self, key, value):
        _check(hip_lib().madicp_ctx_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64(0)
        _check(hip_lib().madicp_ctx_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    # ---- trees ----
    def tree_upload(self, nodes, n_leaves):
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        tid = C.c_int(0)
        _check(hip_lib().madicp_tree_upload(self._h, nodes.ctypes.data_as(C.c_void_p), nodes.shape[0], n_leaves,
                                            C.byref(tid)))
        return tid.value

    def upload(self, host_tree, trusted=False):
        if trusted:
            return self.tree_upload_trusted(host_tree.nodes, host_tree.num_leaves, host_tree.rho2)
        return self.tree_upload(host_tree.nodes, host_tree.num_leaves)

    def tree_upload_trusted(self, nodes, n_leaves, rho2):
        """madicp_tree_upload_trusted: no validation pass (arrays from the product's own builder only)."""
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        tid = C.c_int(0)
        _check(hip_lib().madicp_tree_upload_trusted(self._h, nodes.ctypes.data_as(C.c_void_p), nodes.shape[0], n_leaves,
                                                    float(rho2), C.byref(tid)))
        return tid.value

    def tree_release(self, tid):
        _check(hip_lib().madicp_tree_release(self._h, tid))

    def tree_download(self, tid, n_nodes):
        out = np.empty(n_nodes, dtype=NODE_DTYPE)
        _check(hip_lib().madicp_tree_download(self._h, tid, out.ctypes.data_as(C.c_void_p), n_nodes))
        return out

    def tree_transform(self, tid, R, t):
        R = _f64(R, (9,))
        t = _f64(t, (3,))
        _check(hip_lib().madicp_tree_transform(self._h, tid, R.ctypes.data_as(_dp), t.ctypes.data_as(_dp)))

    def nn_search(self, tid, queries, want=("leaf", "node", "dist", "depth")):
        q = _f64(queries)
        n = q.shape[0]
        leaf = np.empty(n, np.uint32) if "leaf" in want else None
        node = np.empty(n, np.uint32) if "node" in want else None
        dist = np.empty(n, np.float64) if "dist" in want else None
        depth = np.empty(n, np.int32) if "depth" in want else None
        _check(hip_lib().madicp_nn_search(
            self._h, tid, q.ctypes.data_as(_dp), n,
            leaf.ctypes.data_as(_u32p) if leaf is not None else None,
            node.ctypes.data_as(_u32p) if node is not None else None,
            dist.ctypes.data_as(_dp) if dist is not None else None,
            depth.ctypes.data_as(_i32p) if depth is not None else None))
        return dict(leaf=leaf, node=node, dist=dist, depth=depth)

    def nn_search_device(self, tid, d_queries_ptr, n, d_leaf=None, d_node=None, d_dist=None, d_depth=None):
        _check(hip_lib().madicp_nn_search_device_enqueue(self._h, tid, d_queries_ptr, n, d_leaf, d_node, d_dist,
                                                         d_depth))

    # ---- device front-end: clouds, ingest, deskew, tree build ----
    def cloud_upload(self, xyz):
        a = _f64(xyz)
        cid = C.c_int(0)
        _check(hip_lib().madicp_cloud_upload(self._h, a.ctypes.data_as(_dp), a.shape[0], C.byref(cid)))
        return cid.value

    def cloud_release(self, cid):
        _check(hip_lib().madicp_cloud_release(self._h, cid))

    def cloud_size(self, cid):
        n = C.c_int64(0)
        _check(hip_lib().madicp_cloud_size(self._h, cid, C.byref(n)))
        return n.value

    def cloud_download(self, cid):
        n = self.cloud_size(cid)
        out = np.empty((n, 3))
        _check(hip_lib().madicp_cloud_download(self._h, cid, out.ctypes.data_as(_dp), n))
        return out

    def cloud_ingest_f32(self, records, min_range, max_range, kitti_correction):
        """records: (n, stride >= 3) float32, x y z first (a KITTI .bin is (n,4)).  Returns (cloud id, points kept)."""
        r = np.ascontiguousarray(records, dtype=np.float32)
        if r.ndim != 2 or r.shape[1] < 3:
            raise ValueError("records must be (n, >=3) float32")
        cid, kept = C.c_int(0), C.c_int64(0)
        _check(hip_lib().madicp_cloud_ingest_f32(self._h, r.ctypes.data_as(C.POINTER(C.c_float)), r.shape[0], r.shape[1],
                                                 float(min_range), float(max_range), int(bool(kitti_correction)),
                                                 C.byref(cid), C.byref(kept)))
        return cid.value, kept.value

    def cloud_deskew(self, cid, velocity, sensor_hz, want_chunks=False):
        v = _f64(velocity, (6,))
        chunks = np.empty(self.cloud_size(cid), np.int32) if want_chunks else None
        _check(hip_lib().madicp_cloud_deskew(self._h, cid, v.ctypes.data_as(_dp), float(sensor_hz),
                                             chunks.ctypes.data_as(_i32p) if want_chunks else None))
        return chunks

    def tree_build(self, cid, b_max, b_min):
        """MAD-tree of a resident cloud, built on the device.  Returns (tree id, leaves)."""
        tid, nl = C.c_int(0), C.c_int32(0)
        _check(hip_lib().madicp_tree_build(self._h, cid, float(b_max), float(b_min), C.byref(tid), C.byref(nl)))
        return tid.value, nl.value

    def tree_build_begin(self, xyz, b_max, b_min):
        """Look-ahead: copy the scan and enqueue its construction on the library's build stream; returns at once."""
        a = _f64(xyz)
        _check(hip_lib().madicp_tree_build_begin(self._h, a.ctypes.data_as(_dp), a.shape[0], float(b_max), float(b_min)))

    def tree_build_end(self):
        """... and collect it.  Returns (tree id, leaves)."""
        tid, nl = C.c_int(0), C.c_int32(0)
        _check(hip_lib().madicp_tree_build_end(self._h, C.byref(tid), C.byref(nl)))
        return tid.value, nl.value

    def tree_build_cancel(self):
        _check(hip_lib().madicp_tree_build_cancel(self._h))

    def tree_info(self, tid):
        nn, nl = C.c_int32(0), C.c_int32(0)
        _check(hip_lib().madicp_tree_info(self._h, tid, C.byref(nn), C.byref(nl)))
        return nn.value, nl.value

    def tree_build_stats(self):
        out = np.zeros(130, np.int32)
        _check(hip_lib().madicp_tree_build_stats(self._h, out.ctypes.data_as(_i32p)))
        return dict(max_level=int(out[0]), lane_subtrees=int(out[1]), wave_nodes=out[2:66].copy(), chip_nodes=out[66:130].copy())

    def tree_build_points(self, n):
        """the points of the last synchronous device build in the order the construction left them (diagnostics)"""
        out = np.empty((int(n), 3))
        _check(_aid("madicp_debug_tree_build_points")(self._h, out.ctypes.data_as(_dp), int(n)))
        return out

    # ---- moving ----
    def moving_upload(self, leaf_means):
        m = _f64(leaf_means)
        mid = C.c_int(0)
        _check(hip_lib().madicp_moving_upload(self._h, m.ctypes.data_as(_dp), m.shape[0], C.byref(mid)))
        return mid.value

    def moving_update(self, mid, leaf_means):
        m = _f64(leaf_means)
        _check(hip_lib().madicp_moving_update(self._h, mid, m.ctypes.data_as(_dp), m.shape[0]))

    def moving_release(self, mid):
        _check(hip_lib().madicp_moving_release(self._h, mid))

    # ---- registration ----
    @staticmethod
    def _ids(ids):
        arr = (C.c_int * len(ids))(*[int(i) for i in ids])
        return arr

    def icp_linearize(self, mid, tree_ids, T, params, L, want_corr=True):
        K = len(tree_ids)
        X = pose12(T)
        H, b = np.empty((6, 6)), np.empty(6)
        corr = np.empty((K, L), np.uint32) if want_corr else None
        matched = np.empty(L, np.uint8)
        visits = C.c_uint64(0)
        p = IcpParams(*params)
        _check(hip_lib().madicp_icp_linearize(self._h, mid, self._ids(tree_ids), K, X.ctypes.data_as(_dp), C.byref(p),
                                              H.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                              corr.ctypes.data_as(_u32p) if want_corr else None,
                                              matched.ctypes.data_as(_u8p), C.byref(visits)))
        return dict(H=H, b=b, corr=corr, matched=matched, visits=visits.value)

    def icp_register(self, mid, tree_ids, T, params, n_iters, L):
        K = len(tree_ids)
        X = pose12(T)
        H, b = np.empty((6, 6)), np.empty(6)
        matched = np.empty(L, np.uint8)
        X_iters = np.empty((n_iters, 12))
        visits = C.c_uint64(0)
        p = IcpParams(*params)
        _check(hip_lib().madicp_icp_register(self._h, mid, self._ids(tree_ids), K, X.ctypes.data_as(_dp), C.byref(p),
                                             n_iters, H.ctypes.data_as(_dp), b.ctypes.data_as(_dp),
                                             matched.ctypes.data_as(_u8p), X_iters.ctypes.data_as(_dp),
                                             C.byref(visits)))
        return dict(T=pose44(X), X=X, H=H, b=b, matched=matched, X_iters=X_iters, visits=visits.value)

    # ---- streamed registrations (new scan in -> X / H / b / flags out) ----
    def stream_submit(self, leaf_means, tree_ids, T, params, n_iters):
        m = _f64(leaf_means)
        X = pose12(T)
        p = IcpParams(*params)
        tk = C.c_int(-1)
        _check(hip_lib().madicp_stream_submit(self._h, m.ctypes.data_as(_dp), m.shape[0], self._ids(tree_ids), len(tree_ids),
                                              X.ctypes.data_as(_dp), C.byref(p), n_iters, C.byref(tk)))
        return tk.value

    def stream_submit_tree(self, moving_tid, tree_ids, T, params, n_iters):
        X = pose12(T)
        p = IcpParams(*params)
        tk = C.c_int(-1)
        _check(hip_lib().madicp_stream_submit_tree(self._h, moving_tid, self._ids(tree_ids), len(tree_ids),
                                                   X.ctypes.data_as(_dp), C.byref(p), n_iters, C.byref(tk)))
        return tk.value

    def stream_collect(self, ticket, L=0):
        X, H, b = np.empty(12), np.empty((6, 6)), np.empty(6)
        matched = np.empty(L, np.uint8) if L else None
        nm = C.c_int32(0)
        visits = C.c_uint64(0)
        _check(hip_lib().madicp_stream_collect(self._h, ticket, X.ctypes.data_as(_dp), H.ctypes.data_as(_dp),
                                               b.ctypes.data_as(_dp), matched.ctypes.data_as(_u8p) if L else None,
                                               C.byref(nm), C.byref(visits)))
        return dict(T=pose44(X), X=X, H=H, b=b, matched=matched, n_matched=nm.value, visits=visits.value)

    # ---- measurement aids ----
    def nn_time_descend(self, tid, queries, reps=20):
        """(avg microseconds per nn_descend launch over the queries, internal nodes visited per launch)."""
        q = _f64(queries)
        us = C.c_double(0.0)
        depth = C.c_uint64(0)
        _check(_aid("madicp_nn_time_descend")(self._h, tid, q.ctypes.data_as(_dp), q.shape[0], reps, C.byref(us),
                                                C.byref(depth)))
        return us.value, depth.value

    def gather16_us(self, region_bytes, n_gathers, seed=1, reps=3):
        """avg microseconds per launch of n_gathers random 16-byte loads over a region (madicp_debug_gather16)."""
        g = C.c_double(0.0)
        _check(_aid("madicp_debug_gather16")(self._h, int(region_bytes), int(n_gathers), int(seed), int(reps), C.byref(g)))
        return g.value

    def stream_copy_gbs(self, nbytes=1 << 30, reps=10):
        g = C.c_double(0.0)
        _check(_aid("madicp_debug_stream_copy")(self._h, nbytes, reps, C.byref(g)))
        return g.value

    def icp_register_batch_enqueue(self, mids, tree_ids, X0, params, n_iters):
        X0 = _f64(X0, (len(mids), 12))
        p = IcpParams(*params)
        _check(hip_lib().madicp_icp_register_batch_enqueue(self._h, len(mids), self._ids(mids), self._ids(tree_ids),
                                                           len(tree_ids), X0.ctypes.data_as(_dp), C.byref(p), n_iters))

    def icp_time_linearize(self, mids, tree_ids, X0, params, n_launches=50):
        """(avg microseconds per first-round icp_round launch, visits per launch per scan) — see madicp_icp_time_linearize."""
        X0 = _f64(X0, (len(mids), 12))
        p = IcpParams(*params)
        us = C.c_double(0.0)
        visits = np.zeros(len(mids), np.uint64)
        _check(_aid("madicp_icp_time_linearize")(self._h, len(mids), self._ids(mids), self._ids(tree_ids), len(tree_ids),
                                                   X0.ctypes.data_as(_dp), C.byref(p), n_launches, C.byref(us),
                                                   visits.ctypes.data_as(_u64p)))
        return us.value, visits

    def icp_time_registration(self, mids, tree_ids, X0, params, n_iters, reps=30):
        """(avg us per icp_round launch over a registration's rounds, us of icp_final, visits per round per scan,
        nodes really walked per round per scan)."""
        X0 = _f64(X0, (len(mids), 12))
        p = IcpParams(*params)
        lin, sol = C.c_double(0.0), C.c_double(0.0)
        visits = np.zeros(len(mids), np.uint64)
        walked = np.zeros(len(mids), np.uint64)
        _check(_aid("madicp_icp_time_registration")(self._h, len(mids), self._ids(mids), self._ids(tree_ids), len(tree_ids),
                                                      X0.ctypes.data_as(_dp), C.byref(p), n_iters, reps, C.byref(lin),
                                                      C.byref(sol), visits.ctypes.data_as(_u64p), walked.ctypes.data_as(_u64p)))
        return lin.value, sol.value, visits, walked

    def icp_fetch(self, n_scans):
        X, H, b = np.empty((n_scans, 12)), np.empty((n_scans, 6, 6)), np.empty((n_scans, 6))
        nm = np.empty(n_scans, np.int32)
        visits = np.empty(n_scans, np.uint64)
        _check(hip_lib().madicp_icp_fetch(self._h, n_scans, X.ctypes.data_as(_dp), H.ctypes.data_as(_dp),
                                          b.ctypes.data_as(_dp), nm.ctypes.data_as(_i32p),
                                          visits.ctypes.data_as(_u64p)))
        return dict(X=X, H=H, b=b, n_matched=nm, visits=visits)

    def moving_update_async(self, mid, leaf_means):
        """madicp_moving_update on the copy stream (beside the batch in flight; ordered by the library)"""
        m = _f64(leaf_means)
        _check(hip_lib().madicp_moving_update_async(self._h, mid, m.ctypes.data_as(_dp), m.shape[0]))

    def icp_publish_enqueue(self, n_scans):
        """ticket for the results of the batch enqueued last (carried to the host by a kernel behind it)"""
        tk = C.c_int(-1)
        _check(hip_lib().madicp_icp_publish_enqueue(self._h, n_scans, C.byref(tk)))
        return tk.value

    def icp_publish_collect(self, ticket, n_scans):
        X, H, b = np.empty((n_scans, 12)), np.empty((n_scans, 6, 6)), np.empty((n_scans, 6))
        nm = np.empty(n_scans, np.int32)
        visits = np.empty(n_scans, np.uint64)
        _check(hip_lib().madicp_icp_publish_collect(self._h, ticket, n_scans, X.ctypes.data_as(_dp), H.ctypes.data_as(_dp),
                                                    b.ctypes.data_as(_dp), nm.ctypes.data_as(_i32p), visits.ctypes.data_as(_u64p)))
        return dict(X=X, H=H, b=b, n_matched=nm, visits=visits)

    def icp_fetch_matched(self, scan, L):
        out = np.empty(L, np.uint8)
        _check(hip_lib().madicp_icp_fetch_matched(self._h, scan, out.ctypes.data_as(_u8p), L))
        return out

    def icp_register_batch(self, mids, tree_ids, X0, params, n_iters):
        self.icp_register_batch_enqueue(mids, tree_ids, X0, params, n_iters)
        return self.icp_fetch(len(mids))

    # ---- multi-GPU ----
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        _check(hip_lib().madicp_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, n_ranks, rank):
        buf = (C.c_uint8 * 128)(*unique_id)
        _check(hip_lib().madicp_comm_init(self._h, buf, n_ranks, rank))

    def comm_init_host(self, n_ranks, rank, all_reduce):
        """Host-staged transport (madicp_comm_init_host): `all_reduce(array, kind)` must reduce the numpy array IN PLACE
        over all ranks (kind REDUCE_SUM_F64: float64 sum; REDUCE_MAX_U8: uint8 max).  Exceptions become MADICP_ERR_COMM."""
        def _cb(_user, buf, count, kind):
            try:
                ctype = C.c_double if kind == REDUCE_SUM_F64 else C.c_uint8
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(ctype)), shape=(count,))
                all_reduce(arr, kind)
                return 0
            except Exception as e:  # noqa: BLE001 — must not unwind through the C frame
                self._host_ar_error = e
                return 1

        self._host_ar_cb = HOST_ALLREDUCE_FN(_cb)  # (kept alive as long as the library may call it)
        _check(hip_lib().madicp_comm_init_host(self._h, n_ranks, rank, self._host_ar_cb, None))

    def p2p_export(self):
        """This rank's mailbox as a 64-byte hipIpcMemHandle_t (madicp_p2p_export; allocated on first call)."""
        buf = (C.c_uint8 * 64)()
        _check(hip_lib().madicp_p2p_export(self._h, buf))
        return bytes(buf)

    def p2p_attach(self, handles, n_ranks, rank):
        """Map every rank's mailbox: `handles` = the ranks' 64-byte handles in rank order (madicp_p2p_attach)."""
        blob = b"".join(handles)
        if len(blob) != 64 * n_ranks:
            raise ValueError("expected %d handles of 64 bytes" % n_ranks)
        buf = (C.c_uint8 * len(blob))(*blob)
        _check(hip_lib().madicp_p2p_attach(self._h, buf, n_ranks, rank))

    def p2p_detach(self):
        _check(hip_lib().madicp_p2p_detach(self._h))

    def comm_destroy(self):
        _check(hip_lib().madicp_comm_destroy(self._h))
