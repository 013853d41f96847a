"""Seeded KITTI-shaped synthetic LiDAR scans.

The reference ships no dataset and there is no network, so benchmarks and parity tests use a ray-cast
street scene: a spinning 64-beam sensor (elevation +2.0 .. -24.8 deg, 1875 azimuth steps -> 120 000 rays)
at 1.73 m above a ground plane, two facades at y = +-10 m, end walls, and a few dozen boxes (cars, poles,
kiosks) so that surface normals vary.  Gaussian range noise, returns kept for 0.7 <= r <= 120 m
(mad_icp/configurations/datasets/kitti.cfg:2-3 in the reference).  Everything is numpy + default_rng(seed),
fp64, sensor-frame (N,3) output — the shape `Pipeline.compute` / `MADtree.build` take.
"""
import numpy as np

SENSOR_HEIGHT = 1.73
N_BEAMS = 64
N_AZIMUTH = 1875
EL_TOP_DEG, EL_BOTTOM_DEG = 2.0, -24.8
R_MIN, R_MAX = 0.7, 120.0


class Scene:
    def __init__(self, seed=0, n_boxes=40, x_range=(-40.0, 260.0), half_width=10.0, facade_height=14.0):
        rng = np.random.default_rng(seed)
        self.x_range = x_range
        self.half_width = half_width
        self.facade_height = facade_height
        # boxes: centre x, centre y, size x, size y, height
        cx = rng.uniform(x_range[0] + 5, x_range[1] - 5, n_boxes)
        side = rng.choice([-1.0, 1.0], n_boxes)
        cy = side * rng.uniform(3.0, half_width - 1.5, n_boxes)
        kind = rng.integers(0, 3, n_boxes)
        sx = np.where(kind == 0, rng.uniform(3.5, 4.8, n_boxes), np.where(kind == 1, 0.3, rng.uniform(1.5, 3.0, n_boxes)))
        sy = np.where(kind == 0, rng.uniform(1.6, 2.0, n_boxes), np.where(kind == 1, 0.3, rng.uniform(1.5, 3.0, n_boxes)))
        h = np.where(kind == 0, rng.uniform(1.4, 1.9, n_boxes), np.where(kind == 1, rng.uniform(4.0, 8.0, n_boxes),
                                                                      rng.uniform(2.2, 3.2, n_boxes)))
        self.boxes = np.stack([cx, cy, sx, sy, h], axis=1)

    def raycast(self, origin, dirs):
        """origin (3,), dirs (N,3) unit, world frame -> range (N,), inf where nothing is hit."""
        o = origin
        d = dirs
        t_best = np.full(d.shape[0], np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            # ground z = 0
            t = -o[2] / d[:, 2]
            ok = (d[:, 2] < 0) & (t > 0)
            t_best = np.where(ok & (t < t_best), t, t_best)
            # facades y = +-half_width, z in [0, facade_height]
            for ywall in (self.half_width, -self.half_width):
                t = (ywall - o[1]) / d[:, 1]
                z = o[2] + t * d[:, 2]
                ok = (t > 0) & (z >= 0) & (z <= self.facade_height)
                t_best = np.where(ok & (t < t_best), t, t_best)
            # end walls
            for xwall in self.x_range:
                t = (xwall - o[0]) / d[:, 0]
                z = o[2] + t * d[:, 2]
                y = o[1] + t * d[:, 1]
                ok = (t > 0) & (z >= 0) & (z <= self.facade_height) & (np.abs(y) <= self.half_width)
                t_best = np.where(ok & (t < t_best), t, t_best)
            # boxes (slab method).  Only the rays whose heading lies inside the box's angular extent as seen from the
            # sensor can hit it (a conservative superset, margin 1e-6 rad; every ray when the sensor stands over the
            # footprint): the result is that of testing every ray against every box, at a tenth of the work.
            heading = np.arctan2(d[:, 1], d[:, 0])
            for cx, cy, sx, sy, h in self.boxes:
                lo = np.array([cx - sx / 2, cy - sy / 2, 0.0])
                hi = np.array([cx + sx / 2, cy + sy / 2, h])
                if lo[0] - 1e-9 <= o[0] <= hi[0] + 1e-9 and lo[1] - 1e-9 <= o[1] <= hi[1] + 1e-9:
                    idx = np.arange(d.shape[0])
                else:
                    ca = np.arctan2(np.array([lo[1], lo[1], hi[1], hi[1]]) - o[1], np.array([lo[0], hi[0], lo[0], hi[0]]) - o[0])
                    mid = np.arctan2(cy - o[1], cx - o[0])
                    rel = np.angle(np.exp(1j * (ca - mid)))  # corner headings relative to the centre's: within (-pi/2, pi/2)
                    dh = np.angle(np.exp(1j * (heading - mid)))
                    idx = np.nonzero((dh >= rel.min() - 1e-6) & (dh <= rel.max() + 1e-6))[0]
                if idx.size == 0:
                    continue
                tn = tf = None
                for a in range(3):
                    da = d[idx, a]
                    t1 = (lo[a] - o[a]) / da
                    t2 = (hi[a] - o[a]) / da
                    mn, mx = np.minimum(t1, t2), np.maximum(t1, t2)
                    tn = mn if tn is None else np.fmax(tn, mn)  # (fmax / fmin: NaN from 0/0 ignored, as nanmax / nanmin)
                    tf = mx if tf is None else np.fmin(tf, mx)
                ok = (tn <= tf) & (tn > 0)
                cur = t_best[idx]
                t_best[idx] = np.where(ok & (tn < cur), tn, cur)
        return t_best


def sensor_rays(n_beams=N_BEAMS, n_azimuth=N_AZIMUTH):
    el = np.deg2rad(np.linspace(EL_TOP_DEG, EL_BOTTOM_DEG, n_beams))
    az = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False)
    azg, elg = np.meshgrid(az, el, indexing="ij")  # azimuth-major: like a spinning sensor's packet order
    ce = np.cos(elg)
    return np.stack([ce * np.cos(azg), ce * np.sin(azg), np.sin(elg)], axis=-1).reshape(-1, 3)


def path_pose(s):
    """Sensor pose (4x4, sensor->world) at arc parameter s [m] along a gently curving street path."""
    x = s
    y = 1.5 * np.sin(0.05 * s)
    yaw = np.arctan(1.5 * 0.05 * np.cos(0.05 * s))
    c, sn = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[:3, :3] = [[c, -sn, 0], [sn, c, 0], [0, 0, 1]]
    T[:3, 3] = [x, y, SENSOR_HEIGHT]
    return T


def render_scan(scene, T_sensor_to_world, seed, sigma=0.02, n_beams=N_BEAMS, n_azimuth=N_AZIMUTH):
    """One scan in the SENSOR frame, (N,3) float64; N <= n_beams*n_azimuth (no-return rays dropped)."""
    rng = np.random.default_rng(seed)
    d_s = sensor_rays(n_beams, n_azimuth)
    R = T_sensor_to_world[:3, :3]
    d_w = d_s @ R.T
    r = scene.raycast(T_sensor_to_world[:3, 3], d_w)
    r = r + rng.normal(0.0, sigma, r.shape)
    keep = np.isfinite(r) & (r >= R_MIN) & (r <= R_MAX)
    return np.ascontiguousarray(d_s[keep] * r[keep, None])


def perturbation(seed, trans=0.3, rot_deg=1.0):
    """A fixed small rigid perturbation (4x4): |t| = trans, rotation angle = rot_deg about a random axis."""
    rng = np.random.default_rng(seed)
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    tdir = rng.normal(size=3)
    tdir /= np.linalg.norm(tdir)
    th = np.deg2rad(rot_deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    T = np.eye(4)
    T[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    T[:3, 3] = trans * tdir
    return T


def make_problem(n_keyframes, seed=0, spacing=3.0, query_offset=1.2, n_beams=N_BEAMS, n_azimuth=N_AZIMUTH, n_queries=1,
                 query_stream=0):
    """Keyframe scans + query scans for a registration benchmark.

    Returns dict(keyframe_scans [K x (N,3) sensor frame], keyframe_poses [K x 4x4], query_scans [Q x (N,3)],
    query_gt [Q x 4x4], query_guess [Q x 4x4]).  Keyframes sit `spacing` metres apart; query q is rendered
    `query_offset` (+0.35 q) metres past the last keyframe; the guess is GT composed with a seeded perturbation.
    `query_stream` selects an independent set of query scans (other noise, other perturbation, pose shifted by 5 cm
    per stream) over the same keyframes — one stream per rank in bench.py's replica mode.
    """
    scene = Scene(seed)
    kf_scans, kf_poses = [], []
    for k in range(n_keyframes):
        T = path_pose(k * spacing)
        kf_poses.append(T)
        kf_scans.append(render_scan(scene, T, seed * 1000 + k, n_beams=n_beams, n_azimuth=n_azimuth))
    q_scans, q_gt, q_guess = [], [], []
    for q in range(n_queries):
        T = path_pose((n_keyframes - 1) * spacing + query_offset + 0.35 * q + 0.05 * query_stream)
        q_gt.append(T)
        q_scans.append(render_scan(scene, T, seed * 1000 + 500 + q + 7919 * query_stream, n_beams=n_beams, n_azimuth=n_azimuth))
        q_guess.append(T @ perturbation(seed * 1000 + 900 + q + 7919 * query_stream))
    return dict(keyframe_scans=kf_scans, keyframe_poses=kf_poses, query_scans=q_scans, query_gt=q_gt,
                query_guess=q_guess)


def make_query_streams(n_keyframes, seed=0, n_streams=8, spacing=3.0, query_offset=1.2, n_beams=N_BEAMS, n_azimuth=N_AZIMUTH):
    """`n_streams` independent query scans of the SAME difficulty for the keyframes of make_problem(n_keyframes, seed):
    stream s is exactly make_problem(..., query_stream=s)'s query 0 — rendered `query_offset` (+ 5 cm per stream) past
    the last keyframe, its own range noise and its own 0.3 m / 1 deg perturbation of the guess — without re-rendering the
    keyframes.  Stream 0 is the BASELINE configs[2] query.  Returns (scans, gt poses, guesses)."""
    scene = Scene(seed)
    scans, gts, guesses = [], [], []
    for st in range(n_streams):
        T = path_pose((n_keyframes - 1) * spacing + query_offset + 0.05 * st)
        gts.append(T)
        scans.append(render_scan(scene, T, seed * 1000 + 500 + 7919 * st, n_beams=n_beams, n_azimuth=n_azimuth))
        guesses.append(T @ perturbation(seed * 1000 + 900 + 7919 * st))
    return scans, gts, guesses
