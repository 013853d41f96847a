"""The reference's OWN sensitivity to last-bit changes, measured with the oracle (test infrastructure, like oracle_lib).

MAD-tree construction is chaotic in the last bit of its input: a leaf's representative is the member nearest to the centroid
(mad_tree.cpp:76-86), and the two members of a two-point leaf are equally far from their midpoint up to rounding, so a
1-ulp change anywhere upstream flips representatives by up to b_max.  With `deskew = true` the cloud a tree is built from is
a function of the two previous POSES (pipeline.cpp:79-123, :138-139), and the reference's poses already depend in their last
bits on `num_threads` (the per-thread adders are summed in thread order, mad_icp.cpp:106-109).  So two correct
implementations of the deskewed pipeline drift apart by what the reference drifts apart from ITSELF under such changes.

`self_envelope` measures exactly that: the oracle pipeline against itself with other thread counts and with ONE coordinate of
ONE point of one early cloud moved by one ulp.  A product path (host builder or device front-end) is then held INSIDE that
envelope — if it were not, the difference would be a bug, not chaos."""
import numpy as np

import oracle_lib as O
from fixtures import B_MAX, B_MIN, B_RATIO, RHO_KER


def pose_dev(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    return float(np.linalg.norm(d[:3, 3])), float(np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)))


def ulp_jitter(a, rng, ulps=1):
    """Every coordinate moved by -ulps .. +ulps units in the last place (np.nextafter steps: exact zeros become denormals, not
    the NaN an integer view would make of them)."""
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    k = rng.integers(-ulps, ulps + 1, size=a.shape)
    for _ in range(ulps):
        up, dn = k > 0, k < 0
        a[up] = np.nextafter(a[up], np.inf)
        a[dn] = np.nextafter(a[dn], -np.inf)
        k = k - np.sign(k)
    return a


def oracle_drive(scans, deskew, threads=4, ulp_at=None, ulp_all=None, num_keyframes=16, p_th=0.8):
    """Poses of the oracle pipeline over `scans`.  ulp_at = (frame, point, coordinate): that one double moved by one ulp;
    ulp_all = seed: EVERY coordinate of EVERY cloud moved by -1, 0 or +1 ulp (seeded)."""
    p = O.Pipeline(10.0, bool(deskew), B_MAX, RHO_KER, p_th, B_MIN, B_RATIO, num_keyframes, threads, False)
    rng = np.random.default_rng(ulp_all) if ulp_all is not None else None
    poses, kf = [], []
    for i, s in enumerate(scans):
        if ulp_at is not None and ulp_at[0] == i:
            s = np.ascontiguousarray(s, dtype=np.float64).copy()
            j = ulp_at[1] % s.shape[0]
            s[j, ulp_at[2]] = np.nextafter(s[j, ulp_at[2]], np.inf)
        if rng is not None:
            s = ulp_jitter(s, rng)
        p.compute(0.1 * i, s)
        poses.append(p.currentPose().copy())
        kf.append(p.keyframeID())
    return poses, kf


# Three kinds of last-bit change, each of which a correct implementation may differ from the reference by:
#   * another num_threads — another summation order of the per-thread adders (mad_icp.cpp:106-109): the POSES' last bits;
#   * one coordinate of one point by one ulp — the smallest change of an input there is;
#   * every coordinate of every cloud by at most one ulp — what the device builder's own last-bit differences amount to: its
#     trees have the reference's topology and leaf representatives but the last bits of the larger nodes' sums differ (leaf
#     normals to 6e-11), so its poses differ from the reference's at 1e-13 already WITHOUT deskew, and with deskew every point of
#     the next compensated cloud moves by some ulps.
VARIANTS = (("threads=1", dict(threads=1)), ("threads=2", dict(threads=2)), ("threads=3", dict(threads=3)),
            ("threads=8", dict(threads=8)), ("1 ulp, cloud 0, point 5, x", dict(ulp_at=(0, 5, 0))),
            ("1 ulp, cloud 1, point 77, y", dict(ulp_at=(1, 77, 1))), ("1 ulp, cloud 2, point 1234, z", dict(ulp_at=(2, 1234, 2))),
            ("1 ulp, cloud 1, point 9, z; threads=2", dict(ulp_at=(1, 9, 2), threads=2)),
            ("every coordinate -1/0/+1 ulp, seed 1", dict(ulp_all=1)), ("every coordinate -1/0/+1 ulp, seed 2", dict(ulp_all=2)),
            ("every coordinate -1/0/+1 ulp, seed 3", dict(ulp_all=3)), ("every coordinate -1/0/+1 ulp, seed 4", dict(ulp_all=4)))


def self_envelope(scans, deskew, base_threads=4, variants=VARIANTS, **kw):
    """(base poses, base keyframe ids, per-variant per-frame translation deviation [V, F], rotation deviation [V, F])."""
    base, kf = oracle_drive(scans, deskew, base_threads, **kw)
    dt = np.zeros((len(variants), len(scans)))
    da = np.zeros_like(dt)
    for v, (_, opts) in enumerate(variants):
        o = dict(threads=base_threads)
        o.update(opts)
        poses, _ = oracle_drive(scans, deskew, **o, **kw)
        for i, (a, b) in enumerate(zip(base, poses)):
            dt[v, i], da[v, i] = pose_dev(a, b)
    return base, kf, dt, da


RANGE = 10.0  # metres: a rotation of a rad moves a point at this range (the street's half width) by RANGE * a


def combined(dt, da):
    """One number per pose deviation: translation + the displacement the rotation gives a point at RANGE.  (Translation and
    rotation deviations are NOT proportional from sample to sample — along the street the scene constrains translation
    weakly, so a variant can move by millimetres without turning by 1e-8 rad — which is why the envelope is taken on this
    sum and not on the two separately.)"""
    return np.asarray(dt) + RANGE * np.asarray(da)


def running_bound(dev):
    """What a path is held to at frame i: the largest deviation any variant has shown up to and including frame i."""
    return np.maximum.accumulate(dev.max(axis=0))
