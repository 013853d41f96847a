"""The device front-end against the ORACLE directly — not against the product's own host path.

Round 5: `Pipeline` builds the scan's MAD-tree on the MI355X BY DEFAULT when `deskew = false` (csrc/host/pipeline.cpp; the
host builder is `MAD_ICP_GPU_BUILD=0` / `setDeviceFrontEnd(False)`), so an unmodified caller (apps/mad_icp.py, bin_runner)
runs this path.  What is held here:

  * deskew = false — five scenes x 50 full-size frames, different speeds and promotion thresholds: the DEFAULT Pipeline within
    1e-5 m / 1e-5 rad of the oracle pipeline (oracle/, restating pipeline.cpp:125-265 + mad_tree.cpp:47-130) at EVERY frame, with
    identical currentID / keyframeID / isMapUpdated.
  * deskew = true — the reference does not reproduce ITSELF there beyond millimetres (tests/test_oracle_sensitivity.py,
    tests/envelope.py): the oracle pipeline is run against itself with other thread counts and with one coordinate of one point
    moved by one ulp, and BOTH product paths (host builder + host deskew, device deskew + device builder) must stay inside three
    times the deviation the reference has shown from itself up to that frame.  Outside it the difference would be a bug, not
    chaos.  Errors against ground truth are compared too.
  * the device builder on DESKEWED clouds (azimuth-sorted: another point order than a ring-ordered scan's) and on the sixty
    random small clouds, topology compared EXACTLY with the host builder's (== the oracle's, tests/test_host_builder.py): every
    cloud that differs is listed, and must be one whose topology the reference itself does not keep under a 1-ulp change of
    the input (a decision within rounding of its threshold).
"""
import time

import numpy as np
import pytest

import envelope as E
import oracle_lib as O
from fixtures import B_MAX, B_MIN, B_RATIO, RHO_KER, full_scan
from mad_icp_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pypeline(natives):
    from mad_icp.src.pybind import pypeline as m

    return m


# ---- (a) deskew = false: the default Pipeline IS the device front-end, and it is the oracle's trajectory -----------------------
@pytest.mark.parametrize("scene_seed,step,p_th,kf", [(0, 1.0, 0.8, 16), (1, 1.3, 0.85, 16), (3, 0.7, 0.8, 8), (4, 2.0, 0.9, 16),
                                                     (9, 1.0, 0.95, 4)])
def test_default_pipeline_builds_on_the_device_and_matches_the_oracle(pypeline, scene_seed, step, p_th, kf, capsys):
    n_frames = 50
    scene = synth.Scene(scene_seed)
    args = (10.0, False, B_MAX, RHO_KER, p_th, B_MIN, B_RATIO, kf, 16, False)
    gp, op = pypeline.Pipeline(*args), O.Pipeline(*args)
    assert gp.deviceFrontEnd()  # nobody asked for it: the default
    worst_t = worst_a = 0.0
    promotions, t_frame = 0, []
    for i in range(n_frames):
        s = synth.render_scan(scene, synth.path_pose(step * i), 4000 + 97 * scene_seed + i)
        t = time.perf_counter()
        gp.compute(0.1 * i, s)
        t_frame.append(time.perf_counter() - t)
        op.compute(0.1 * i, s)
        dt, da = E.pose_dev(op.currentPose(), np.asarray(gp.currentPose()))
        worst_t, worst_a = max(worst_t, dt), max(worst_a, da)
        assert dt <= 1e-5 and da <= 1e-5, (i, dt, da)
        assert gp.currentID() == op.currentID() and gp.keyframeID() == op.keyframeID(), (i, gp.keyframeID(), op.keyframeID())
        assert gp.isMapUpdated() == op.isMapUpdated(), i
        if i > 0:
            assert abs(gp.lastInliersRatio() - op.lastInliersRatio()) < 2e-3
        promotions += int(gp.isMapUpdated())
    assert promotions >= 3
    gt = np.linalg.inv(synth.path_pose(0.0)) @ synth.path_pose(step * (n_frames - 1))
    assert np.linalg.norm(np.asarray(gp.currentPose())[:3, 3] - gt[:3, 3]) < 0.25
    with capsys.disabled():
        print("\n[default Pipeline (device front-end) vs oracle pipeline, scene %d, %d frames x %d points, %.1f m/frame, p_th %.2f, "
              "%d keyframes] worst %.2e m / %.2e rad; %d promotions; %.2f ms per frame"
              % (scene_seed, n_frames, s.shape[0], step, p_th, kf, worst_t, worst_a, promotions, 1e3 * np.mean(t_frame[2:])))


# ---- (c) deskew = true: both product paths inside the reference's own envelope -----------------------------------------------
def _product_drive(pypeline, scans, device):
    p = pypeline.Pipeline(10.0, True, B_MAX, RHO_KER, 0.8, B_MIN, B_RATIO, 16, 16, False)
    assert p.deviceFrontEnd()  # (round 6: the default for deskewed datasets too — tests/test_gpu_deskew_one_step.py is its bar)
    p.setDeviceFrontEnd(device)
    poses, kf = [], []
    for i, s in enumerate(scans):
        p.compute(0.1 * i, s)
        poses.append(np.asarray(p.currentPose()).copy())
        kf.append(p.keyframeID())
    return poses, kf


def _jitter(scans, seed=0):
    """1e-7 m of noise: no two points share an azimuth any more (a real sensor's noise does the same), so std::sort's
    unspecified order among ties — the synthetic scans hold 64 points per azimuth column — is out of the picture"""
    rng = np.random.default_rng(seed)
    return [sc + rng.normal(scale=1e-7, size=sc.shape) for sc in scans]


def _hold_inside_envelope(tag, scans, pypeline, capsys, variants=E.VARIANTS, gt=None, head=0):
    """Both product paths against the oracle pipeline (4 threads), each inside the envelope of the variants that change what
    that path changes:
      * host path (host deskew + host builder: the reference's trees bit for bit from bit-identical clouds) — what differs from
        the reference is the last bits of the POSES (summation order, fused multiply-adds after the gate): the envelope of the
        thread-count and single-coordinate variants;
      * device front-end — its builder also differs in the last bits of the larger nodes' sums (poses at 1e-13 without deskew):
        the envelope of all variants, the every-coordinate ones included.
    `head` > 0 (long drives, where a variant is expensive): the first `head` frames are sampled with ALL of E.VARIANTS, the
    whole drive with `variants` — the onset of the amplification is the heavy-tailed part and wants the larger sample."""
    base, kf_o, dt, da = E.self_envelope(scans, deskew=True, base_threads=4, variants=variants)
    few = np.array(["ulp_all" not in o for _, o in variants])
    comb = E.combined(dt, da)
    m_all, m_few, t_all = comb.max(axis=0), comb[few].max(axis=0), dt.max(axis=0)
    if head:
        _, _, dt_h, da_h = E.self_envelope(scans[:head], deskew=True, base_threads=4, variants=E.VARIANTS)
        few_h = np.array(["ulp_all" not in o for _, o in E.VARIANTS])
        comb_h = E.combined(dt_h, da_h)
        m_all[:head] = np.maximum(m_all[:head], comb_h.max(axis=0))
        m_few[:head] = np.maximum(m_few[:head], comb_h[few_h].max(axis=0))
        t_all[:head] = np.maximum(t_all[:head], dt_h.max(axis=0))
    bound_all, bound_few = np.maximum.accumulate(m_all), np.maximum.accumulate(m_few)  # metres: translation + 10 m x rotation
    bt = np.maximum.accumulate(t_all)
    rows = []
    # (Both paths are held to the envelope of ALL variants.  Holding the host path to the thread / single-coordinate variants
    # alone — the kind of change it makes — was tried: its deviations sit in that envelope from frame ~4 on, but the ONSET of
    # the amplification is heavy-tailed — one drive showed 3e-3 m at frame 2 where eight such variants showed at most 3e-4 at
    # frame 2 and 7e-3 at frame 3 — and eight samples do not bound it; both envelopes are printed.)
    for name, device, bound in (("host path", False, bound_all), ("device front-end", True, bound_all)):
        poses, kf = _product_drive(pypeline, scans, device)
        d = np.array([E.pose_dev(a, b) for a, b in zip(base, poses)])
        rows.append((name, d, poses, kf, bound))
    with capsys.disabled():
        fr = sorted(set([1, 2, 3, 4, 5, 6, 8, 11] + list(range(19, len(scans), 10)) + [len(scans) - 1]))
        fr = [f for f in fr if f < len(scans)]
        print("\n[%s: deskew=True, %d frames x %d points] deviation from the oracle pipeline (4 threads): translation + 10 m x "
              "rotation, metres" % (tag, len(scans), scans[0].shape[0]))
        print("  %-56s %s" % ("frame", " ".join("%7d" % f for f in fr)))
        print("  %-56s %s" % ("oracle vs ITSELF (threads, one coordinate), running max", " ".join("%7.0e" % bound_few[f] for f in fr)))
        print("  %-56s %s" % ("product host path", " ".join("%7.0e" % E.combined(rows[0][1][f, 0], rows[0][1][f, 1]) for f in fr)))
        print("  %-56s %s" % ("oracle vs ITSELF (+ every coordinate), running max", " ".join("%7.0e" % bound_all[f] for f in fr)))
        print("  %-56s %s" % ("product device front-end", " ".join("%7.0e" % E.combined(rows[1][1][f, 0], rows[1][1][f, 1]) for f in fr)))
    for name, d, poses, kf, bound in rows:
        # frames 0 and 1 are not deskewed (pipeline.cpp:138-139 needs two poses): the north-star bar itself
        assert (d[:2, 0] <= 1e-5).all() and (d[:2, 1] <= 1e-5).all(), (name, d[:2])
        m = E.combined(d[:, 0], d[:, 1])
        over = np.flatnonzero(m > 3.0 * bound + 2e-5)
        assert over.size == 0, (tag, name, [(int(f), float(d[f, 0]), float(d[f, 1]), float(bound[f])) for f in over[:5]])
        if gt is not None:  # ... and as good against ground truth as the reference (2 % + 1 mm)
            e_o = np.array([np.linalg.norm((np.linalg.inv(g) @ b)[:3, 3]) for g, b in zip(gt, base)])
            e_p = np.array([np.linalg.norm((np.linalg.inv(g) @ b)[:3, 3]) for g, b in zip(gt, poses)])
            assert e_p[-1] <= 1.02 * e_o[-1] + 3.0 * bt[-1] + 1e-3
            assert np.sqrt((e_p ** 2).mean()) <= 1.02 * np.sqrt((e_o ** 2).mean()) + 3.0 * bt[-1] + 1e-3
        # same keyframes, except where a promotion decision itself sits inside the envelope: at most one frame in fifty apart
        assert np.sum(np.array(kf) != np.array(kf_o)) <= max(1, len(scans) // 50), (name, kf, kf_o)
    return bt


@pytest.mark.parametrize("size,jitter", [("reduced", False), ("reduced", True), ("full", False), ("full", True)])
def test_deskewed_paths_stay_inside_the_references_own_envelope(pypeline, size, jitter, capsys):
    scene = synth.Scene(0)
    kw = dict(n_beams=32, n_azimuth=600) if size == "reduced" else {}
    scans = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i, **kw) if kw else full_scan(0, 1.0 * i, 100 + i)
             for i in range(12)]
    if jitter:
        scans = _jitter(scans)
    bt = _hold_inside_envelope("%s%s" % (size, ", distinct azimuths" if jitter else ", tied azimuths"), scans, pypeline, capsys)
    assert bt[-1] >= 1e-5  # (the envelope is not empty: the reference did move)


@pytest.mark.parametrize("jitter", [False, True])
def test_deskewed_long_drive_inside_the_envelope(pypeline, jitter, capsys):
    """100 full-size frames, 1 m per frame (the drive of the former 5e-2 m bar between the product's two paths): each path
    against the oracle, inside the oracle's own envelope, with the error against ground truth of the reference's.  Five
    variants (two thread counts, one single-coordinate 1-ulp change, two every-coordinate ones) make the envelope over the whole
    drive, all twelve over its first twelve frames."""
    n = 100
    scans = [full_scan(0, 1.0 * i, 100 + i) for i in range(n)]
    if jitter:
        scans = _jitter(scans, 3)
    T0inv = np.linalg.inv(synth.path_pose(0.0))
    gt = [T0inv @ synth.path_pose(1.0 * i) for i in range(n)]
    variants = (E.VARIANTS[0], E.VARIANTS[1], E.VARIANTS[4], E.VARIANTS[8], E.VARIANTS[9])
    _hold_inside_envelope("long drive%s" % (", distinct azimuths" if jitter else ", tied azimuths"), scans, pypeline, capsys,
                          variants=variants, gt=gt, head=12)


# ---- (d) the device builder on deskewed clouds and on small degenerate clouds, topology compared exactly ------------------------
def _build_pair(ctx, pts, b_max=B_MAX, b_min=B_MIN):
    ht = capi.HostTree(pts, b_max, b_min, 2)
    cid = ctx.cloud_upload(pts)
    tid, nl = ctx.tree_build(cid, b_max, b_min)
    nodes = ctx.tree_download(tid, 2 * nl - 1)
    ctx.tree_release(tid)
    ctx.cloud_release(cid)
    return ht, nodes


def _same_topology(nodes, ht):
    return nodes.shape[0] == ht.nodes.shape[0] and np.array_equal(nodes["right"], ht.nodes["right"])


def _reference_keeps_its_topology(pts, b_max, b_min, seed, trials=8, ulps=(1, 4, 16)):
    """Does the reference's OWN tree (host builder == oracle) survive a change of every coordinate by a few ulps?  The device
    builder adds the members of a node of more than 32 points in another shape than the reference's serial chain
    (tree_build.hip.h): its centroids differ from the reference's by the rounding of a sum — a few ulps, nine on a 246-point
    node whose members share a coordinate.  Returns [(ulps, trees kept, trials)]."""
    base = capi.HostTree(pts, b_max, b_min, 2).nodes["right"]
    out = []
    for u in ulps:
        rng = np.random.default_rng(seed + 7919 * u)
        kept = 0
        for _ in range(trials):
            r = capi.HostTree(E.ulp_jitter(pts, rng, u), b_max, b_min, 2).nodes["right"]
            kept += int(r.shape == base.shape and np.array_equal(r, base))
        out.append((u, kept, trials))
    return out


def test_device_builder_on_deskewed_clouds(mctx, capsys):
    """Every builder-vs-builder test so far fed ring-ordered scans.  A deskewed cloud arrives sorted by azimuth
    (pipeline.cpp:88-92) — another member order into every sum — and motion-compensated.  Twelve of them (two scenes, the
    oracle's own deskew with the drive's motion), tied and distinct azimuths: the host builder's topology, EVERY leaf
    representative at its ordinal, the construction's member order row for row."""
    ctx = mctx  # (the measurement build's context: madicp_debug_tree_build_points below)
    total = diff = 0
    for sc in (0, 3):
        scene = synth.Scene(sc)
        for i in range(1, 7):
            raw = synth.render_scan(scene, synth.path_pose(1.1 * i), 8100 + 17 * sc + i)
            if i % 2:
                raw = raw + np.random.default_rng(i).normal(scale=1e-7, size=raw.shape)
            Tp = np.linalg.inv(synth.path_pose(0.0)) @ synth.path_pose(1.1 * (i - 1))
            Tn = np.linalg.inv(synth.path_pose(0.0)) @ synth.path_pose(1.1 * i)
            pts, _ = O.deskew(raw, Tp, Tn, 10.0)
            az = np.arctan2(pts[:, 1], pts[:, 0])
            ht, nodes = _build_pair(ctx, pts)
            assert _same_topology(nodes, ht), (sc, i)
            leaf = nodes["right"] == 0
            same = np.all(nodes["mean"][leaf].view(np.uint64) == ht.nodes["mean"][leaf].view(np.uint64), axis=1)
            total += int(leaf.sum())
            diff += int((~same).sum())
            d_order = ctx.tree_build_points(pts.shape[0])
            h_order, _ = capi.host_tree_points(pts, B_MAX, B_MIN, 2)
            same_row = np.all(d_order.view(np.uint64) == h_order.view(np.uint64), axis=1)
            reps = set(map(bytes, np.ascontiguousarray(ht.nodes["mean"][leaf]).view(np.uint8).reshape(-1, 24)))
            assert all(bytes(h_order[j].view(np.uint8)) in reps for j in np.flatnonzero(~same_row)), (sc, i)
            assert np.abs(nodes["bbox0"][leaf] - ht.nodes["bbox0"][leaf]).max() <= 1e-9
            assert np.abs(nodes["dir"][leaf] - ht.nodes["dir"][leaf]).max() <= 1e-8
            del az
    with capsys.disabled():
        print("\n[device vs host builder on 12 DESKEWED clouds] topology identical on all; leaf representatives that differ: %d of %d"
              % (diff, total))
    assert diff == 0


def test_device_builder_topology_on_random_small_clouds_exactly(ctx, capsys):
    """The sixty random small clouds of tests/test_gpu_frontend.py (blobs, sheets, lines, duplicates; 1 .. 400 points; random
    thresholds), topology compared EXACTLY.  The device builder may only differ from the reference's tree where the reference's
    tree is itself not determined by the input to better than a few ulps: every cloud that differs is listed with what a change
    of every coordinate by up to 1, 4 and 16 ulps does to the HOST builder's (== the oracle's) topology, and a cloud whose
    reference topology survives all of them while the device's differs fails the test.  (What differs on these clouds, measured:
    exactly collinear points — the SIGN of the split direction, an eigenvector of a covariance with two zero eigenvalues, is left
    to the rounding of the centroid, and with the opposite sign the two children swap places.)"""
    r2 = np.random.default_rng(77)
    listing, unexplained = [], []
    n_same = 0
    for i in range(60):
        n = int(r2.integers(1, 400))
        kind = int(r2.integers(4))
        c = r2.normal(size=(n, 3)) * r2.choice([0.01, 0.3, 5.0], size=3)
        if kind == 1:
            c[:, 2] = 0.0
        elif kind == 2:
            c[:, 1:] = 0.0
        elif kind == 3:
            c = np.repeat(c[: max(1, n // 4)], 4, axis=0)
        c = c + r2.normal(size=3) * 10.0
        b_max, b_min = float(r2.choice([1e-5, 0.05, 0.2, 1.0])), float(r2.choice([0.01, 0.1, 0.5]))
        r2.integers(3)
        ht, nodes = _build_pair(ctx, c, b_max, b_min)
        if _same_topology(nodes, ht):
            n_same += 1
            continue
        kept = _reference_keeps_its_topology(c, b_max, b_min, 1000 + i)
        m = min(nodes.shape[0], ht.nodes.shape[0])
        neq = np.flatnonzero(nodes["right"][:m] != ht.nodes["right"][:m])
        first = int(neq[0]) if neq.size else m
        mirrored = first < m and bool(np.dot(nodes["dir"][first], ht.nodes["dir"][first]) < 0)
        listing.append("cloud %2d: %3d points, %s, b_max %g, b_min %g: device %d leaves, reference %d; first differing node %d%s; the "
                       "reference keeps its own topology in %s perturbations of every coordinate"
                       % (i, c.shape[0], ("blob", "sheet", "line", "duplicates")[kind], b_max, b_min, (nodes.shape[0] + 1) // 2,
                          ht.num_leaves, first, " (split direction of opposite SIGN: the children are mirrored)" if mirrored else "",
                          ", ".join("%d of %d +-%d-ulp" % (k_, t_, u_) for u_, k_, t_ in kept)))
        if all(k_ == t_ for _, k_, t_ in kept):
            unexplained.append(i)
    with capsys.disabled():
        print("\n[device vs host builder, 60 random small clouds, exact topology] identical on %d; differing:" % n_same)
        for line in listing:
            print("  " + line)
    assert not unexplained, unexplained
    assert n_same >= 54
