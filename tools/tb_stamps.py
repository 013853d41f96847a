#!/usr/bin/env python3
"""Phase timing inside tb_level (device tree builder) from in-kernel wall-clock stamps — development tool, GPU box only.
Uses a library built with -DMADICP_TB_STAMPS (MADICP_HIP_LIB points at it): lane 0 of every wave of the first 512
workgroups stores the 100 MHz wall clock at a few points.  Printed per level, in microseconds after the first wave's
start: median and maximum over the waves that passed each point."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import capi, synth  # noqa: E402

pb = synth.make_problem(1, seed=1, n_queries=1)
scan = pb["query_scans"][0]
ctx = capi.Context(0)
cid = ctx.cloud_upload(scan)
for _ in range(3):
    t, nl = ctx.tree_build(cid, 0.2, 0.1)
    ctx.tree_release(t)
lib = capi.hip_lib()
lib.madicp_debug_tb_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert lib.madicp_debug_tb_stamps(ctx._h, None, 1) == 0
t, nl = ctx.tree_build(cid, 0.2, 0.1)
buf = np.zeros(24 * 512 * 4 * 16, np.uint64)
assert lib.madicp_debug_tb_stamps(ctx._h, buf.ctypes.data, 0) == 0
s = buf.reshape(24, 2048, 16).astype(np.int64)
st = ctx.tree_build_stats()
cols = [(0, "start"), (3, "w:node"), (4, "w:sums"), (5, "w:eig"), (6, "w:sweep"), (7, "w:reduce"), (10, "w:barrier"), (8, "w:alloc"), (9, "w:emit"),
        (10, "l:sums"), (11, "l:eig"), (12, "l:sweep"), (13, "l:leaf"), (14, "l:alloc"), (15, "l:emit"), (1, "end")]
# chip levels 0-5: tb_chip_stats (slots 8-14) and tb_chip_scatter (slots 0-7, 1 = end), median / max over the wavefronts,
# microseconds after the kernel's own first wavefront
names_st = [(8, "start"), (9, "node"), (10, "sums joined"), (11, "eig"), (12, "V seen"), (13, "sides+ext"), (14, "tables out")]
names_sc = [(0, "start"), (2, "node"), (3, "part2 scan"), (4, "sides+scan"), (5, "plan,tab req"), (6, "child sums"), (7, "stores"), (1, "alloc/end")]
for lv in range(0, 6):
    for title, names in (("stats", names_st), ("scatter", names_sc)):
        a = s[lv]
        st0 = a[:, names[0][0]][a[:, names[0][0]] > 0]
        if st0.size == 0:
            continue
        t0 = st0.min()
        row = []
        for k, nm in names:
            v = a[:, k][a[:, k] > 0]
            row.append("%s %5.2f/%5.2f" % (nm, np.median(v - t0) / 100.0, (v.max() - t0) / 100.0) if v.size else nm + " -")
        print("chip level %d %-7s | %s" % (lv, title, " | ".join(row)))
print("level waveN | " + " ".join("%13s" % n for _, n in cols) + "   (median/max us)")
for lv in range(0, min(st["max_level"] + 2, 24)):
    a = s[lv]
    starts = a[:, 0][a[:, 0] > 0]
    if starts.size == 0:
        continue
    t0 = starts.min()
    row = []
    for k, _ in cols:
        v = a[:, k][a[:, k] > 0]
        row.append("%6.2f/%6.2f" % (np.median(v - t0) / 100.0, (v.max() - t0) / 100.0) if v.size else "            -")
    print("%5d %5d | %s" % (lv, st["wave_nodes"][lv], " ".join(row)))
ctx.close()
