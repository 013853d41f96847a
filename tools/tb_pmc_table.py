#!/usr/bin/env python3
"""One line per tb_* kernel from a tools/pmc_summary.py file (tools/measure_r6.sh pmc): what a wavefront of the device tree
builder does with its life.  usage: tools/tb_pmc_table.py gpurun_out/<tag>/tree_build_pmc_summary.md

  wait     = SQ_WAIT_ANY / SQ_WAVE_CYCLES          wavefront cycles spent waiting (memory, barrier, dependency)
  issue    = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES   ... issuing anything
  valu     = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES  ... with the vector ALU active
  L2 hit   = TCC_HIT_sum / TCC_REQ_sum
  fetch / write: FETCH_SIZE / WRITE_SIZE in KB per launch as the counters give them (no correction applied: see the guide's
                 note on 64-byte / 128-byte fetch accounting; the order of magnitude is the point here)"""
import re
import sys

cur, tab = None, {}
for line in open(sys.argv[1]):
    m = re.match(r"## `(?:madicp::)?(?:tb::)?([a-z_0-9]+)", line)
    if m:
        cur = m.group(1)
        tab[cur] = {}
        continue
    m = re.match(r"\| ([A-Za-z_0-9]+) \| (\d+) \| ([0-9.]+) \|", line)
    if m and cur:
        tab[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
print("| kernel | launches per pass | waves | wait | issue | valu | L2 requests | L2 hit | fetch KB | write KB | VALU insts | VMEM rd / wr insts |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for k, c in tab.items():
    if not k.startswith("tb_"):
        continue
    g = lambda n: c.get(n, (0, 0.0))[1]
    wc = max(g("SQ_WAVE_CYCLES"), 1.0)
    req = max(g("TCC_REQ_sum"), 1.0)
    print("| `%s` | %d | %.0f | %.2f | %.2f | %.2f | %.0f | %.2f | %.0f | %.0f | %.0f | %.0f / %.0f |" % (
        k, c.get("SQ_WAVES", (0, 0))[0], g("SQ_WAVES"), g("SQ_WAIT_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc,
        g("TCC_REQ_sum"), g("TCC_HIT_sum") / req, g("FETCH_SIZE"), g("WRITE_SIZE"), g("SQ_INSTS_VALU"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR")))
