#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_c4.sh per kernel: mean of each counter over dispatches, plus
the derived ratios that matter for the GEMM kernels (matrix-pipe busy share, wait share, LDS conflict share, L2 hit
rate).  `python tools/pmc_summary.py <dir> [out.json]` prints a table and optionally writes profiles-ready JSON."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
import os
newest = {}          # one CSV per pass directory: gpurun merges results, older runs may still lie beside the newest
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    d = os.path.dirname(f)
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in newest.values():
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"dispatches": max(len(v) for v in cs.values()), "counters": {c: round(v, 1) for c, v in sorted(m.items())}}
    ratios = {}
    if m.get("SQ_WAVE_CYCLES"):
        wc = m["SQ_WAVE_CYCLES"]
        # units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
        # counts cycles.  mfma_busy_cycles_per_wave_cycle x (resident waves per SIMD) = share of time the matrix pipe is busy.
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            ratios["mfma_busy_cycles_per_wave_cycle"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 4)
        for name, key in (("wait_any_per_wave_cycle", "SQ_WAIT_ANY"),
                          ("wait_inst_per_wave_cycle", "SQ_WAIT_INST_ANY"), ("active_inst_per_wave_cycle", "SQ_ACTIVE_INST_ANY")):
            if key in m:
                ratios[name] = round(m[key] / wc, 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        ratios["lds_conflict_share"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
        ratios["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    d["ratios"] = ratios
    out[k] = d
for k, d in out.items():
    print(k[:100])
    print("   dispatches %d  ratios %s" % (d["dispatches"], d["ratios"]))
if len(sys.argv) > 2:
    json.dump({"note": "rocprofv3 --pmc, separate passes (tools/pmc_c4.sh); means per dispatch of each kernel", "kernels": out},
              open(sys.argv[2], "w"), indent=1)
