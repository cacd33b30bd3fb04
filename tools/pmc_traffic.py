#!/usr/bin/env python3
"""Per-launch HBM traffic from the two rocprofv3 --pmc passes of tools/pmc_traffic.sh.

  python tools/pmc_traffic.py <out dir> <config> <bench JSON of the same config (its "launch_order")>

Launches of one step are told apart by their position between two k_prep dispatches: the profiled run submits every step on its
own (--single-step), so the n-th dispatch after a k_prep is the n-th launch of bench's `launch_order` (any config: MLP, C4, conv
branch).  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and gfx950's FETCH_SIZE tallies 128-byte
requests at 64 bytes (MI355X_MICROARCH.md, "HBM").  Writes profiles-ready JSON to <out>/traffic.json, with bench's algorithmic
bytes per launch beside the counters."""
import collections
import csv
import glob
import json
import sys

out, cfg, order_file = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

archs, B, dtype, label = bench.CONFIGS[cfg]
names = json.loads(open(order_file).read().strip().splitlines()[-1])["launch_order"]


def per_position(counter, sub):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    steps, cur = [], None
    for r in rows:
        n = r["Kernel_Name"]
        if "k_prep" in n:
            if cur is not None:
                steps.append(cur)
            cur = [("prep", float(r["Counter_Value"]))]
        elif cur is not None and "avae" in n:               # every kernel of the library lives in namespace avae
            cur.append((None, float(r["Counter_Value"])))
    steps = [s for s in steps[12:] if len(s) == len(names) + 1]          # skip warm-up; whole steps only
    seq = collections.defaultdict(list)
    for s in steps:
        for pos, (_, v) in enumerate(s):
            seq[pos].append(v)
    return {k: sum(v) / len(v) for k, v in seq.items()}, len(steps)


fetch, nf = per_position("FETCH_SIZE", "fetch")
write, nw = per_position("WRITE_SIZE", "write")
work, P = bench.launch_work(archs, B, 2 if dtype == "bf16" else 4)
res = {"config": label, "steps_averaged": [nf, nw],
       "note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, mean over steps; rocprofv3 --pmc, separate passes; "
               "algorithmic_bytes = bench.launch_work (a fused a+b launch: the sum of its parts, minus the gradient re-read for wgrad+adam)",
       "launches": {}}
for pos, name in enumerate(["prep"] + names):
    f, w = fetch.get(pos, 0.0), write.get(pos, 0.0)
    parts = name.split("+")
    alg = sum(work[p][0] for p in parts) if all(p in work for p in parts) else None
    if alg is not None and parts == ["wgrad", "adam"]:
        alg -= 4 * P
    res["launches"][name] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes": int((2 * f + w) * 1024), "algorithmic_bytes": alg}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
tot = sum(v["hbm_bytes"] for v in res["launches"].values())
alg = sum(v["algorithmic_bytes"] or 0 for v in res["launches"].values())
for k, v in res["launches"].items():
    print("%-28s fetch %9.1f KiB  write %9.1f KiB  hbm %8.2f MB  algorithmic %s" % (k, v["FETCH_SIZE_KiB"], v["WRITE_SIZE_KiB"], v["hbm_bytes"] / 1e6,
                                                                                 "%.2f MB" % (v["algorithmic_bytes"] / 1e6) if v["algorithmic_bytes"] else "-"))
print("step total %.2f MB moved, %.2f MB algorithmic" % (tot / 1e6, alg / 1e6))
