#!/usr/bin/env python3
"""Per-launch durations of the train step from a rocprofv3 --kernel-trace CSV.

  python tools/per_launch.py <kernel_trace.csv> <names.json|bench-json-line> [<log of the profiled bench run>] > profiles/rNN_<cfg>_per_launch.json

k_grouped runs every GEMM launch of the step, so rocprofv3's per-kernel-name statistics lump them
together.  A graph replay issues the launches in a fixed order (fwd_*, bwd_*, wgrad*, adam), so the n-th
dispatch after a k_adam is the n-th launch of the plan; the staging kernel k_prep runs once per replay
(16 or 4 steps) or once per single step and is reported on its own.  Steps whose dispatch count differs are
skipped.
"""
import csv
import json
import sys

ORDER_HINT = ["fwd_enc", "conv_enc", "conv_head", "fwd_head", "fwd_dec", "conv_dec", "conv_out", "fwd_out_loss",
              "bwd_out", "conv_bwd_dec", "bwd_dec", "bwd_head", "conv_bwd_enc", "bwd_enc", "wgrad"]


def main():
    trace, order_file = sys.argv[1], sys.argv[2]
    oj = json.load(open(order_file))                      # {"order": [...]} or a bench.py JSON line (its "launch_order")
    names = oj["order"] if "order" in oj else oj["launch_order"]          # launches of one step in order, ending with "adam"
    rows = [r for r in csv.DictReader(open(trace)) if any(k in r["Kernel_Name"] for k in ("k_prep", "k_grouped", "k_small", "k_gather", "k_col2im", "k_adam", "k_thin", "k_wadj", "k_gperm", "k_rowsum",
                                                                                      "k_colsum", "k_reduce", "k_sums"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))            # the CSV is not in execution order
    # a step ends with k_adam (or with the weight-gradient launch that carries it); the staging kernel k_prep runs once per step or once per replay of 16 / 4 steps
    steps, cur, preps = [], [], []
    for r in rows:
        if "k_prep" in r["Kernel_Name"]:
            preps.append(r)
            continue
        cur.append(r)
        n_ = r["Kernel_Name"]
        if "k_adam" in n_ or "k_small_tn" in n_:      # (k_small_tn: the small nets' weight-gradient launch with the optimiser in its epilogue)
            steps.append(cur)
            cur = []
    steps = [s for s in steps if len(s) == len(names)]
    steps = steps[len(steps) // 5:]                      # drop the first fifth (warm-up)
    preps = preps[len(preps) // 5:]

    def dur(r):
        return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3

    def wgs(r):
        return (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) * (int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]))
    out = {}
    for j, name in enumerate(names):
        d = [dur(s[j]) for s in steps]
        out[name] = {"calls": len(d), "avg_us": round(sum(d) / len(d), 2), "min_us": round(min(d), 2), "workgroups": wgs(steps[0][j])}
    pd = [dur(r) for r in preps]
    # both lists lost their first fifth, so their ratio is the number of steps one staging launch serves
    prep = {"calls": len(pd), "avg_us": round(sum(pd) / len(pd), 2), "min_us": round(min(pd), 2),
            "workgroups": max(wgs(r) for r in preps), "steps_per_call": round(len(steps) / len(pd), 1),
            "avg_us_per_step": round(sum(pd) / len(steps), 2)}
    res = {"source": "rocprofv3 --kernel-trace, %d graph-replayed steps" % len(steps), "prep": prep,
           "launches": out, "sum_avg_us": round(sum(v["avg_us"] for v in out.values()) + prep["avg_us_per_step"], 2)}
    # The profiler slows the run down (and DVFS differs): say by how much, so that nobody reads the sum as the step time.
    res["unprofiled_ms_per_step"] = oj.get("ms_per_step")
    if len(sys.argv) > 3:
        for line in open(sys.argv[3]):
            if line.startswith("{") and '"ms_per_step"' in line:
                res["profiled_run_ms_per_step"] = json.loads(line)["ms_per_step"]
    res["note"] = ("kernel durations under rocprofv3 (dispatch begin -> end); the profiled run is slower than the unprofiled one "
                   "(profiled_run_ms_per_step vs unprofiled_ms_per_step) and the inter-kernel gaps are not in sum_avg_us: "
                   "these averages rank the launches, they do not decompose the unprofiled step")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
