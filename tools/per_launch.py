#!/usr/bin/env python3
"""Per-launch durations of the train step from a rocprofv3 --kernel-trace CSV.

  python tools/per_launch.py <kernel_trace.csv> <names.json|bench-json-line> > profiles/rNN_<cfg>_per_launch.json

k_grouped runs every GEMM launch of the step, so rocprofv3's per-kernel-name statistics lump them
together.  A graph replay issues the launches in a fixed order (k_prep, then the launches in the order
bench.py's `kernels_us` names them in the plan: fwd_*, bwd_*, wgrad*), so the n-th dispatch after a
k_prep is the n-th launch of the plan.  Steps whose dispatch count differs (warm-up, the eager timing
pass) are skipped.
"""
import csv
import json
import sys

ORDER_HINT = ["fwd_enc", "conv_enc", "conv_head", "fwd_head", "fwd_dec", "conv_dec", "conv_out", "fwd_out_loss",
              "bwd_out", "conv_bwd_dec", "bwd_dec", "bwd_head", "conv_bwd_enc", "bwd_enc", "wgrad"]


def main():
    trace, order_file = sys.argv[1], sys.argv[2]
    names = json.load(open(order_file))["order"]
    rows = [r for r in csv.DictReader(open(trace)) if "k_prep" in r["Kernel_Name"] or "k_grouped" in r["Kernel_Name"]
            or "k_gather" in r["Kernel_Name"] or "k_col2im" in r["Kernel_Name"] or "k_adam" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))            # the CSV is not in execution order
    steps, cur = [], None
    for r in rows:
        if "k_prep" in r["Kernel_Name"]:
            if cur is not None:
                steps.append(cur)
            cur = [r]
        elif cur is not None:
            cur.append(r)
    if cur:
        steps.append(cur)
    steps = [s for s in steps if len(s) == len(names) + 1]
    steps = steps[len(steps) // 5:]                      # drop the first fifth (warm-up)
    out = {}
    for j, name in enumerate(["prep"] + names):
        d = [(int(s[j]["End_Timestamp"]) - int(s[j]["Start_Timestamp"])) / 1e3 for s in steps]
        out[name] = {"calls": len(d), "avg_us": round(sum(d) / len(d), 2), "min_us": round(min(d), 2),
                     "workgroups": (int(steps[0][j]["Grid_Size_X"]) // int(steps[0][j]["Workgroup_Size_X"]))
                                   * (int(steps[0][j]["Grid_Size_Y"]) // int(steps[0][j]["Workgroup_Size_Y"]))}
    span = [(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e3 for s in steps]
    print(json.dumps({"source": "rocprofv3 --kernel-trace, %d graph-replayed steps" % len(steps),
                      "launches": out, "sum_avg_us": round(sum(v["avg_us"] for v in out.values()), 2),
                      "step_span_avg_us": round(sum(span) / len(span), 2)}, indent=1))


if __name__ == "__main__":
    main()
