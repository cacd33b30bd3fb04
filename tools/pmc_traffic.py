#!/usr/bin/env python3
"""Per-launch HBM traffic from the two rocprofv3 --pmc passes of tools/pmc_traffic.sh.

Launches of one step are told apart by their position between two k_prep dispatches (every GEMM
launch runs the same kernel, k_grouped).  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the
counters are in KiB and gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes
(MI355X_MICROARCH.md, "HBM").  Writes profiles-ready JSON to <out>/traffic.json."""
import collections
import csv
import glob
import json
import sys

out, cfg = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "c2")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

archs, B, dtype, label = bench.CONFIGS[cfg]


def per_position(counter, sub):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (out, sub), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seq, pos, steps = collections.defaultdict(list), -1, 0
    for r in rows:
        n = r["Kernel_Name"]
        if "k_prep" in n:
            pos, steps = 0, steps + 1
            if steps > 12:                      # skip warm-up
                seq["prep"].append(float(r["Counter_Value"]))
            continue
        if pos < 0:
            continue
        if "k_grouped" in n or "k_small" in n or "k_adam" in n:
            pos += 1
            if steps > 12:
                seq[pos].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in seq.items() if v}


fetch = per_position("FETCH_SIZE", "fetch")
write = per_position("WRITE_SIZE", "write")
L = len(archs[0]["n_hidden"])
cd = lambda a, b: -(-a // b)
latent_alone = (sum(cd(B, 256) * cd(na["n_input"], 64) for na in archs) >= 192 and sum(cd(B, 128) * cd(na["n_input"], 128) for na in archs) >= 192)
import os
# small nets: fwd_dec1 and bwd_head ride in the launch that produces their input (avae_host.hip::fuse_tail; AVAE_NO_TAIL=1 keeps them apart)
tail = not latent_alone and not os.environ.get("AVAE_NO_TAIL")
names = (["fwd_enc%d" % (k + 1) for k in range(L)] + (["fwd_head+fwd_dec1"] if tail else ["fwd_head"]) + (["latent"] if latent_alone else [])
         + ["fwd_dec%d" % (k + 1) for k in range(1 if tail else 0, L)] + ["fwd_out_loss", "bwd_out"]
         + ["bwd_dec%d" % (k + 1) for k in range(L - 1, 0, -1)] + (["bwd_dec1_latent+bwd_head"] if tail else ["bwd_dec1_latent", "bwd_head"])
         + ["bwd_enc%d" % (k + 1) for k in range(L - 1, 0, -1)])
fused_adam = tail and not os.environ.get("AVAE_NO_ADAM_FUSE") and not os.environ.get("AVAE_NO_LEAN")      # small nets: "wgrad+adam" is ONE launch
if fused_adam:
    names += ["wgrad+adam"]
else:
    n_wg = len([k for k in fetch if k != "prep"]) - len(names) - 1            # the step ends with k_adam
    names += (["wgrad"] if n_wg == 1 else ["wgrad%d" % (i + 1) for i in range(n_wg)]) + ["adam"]
res = {"config": label, "note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, mean over steps; rocprofv3 --pmc, separate passes",
       "launches": {}}
for key in ["prep"] + list(range(1, len(names) + 1)):
    name = key if key == "prep" else names[key - 1]
    f, w = fetch.get(key, 0.0), write.get(key, 0.0)
    res["launches"][name] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes": int((2 * f + w) * 1024)}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
tot = sum(v["hbm_bytes"] for v in res["launches"].values())
for k, v in res["launches"].items():
    print("%-16s fetch %9.1f KiB  write %9.1f KiB  hbm %8.2f MB" % (k, v["FETCH_SIZE_KiB"], v["WRITE_SIZE_KiB"], v["hbm_bytes"] / 1e6))
print("step total %.2f MB" % (tot / 1e6))
