#!/usr/bin/env python3
"""HBM bytes per launch of a step kernel from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in SEPARATE passes),
corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: the counters are in KB; on
gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads and is doubled; WRITE_SIZE is exact.

Two runs with different epochs-per-launch separate what a launch pays once (the per-tag state, read and written) from
what it pays per epoch (measurements in, poses out):  bytes(E) = fixed + E * per_epoch.  The result is merged into
profiles/traffic_latest.json under the kernel's name; bench.py rebuilds `roofline.traffic` from it for the launch length
it actually timed and labels it as static.

    python tools/traffic_summary.py --match k_step_imu9 --key "k_step_imu9<double,float,8>" --tags 65536 \
        --run 25 DIR_FETCH DIR_WRITE --run 5 DIR_FETCH DIR_WRITE --note "..." --merge profiles/traffic_latest.json
"""
import csv
import glob
import json
import os
import sys


def mean_counter(d, counter, match):
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter or match not in row.get("Kernel_Name", ""):
                continue
            k = (f, row["Dispatch_Id"])
            vals[k] = vals.get(k, 0.0) + float(row["Counter_Value"])
    v = sorted(vals.values())
    if not v:
        raise SystemExit(f"no {counter} rows for kernels matching {match!r} under {d}")
    # launches of the timed region dominate; drop shorter launches (warm-up, tails) when the launch sizes differ
    top = [x for x in v if x > 0.8 * v[-1]]
    return sum(top) / len(top), len(top)


def main():
    a = sys.argv[1:]
    opt = {"--match": None, "--key": None, "--tags": "65536", "--note": "", "--merge": None}
    runs = []
    i = 0
    while i < len(a):
        if a[i] == "--run":
            runs.append((int(a[i + 1]), a[i + 2], a[i + 3]))
            i += 4
        elif a[i] in opt:
            opt[a[i]] = a[i + 1]
            i += 2
        else:
            raise SystemExit("unknown argument " + a[i])
    match, key, T = opt["--match"], opt["--key"] or opt["--match"], int(opt["--tags"])
    pts = []
    for E, dfetch, dwrite in runs:
        fkb, nf = mean_counter(dfetch, "FETCH_SIZE", match)
        wkb, nw = mean_counter(dwrite, "WRITE_SIZE", match)
        pts.append({"epochs_per_launch": E, "FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "launches_averaged": [nf, nw],
                    "read_bytes": fkb * 1024 * 2, "write_bytes": wkb * 1024, "bytes": fkb * 1024 * 2 + wkb * 1024})
    if len(pts) >= 2:
        p, q = pts[0], pts[-1]
        per_epoch = (p["bytes"] - q["bytes"]) / (p["epochs_per_launch"] - q["epochs_per_launch"])
        fixed = p["bytes"] - per_epoch * p["epochs_per_launch"]
    else:
        raise SystemExit("need two --run entries with different epochs per launch")
    entry = {"bytes_per_launch_fixed": fixed, "bytes_per_epoch": per_epoch, "tags": T,
             "fixed_bytes_per_tag": fixed / T, "per_epoch_bytes_per_tag": per_epoch / T,
             "measured_with": opt["--note"],
             "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                       "(MI355X_MICROARCH.md, HBM section); counters are KB; FETCH_SIZE doubled (gfx950 correction), "
                       "WRITE_SIZE exact; mean over the full-size launches of each run; two launch lengths -> fixed + per-epoch",
             "runs": pts}
    out = {}
    if opt["--merge"] and os.path.exists(opt["--merge"]):
        out = json.load(open(opt["--merge"]))
    out[key] = entry
    if opt["--merge"]:
        json.dump(out, open(opt["--merge"], "w"), indent=1)
    json.dump({key: entry}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
