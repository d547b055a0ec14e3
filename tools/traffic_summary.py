#!/usr/bin/env python3
"""HBM bytes per launch of the bench kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: the counters are in KB; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads and is doubled; WRITE_SIZE is exact.

    python tools/traffic_summary.py DIR_FETCH DIR_WRITE --match k_step_imu9 [--epochs-per-launch 25] > profiles/traffic_latest.json
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
    # launches of the timed region dominate; drop a shorter tail launch if the launch sizes differ
    top = [x for x in v if x > 0.8 * v[-1]]
    return sum(top) / len(top), len(top), len(v)


def main():
    a = [x for x in sys.argv[1:]]
    match, epl = "k_step_imu9", 25
    if "--match" in a:
        i = a.index("--match"); match = a[i + 1]; del a[i:i + 2]
    if "--epochs-per-launch" in a:
        i = a.index("--epochs-per-launch"); epl = int(a[i + 1]); del a[i:i + 2]
    dfetch, dwrite = a[0], a[1]
    fkb, nf, _ = mean_counter(dfetch, "FETCH_SIZE", match)
    wkb, nw, _ = mean_counter(dwrite, "WRITE_SIZE", match)
    read, write = fkb * 1024 * 2, wkb * 1024
    T = 65536
    out = {
        "hbm_bytes_per_launch": read + write,
        "launch": f"{match}, {epl} epochs x {T} tags per launch (bench.py default)",
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                  "(MI355X_MICROARCH.md, HBM section); counters are KB; FETCH_SIZE doubled (gfx950 correction), "
                  "WRITE_SIZE exact; mean over the full-size launches of the run",
        "raw": {"FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "launches_averaged": [nf, nw]},
        "corrected_bytes": {"read": read, "write": write},
        "algorithmic_bytes_per_launch": 544 * T * epl,
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
