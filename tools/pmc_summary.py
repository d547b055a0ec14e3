#!/usr/bin/env python3
"""Mean per-dispatch PMC counters per kernel from rocprofv3 --pmc output directories.

    python tools/pmc_summary.py gpurun_out/pmc_passA gpurun_out/pmc_passB [--match k_step] > profiles/xyz.json

Each directory is what `rocprofv3 --kernel-trace --pmc C1 C2 ... -d DIR --output-format csv -- prog` wrote. Counters of
different passes are merged per kernel name (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_* count quad-cycles on gfx950).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    acc = defaultdict(lambda: defaultdict(list))
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(dict)
            for row in csv.DictReader(open(f)):
                name = row.get("Kernel_Name", "")
                if match and match not in name:
                    continue
                key = (name, row.get("Dispatch_Id"))
                per_dispatch[key][row["Counter_Name"]] = per_dispatch[key].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            for (name, _), ctrs in per_dispatch.items():
                for c, v in ctrs.items():
                    acc[name][c].append(v)
    out = {}
    for name, ctrs in acc.items():
        out[name] = {c: sum(v) / len(v) for c, v in sorted(ctrs.items())}
        out[name]["_dispatches"] = max(len(v) for v in ctrs.values())
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
