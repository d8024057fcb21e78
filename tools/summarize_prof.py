#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / stats / PMC counter collection) into small per-kernel summaries.

    python tools/summarize_prof.py <rocprof_out_dir> <summary_json>
Per kernel name: dispatch count, total/average duration (kernel trace) and, for --pmc passes, the per-launch average
of every collected counter.  gfx950 corrections (MI355X_MICROARCH.md, HBM section) are applied by the consumer
(profiles/README.md): FETCH_SIZE is reported in KiB-like units of 1024 B and under-counts wide coalesced reads by 2x.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("idxtts::", "").strip()


def main(src, dst):
    out = {}
    for path in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        agg = defaultdict(lambda: [0, 0])
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if os.environ.get("SUMMARIZE_BY_GRID"):     # split one kernel's launches by launch geometry
                    k += " grid=%s,%s wg=%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?"))
                agg[k][0] += 1
                agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        out["kernel_trace"] = {k: {"calls": v[0], "total_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3} for k, v in agg.items()}
    for path in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                c = agg[k][r["Counter_Name"]]
                c[0] += 1
                c[1] += float(r["Counter_Value"])
        out["counters"] = {k: {cn: {"launches": v[0], "avg_per_launch": v[1] / v[0], "total": v[1]} for cn, v in d.items()} for k, d in agg.items()}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", dst, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
