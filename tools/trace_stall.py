#!/usr/bin/env python3
"""Summarise a rocprofv3 --hip-trace --kernel-trace csv dump: the longest HIP API calls (the early-process stall)."""
import csv, glob, os, sys
root = sys.argv[1]
for f in glob.glob(os.path.join(root, "**", "*hip_api_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), "calls")
    for r in rows:
        r["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    big = sorted(rows, key=lambda r: -r["dur"])[:25]
    print("longest calls: dur_us  t_since_start_ms  function")
    for r in big:
        print("%10.1f  %10.1f  %s" % (r["dur"], (int(r["Start_Timestamp"]) - t0) / 1e6, r["Function"]))
    agg = {}
    for r in rows:
        a = agg.setdefault(r["Function"], [0, 0.0]); a[0] += 1; a[1] += r["dur"]
    print("by function: calls total_ms")
    for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
        print("%-40s %8d %10.2f" % (k, n, d / 1e3))
