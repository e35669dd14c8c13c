#!/usr/bin/env python3
"""Per-(kernel, grid size) SQ counters of the scan's step launches from one rocprofv3 PMC pass (developer tool):

    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
              --kernel-trace --output-format csv -d D -o w -- python3 tools/stall_probe.py 3
    python tools/pmc_widths.py D/w_counter_collection.csv out.json

Question: the launches with more than 8 row blocks cost ~2 us more than the narrow ones -- is that
workgroup DISPATCH (then wave-cycles per wave stay flat and GUI_ACTIVE grows with the workgroup count) or contention
INSIDE the CU / L2 (then cycles per wave grow)?  SQ_WAVE_CYCLES etc. count quad-cycles summed over waves."""
import collections, csv, json, sys

src, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(src) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "step" not in name:
            continue
        short = name[name.index("::") + 2:][:28] if "::" in name else name[:28]
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 256)) or 256)
        key = "%s|%d" % (short, grid // max(wg, 1))
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[key] += 1
res = {}
for k in sorted(acc, key=lambda s: (s.split("|")[0], int(s.split("|")[1]))):
    v, n = acc[k], max(cnt[k], 1)
    waves = v.get("SQ_WAVES", 0.0) / n
    e = {"launches": n, "workgroups": int(k.split("|")[1]), "waves_per_launch": round(waves, 1),
         "gui_active_cycles_per_xcd": round(v.get("GRBM_GUI_ACTIVE", 0.0) / n / 8.0, 1)}
    for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if c in v:
            e[c.lower() + "_per_wave_x4"] = round(4.0 * v[c] / n / max(waves, 1.0), 1) if c != "SQ_BUSY_CYCLES" else round(v[c] / n, 1)
    res[k] = e
    print(k, e)
json.dump(res, open(out, "w"), indent=1)
