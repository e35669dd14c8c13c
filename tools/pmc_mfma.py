#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass (developer tool).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d D -o m -- python3 bench.py ...
    python tools/pmc_mfma.py D/m_counter_collection.csv out.json

util = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): the share of the kernel's
active cycles in which a SIMD's matrix pipe is busy (both counters are sums over the chip; calibrated on
a 4096^3 fp32 GEMM: 51 % by counters against 80 / 157 TFLOP/s by time).  PMC mode runs every dispatch
isolated, so these are utilisations of the kernels on their own, not of the step."""
import collections
import csv
import json
import sys


def main():
    src, out = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    with open(src) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"][:90]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
    res = {}
    for k, v in acc.items():
        busy, gui = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or gui <= 0:
            continue
        res[k] = {"calls": cnt[k], "mfma_busy_cycles_per_call": round(busy / cnt[k]), "gui_active_cycles_per_call": round(gui / cnt[k]),
                  "mfma_util": round((busy / 1024.0) / (gui / 8.0), 4)}
    import hashlib, os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seq-recommendations_amd", "libseqrec_hip.so")
    res["_lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]      # bench.py drops figures taken on another build
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(((k, v) for k, v in res.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["mfma_util"]):
        print("%-90s util %.3f  calls %d" % (k, v["mfma_util"], v["calls"]))


if __name__ == "__main__":
    main()
