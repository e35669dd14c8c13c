#!/usr/bin/env python3
"""Per-kernel mean HBM traffic per launch from two rocprofv3 PMC passes (developer tool).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -o w -- python3 bench.py ...
    python tools/pmc_traffic.py A/f_counter_collection.csv B/w_counter_collection.csv out.json

FETCH_SIZE / WRITE_SIZE are in KiB; fetch_kib_x2 applies the gfx950 correction for wide coalesced
reads (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half the bytes of such streams)."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"][:90]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    fa, wa = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fa) | set(wa)):
        n = fa.get(k, [0, 0.0])[0] or wa.get(k, [0, 0.0])[0]
        f = fa[k][1] / fa[k][0] if k in fa and fa[k][0] else 0.0
        w = wa[k][1] / wa[k][0] if k in wa and wa[k][0] else 0.0
        res[k] = {"calls": n, "fetch_kib_raw": round(f, 2), "fetch_kib_x2": round(2 * f, 2), "write_kib": round(w, 2)}
    import hashlib, os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seq-recommendations_amd", "libseqrec_hip.so")
    res["_lib_sha16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]      # bench.py drops figures taken on another build
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, len(res) - 1, "kernels")


if __name__ == "__main__":
    main()
