#!/usr/bin/env python3
"""Pivot tools/bench_gemm2.py logs: rows = (shape, tile, splitk), columns = the '== label' sections; cell = median us."""
import re, sys
from collections import OrderedDict
sec, rows, secs = None, OrderedDict(), []
for line in open(sys.argv[1]):
    if line.startswith("=="):
        sec = line[2:].strip(); secs.append(sec); continue
    m = re.match(r"(\S+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) (v\d) tile=(\d) splitk=\s*(\d+) : med\s+([\d.]+) us", line)
    if m:
        rows.setdefault((m.group(1), m.group(5), int(m.group(6)), int(m.group(7))), {})[sec] = float(m.group(8))
print("%-12s %-3s %4s %3s " % ("shape", "", "tile", "sk") + " ".join("%10s" % s[:10] for s in secs))
for k, v in rows.items():
    print("%-12s %-3s %4d %3d " % k + " ".join("%10.1f" % v[s] if s in v else "%10s" % "-" for s in secs))
