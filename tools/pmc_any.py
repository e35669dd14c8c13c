#!/usr/bin/env python3
"""Average every counter of rocprofv3 counter_collection.csv files per kernel-name substring:  pmc_any.py <substr> <csv>..."""
import csv, sys
from collections import defaultdict
sub = sys.argv[1]
acc, cnt = defaultdict(float), defaultdict(int)
for path in sys.argv[2:]:
    for row in csv.DictReader(open(path)):
        if sub not in row.get("Kernel_Name", ""):
            continue
        k = row["Counter_Name"]
        acc[k] += float(row["Counter_Value"]); cnt[k] += 1
for k in sorted(acc):
    print("%-34s %16.0f   (%d samples)" % (k, acc[k] / cnt[k], cnt[k]))
