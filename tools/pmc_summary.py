"""Average PMC counter values per kernel from rocprofv3 counter_collection CSVs (tools/pmc_run.sh).
usage: python tools/pmc_summary.py <kernel-substring> <dir-or-csv> [...]"""
import collections
import csv
import glob
import os
import sys

sub = sys.argv[1]
for path in sys.argv[2:]:
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:40], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (name, grid), cs in acc.items():
            n = max(len(v) for v in cs.values())
            if n < 4:
                continue
            print(f"{name} grid={grid} launches={n}")
            for c, v in sorted(cs.items()):
                v = v[len(v) // 4:]          # steady state
                print(f"    {c:34s} {sum(v) / len(v):16.1f}")
