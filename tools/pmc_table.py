"""Per-kernel HBM traffic table from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; see tools/pmc_traffic.py):
average KB per launch for every kernel with >= 5 launches.  usage: python tools/pmc_table.py <f.csv> <w.csv>"""
import collections
import csv
import sys


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[(r["Kernel_Name"][:48], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print(f"{'kernel':48s} {'grid':>8s} {'n':>4s} {'fetch x2 MB':>12s} {'write MB':>9s} {'total MB':>9s}")
for k in sorted(f, key=lambda k: -(2 * sum(f[k]) / len(f[k]) + (sum(w[k]) / len(w[k]) if k in w else 0))):
    if len(f[k]) < 5:
        continue
    a = 2 * sum(f[k]) / len(f[k]) / 1024
    b = sum(w[k]) / len(w[k]) / 1024 if k in w else 0.0
    print(f"{k[0]:48s} {k[1]:8d} {len(f[k]):4d} {a:12.2f} {b:9.2f} {a + b:9.2f}")
