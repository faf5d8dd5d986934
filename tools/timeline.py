"""Print the timeline of one steady-state step from a `rocprofv3 --kernel-trace` CSV: start / end / duration (us),
hardware queue and kernel, so that cross-queue gaps on the critical path are visible.
usage: python tools/timeline.py <..._kernel_trace.csv> [first-kernel-substring] [step-index]"""
import csv
import sys


def main():
    path = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_colstats"
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    i0, i1 = idx[k], idx[k + 1]
    t0 = int(rows[i0]["Start_Timestamp"])
    for r in rows[i0:i1 + 1]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{s / 1e3:8.1f} {e / 1e3:8.1f} {(e - s) / 1e3:6.1f}  q{r['Queue_Id']}  {r['Kernel_Name'][:56]}  grid={r['Grid_Size_X']}")


if __name__ == "__main__":
    main()
