"""Per-step statistics from a rocprofv3 --kernel-trace CSV of bench.py: step period (start of one k_mid_fwd_fused to the next),
and for every kernel of a step its start offset / duration, averaged over the steady-state steps (median).
usage: python tools/step_stats.py <kernel_trace.csv> [n_kernels_first_steps_to_skip]"""
import csv
import statistics as st
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# step marker: the fused middle of the forward pass (one launch per step in every configuration); offsets are relative to it
starts = [i for i, r in enumerate(rows) if "k_mid_fwd_fused" in r["Kernel_Name"]]
starts = starts[len(starts) // 3:]                      # steady state (graph replays)
per = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3 for a, b in zip(starts[:-1], starts[1:])]
print(f"steps {len(per)}  period us: median {st.median(per):.1f}  mean {st.mean(per):.1f}  p10 {sorted(per)[len(per) // 10]:.1f}  p90 {sorted(per)[9 * len(per) // 10]:.1f}")
hist = {}
for p in per:
    hist[int(p // 10) * 10] = hist.get(int(p // 10) * 10, 0) + 1
print("period histogram (10 us bins):", dict(sorted(hist.items())))
agg = {}
fast = sorted(per)[len(per) // 10]                      # the replayed-graph population (the eager passes of bench.py are slower)
n_fast = 0
for (a, b), p in zip(zip(starts[:-1], starts[1:]), per):
    if p > 1.2 * fast:
        continue
    n_fast += 1
    t0 = int(rows[a]["Start_Timestamp"])
    seen = {}
    for r in rows[a:b]:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]
        k = seen.get(n, 0)
        seen[n] = k + 1
        agg.setdefault((n, k, r["Queue_Id"]), []).append(((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
out = sorted(((st.median(x[0] for x in v), st.median(x[1] for x in v), len(v), k) for k, v in agg.items()))
for s, d, n, (name, k, q) in out:
    if n > n_fast // 2:
        print(f"{s:8.1f} {s + d:8.1f} {d:6.1f}  q{q}  {name}#{k}  (n={n})")
