"""Per-kernel times of one training step (eager, every kernel bracketed by HIP events on its stream: the library's
hlvae_prof facility) plus the step time of the replayed HIP graph, for quick A/B runs on a GPU box.

    python tools/kernel_times.py [--workload d4|tabular] [--rows N] [--batch B] [--kl normal|gp] [--conv] [--sharded]
                                 [--env NAME=v1,v2,...]      # repeat the measurement for each value of an environment
                                                             # variable the library reads at launch time (A/B switches)
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench as B_   # noqa: E402
from hlvae_amd import _lib  # noqa: E402


def measure(a, label):
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset
    from hlvae_amd.parallel import DataParallel
    dev = torch.device("cuda:0")
    if a.spec == "d4sorted":      # D4's variables grouped by kind (972 cat5 then 324 real): homogeneous 16-variable tiles
        from hlvae_amd import synthetic
        n_subj = (a.rows or 1000) // 20
        src = synthetic.make_tabular(n_rows=n_subj * 20, T=20, seed=100, spec=[("cat", 5)] * 972 + [("real", 1)] * 324)
    elif a.spec == "tabsorted":   # the 64-feature mix grouped by kind (no interleave)
        from hlvae_amd import synthetic
        n_subj = (a.rows or 65536) // 16
        src = synthetic.make_tabular(n_rows=n_subj * 16, T=16, seed=100, spec=synthetic.tabular_type_spec(interleave=False), expanded=False)
    else:
        src, n_subj = B_.make_source(a, 0)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    torch.manual_seed(0)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=a.conv, max_batch=a.batch, materialize_samples=False,
                  group_variables=not a.no_group).to(dev)
    gp = None
    if a.kl == "gp":
        from hlvae_amd.elbo_functions import GPPriorHIP
        gp = GPPriorHIP.from_reference_config(model, src, n_subj, dev)
    tr = ELBOTrainer(model, P_total=n_subj, kl=None if a.kl == "none" else a.kl, gp=gp, max_batch=a.batch,
                     dp=DataParallel.single() if a.sharded else None, metrics=True)
    ring = B_.build_ring(src, a.batch, 4)
    if hasattr(src, "raw"):
        ds = CompactDataset.from_raw(src.raw, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    else:
        ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    for b in ring:
        b["rows_dev"] = torch.tensor(b["rows"].astype(np.int32), device=dev)
        b["groups_dev"] = torch.tensor(b["groups"], device=dev) if a.kl == "gp" else None
    lib = _lib.load()
    b0 = ring[0]
    for _ in range(3):
        tr.step_rows(ds, b0["rows_dev"], b0["P_batch"], groups=b0["groups_dev"])
    torch.cuda.synchronize()
    lib.hlvae_prof_enable(1)
    n = a.n
    for _ in range(n):
        tr.step_rows(ds, b0["rows_dev"], b0["P_batch"], groups=b0["groups_dev"])
    lib.hlvae_prof_enable(0)
    buf = C.create_string_buffer(1 << 16)
    _lib.check(lib.hlvae_prof_report(buf, len(buf)), "prof_report")
    rowsT = []
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.split()
        rowsT.append((1e3 * float(tot) / n, int(cnt) / n, name))
    rowsT.sort(reverse=True)
    print(f"== {label}: eager per-kernel (us per step, launches per step)")
    for us, k, name in rowsT:
        print(f"   {us:8.2f}  x{k:<4.1f} {name}")
    print(f"   {sum(r[0] for r in rowsT):8.2f}  sum")
    # replayed graph (4 chained pipelined steps), like bench.py
    if not a.no_graph:
        R = [b["rows_dev"] for b in ring]
        G = [b["groups_dev"] for b in ring]
        PB = [b["P_batch"] for b in ring]
        nx = [R[(i + 1) % 4] for i in range(4)]
        pf = not a.conv
        tr.capture_rows("ring", ds, R, PB, next_rows=nx if pf else None, groups=G)
        if pf:
            tr.prime_rows(ds, R[0])
        for _ in range(5):
            tr.replay("ring")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = max(10, a.n)
        for _ in range(reps):
            tr.replay("ring")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (4 * reps)
        print(f"   graph replay: {1e3 * dt:.4f} ms/step, {a.batch / dt:.0f} rows/s")
    model._release_device_state()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="d4")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--kl", default="normal")
    ap.add_argument("--conv", action="store_true")
    ap.add_argument("--sharded", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--feed", default="compact")
    ap.add_argument("-n", type=int, default=30)
    ap.add_argument("--env", default=None)
    ap.add_argument("--spec", default=None)
    ap.add_argument("--no-group", action="store_true", help="head kernel walks the variables in their own order")
    a = ap.parse_args()
    if a.env:
        name, vals = a.env.split("=")
        for v in vals.split(","):
            os.environ[name] = v
            measure(a, f"{name}={v}")
    else:
        measure(a, "baseline")


if __name__ == "__main__":
    main()
