"""Build profiles/r1_pmc_traffic.json from two rocprofv3 PMC passes of `bench.py --no-graph`:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -o f -- python3 bench.py --no-graph ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirW> -o w -- python3 bench.py --no-graph ...
Counters are in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE undercounts 16-B-per-lane
coalesced reads by 2x; WRITE_SIZE is exact for 16-B stores and float atomics.
usage: python tools/pmc_traffic.py <dirF>/f_counter_collection.csv <dirW>/w_counter_collection.csv > profiles/r1_pmc_traffic.json"""
import collections
import csv
import json
import sys

# bench.py roofline label <- (kernel-name substring, grid-size predicate)
KERNELS = {
    "y_heads_loglik": ("k_y_heads", None),
    "adam_wy_early": ("k_adam_tiled", "max"),              # the larger of the two k_adam_tiled grids of a step
    "adam_weights_shadows": ("k_adam_tiled", "min"),
    "normalize_pack": ("k_pack_compact", None),
    "dW1_dWd_dWmu": ("k_gemm_f32_group", None),
    "conv_enc_bwd": ("k_conv_enc_bwd", None),
    "gp_param_grad": ("k_gp_param_grad", None),
}


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[r["Kernel_Name"]][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


def pick(acc, sub, which):
    for name, grids in acc.items():
        if sub in name:
            sizes = sorted(g for g, v in grids.items() if len(v) >= 5)        # steady-state launches only
            if not sizes:
                continue
            g = sizes[-1] if which in (None, "max") else sizes[0]
            v = grids[g]
            return name, g, sum(v) / len(v)
    return None


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) of "
                     "bench.py --no-graph --no-cpu-baseline --steps 20 --warmup 5 on MI355X; tools/pmc_traffic.py",
           "correction": "FETCH_SIZE x2 for 16-B-per-lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section); "
                         "WRITE_SIZE exact for 16-B stores and float atomics; counters are in KB",
           "kernels": {}}
    for label, (sub, which) in KERNELS.items():
        a, b = pick(f, sub, which), pick(w, sub, which)
        if a is None or b is None:
            continue
        out["kernels"][label] = {"kernel": a[0][:60], "grid": a[1], "FETCH_SIZE_KB": a[2], "WRITE_SIZE_KB": b[2],
                                 "traffic_bytes_per_launch": int(round((2.0 * a[2] + b[2]) * 1024))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
