"""Build profiles/r3_pmc_traffic.json (one entry per workload, the key bench.py looks up) from two rocprofv3 PMC passes of
`bench.py --no-graph` per workload:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -o f -- python3 bench.py --no-graph ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dirW> -o w -- python3 bench.py --no-graph ...
Counters are in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE undercounts 16-B-per-lane
coalesced reads by 2x; WRITE_SIZE is exact for 16-B stores and float atomics.
usage: python tools/pmc_traffic.py abi=<library ABI version> <workload_key>=<fetch csv>,<write csv> [...] > profiles/r3_pmc_traffic.json
(bench.py refuses the file when its "abi" differs from the library it runs with)"""
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
    "gp_param_grad": ("k_gp_param_grad", "max"),
    "gp_chain": ("k_gp_chain_rb<1>", None),
    "gp_spd_inv": ("k_gp_spd_inv", None),
    "gp_gemm": ("k_gp_gemm", "max"),
    "gp_subject_fwd": ("k_gp_subject_fwd", None),
    "gp_subject_bwd": ("k_gp_subject_bwd", None),
    "dWy": ("k_gemm_f32<", "max"),
    "dWy_adam": ("k_gemm_adam", None),
    "dU_splitk": ("k_gemm_splitk", "min"),
    "enc1_splitk": ("k_gemm_splitk", "max"),
    "mid_fwd_fused": ("k_mid_fwd_fused", None),
    "mid_bwd_fused": ("k_mid_bwd_fused", None),
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
    out = {"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) of "
                     "bench.py --no-graph --no-cpu-baseline --steps 20 --warmup 5 <workload flags> on MI355X; tools/pmc_traffic.py",
           "correction": "FETCH_SIZE x2 for 16-B-per-lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section); "
                         "WRITE_SIZE exact for 16-B stores and float atomics; counters are in KB"}
    for arg in sys.argv[1:]:
        key, files = arg.split("=")
        if key == "abi":
            out["abi"] = int(files)
            continue
        ff, fw = files.split(",")
        f, w = per_kernel(ff, "FETCH_SIZE"), per_kernel(fw, "WRITE_SIZE")
        kern = {}
        for label, (sub, which) in KERNELS.items():
            a, b = pick(f, sub, which), pick(w, sub, which)
            if a is None or b is None:
                continue
            kern[label] = {"kernel": a[0][:60], "grid": a[1], "FETCH_SIZE_KB": a[2], "WRITE_SIZE_KB": b[2],
                           "traffic_bytes_per_launch": int(round((2.0 * a[2] + b[2]) * 1024))}
        out[key] = {"kernels": kern}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
