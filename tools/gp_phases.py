"""clock64() phase profile of k_gp_subject_fwd / k_gp_subject_bwd (one GP training step, BASELINE configs[4] shapes): mean clocks per
phase of thread 0 over the first 2048 workgroups."""
import ctypes as C, os, sys
os.environ.setdefault("HL_GP_SERIAL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B_
from hlvae_amd import _lib
lib = _lib.load()
dbg = C.CDLL(_lib.LIB_PATH).hlvae_debug_gp_clk
dbg.argtypes = [C.c_void_p, C.c_int]
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-also", "--no-graph", "--no-in-step", "--workload", "d4", "--rows", "50000", "--batch", "1024",
            "--kl", "gp", "--steps", "6", "--warmup", "4"]
assert dbg(None, 1) == 0
B_.main()
buf = (C.c_longlong * (2 * 2048 * 10))()
assert dbg(buf, 0) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(2, 2048, 10).astype(np.float64)
names = [["staging issued", "staged + residual", "covariance pairs", "Gauss-Jordan", "iB, K0 written", "v, g_mu, g_lv", "V = iB Ks", "u / P1 atomics"],
         ["V_s, Y_s staged", "iB w", "pair loop", "flush"]]
for k, kn in enumerate(("k_gp_subject_fwd", "k_gp_subject_bwd")):
    x = a[k]
    ok = x[:, 0] > 0
    print(f"{kn}: {int(ok.sum())} workgroups stamped; span first start -> last end {x[ok][:, :len(names[k]) + 1].max() - x[ok, 0].min():.0f} clocks")
    for i, nm in enumerate(names[k]):
        d = x[ok, i + 1] - x[ok, i]
        print(f"   {nm:20s} mean {d.mean():9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
    print(f"   {'total':20s} mean {(x[ok, len(names[k])] - x[ok, 0]).mean():9.0f}")
    if k == 1:
        d = x[ok, 5] - x[ok, 2]
        print(f"   (thread 0: iB w -> its dot products done: mean {d[d > 0].mean():9.0f})")
