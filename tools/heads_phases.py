"""clock64() phase profile of k_y_heads (HL_HEADS_CLK=1): per-phase mean over all waves, and the spread of start / end."""
import ctypes as C, os, sys
os.environ["HL_HEADS_CLK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B_
from hlvae_amd import _lib, synthetic
from hlvae_amd.HLVAE import HLVAE
from hlvae_amd.training import ELBOTrainer
from hlvae_amd.datafeed import CompactDataset
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
src = synthetic.make_d4(n_subjects=batch // 20 + 2, T=20, seed=100)
model = HLVAE([src.cov_dim_ext, [500], 32, [500], 5], src.types_info, src.n_variables, conv=False, max_batch=batch, materialize_samples=False).to(dev)
tr = ELBOTrainer(model, P_total=50, kl="normal", max_batch=batch, metrics=True)
ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
rows = torch.arange(batch, dtype=torch.int32, device=dev)
for _ in range(5):
    tr.step_rows(ds, rows, 26)
torch.cuda.synchronize()
lib = _lib.load()
lib._handle  # noqa
fn = C.CDLL(_lib.LIB_PATH).hlvae_debug_heads_clk
grid = 81 * (batch // 64)
buf = (C.c_longlong * (48 * grid))()
n = fn(buf, grid)
a = np.frombuffer(buf, dtype=np.int64).reshape(grid, 4, 12)[:n].astype(np.float64)
t0 = a[..., 0].min()
names = ["prefetch issue", "gemm main loop", "acc -> lds", "heads + loglik", "row sums", "dyT + d by", "barrier", "grad shuffle-reduce",
         "dy row-major", "barrier", "atomics"]
print(f"workgroups {n}; first start 0, last start {a[..., 0].max() - t0:.0f}, last end {a[..., 11].max() - t0:.0f} (clock64 ticks, 100 MHz => x10 ns)")
for i, nm in enumerate(names):
    d = a[..., i + 1] - a[..., i]
    print(f"  {nm:22s} mean {d.mean():8.1f}  p10 {np.percentile(d, 10):8.1f}  p90 {np.percentile(d, 90):8.1f}")
print(f"  {'total per wave':22s} mean {(a[..., 11] - a[..., 0]).mean():8.1f}")
