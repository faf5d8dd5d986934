"""How much does the step time of the SAME captured chain vary from one capture (graph instantiation) to the next?
The HIP graph executor maps the step's streams onto hardware queues when the graph is instantiated; bench runs of identical code
measured 0.120 - 0.131 ms/step.  This captures the 8-step chain N times in one process and times each capture."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B_
from hlvae_amd import synthetic
from hlvae_amd.HLVAE import HLVAE
from hlvae_amd.training import ELBOTrainer
from hlvae_amd.datafeed import CompactDataset

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
src = synthetic.make_d4(n_subjects=50, T=20, seed=100)
torch.manual_seed(0)
model = HLVAE([src.cov_dim_ext, [500], 32, [500], 5], src.types_info, src.n_variables, conv=False, max_batch=512, materialize_samples=False).to(dev)
tr = ELBOTrainer(model, P_total=50, kl="normal", max_batch=512, metrics=True)
ring = B_.build_ring(src, 512, 4)
ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
R = [torch.tensor(b["rows"].astype(np.int32), device=dev) for b in ring]
PB = [b["P_batch"] for b in ring]
nx = [R[(i + 1) % 4] for i in range(4)]
idx = [j % 4 for j in range(8)]
res = []
for c in range(N):
    tr.capture_rows(("v", c), ds, [R[k] for k in idx], [PB[k] for k in idx], next_rows=[nx[k] for k in idx])
    tr.prime_rows(ds, R[0])
    for _ in range(5):
        tr.replay(("v", c))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tr.replay(("v", c))
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 400 * 1e3)
    print(f"capture {c}: {res[-1]:.4f} ms/step", flush=True)
# replay the first capture again: is the time a property of the capture or of the moment?
for c in (0, 1, 0, 1):
    tr.prime_rows(ds, R[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tr.replay(("v", c))
    torch.cuda.synchronize()
    print(f"capture {c} again: {(time.perf_counter() - t0) / 400 * 1e3:.4f} ms/step", flush=True)
print("min %.4f median %.4f max %.4f" % (min(res), float(np.median(res)), max(res)))
