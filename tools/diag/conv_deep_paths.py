"""one optimiser step of the convolutional model with two hidden layers per side: fused tile-Adam path vs the data-parallel (flat Adam)
path at world size 1 -- per-tensor difference of the UPDATE"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from hlvae_amd import synthetic
from hlvae_amd.HLVAE import HLVAE
from hlvae_amd.training import ELBOTrainer
from hlvae_amd.parallel import DataParallel
from hlvae_amd.datafeed import CompactDataset

dev = torch.device("cuda:0")
hid_e, hid_d = ([500, 132], [260, 500]) if len(sys.argv) < 2 or sys.argv[1] == "deep" else ([500], [500])
src = synthetic.make_d4(n_subjects=30, T=20, seed=11)
dims = [src.cov_dim_ext, hid_e, 32, hid_d, 5]
ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
rows = torch.tensor(np.arange(512).astype(np.int32), device=dev)
eps = torch.randn(512, 32, generator=torch.Generator().manual_seed(40)).to(dev)
out = []
for dp in (None, DataParallel.single()):
    torch.manual_seed(5)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=True, max_batch=512, materialize_samples=False).to(dev)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tr = ELBOTrainer(model, P_total=30, kl="normal", max_batch=512, dp=dp, metrics=True)
    tr.step_rows(ds, rows, 26, eps=eps)
    nll = float(tr.scalars()["nll_sum"])
    after = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}
    torch.cuda.synchronize()
    out.append((nll, before, after, grads))
(n0, b0, a0, g0), (n1, b1, a1, g1) = out
print("nll", n0, n1)
for k in a0:
    if a0[k].numel() == 0:
        continue
    u0, u1 = (a0[k] - b0[k]).double(), (a1[k] - b1[k]).double()
    du = float((u0 - u1).abs().max())
    gd = float((g0[k].double() - g1[k].double()).abs().max() / (g0[k].double().abs().max() + 1e-30)) if k in g0 and k in g1 else -1
    print(f"{k:45s} |upd| {float(u0.abs().max()):.3e} / {float(u1.abs().max()):.3e}  max diff {du:.3e}  grad rel diff {gd:.2e}")
