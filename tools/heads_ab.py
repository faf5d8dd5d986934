"""Bit-level A/B of the head kernel's outputs (HL_HEADS_CORE=1 / default in separate processes): prints sha1 of dY (both layouts),
log_p_x, x_hat and the dense gradients after identical steps (learning rate 0)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hlvae_amd import synthetic
from hlvae_amd.HLVAE import HLVAE
from hlvae_amd.training import ELBOTrainer
from hlvae_amd.datafeed import CompactDataset
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
src = synthetic.make_d4(n_subjects=B // 20 + 5, T=20, seed=11)
dims = [src.cov_dim_ext, [500], 32, [500], 5]
ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
rows = torch.tensor(np.arange(B).astype(np.int32), device=dev)
eps = torch.randn(B, 32, generator=torch.Generator().manual_seed(40)).to(dev)
torch.manual_seed(5)
model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=B, materialize_samples=False).to(dev)
tr = ELBOTrainer(model, P_total=30, kl="normal", max_batch=B, lr=0.0, metrics=True)
for it in range(3):
    tr.step_rows(ds, rows, 26, eps=eps)
    torch.cuda.synchronize()
    t = model._ws_t
    d = model._dims
    out = {k: hashlib.sha1((t[k].view(torch.int16) if t[k].dtype == torch.bfloat16 else t[k]).cpu().numpy().tobytes()).hexdigest()[:12] for k in ("dy", "dyT", "log_p_x", "xhat")}
    out["G"] = hashlib.sha1(t["G"][int(d.atomic_region):int(d.arena_size)].cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"core={os.environ.get('HL_HEADS_CORE', '2')} rows={B} step {it}: nll {float(tr.scalars()['nll_sum']):.6f}", out)
