"""time hlvae_adam_step (the all-in-one optimiser launch of the data-parallel path) on the D4 model"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hlvae_amd import synthetic
from hlvae_amd.HLVAE import HLVAE
from hlvae_amd.training import ELBOTrainer

dev = torch.device("cuda:0")
src = synthetic.make_d4(n_subjects=4, T=5, seed=0)
model = HLVAE([src.cov_dim_ext, [500], 32, [500], 5], src.types_info, src.n_variables, conv=False, max_batch=512,
              materialize_samples=False).to(dev)
tr = ELBOTrainer(model, P_total=10, kl="normal", max_batch=512)
for _ in range(5):
    tr.opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    tr.opt.step()
e1.record()
torch.cuda.synchronize()
print("adam_step all-in-one: %.1f us" % (e0.elapsed_time(e1) * 10))
