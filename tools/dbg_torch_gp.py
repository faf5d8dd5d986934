"""Reproducer: torch-ROCm 2.10 batched Cholesky path (GPPrior, the torch statement of the GP bound) returns NaN on the GPU after
device memory was filled with NaN and freed -- no kernel of this repository is involved.  The product path (GPPriorHIP) does
not use torch.linalg.  Run: python tools/dbg_torch_gp.py"""
import sys, torch
sys.path.insert(0, '.')
import hlvae_amd
from hlvae_amd.elbo_functions import GPPrior
dev = torch.device('cuda:0')
torch.manual_seed(0)
L, Q, M = 6, 6, 20
Ts = [6] * 8
rows = []
for s_, T in enumerate(Ts):
    for t in range(T):
        rows.append([float(t), float(t - 2) if s_ % 2 else 0.0, float(s_ + 3), float(s_ % 2), float(s_ % 2), float((s_ // 2) % 2)])
x = torch.tensor(rows, dtype=torch.float64)
x = x[torch.randperm(x.shape[0])]
B = x.shape[0]
mu = torch.randn(B, L); lv = 0.5 * torch.randn(B, L) - 1.0
def run(device, poison):
    if poison:
        junk = [torch.full((1 << 20,), float('nan'), dtype=torch.float64, device=dev) for _ in range(64)]
        small = [torch.full((n,), float('nan'), dtype=torch.float64, device=dev) for n in (64, 512, 4096, 32768, 262144) for _ in range(32)]
        del junk, small
    ref = GPPrior(L, x.to(device), M, 2, N_total=777, seed=4)
    ref.kl_and_grads(mu.to(device), lv.to(device), x.to(device), 40, len(Ts))
    return float(ref.last_kld)
print("cpu", run("cpu", False))
print("gpu fresh", run(dev, False))
print("gpu poisoned", run(dev, True))
print("gpu poisoned again", run(dev, True))
