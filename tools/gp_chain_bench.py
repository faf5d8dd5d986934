"""k_gp_chain alone against the launches it replaces (5 x k_gp_bmm + natgrad + rsym), same inputs: times (torch events, median of
20) and the largest differences of the outputs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hlvae_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
L, M = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 120
g = torch.Generator(device="cpu").manual_seed(0)
def spd():
    a = torch.randn(L, M, M, generator=g, dtype=torch.float64)
    return (a @ a.transpose(1, 2) / M + torch.eye(M, dtype=torch.float64)).to(dev)
iK, W, H, iH = spd(), spd(), spd(), spd()
HiK = (H @ iK).contiguous()
m, P1, u = (torch.randn(L, M, generator=g, dtype=torch.float64).to(dev) for _ in range(3))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
e = lambda *s: torch.empty(*s, dtype=torch.float64, device=dev)
def outs():
    return dict(T1=e(L, M, M), Bm=e(L, M, M), grad_m=e(L, M), grad_H=e(L, M, M), tmp=e(L, M), HiKW=e(L, M, M), Rs=e(L, M, M), T1b=e(L, M, M), G=e(L, M, M))
lr, c, ga, gb = 0.01, 1.7, -1.0, 1.0
p = _lib.ptr
def fused(o):
    _lib.check(lib.hlvae_gp_chain(p(iK), p(W), p(HiK), p(H), p(iH), p(m), p(P1), p(u), C.c_double(lr), C.c_double(c), C.c_double(ga), C.c_double(gb),
                                  M, L, p(o["T1"]), p(o["Bm"]), p(o["grad_m"]), p(o["grad_H"]), p(o["tmp"]), p(o["HiKW"]), p(o["Rs"]), p(o["T1b"]), p(o["G"]), st), "chain")
def bmm(A, B, out, D=None, alpha=1.0, beta=1.0):
    _lib.check(lib.hlvae_gp_bmm(p(A), p(B), p(D), p(out), M, L, C.c_double(alpha), C.c_double(beta), st), "bmm")
def separate(o):
    bmm(iK, W, o["T1"]); bmm(o["T1"], iK, o["Bm"], D=iK)
    _lib.check(lib.hlvae_gp_natgrad(p(o["Bm"]), p(iK), p(iH), p(m), p(P1), C.c_double(lr), M, L, p(o["grad_m"]), p(o["grad_H"]), p(o["tmp"]), st), "natgrad")
    bmm(HiK, W, o["HiKW"])
    _lib.check(lib.hlvae_gp_rsym(p(u), p(m), p(W), p(o["HiKW"]), p(H), C.c_double(c), M, L, p(o["Rs"]), st), "rsym")
    bmm(iK, o["Rs"], o["T1b"]); bmm(o["T1b"], iK, o["G"], D=iK, alpha=ga, beta=gb)
def rowblocks(o):
    _lib.check(lib.hlvae_gp_chain_rb(p(iK), p(W), p(HiK), p(H), p(iH), p(m), p(P1), p(u), C.c_double(lr), C.c_double(c), C.c_double(ga), C.c_double(gb),
                                     M, L, p(o["grad_m"]), p(o["grad_H"]), p(o["tmp"]), p(o["Rs"]), p(o["G"]), st), "chain_rb")
oa, ob, oc = outs(), outs(), outs()
fused(oa); separate(ob); rowblocks(oc); torch.cuda.synchronize()
for k in ("grad_m", "grad_H", "tmp", "Rs", "G"):
    d = (oc[k] - ob[k]).abs().max().item() / max(ob[k].abs().max().item(), 1e-300)
    print(f"  row blocks {k:7s} max rel diff {d:.2e}")
for k in oa:
    d = (oa[k] - ob[k]).abs().max().item() / max(ob[k].abs().max().item(), 1e-300)
    print(f"  {k:7s} max rel diff {d:.2e}")
def timeit(f, o):
    ts = []
    for _ in range(25):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(o); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
print(f"M={M}: k_gp_chain {timeit(fused, oa):.1f} us; k_gp_chain_rb {timeit(rowblocks, oc):.1f} us; separate launches {timeit(separate, ob):.1f} us")
