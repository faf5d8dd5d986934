import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, hlvae_amd
from hlvae_amd import synthetic
from hlvae_amd.HLVAE import HLVAE
from hlvae_amd.training import ELBOTrainer
from hlvae_amd.datafeed import CompactDataset
dev=torch.device('cuda:0')
src = synthetic.make_d4(n_subjects=30, T=20, seed=11)
dims = [src.cov_dim_ext, [500], 32, [500], 5]
ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
rows = torch.tensor(np.arange(512).astype(np.int32), device=dev)
eps = torch.randn(512, 32, generator=torch.Generator().manual_seed(40)).to(dev)
torch.manual_seed(5)
model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=512, materialize_samples=False).to(dev)
tr = ELBOTrainer(model, P_total=30, kl="normal", max_batch=512, lr=0.0, metrics=True)
snaps=[]
for it in range(3):
    tr.step_rows(ds, rows, 26, eps=eps)
    torch.cuda.synchronize()
    snaps.append({k: model._ws_t[k].clone() for k in ("xn","t","mu","z","u","dy","dyT","du","dml","dt","G","log_p_x","slab")})
for k in snaps[0]:
    a,b,c=snaps[0][k],snaps[1][k],snaps[2][k]
    print(k, "run0 vs run1 differing elems:", int((a!=b).sum()), " run1 vs run2:", int((b!=c).sum()))
G=snaps[1]["G"]-snaps[2]["G"]
d=model._dims
for nm,lo,hi in (("small",0,int(d.atomic_region)),("wd",int(d.o_wd),int(d.o_wmu)),("wmu",int(d.o_wmu),int(d.o_wlv)),("w1",int(d.o_w1),int(d.o_wy)),("wy",int(d.o_wy),int(d.arena_size))):
    print(nm, float(G[lo:hi].abs().max()), int((G[lo:hi]!=0).sum()))
a,b=snaps[1]["dy"].float(),snaps[2]["dy"].float()
idx=(a!=b).nonzero()
print("dy diffs (row, col, var, k, a, b):")
for r,c in idx.tolist()[:30]:
    print(r, c, c//5, c%5, float(a[r,c]), float(b[r,c]), " tile_n", (c//5)//16, "v", (c//5)%16, "rg", (r%64)//4, "i", r%4)
