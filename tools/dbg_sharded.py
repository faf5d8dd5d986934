import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, hlvae_amd
from hlvae_amd import synthetic
from hlvae_amd.HLVAE import HLVAE
from hlvae_amd.training import ELBOTrainer
from hlvae_amd.parallel import DataParallel
from hlvae_amd.datafeed import CompactDataset
dev=torch.device('cuda:0')
src = synthetic.make_d4(n_subjects=30, T=20, seed=11)
dims = [src.cov_dim_ext, [500], 32, [500], 5]
ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
rows = [torch.tensor(np.arange(i * 40, i * 40 + 512).astype(np.int32), device=dev) for i in range(2)]
eps = [torch.randn(512, 32, generator=torch.Generator().manual_seed(40 + i)).to(dev) for i in range(3)]
res=[]
import os
modes={'ff':(None,None),'ss':(DataParallel.single(),DataParallel.single()),'fs':(None,DataParallel.single())}[os.environ.get('MODE','fs')]
for dp in modes:
    torch.manual_seed(5)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=512, materialize_samples=False).to(dev)
    tr = ELBOTrainer(model, P_total=30, kl="normal", max_batch=512, dp=dp, metrics=True)
    tr.prime_rows(ds, rows[0])
    for i in range(int(sys.argv[1])):
        tr.step_rows(ds, rows[i % 2], 26, eps=eps[i], prefetch_rows=rows[(i + 1) % 2], prepacked=True)
    model.state_dict(); torch.cuda.synchronize()
    res.append({k: v.detach().clone() for k, v in model.named_parameters()})
for k in res[0]:
    a,b=res[0][k].double(),res[1][k].double()
    if a.numel(): print(k, float((a-b).norm()/max(float(a.norm()),1e-30)), float((a-b).abs().max()))
