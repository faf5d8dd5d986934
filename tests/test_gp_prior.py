"""hl-vae_amd/elbo_functions.py + GP_model.py (batched, padded-subject formulation of the GP-prior KL) against the
fixture produced by the reference's own minibatch_KLD_upper_bound_iter (tests/golden/gp_kl.npz).
Runs on CPU (fp64 torch ops are device independent); the GPU variant is in test_gpu_parity.py."""
import os

import numpy as np
import torch

from hlvae_amd import GP_model, elbo_functions
from tests_common import rel_err

CFG = dict(cat_kernel=[2], bin_kernel=[], sqexp_kernel=[0],
           cat_int_kernel=[{"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                           {"cont_covariate": 1, "cat_covariate": 4}], bin_int_kernel=[])


def build_from_fixture(g, dev="cpu"):
    P, P_b, N, eps, idc, lr = g["scalars"]
    L = g["mu"].shape[1]
    k0, k1 = GP_model.generate_kernel_batched(L, CFG["cat_kernel"], CFG["bin_kernel"], CFG["sqexp_kernel"],
                                              CFG["cat_int_kernel"], CFG["bin_int_kernel"], [], int(idc))
    names = {}
    for name, add in (("k0", k0), ("k1", k1)):
        for ti, sk in enumerate(add.kernels):
            with torch.no_grad():
                sk._log_scale.copy_(torch.tensor(g[f"kp__{name}.{ti}.scale"]))
            names[f"{name}.{ti}.scale"] = sk._log_scale
            facs = list(sk.kernel.factors) if isinstance(sk.kernel, GP_model.ProductKernel) else [sk.kernel]
            for fi, f in enumerate(facs):
                if isinstance(f, GP_model.RbfKernel):
                    with torch.no_grad():
                        f._log_lengthscale.copy_(torch.tensor(g[f"kp__{name}.{ti}.{fi}.ls"]))
                    names[f"{name}.{ti}.{fi}.ls"] = f._log_lengthscale
    return k0.to(dev), k1.to(dev), names, (P, P_b, N, eps, int(idc), lr, L)


def run_case(g, dev="cpu"):
    k0, k1, names, (P, P_b, N, eps, idc, lr, L) = build_from_fixture(g, dev)
    t = lambda k: torch.tensor(g[k], device=dev)
    mu, lv, z = t("mu").requires_grad_(True), t("log_v").requires_grad_(True), t("z").requires_grad_(True)
    lik = GP_model.Likelihoods(L, 1.0).to(dev)
    kld, gm, gH = elbo_functions.minibatch_KLD_upper_bound_iter(k0, k1, lik, L, t("m"), t("H"), t("x"), mu, lv, z, P, P_b,
                                                                N, True, idc, float(eps))
    kld.sum().backward()
    return kld, gm, gH, mu, lv, z, names, lr


def test_gp_kl_matches_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "gp_kl.npz"))
    kld, gm, gH, mu, lv, z, names, lr = run_case(g)
    assert rel_err(kld, g["kld"]) < 1e-11
    assert rel_err(gm, g["grad_m"]) < 1e-9 and rel_err(gH, g["grad_H"]) < 1e-9
    assert rel_err(mu.grad, g["d_mu"]) < 1e-9 and rel_err(lv.grad, g["d_log_v"]) < 1e-9
    assert rel_err(z.grad, g["d_z"]) < 1e-8
    assert set(names) == {k[4:] for k in g.files if k.startswith("kp__")}
    for k, p in names.items():
        assert rel_err(p.grad, g["kg__" + k]) < 1e-8, k
    m_new, H_new = elbo_functions.natural_gradient_step(torch.tensor(g["m"]), torch.tensor(g["H"]), gm.detach(), gH.detach(), float(lr))
    assert rel_err(m_new, g["m_new"]) < 1e-9 and rel_err(H_new, g["H_new"]) < 1e-9


def test_gp_prior_object_runs_and_decreases_kl():
    """GPPrior wiring (Adam on kernels + inducing points, natural-gradient update of (m, H)) on a tiny problem."""
    torch.manual_seed(0)
    L, S, T, Q = 3, 6, 5, 6
    rows = [[float(t), float(t - 2) if s % 2 else 0.0, float(s), float(s % 2), float(s % 2), 0.0] for s in range(S) for t in range(T)]
    x = torch.tensor(rows, dtype=torch.float64)
    gp = elbo_functions.GPPrior(L, x, M=8, id_covariate=2, N_total=x.shape[0])
    mu = 0.1 * torch.randn(S * T, L)
    lv = -1.0 + 0.1 * torch.randn(S * T, L)
    vals = []
    for _ in range(25):
        g_mu, g_lv = gp.kl_and_grads(mu, lv, x, P_total=S, P_batch=S)
        assert g_mu.shape == (S * T, L) and g_mu.dtype == torch.float32 and torch.isfinite(g_lv).all()
        vals.append(float(gp.last_kld))
        gp.optimizer_step()
    assert vals[-1] < vals[0]
