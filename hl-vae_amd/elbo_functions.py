"""GP-prior KL term of the longitudinal ELBO on the GPU.

``minibatch_KLD_upper_bound_iter`` keeps the reference's name, argument order and return triple
(reference elbo_functions.py:196-285) but is organised for the device instead of as a Python loop
over subjects: the subjects of the batch are padded to a common T and every per-subject quantity
(K0_st, B_st = K1_st + sigma^2 I, Cholesky, inverse, the A..E partial sums, the natural-gradient
partials) is ONE batched fp64 operation over [S, L, T, T] / [S, L, T, M].  Padded rows/columns are
turned into an identity block of B_st and masked out of every sum, which leaves the value unchanged.

Round-1 status: the batched linear algebra runs through PyTorch-ROCm (rocBLAS / rocSOLVER) in fp64, with
autograd for the kernel hyper-parameters and inducing points; it is NOT yet hand-written HIP (DESIGN.md
section 7).  The gradients with respect to the VAE outputs (mu, log_var) are returned as fp32 tensors and
enter the hand-written backward kernels through ``hlvae_backward(g_mu, g_lv)``.

``GPPrior`` bundles what reference HLVAE_main.py:208-278 sets up: the two additive kernels, the inducing
points ``zt_list``, the variational parameters (m, H), Adam over kernels + inducing points, and the
natural-gradient update of (m, H) (reference training.py:130-137).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import GP_model


def subject_groups(ids: torch.Tensor):
    """rows -> ([S, Tmax] row index, [S, Tmax] validity) for the subjects present in ``ids`` (sorted by id,
    as torch.unique does in the reference, elbo_functions.py:242)."""
    ids_h = ids.detach().cpu().numpy()
    subj = np.unique(ids_h)
    rows = [np.nonzero(ids_h == s)[0] for s in subj]
    Tmax = max(len(r) for r in rows)
    idx = np.zeros((len(rows), Tmax), dtype=np.int64)
    valid = np.zeros((len(rows), Tmax), dtype=np.float64)
    for i, r in enumerate(rows):
        idx[i, :len(r)] = r
        valid[i, :len(r)] = 1.0
    return torch.as_tensor(idx, device=ids.device), torch.as_tensor(valid, device=ids.device)


def minibatch_KLD_upper_bound_iter(covar_module0, covar_module1, likelihood, latent_dim, m, H, train_xt, mu, log_v, z,
                                   P, P_in_current_batch, N, natural_gradient, id_covariate, eps, groups=None):
    """Unbiased minibatch estimate of the KL upper bound and the natural-gradient terms (grad_m, grad_H).

    covar_module0/1: ``GP_model.AdditiveKernel`` without / with the id covariate;  likelihood: object with
    ``.noise`` [L] (``GP_model.Likelihoods``);  m [L,M,1], H [L,M,M];  train_xt [B,Q];  mu, log_v [B,L];
    z [L,M,Q].  ``groups`` = subject_groups(train_xt[:, id_covariate]) if already known (avoids a host sync)."""
    f64 = torch.float64
    L, M = latent_dim, H.shape[-1]
    dev = train_xt.device
    mu, log_v = mu.to(f64), log_v.to(f64)
    eyeM = torch.eye(M, dtype=f64, device=dev)
    K0xz = covar_module0(train_xt, z)                                    # [L,B,M]
    K0zz = covar_module0(z, z) + eps * eyeM
    LK = torch.linalg.cholesky(K0zz)
    iK = torch.cholesky_solve(eyeM.expand(L, M, M), LK)
    LH = torch.linalg.cholesky(H)
    iH = torch.cholesky_solve(eyeM.expand(L, M, M), LH)
    iKm = iK @ m                                                         # [L,M,1]
    resid = (K0xz @ iKm).squeeze(2) - mu.T                               # [L,B]   (A_part)
    Q = iK @ H @ iK                                                      # (E_part)

    idx, valid = groups if groups is not None else subject_groups(train_xt[:, id_covariate])
    S, T = idx.shape
    xs = train_xt[idx].unsqueeze(1)                                      # [S,1,T,Q]
    pair = valid[:, None, :, None] * valid[:, None, None, :]             # [S,1,T,T]
    eyeT = torch.eye(T, dtype=f64, device=dev)
    noise = likelihood.noise.view(1, L, 1, 1)
    K0s = covar_module0(xs, xs) * pair                                   # [S,L,T,T]
    Bs = (covar_module1(xs, xs) + eyeT * noise) * pair + eyeT * (1.0 - valid)[:, None, :, None]
    LB = torch.linalg.cholesky(Bs)
    iB = torch.cholesky_solve(eyeT.expand(S, L, T, T), LB) * pair
    Ks = (K0xz[:, idx] * valid[None, :, :, None]).permute(1, 0, 2, 3)    # [S,L,T,M]
    iBK = iB @ Ks                                                        # [S,L,T,M]
    W = torch.einsum("sltm,sltn->lmn", Ks, iBK)                          # sum_s Ks^T iB Ks   [L,M,M]
    r = (resid[:, idx] * valid[None]).permute(1, 0, 2)                   # [S,L,T]
    v = torch.einsum("sltu,slu->slt", iB, r)
    A = torch.sum(r * v)
    ev = (torch.exp(log_v)[idx] * valid[:, :, None]).permute(0, 2, 1)    # [S,L,T]
    Bt = torch.sum(torch.diagonal(iB, dim1=-1, dim2=-2) * ev)
    C = 2.0 * torch.sum(torch.log(torch.diagonal(LB, dim1=-1, dim2=-2)))
    D = torch.sum(iB * K0s) - torch.sum(W * iK)
    E = torch.sum(Q * W)
    Fq = torch.sum(log_v)
    kl_u = 0.5 * (torch.sum(iK * H.transpose(-1, -2)) + torch.sum(m * iKm) - L * M
                  + 2.0 * torch.sum(torch.log(torch.diagonal(LK, dim1=-1, dim2=-2)))
                  - 2.0 * torch.sum(torch.log(torch.diagonal(LH, dim1=-1, dim2=-2))))
    kld_total = P / P_in_current_batch * 0.5 * (A + Bt + C + D + E - Fq) + kl_u - L * N / 2.0
    grad_m = grad_H = None
    if natural_gradient:
        mu_s = (mu[idx] * valid[:, :, None]).permute(0, 2, 1)            # [S,L,T]
        P1 = torch.einsum("sltm,slt->lm", iBK, mu_s).unsqueeze(-1)       # sum_s Ks^T iB mu_s
        Bm = iK @ W @ iK + iK
        grad_m = -(iK @ P1) + Bm @ m
        grad_H = 0.5 * (Bm - iH)
    return kld_total.reshape(1), grad_m, grad_H


def natural_gradient_step(m, H, grad_m, grad_H, lr):
    """reference training.py:130-137: update of the variational parameters in natural coordinates."""
    M = H.shape[-1]
    eye = torch.eye(M, dtype=H.dtype, device=H.device).expand_as(H)
    iH = torch.cholesky_solve(eye, torch.linalg.cholesky(H))
    iH_new = iH + lr * (grad_H + grad_H.transpose(-1, -2))
    H_new = torch.cholesky_solve(eye, torch.linalg.cholesky(iH_new))
    m_new = H_new @ (iH @ m - lr * (grad_m - 2.0 * (grad_H @ m)))
    return m_new.detach(), H_new.detach()


class GPPrior:
    """Everything the GP prior needs besides the VAE: kernels, inducing points, (m, H), their optimisers."""

    def __init__(self, latent_dim: int, train_x: torch.Tensor, M: int, id_covariate: int, N_total: int,
                 cat_kernel=(2,), bin_kernel=(), sqexp_kernel=(0,),
                 cat_int_kernel=({"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                                 {"cont_covariate": 1, "cat_covariate": 4}),
                 bin_int_kernel=(), covariate_missing_val=(), natural_gradient_lr: float = 0.01, lr: float = 1e-3,
                 eps: float = 1e-6, seed: int = 0, dp=None):
        dev = train_x.device
        self.L, self.M, self.id_covariate, self.N_total, self.eps = latent_dim, M, id_covariate, N_total, eps
        self.ng_lr, self.dp = natural_gradient_lr, dp
        self.k0, self.k1 = GP_model.generate_kernel_batched(latent_dim, list(cat_kernel), list(bin_kernel), list(sqexp_kernel),
                                                            list(cat_int_kernel), list(bin_int_kernel),
                                                            list(covariate_missing_val), id_covariate)
        self.k0.to(dev), self.k1.to(dev)
        self.likelihood = GP_model.Likelihoods(latent_dim, 1.0, constrain=True).to(dev)   # HLVAE_main.py:211-213
        g = torch.Generator().manual_seed(seed)
        Ntr = train_x.shape[0]
        zt = torch.stack([train_x[torch.randperm(Ntr, generator=g)[:M].to(dev)] for _ in range(latent_dim)])
        self.zt_list = zt.clone().to(torch.float64).requires_grad_(True)                   # HLVAE_main.py:224-229
        self.m = torch.randn(latent_dim, M, 1, generator=g, dtype=torch.float64).to(dev)   # :259
        Hh = (torch.randn(latent_dim, M, M, generator=g, dtype=torch.float64) / 10).to(dev)
        self.H = Hh @ Hh.transpose(-1, -2) + 1e-6 * torch.eye(M, dtype=torch.float64, device=dev)   # :260-263 (+jitter)
        params = list(self.k0.parameters()) + list(self.k1.parameters()) + [self.zt_list]
        self.opt = torch.optim.Adam(params, lr=lr)                                         # :277-278
        self.last_kld = None
        self._grad_m = self._grad_H = None
        self._groups = {}

    @classmethod
    def from_reference_config(cls, model, src, P_total, dev, M=120):
        """shipped configuration (config/hlvae_config_file.txt:24-45): M inducing points drawn from the training covariates"""
        train_x = torch.tensor(src.labels, dtype=torch.float64, device=dev)
        return cls(model.z_dim, train_x, min(M, train_x.shape[0]), src.id_covariate, N_total=train_x.shape[0])

    def kl_and_grads(self, mu: torch.Tensor, log_v: torch.Tensor, train_x: torch.Tensor, P_total: int, P_batch: int,
                     groups=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """KL value (kept in ``last_kld``) and its gradients w.r.t. the encoder outputs, as fp32 [B, L]."""
        if groups is None:       # batch composition is static per resident batch tensor: group once, no host sync afterwards
            key = (train_x.data_ptr(), train_x.shape[0])
            if key not in self._groups:
                self._groups[key] = subject_groups(train_x[:, self.id_covariate])
            groups = self._groups[key]
        mu_ = mu.detach().to(torch.float64).requires_grad_(True)
        lv_ = log_v.detach().to(torch.float64).requires_grad_(True)
        self.opt.zero_grad(set_to_none=True)
        kld, gm, gH = minibatch_KLD_upper_bound_iter(self.k0, self.k1, self.likelihood, self.L, self.m, self.H, train_x,
                                                     mu_, lv_, self.zt_list, P_total, P_batch, self.N_total, True,
                                                     self.id_covariate, self.eps, groups=groups)
        kld.sum().backward()
        self.last_kld = kld.detach()
        self._grad_m, self._grad_H = gm.detach(), gH.detach()
        return mu_.grad.to(torch.float32).contiguous(), lv_.grad.to(torch.float32).contiguous()

    def optimizer_step(self):
        if self.dp is not None:
            raise NotImplementedError("GP prior under data parallelism needs the per-subject partial sums (A..E, P1, W) "
                                      "all-reduced before the non-linear terms (SURVEY.md section 8(e) caveat 2): next round")
        self.opt.step()
        self.m, self.H = natural_gradient_step(self.m, self.H, self._grad_m, self._grad_H, self.ng_lr)
