"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the GP-prior KL term.

float64 restatement of
  * elbo_functions.minibatch_KLD_upper_bound_iter   (reference elbo_functions.py:196-285)
  * the natural-gradient update of (m, H)            (reference training.py:130-137)
  * the additive GP kernels in their gpytorch-free form (reference GP_model.py:27-116; the
    executed reference builds them with gpytorch, kernel_gen.py:199-310, which is absent from
    /root/reference -- an un-vendored, unpinned dependency -- so for the kernel *values*
    parity is anchored on GP_model.py, the in-tree statement of the same kernels).

Pinning: tests/golden/make_golden.py runs the reference's own
minibatch_KLD_upper_bound_iter with GP_model.py kernels behind a thin ``.evaluate()``
adapter and stores value, grad_m, grad_H and autograd gradients as fixtures.

Structure differs from the reference on purpose: subjects are padded to a common T and
processed as one batch (the layout the HIP kernel uses) instead of a Python loop.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

DT = torch.float64
MIN_LOG = -16.0                       # GP_model.py:49,77


def _pos(raw):                        # GP_model.py:57,85: exp(min + softplus(raw - min))
    return torch.exp(MIN_LOG + F.softplus(raw - MIN_LOG))


def raw_of(value: float) -> float:    # GP_model.py:50,78: raw = log(value - exp(min))
    return math.log(value - math.exp(MIN_LOG))


@dataclass
class Term:
    """scale[L] * prod(base kernels).  base = ('cat'|'bin'|'rbf', covariate index)."""
    factors: List[Tuple[str, int]]


@dataclass
class KernelSpec:
    """Additive kernel pair of generate_kernel_batched (GP_model.py:118-208):
    ``k0`` terms do not involve the id covariate, ``k1`` terms do."""
    k0: List[Term] = field(default_factory=list)
    k1: List[Term] = field(default_factory=list)


def spec_from_config(cat_kernel, bin_kernel, sqexp_kernel, cat_int_kernel, bin_int_kernel, id_covariate,
                     covariate_missing_val=()) -> KernelSpec:
    """Same term ORDER as GP_model.generate_kernel_batched builds its ModuleLists (:143-206)."""
    miss = {d["covariate"]: d["mask"] for d in covariate_missing_val}
    spec = KernelSpec()

    def with_mask(f, idx):
        return f + ([("bin", miss[idx])] if idx in miss else [])

    for idx in cat_kernel:
        (spec.k1 if idx == id_covariate else spec.k0).append(Term(with_mask([("cat", idx)], idx)))
    for idx in sqexp_kernel:
        spec.k0.append(Term(with_mask([("rbf", idx)], idx)))
    for idx in bin_kernel:
        spec.k0.append(Term(with_mask([("bin", idx)], idx)))
    for d in cat_int_kernel:
        f = with_mask([("cat", d["cat_covariate"])], d["cat_covariate"]) + \
            with_mask([("rbf", d["cont_covariate"])], d["cont_covariate"])
        (spec.k1 if d["cat_covariate"] == id_covariate else spec.k0).append(Term(f))
    for d in bin_int_kernel:
        f = with_mask([("bin", d["bin_covariate"])], d["bin_covariate"]) + \
            with_mask([("rbf", d["cont_covariate"])], d["cont_covariate"])
        spec.k0.append(Term(f))
    return spec


def init_kernel_params(spec: KernelSpec, L: int, scale=math.log(2), lengthscale=2.5) -> Dict[str, torch.Tensor]:
    """raw parameters: '<k0|k1>.<term>.scale' [L] and '<k0|k1>.<term>.<factor>.ls' [L]."""
    p = {}
    for name, terms in (("k0", spec.k0), ("k1", spec.k1)):
        for ti, t in enumerate(terms):
            p[f"{name}.{ti}.scale"] = torch.full((L,), raw_of(scale), dtype=DT)
            for fi, (kind, _) in enumerate(t.factors):
                if kind == "rbf":
                    p[f"{name}.{ti}.{fi}.ls"] = torch.full((L,), raw_of(lengthscale), dtype=DT)
    return p


def eval_kernel(terms: List[Term], prm: Dict[str, torch.Tensor], name: str, x1, x2):
    """Sum of scaled product kernels -> [L, n1, n2] (or [S, L, n1, n2] for batched x).
    x1: [..., n1, Q], x2: [..., n2, Q] where leading dims broadcast against L placed in front
    of the two matrix dims.  cat: GP_model.py:40-41, bin: :32-33, rbf: :64-69, scale: :92-97."""
    out = 0.0
    for ti, t in enumerate(terms):
        k = None
        for fi, (kind, dim) in enumerate(t.factors):
            a = x1[..., dim].unsqueeze(-1)
            b = x2[..., dim].unsqueeze(-2)
            if kind == "cat":
                f = (a - b == 0).to(DT)
            elif kind == "bin":
                f = (a + b == 2).to(DT)
            else:
                ls = _pos(prm[f"{name}.{ti}.{fi}.ls"])[:, None, None]
                f = torch.exp(-((a - b) ** 2) / (2 * ls ** 2))
            k = f if k is None else k * f
        out = out + _pos(prm[f"{name}.{ti}.scale"])[:, None, None] * k
    return out


def group_subjects(train_x, id_covariate):
    """rows -> (subject list, [S, Tmax] row index (padded with 0), [S, Tmax] validity)."""
    ids = train_x[:, id_covariate]
    subj = torch.unique(ids)                                             # elbo_functions.py:242
    rows = [torch.nonzero(ids == s).flatten() for s in subj]
    Tmax = max(len(r) for r in rows)
    idx = torch.zeros(len(rows), Tmax, dtype=torch.long)
    valid = torch.zeros(len(rows), Tmax, dtype=DT)
    for i, r in enumerate(rows):
        idx[i, :len(r)] = r
        valid[i, :len(r)] = 1.0
    return subj, idx, valid


def minibatch_kld_upper_bound_iter(spec: KernelSpec, kprm, noise, latent_dim, m, H, train_xt, mu, log_v, z,
                                   P, P_in_current_batch, N, natural_gradient, id_covariate, eps):
    """elbo_functions.py:196-285.  noise: [L] (likelihood.noise_covar.noise).  Returns
    (kld_total [1], grad_m [L,M,1] | None, grad_H [L,M,M] | None)."""
    L, M = latent_dim, H.shape[-1]
    eyeM = torch.eye(M, dtype=DT)
    K0xz = eval_kernel(spec.k0, kprm, "k0", train_xt[None], z)           # [L,B,M]   :222
    K0zz = eval_kernel(spec.k0, kprm, "k0", z, z) + eps * eyeM           # :223-224
    LK0zz = torch.linalg.cholesky(K0zz)                                  # :225
    iK0zz = torch.cholesky_solve(eyeM.expand(L, M, M), LK0zz)            # :226
    LH = torch.linalg.cholesky(H)                                        # :227
    iH = torch.cholesky_solve(eyeM.expand(L, M, M), LH)                  # :228
    A_part = (K0xz @ iK0zz @ m).squeeze(2) - mu.T                        # [L,B]     :230
    E_part = iK0zz @ H @ iK0zz                                           # :231

    _, idx, valid = group_subjects(train_xt, id_covariate)
    S, T = idx.shape
    xs = train_xt[idx]                                                   # [S,T,Q]
    vv = valid[:, None, :, None] * valid[:, None, None, :]               # [S,1,T,T]
    eyeT = torch.eye(T, dtype=DT)
    K0_st = eval_kernel(spec.k0, kprm, "k0", xs[:, None], xs[:, None]) * vv            # [S,L,T,T] :248
    B_st = eval_kernel(spec.k1, kprm, "k1", xs[:, None], xs[:, None]) + eyeT * noise[None, :, None, None]  # :249-250
    B_st = B_st * vv + eyeT * (1.0 - valid)[:, None, :, None]            # padded rows/cols -> identity block
    LB = torch.linalg.cholesky(B_st)                                     # :251
    iB = torch.cholesky_solve(eyeT.expand(S, L, T, T), LB)               # :252
    iBv = iB * vv
    Kxz = K0xz[:, idx] * valid[None, :, :, None]                         # [L,S,T,M] :253
    Kxz = Kxz.permute(1, 0, 2, 3)                                        # [S,L,T,M]
    KiBK = torch.einsum("slik,slij,sljm->lkm", Kxz, iBv, Kxz)            # sum over subjects of :254
    a = (A_part[:, idx] * valid[None]).permute(1, 0, 2)                  # [S,L,T]
    A = torch.einsum("sli,slij,slj->", a, iBv, a)                        # :256
    ev = (torch.exp(log_v)[idx] * valid[:, :, None]).permute(0, 2, 1)    # [S,L,T]
    Bt = torch.sum(torch.diagonal(iBv, dim1=-1, dim2=-2) * ev)           # :257
    C = 2 * torch.sum(torch.log(torch.diagonal(LB, dim1=-2, dim2=-1)))   # :258 (padded diag = 1)
    Dt = torch.sum(iBv * K0_st) - torch.sum(KiBK * iK0zz)                # :259
    E = torch.sum(E_part * KiBK)                                         # :260
    Fq = torch.sum(log_v)                                                # :268
    tr1 = torch.sum(iK0zz * H.transpose(-1, -2))                         # :271
    qf1 = torch.sum(m * (iK0zz @ m))                                     # :272
    logdetK = 2 * torch.sum(torch.log(torch.diagonal(LK0zz, dim1=-1, dim2=-2)))   # :273
    logdetH = 2 * torch.sum(torch.log(torch.diagonal(LH, dim1=-1, dim2=-2)))      # :274
    kld_qu_pu = 0.5 * (tr1 + qf1 - L * M + logdetK - logdetH)            # :275
    kld_total = P / P_in_current_batch * 0.5 * (A + Bt + C + Dt + E - Fq) + kld_qu_pu - L * N / 2   # :277
    grad_m = grad_H = None
    if natural_gradient:
        mu_p = (mu[idx] * valid[:, :, None]).permute(0, 2, 1)            # [S,L,T]
        ng_P1 = torch.einsum("sltm,sltu,slu->lm", Kxz, iBv, mu_p).unsqueeze(-1)   # :263-265
        Bm = iK0zz @ KiBK @ iK0zz + iK0zz                                # :281
        grad_m = -(iK0zz @ ng_P1) + Bm @ m                               # :282
        grad_H = 0.5 * (-iH + Bm)                                        # :283
    return kld_total.reshape(1), grad_m, grad_H


def natural_gradient_update(m, H, grad_m, grad_H, lr):
    """training.py:130-137."""
    M = H.shape[-1]
    eye = torch.eye(M, dtype=DT).expand_as(H)
    iH = torch.cholesky_solve(eye, torch.linalg.cholesky(H))
    iH_new = iH + lr * (grad_H + grad_H.transpose(-1, -2))
    H_new = torch.cholesky_solve(eye, torch.linalg.cholesky(iH_new)).detach()
    m_new = (H_new @ (iH @ m - lr * (grad_m - 2 * (grad_H @ m)))).detach()
    return m_new, H_new


def batch_predict_varying_T(spec: KernelSpec, kprm, noise, latent_dim, prediction_x, test_x, mu, z, id_covariate, eps):
    """GP posterior mean of the latent at new covariates (reference utils.py:99-191; the imputation / prediction entry of the
    evaluation surface).  prediction_x [Np,Q] with encoder means mu [Np,L]; test_x [Nt,Q]; z [L,M,Q].  Returns Z_pred [Nt,L].
    (torch.solve(B, A) of the reference is torch.linalg.solve(A, B).)"""
    L, M = latent_dim, z.shape[1]
    eyeM = torch.eye(M, dtype=DT)
    K0xz = eval_kernel(spec.k0, kprm, "k0", prediction_x[None], z)                        # :127
    K0zz = eval_kernel(spec.k0, kprm, "k0", z, z) + eps * eyeM                            # :128,131
    K0Xz = eval_kernel(spec.k0, kprm, "k0", test_x[None], z)                              # :129
    K0zx = K0xz.transpose(-1, -2)
    H = K0zz.clone()
    ids = prediction_x[:, id_covariate]
    subjects = torch.unique(ids).tolist()                                                 # :136
    iB_mu = torch.zeros(L, prediction_x.shape[0], 1, dtype=DT)
    iBs = []
    for s in subjects:                                                                    # :138-160
        ind = ids == s
        x_st = prediction_x[ind]
        T = x_st.shape[0]
        B_st = eval_kernel(spec.k1, kprm, "k1", x_st[None], x_st[None]) + torch.eye(T, dtype=DT) * noise.view(L, 1, 1)
        iB = torch.cholesky_solve(torch.eye(T, dtype=DT).expand(L, T, T), torch.linalg.cholesky(B_st))
        Ks = K0xz[:, ind]
        H = H + Ks.transpose(-1, -2) @ (iB @ Ks)                                          # :156-157
        iB_mu[:, ind] = iB @ mu[ind].T.unsqueeze(2)                                       # :158
        iBs.append(iB)
    t1 = K0xz @ torch.linalg.solve(H, K0zx @ iB_mu)                                       # :162
    t2 = torch.zeros_like(iB_mu)
    for i, s in enumerate(subjects):                                                      # :164-166
        ind = ids == s
        t2[:, ind] = iBs[i] @ t1[:, ind]
    mu_tilde = iB_mu - t2                                                                 # :167
    a = K0Xz @ torch.linalg.solve(K0zz, K0zx @ mu_tilde)                                  # :169
    tids = test_x[:, id_covariate]
    test_subjects = torch.unique(tids)
    msk = torch.isin(ids, test_subjects)                                                  # :171-172
    b = torch.zeros(L, test_x.shape[0], 1, dtype=DT)
    for s in test_subjects.tolist():                                                      # :175-186
        ind = tids == s
        K1Xx = eval_kernel(spec.k1, kprm, "k1", test_x[ind][None], prediction_x[msk][None])
        b[:, ind] = K1Xx @ mu_tilde[:, msk]
    return (a + b).squeeze(2).T                                                           # :188
