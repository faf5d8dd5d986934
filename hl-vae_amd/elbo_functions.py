"""GP-prior KL term of the longitudinal ELBO on the GPU.

``minibatch_KLD_upper_bound_iter`` keeps the reference's name, argument order and return triple
(reference elbo_functions.py:196-285) but is organised for the device instead of as a Python loop
over subjects: the subjects of the batch are padded to a common T and every per-subject quantity
(K0_st, B_st = K1_st + sigma^2 I, Cholesky, inverse, the A..E partial sums, the natural-gradient
partials) is ONE batched fp64 operation over [S, L, T, T] / [S, L, T, M].  Padded rows/columns are
turned into an identity block of B_st and masked out of every sum, which leaves the value unchanged.

Two implementations of the same mathematics live here:

* ``minibatch_KLD_upper_bound_iter`` / ``GPPrior``: batched fp64 PyTorch ops with autograd (device independent; the
  readable statement, pinned to the reference fixture on CPU and used as the cross-check of the HIP path).
* ``GPPriorHIP``: the structured pieces as hand-written fp64 HIP kernels behind the C ABI (csrc/gp.hip: kernel
  matrices, LDS-resident Cholesky/inverse, the fused per-(subject, latent) block kernel, the chain rule into
  hyper-parameters and inducing points) with ANALYTIC gradients -- no autograd graph -- and plain library GEMMs
  (torch.matmul -> rocBLAS) for the [L,M,M] / [L,B,M] products in between.

The gradients with respect to the VAE outputs (mu, log_var) are returned as fp32 [B, L] tensors and enter the
hand-written backward kernels through ``hlvae_backward(g_mu, g_lv)``.

``GPPrior`` bundles what reference HLVAE_main.py:208-278 sets up: the two additive kernels, the inducing
points ``zt_list``, the variational parameters (m, H), Adam over kernels + inducing points, and the
natural-gradient update of (m, H) (reference training.py:130-137).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import contextlib
import torch

from . import GP_model


class _GroupCache:
    """Subject grouping of resident batch tensors, computed once per tensor (torch.unique is a host sync and cannot be
    captured in a HIP graph).  An entry keeps a reference to its tensor, so the allocator cannot hand the same address to a
    different batch while the entry lives; least-recently-used entries are dropped beyond ``capacity``."""

    def __init__(self, capacity: int = 32):
        from collections import OrderedDict
        self.capacity, self.d = capacity, OrderedDict()

    def get(self, train_x: torch.Tensor, build):
        key = (train_x.data_ptr(), tuple(train_x.shape), train_x._version)
        hit = self.d.get(key)
        if hit is not None:
            self.d.move_to_end(key)
            return hit[1]
        val = build(train_x)
        self.d[key] = (train_x, val)
        while len(self.d) > self.capacity:
            self.d.popitem(last=False)
        return val


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
        self._groups = _GroupCache()

    @classmethod
    def from_reference_config(cls, model, src, P_total, dev, M=120):
        """shipped configuration (config/hlvae_config_file.txt:24-45): M inducing points drawn from the training covariates"""
        train_x = torch.tensor(src.labels, dtype=torch.float64, device=dev)
        return cls(model.z_dim, train_x, min(M, train_x.shape[0]), src.id_covariate, N_total=train_x.shape[0])

    def kl_and_grads(self, mu: torch.Tensor, log_v: torch.Tensor, train_x: torch.Tensor, P_total: int, P_batch: int,
                     groups=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """KL value (kept in ``last_kld``) and its gradients w.r.t. the encoder outputs, as fp32 [B, L]."""
        if groups is None:       # batch composition is static per resident batch tensor: group once, no host sync afterwards
            groups = self._groups.get(train_x, lambda t: subject_groups(t[:, self.id_covariate]))
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
            raise NotImplementedError("the torch statement of the GP prior is single-process; data parallel: GPPriorHIP(dp=...)")
        self.opt.step()
        self.m, self.H = natural_gradient_step(self.m, self.H, self._grad_m, self._grad_H, self.ng_lr)


# ======================================================================================================================
#  HIP path
# ======================================================================================================================
import ctypes as _C

from . import _lib


def _spec_from_config(cat_kernel, bin_kernel, sqexp_kernel, cat_int_kernel, bin_int_kernel, covariate_missing_val, id_covariate):
    """(terms without id covariate, terms with it) as lists of factor lists [(kind, dim), ...], in the term order of
    GP_model.generate_kernel_batched."""
    miss = {d["covariate"]: d["mask"] for d in covariate_missing_val}

    def wm(f, idx):
        return f + ([(_lib.GP_BIN, miss[idx])] if idx in miss else [])

    k0, k1 = [], []
    for idx in cat_kernel:
        (k1 if idx == id_covariate else k0).append(wm([(_lib.GP_CAT, idx)], idx))
    for idx in sqexp_kernel:
        k0.append(wm([(_lib.GP_RBF, idx)], idx))
    for idx in bin_kernel:
        k0.append(wm([(_lib.GP_BIN, idx)], idx))
    for d in cat_int_kernel:
        f = wm([(_lib.GP_CAT, d["cat_covariate"])], d["cat_covariate"]) + wm([(_lib.GP_RBF, d["cont_covariate"])], d["cont_covariate"])
        (k1 if d["cat_covariate"] == id_covariate else k0).append(f)
    for d in bin_int_kernel:
        k0.append(wm([(_lib.GP_BIN, d["bin_covariate"])], d["bin_covariate"]) + wm([(_lib.GP_RBF, d["cont_covariate"])], d["cont_covariate"]))
    return k0, k1


def _pack_spec(terms, slot0):
    """list of factor lists -> (HlvaeGpKernel, names of the hyper-parameter rows it introduced, next free row)"""
    k = _lib.HlvaeGpKernel()
    assert len(terms) <= _lib.GP_MAX_TERMS
    k.n_terms = len(terms)
    names, slot = [], slot0
    for t, fac in enumerate(terms):
        assert len(fac) <= _lib.GP_MAX_FACTORS
        k.scale_slot[t] = slot
        names.append((t, None))
        slot += 1
        k.n_factors[t] = len(fac)
        for f, (kind, dim) in enumerate(fac):
            k.kind[t][f], k.dim[t][f], k.ls_slot[t][f] = kind, dim, -1
            if kind == _lib.GP_RBF:
                k.ls_slot[t][f] = slot
                names.append((t, f))
                slot += 1
    return k, names, slot


class GPPriorHIP:
    """GP prior with hand-written HIP kernels and analytic gradients (see module docstring).  Same state and update
    rules as ``GPPrior``; hyper-parameters are one fp64 tensor ``prm`` [rows, L] of RAW values (row order: terms of the
    id-free kernel then of the id kernel, each: scale, then its RBF lengthscales).  ``prm`` and ``zt_list`` are views of
    one flat arena that a single fused Adam kernel updates; every buffer of a step is preallocated, so the step can be
    captured in a HIP graph.

    Data parallel (``dp`` = hlvae_amd.parallel.DataParallel; whole subjects per rank, SURVEY.md section 8(e) caveat 2):
    hyper-parameters, inducing points, m, H are replicated; each rank evaluates its subjects; the bound is linear in the
    per-subject sums W = sum Ks^T iB Ks, P1 = sum Ks^T iB mu_s, u = sum Ks^T iB a_s, so ONE all-reduce of the packed
    [W | P1 | u | bound] buffer (L M^2 fp64, 3.7 MB at L=32, M=120) before the non-linear terms, and one of the small
    hyper-parameter gradient arena, reproduce the single-process step; the M x M inversions are replicated work."""

    def __init__(self, latent_dim, train_x, M, id_covariate, N_total, cat_kernel=(2,), bin_kernel=(), sqexp_kernel=(0,),
                 cat_int_kernel=({"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                                 {"cont_covariate": 1, "cat_covariate": 4}),
                 bin_int_kernel=(), covariate_missing_val=(), natural_gradient_lr=0.01, lr=1e-3, eps=1e-6, seed=0, dp=None):
        import math
        dev = train_x.device
        self.dp = dp
        if dev.type != "cuda":
            raise RuntimeError("GPPriorHIP runs on the GPU only (no CPU fallback); GPPrior is the device-independent statement")
        self.L, self.M, self.id_covariate, self.N_total, self.eps, self.ng_lr = latent_dim, M, id_covariate, N_total, eps, natural_gradient_lr
        self.lr = lr
        self.Q = Q = train_x.shape[1]
        L = latent_dim
        t0, t1 = _spec_from_config(list(cat_kernel), list(bin_kernel), list(sqexp_kernel), list(cat_int_kernel),
                                   list(bin_int_kernel), list(covariate_missing_val), id_covariate)
        self.k0, names0, n = _pack_spec(t0, 0)
        self.k1, names1, n = _pack_spec(t1, n)
        self.n_slots = n
        self.slot_names = [("k0",) + x for x in names0] + [("k1",) + x for x in names1]
        raw = lambda v: math.log(v - math.exp(-16.0))
        init = [raw(math.log(2)) if f is None else raw(2.5) for (_, _, f) in self.slot_names]        # GP_model.py:44,72
        f64 = dict(dtype=torch.float64, device=dev)
        n_theta = n * L + L * M * Q
        self._theta = torch.zeros(n_theta, **f64)                  # [hyper-parameters | inducing points]
        self._gtheta = torch.zeros(n_theta, **f64)                 # gradients (zeroed by the Adam kernel after use)
        self._adam_m, self._adam_v = torch.zeros(n_theta, **f64), torch.zeros(n_theta, **f64)
        self._adam_step = torch.zeros(2, dtype=torch.int64, device=dev)    # {steps done, ticket}
        self.prm = self._theta[:n * L].view(n, L)
        self.zt_list = self._theta[n * L:].view(L, M, Q)
        self.prm.copy_(torch.tensor(init, **f64)[:, None].expand(n, L))
        g = torch.Generator().manual_seed(seed)
        Ntr = train_x.shape[0]
        self.zt_list.copy_(torch.stack([train_x[torch.randperm(Ntr, generator=g)[:M].to(dev)] for _ in range(L)]))
        self.prm.grad = self._gtheta[:n * L].view(n, L)
        self.zt_list.grad = self._gtheta[n * L:].view(L, M, Q)
        self._hyp = torch.zeros(3, n, L, **f64)
        # one buffer [K0zz + jitter | H | iK]: [K0zz | H] is inverted in one batched launch when no factorisation is cached,
        # and [H | iK] is where the end-of-step inversion writes its two results directly (no copies)
        self._big = torch.zeros(3 * L, M, M, **f64)
        self._KH = self._big[:2 * L]
        self.H = self._big[L:2 * L]
        self.m = torch.randn(L, M, 1, generator=g, dtype=torch.float64).to(dev)
        Hh = (torch.randn(L, M, M, generator=g, dtype=torch.float64) / 10).to(dev)
        self.H.copy_(Hh @ Hh.transpose(-1, -2) + 1e-6 * torch.eye(M, **f64))
        self.noise = torch.ones(L, **f64)                                                           # HLVAE_main.py:211-213
        self.fail = torch.zeros(1, dtype=torch.int32, device=dev)
        # factorisation cache: the end of a step inverts [iH_new | K0zz of the UPDATED hyper-parameters] in one launch, so
        # the next step starts with iK, iH and both log-determinants in hand (one batched inversion per step, not two)
        self._KH2 = torch.zeros(2 * L, M, M, **f64)               # input of that inversion: [iH (updated in place) | K0zz]
        self._HiK = self._big[L:]                                 # its output: [H_new | iK]
        self._ld2 = torch.zeros(2 * L, **f64)                     # [log det iH_new | log det K0zz]
        self._iK, self._iHb = self._big[2 * L:], self._KH2[:L]
        self._ldK, self._ldH = self._ld2[L:], torch.zeros(L, **f64)
        self._fact_key = None
        self._xchg = torch.zeros(L * M * M + 2 * L * M + 1, **f64)    # [W | P1 | u | bound]: the one DP exchange buffer
        import os as _os
        self._fuse_sums = _os.environ.get("HL_GP_FUSE", "1") != "0"  # residual + P1 + u inside k_gp_subject_fwd (HL_GP_FUSE=0: separate launches, A/B)
        self.last_kld = self._xchg[-1:]
        self._groups = _GroupCache()
        self._grad_m = self._grad_H = self._iH = self._tmp = None
        self._bufs, self._mm, self._side, self._pending = {}, None, None, False
        self._prep, self._prep_stream, self._tail_pending = None, None, False
        self._ahead_stream, self._ahead_bufs, self._ahead = None, {}, None
        self._serial = _os.environ.get("HL_GP_SERIAL", "0") == "1"
        self._balance = int(_os.environ.get("HL_GP_BALANCE", "2"))        # where the chain rule through K0xz runs (kl_and_grads)
        # the two chains behind the per-subject kernel fork from ITS event (0.591 ms at configs[4]) or from the caller's stream behind
        # that event (0.607: their first launches then start behind a cross-queue signal of the caller's queue; without the early
        # fork 0.610).  Inside a capture with the deferred state update only the second form survives hipStreamEndCapture.
        self._fork_direct = _os.environ.get("HL_GP_FORK_DIRECT", "1") != "0"
        self._defer_capture = False
        self._a_first = _os.environ.get("HL_GP_A_FIRST", "0") != "0"       # chain A queued before chain C (kl_and_grads)
        self._early = _os.environ.get("HL_GP_EARLY", "1") != "0"          # per-subject kernel forked behind the ENCODER (kl_and_grads(after=...))
        self._chain = int(_os.environ.get("HL_GP_CHAIN", "2"))            # the M x M algebra behind W: 0 separate launches, 1 k_gp_chain, 2 k_gp_chain_rb
        self._split_kzz = _os.environ.get("HL_GP_SPLIT", "1") != "0"   # K0zz gradient behind chain C (kl_and_grads)
        if dp is not None:                     # inducing points are drawn from rank-local covariates: replicate rank 0's state
            for t in (self._theta, self.m, self._KH):
                dp.broadcast_(t)
        self._transform()

    @classmethod
    def from_reference_config(cls, model, src, P_total, dev, M=120, dp=None):
        train_x = torch.tensor(src.labels, dtype=torch.float64, device=dev)
        return cls(model.z_dim, train_x, min(M, train_x.shape[0]), src.id_covariate, N_total=train_x.shape[0], dp=dp)

    # ---- thin wrappers over the C ABI ---------------------------------------------------------------------------
    def _stream(self):
        return _C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _transform(self):
        """positive hyper-parameters, d pos / d raw / pos and 1 / pos^2, once per step (GP_model.py:57,85)"""
        _lib.check(_lib.load().hlvae_gp_transform(_lib.ptr(self.prm), self.n_slots, self.L, _lib.ptr(self._hyp), self._stream()),
                   "gp_transform")
        return self._hyp

    def kernel_matrix(self, k, x1, x2, jitter=0.0, out=None):
        L, Q = self.L, self.Q
        pl1, pl2 = int(x1.dim() == 3), int(x2.dim() == 3)
        n1, n2 = x1.shape[-2], x2.shape[-2]
        if out is None:
            out = torch.empty(L, n1, n2, dtype=torch.float64, device=x1.device)
        _lib.check(_lib.load().hlvae_gp_kernel_matrix(_C.byref(k), _lib.ptr(self._hyp), self.n_slots, L, Q, _lib.ptr(x1), n1, pl1,
                                                      _lib.ptr(x2), n2, pl2, _C.c_double(jitter), _lib.ptr(out), self._stream()),
                   "gp_kernel_matrix")
        return out

    def chol_inv(self, A, out=None):
        n, N = A.shape[0], A.shape[-1]
        inv, logdet = out if out is not None else (torch.empty_like(A), torch.empty(n, dtype=torch.float64, device=A.device))
        _lib.check(_lib.load().hlvae_gp_chol_inv(_lib.ptr(A), n, N, _lib.ptr(inv), _lib.ptr(logdet), _lib.ptr(self.fail),
                                                 self._stream()), "gp_chol_inv")
        return inv, logdet

    def _group(self, train_x):
        def build(t):
            idx, valid = subject_groups(t[:, self.id_covariate])
            return torch.where(valid > 0, idx, torch.full_like(idx, -1)).to(torch.int32).contiguous()

        return self._groups.get(train_x, build)

    # ---- the KL bound, its gradients, the natural-gradient terms ------------------------------------------------
    def bmm(self, A, B, D=None, alpha=1.0, beta=1.0):
        """alpha A @ B + beta D for dense [L, M, M] fp64 operands, on the fp64 matrix cores (csrc/gp.hip: k_gp_bmm)"""
        L, M = self.L, self.M
        for t in (A, B) + (() if D is None else (D,)):
            if t.shape != (L, M, M) or not t.is_contiguous() or t.dtype != torch.float64:
                raise ValueError("GPPriorHIP.bmm: dense contiguous [L, M, M] fp64 operands")
        out = torch.empty(L, M, M, dtype=torch.float64, device=A.device)
        _lib.check(_lib.load().hlvae_gp_bmm(_lib.ptr(A), _lib.ptr(B), _lib.ptr(D), _lib.ptr(out), M, L, _C.c_double(alpha),
                                            _C.c_double(beta), self._stream()), "gp_bmm")
        return out

    # ---- device buffers of a step, allocated once per batch geometry (graph capture keeps reading the same addresses) --------
    def _step_buffers(self, B, S, T, dev):
        key = (B, S, T)
        buf = self._bufs.get(key)
        if buf is None:
            L, M = self.L, self.M
            f64 = dict(dtype=torch.float64, device=dev)
            e = lambda *sh: torch.empty(*sh, **f64)
            buf = dict(Kxz=e(L, B, M), V=e(L, B, M), Y=e(L, B, M), G_Kxz=e(L, B, M), v=e(L, B), resid=e(L, B),
                       iB=e(S, L, T, T), K0s=e(S, L, T, T), part=e(S, L, 4),
                       g_mu=torch.empty(B, L, dtype=torch.float32, device=dev), g_lv=torch.empty(B, L, dtype=torch.float32, device=dev))
            self._bufs[key] = buf
            # (never evicted: a captured HIP graph has these addresses baked in -- dropping the oldest geometry once a ninth
            #  appeared let a later replay of an earlier graph read and write freed memory.  One entry costs ~5 L B M fp64:
            #  8 MB at the shipped configuration; data with many distinct (rows, subjects, T) geometries pays that per geometry.)
        if self._mm is None:
            L, M = self.L, self.M
            f64 = dict(dtype=torch.float64, device=dev)
            self._mm = {k: torch.empty(L, M, M, **f64) for k in ("HiK", "N1", "T1", "T1b", "Bm", "HiKW", "Rs", "G_Kzz", "grad_H")}
            self._mm.update({k: torch.empty(L, M, 1, **f64) for k in ("iKm", "grad_m", "tmp")})
        return buf

    def _gemm(self, A, transA, Bm_, C, Mo, N, K, alpha=1.0, beta=0.0, D=None):
        """C[l] = alpha op(A[l]) B[l] + beta D[l] for dense contiguous [L, ., .] fp64 operands (csrc/gp.hip: k_gp_gemm)"""
        lda, ldb = A.shape[2], Bm_.shape[2]
        _lib.check(_lib.load().hlvae_gp_gemm(_lib.ptr(A), lda, A.stride(0), int(transA), _lib.ptr(Bm_), ldb, Bm_.stride(0),
                                             _lib.ptr(D), N if D is not None else 0, D.stride(0) if D is not None else 0,
                                             _lib.ptr(C), N, C.stride(0), Mo, N, K, self.L, _C.c_double(alpha), _C.c_double(beta),
                                             self._stream()), "gp_gemm")
        return C

    def _bmm_into(self, A, B, out, D=None, alpha=1.0, beta=1.0):
        L, M = self.L, self.M
        _lib.check(_lib.load().hlvae_gp_bmm(_lib.ptr(A), _lib.ptr(B), _lib.ptr(D), _lib.ptr(out), M, L, _C.c_double(alpha),
                                            _C.c_double(beta), self._stream()), "gp_bmm")
        return out

    def _bmv(self, A, x, out, y=None, alpha=1.0, beta=0.0):
        _lib.check(_lib.load().hlvae_gp_bmv(_lib.ptr(A), _lib.ptr(x), _lib.ptr(y), _lib.ptr(out), self.M, self.L, _C.c_double(alpha),
                                            _C.c_double(beta), self._stream()), "gp_bmv")
        return out

    def _spd_inv(self, A, inv, logdet, n_neg=0, logdet_neg=None):
        _lib.check(_lib.load().hlvae_gp_spd_inv2(_lib.ptr(A), A.shape[0], A.shape[-1], _lib.ptr(inv), _lib.ptr(logdet), n_neg,
                                                 _lib.ptr(logdet_neg), _lib.ptr(self.fail), self._stream()), "gp_spd_inv2")

    def check(self):
        """raises if an SPD inversion met a non-positive pivot since the last check (the reference's torch.cholesky raises at
        once, elbo_functions.py:225-228; here the flag lives on the device and is read where the host reads scalars anyway)"""
        self.join()
        if int(self.fail.item()) != 0:
            self.fail.zero_()
            raise RuntimeError("GPPriorHIP: a covariance (K0zz, H, iH_new or a subject block) is not positive definite")

    def prepare(self, labels, rows=None, groups=None, ahead=False):
        """Everything of the bound that depends on the prior's own state and on the batch's COVARIATES only -- transformed
        hyper-parameters, K0xz, (the factorisations when none were left behind), iK m, H iK, iK - iK H iK, the cleared
        accumulators -- queued on a stream of the prior's, forked from the caller's here.  Called at the top of a training step
        (ELBOTrainer) it runs UNDER the VAE's forward pass (encoder, fused middle, head kernel: ~75 us at 1024 rows) instead of
        between it and the per-subject kernel (~80 us of the step's critical path, round-3 timeline); ``kl_and_grads`` joins it.
        labels [N, Q] covariates; rows (int32 device tensor): the batch's rows of ``labels`` (the gather runs on the side stream
        too); groups: the subject structure, as for kl_and_grads.
        ahead: the caller asserts that ``compute_ahead`` ran for exactly this batch behind the last optimiser step (the same
        contract as a pre-packed input batch): its gathered covariates and K0xz are taken as they are."""
        dev = labels.device
        main = torch.cuda.current_stream(dev)
        if self._prep_stream is None:
            self._prep_stream = torch.cuda.current_stream(dev) if self._serial else torch.cuda.Stream(device=dev)
        sP = self._prep_stream
        sP.wait_stream(main)
        B = labels.shape[0] if rows is None else rows.shape[0]
        # (groups from the sampler only: the host-side grouping cache is keyed by the covariate tensor, and the ahead buffer is reused)
        use_ahead = (ahead and rows is not None and groups is not None and self._ahead == B and B in self._ahead_bufs
                     and self._fact_key == (self._theta._version, self._KH._version))
        if use_ahead:
            sP.wait_stream(self._ahead_stream)
        with torch.cuda.stream(sP):
            if use_ahead:
                x, Kxz = self._ahead_bufs[B]
            else:
                x = labels if rows is None else labels.index_select(0, rows.long())
                x, Kxz = x.contiguous(), None
            idx = groups if groups is not None else self._group(x)
            st_ = self._prepare_state(x, idx.shape[0], idx.shape[1], x.shape[0], dev, Kxz=Kxz)
        self._prep = (x, idx) + st_
        return x

    def compute_ahead(self, labels, rows, fork_from=None):
        """Covariate gather + K0xz of a FOLLOWING batch on a stream of their own, forked from the caller's here.  Called by
        ``optimizer_step(next_batch=...)`` right behind the hyper-parameter transform: K0xz of the next batch needs the updated
        hyper-parameters and inducing points but not the batched inversion, so its 31 MB kernel matrix (55 us inside the step at
        configs[4]) runs BESIDE the 82 us inversion instead of behind it (round-3 timeline: the state update -> K0xz ->
        iK-products chain is the step's loop-carried critical path).  ``prepare(..., ahead=True)`` picks the buffers up."""
        dev = labels.device
        main = fork_from if fork_from is not None else torch.cuda.current_stream(dev)
        if self._ahead_stream is None:
            self._ahead_stream = torch.cuda.current_stream(dev) if self._serial else torch.cuda.Stream(device=dev)
        sK = self._ahead_stream
        sK.wait_stream(main)
        B = rows.shape[0]
        ab = self._ahead_bufs.get(B)
        if ab is None:          # (never evicted: captured graphs hold the addresses, as for _step_buffers)
            ab = (torch.empty(B, labels.shape[1], dtype=torch.float64, device=dev),
                  torch.empty(self.L, B, self.M, dtype=torch.float64, device=dev))
            self._ahead_bufs[B] = ab
        with torch.cuda.stream(sK):
            torch.index_select(labels, 0, rows.long(), out=ab[0])
            self.kernel_matrix(self.k0, ab[0], self.zt_list, out=ab[1])
        self._ahead = B

    def join_ahead(self):
        """the caller's stream waits for a compute_ahead nobody has consumed yet (end of a captured chain: every forked stream
        must be back on the capturing one)"""
        if self._ahead_stream is not None and self._ahead is not None:
            torch.cuda.current_stream(self.zt_list.device).wait_stream(self._ahead_stream)

    def prime_ahead(self, labels, rows):
        """``compute_ahead`` outside a training step (before the first step of a pipelined sequence / the first replay of a
        captured chain): transforms the hyper-parameters first"""
        self.join()
        self._transform()
        self.compute_ahead(labels, rows)

    def _prepare_state(self, x, S, T, B, dev, Kxz=None):
        """the state-only launches (current stream); returns (buf, hyp, Kxz, iKm, HiK, N1).  Kxz given: computed ahead, behind the
        transform of the last optimiser step (the planes in self._hyp are current)"""
        L, M = self.L, self.M
        k0, z = self.k0, self.zt_list
        buf = self._step_buffers(B, S, T, dev)
        mm = self._mm
        if Kxz is None:
            hyp = self._transform()
            Kxz = self.kernel_matrix(k0, x, z, out=buf["Kxz"])
        else:
            hyp = self._hyp
        if self._fact_key != (self._theta._version, self._KH._version):
            # no factorisation left behind by the previous optimiser step (first step, or parameters touched since): K0zz
            # (written next to H) and H inverted straight into their homes
            self.kernel_matrix(k0, z, z, jitter=self.eps, out=self._KH[:L])
            self._spd_inv(self._KH[:L], self._iK, self._ldK)
            self._spd_inv(self.H, self._iHb, self._ldH)
        iK = self._iK
        iKm = self._bmv(iK, self.m, mm["iKm"])                               # [L,M,1]
        HiK = self._bmm_into(self.H, iK, mm["HiK"])
        N1 = self._bmm_into(iK, HiK, mm["N1"], D=iK, alpha=-1.0, beta=1.0)   # iK - iK H iK
        if self._fuse_sums:
            LMM, LM = L * M * M, L * M
            self._xchg[:LMM + 2 * LM].zero_()                                # W (hlvae_gp_gemm_acc), P1, u (the per-subject kernel) accumulate
        return buf, hyp, Kxz, iKm, HiK, N1

    def kl_and_grads(self, mu, log_v, train_x, P_total, P_batch, groups=None, join=True, after=None):
        """mu, log_v: fp32 [B, L] (the workspace tensors of the VAE); returns fp32 [B, L] gradients.  Hyper-parameter
        and inducing-point gradients are left in ``prm.grad`` / ``zt_list.grad``.  Every product, reduction and element-wise
        step runs in the kernels of csrc/gp.hip; no host synchronisation when ``groups`` comes from the sampler.
        train_x None: the batch ``prepare`` was called for.
        after: an event the caller recorded on its stream when mu / log_v were final (ELBOTrainer: behind the encoder, BEFORE it
        queued the decoder): the per-subject kernel is then forked from that point -- it runs on the prior's preparation stream
        beside the VAE's head kernel instead of behind it (round-3 timeline: 50 us of the step's critical path at 1024 rows) --
        and the caller's stream waits for g_mu / g_lv here."""
        lib, st = _lib.load(), self._stream()
        L, M, Q, B = self.L, self.M, self.Q, mu.shape[0]
        dev = mu.device
        c = float(P_total) / float(P_batch)
        prep, self._prep = self._prep, None
        early = False
        if prep is not None and (train_x is None or train_x is prep[0]):
            x, idx, buf, hyp, Kxz, iKm, HiK, N1 = prep
            early = after is not None and self._early and self._fuse_sums and not self._serial
            if not early:
                torch.cuda.current_stream(dev).wait_stream(self._prep_stream)
            x.record_stream(torch.cuda.current_stream(dev))                  # (allocated on the side stream, read on this one)
            if x.shape[0] != B:
                raise ValueError(f"kl_and_grads: prepare() saw {x.shape[0]} rows, the encoder outputs have {B}")
        else:
            if train_x is None:
                raise ValueError("kl_and_grads(train_x=None) needs a preceding prepare()")
            self.join_tail()
            if prep is not None:                                             # prepared for another batch: drop it (ordered, unused)
                torch.cuda.current_stream(dev).wait_stream(self._prep_stream)
            x = train_x.contiguous()
            # subject structure of the batch: [S, T] batch rows of each subject, -1 = padding.  The sampler knows it when it builds
            # the batch (datafeed.subject_index, no device work); from a bare covariate tensor it costs a host round trip, cached
            idx = groups if groups is not None else self._group(x)
            buf, hyp, Kxz, iKm, HiK, N1 = self._prepare_state(x, idx.shape[0], idx.shape[1], B, dev)
        if idx.dtype != torch.int32 or not idx.is_contiguous() or idx.dim() != 2:
            raise ValueError("groups: contiguous int32 [S, T] tensor of batch-row indices, -1 = padding")
        S, T = idx.shape
        if mu.dtype != torch.float32 or log_v.dtype != torch.float32 or not mu.is_contiguous() or not log_v.is_contiguous():
            mu, log_v = mu.to(torch.float32).contiguous(), log_v.to(torch.float32).contiguous()
        k0, k1, z = self.k0, self.k1, self.zt_list
        mm = self._mm
        iK, iH, ldK, ldH = self._iK, self._iHb, self._ldK, self._ldH
        self._iH = iH
        LMM, LM = L * M * M, L * M
        W = self._xchg[:LMM].view(L, M, M)
        P1 = self._xchg[LMM:LMM + LM].view(L, M, 1)
        u = self._xchg[LMM + LM:LMM + 2 * LM].view(L, M, 1)
        fused = self._fuse_sums
        if fused:
            # the residual a = K0xz iK m - mu^T and the two matrix^T-vector sums P1 = V^T mu, u = K0xz^T v (elbo_functions.py:
            # 262-266) are computed inside the per-subject kernel from its LDS tile of K0xz (round 3): three launches and three
            # 31 MB passes less (gp_resid 9 us on the critical path, the two gemv_t 76 + 32 us on the natural-gradient chain)
            resid = None
        else:
            resid = buf["resid"]                                             # K0xz iK m - mu^T   [L,B]
            _lib.check(lib.hlvae_gp_resid(_lib.ptr(Kxz), _lib.ptr(iKm), _lib.ptr(mu), L, B, M, _lib.ptr(resid), st), "gp_resid")
        iB, K0s, V, v, part, g_mu, g_lv = (buf[k] for k in ("iB", "K0s", "V", "v", "part", "g_mu", "g_lv"))
        main = torch.cuda.current_stream(dev)
        if early:
            self._prep_stream.wait_event(after)
            fwd_ctx = torch.cuda.stream(self._prep_stream)
        else:
            fwd_ctx = contextlib.nullcontext()
        with fwd_ctx:
          st = self._stream()
          _lib.check(lib.hlvae_gp_subject_fwd(_C.byref(k0), _C.byref(k1), _lib.ptr(hyp), self.n_slots, L, Q, _lib.ptr(x),
                                            _lib.ptr(self.noise), _lib.ptr(idx), S, T, _lib.ptr(Kxz), B, M, _lib.ptr(resid),
                                            _lib.ptr(log_v), _C.c_double(c), _lib.ptr(iB), _lib.ptr(K0s), _lib.ptr(V), _lib.ptr(v),
                                            _lib.ptr(part), _lib.ptr(g_mu), _lib.ptr(g_lv), _lib.ptr(iKm) if fused else None,
                                            _lib.ptr(mu) if fused else None, _lib.ptr(u) if fused else None,
                                            _lib.ptr(P1) if fused else None, st), "gp_subject_fwd")
        st = None
        # g_mu / g_lv -- all the VAE's backward pass needs -- are final here.  What follows (the bound's value, the natural-gradient
        # terms, the chain rule into hyper-parameters and inducing points) is two independent chains of latency-bound kernels:
        # they run side by side on two streams of ours, and with join = False also beside whatever the caller queues next on
        # its own stream (ELBOTrainer: the VAE's backward pass + optimiser); optimizer_step() / join() wait for them.
        sA, sC = self._streams(dev)
        if early:
            # the caller's stream picks g_mu / g_lv up, and the two chains fork from IT (not from the event itself: a side stream
            # that waits on another side stream's event and is later joined back into that stream -- optimizer_step(defer=True) --
            # crashes ROCm 7.2's hipStreamEndCapture, tools/repro/capture_forkjoin.py; the caller's head kernel, which the chains
            # now also wait for, is over long before the per-subject kernel)
            evF = torch.cuda.Event()
            evF.record(self._prep_stream)
            main.wait_event(evF)
        if early and self._fork_direct and not self._defer_capture:
            sA.wait_event(evF)
            sC.wait_event(evF)
        else:
            sA.wait_stream(main)
            sC.wait_stream(main)
        world = 1 if self.dp is None else self.dp.world
        gprm, gz = self.prm.grad, self.zt_list.grad                          # zero here: the Adam kernel cleans them
        balance = self._balance if (self._chain and M % 4 == 0) else 0
        evW = evY = None
        Rs, G_Kzz_s = mm["Rs"], mm["G_Kzz"]

        def chain_c():
            nonlocal evY
            with torch.cuda.stream(sC):      # chain C: gradient w.r.t. K0xz and the subject blocks
                st = self._stream()
                Y = self._gemm(V, False, N1, buf["Y"], B, M, M)                  # V (iK - Q)   [L,B,M]  (local rows)
                if balance == 1:
                    # round 3: the chain rule through K0xz forms G_Kxz = c [ v (iK m)^T - Y ] on the fly (no k_gp_gkxz launch, no 31 MB
                    # matrix written and read back) and may run on chain A behind the M x M algebra (HL_GP_BALANCE=1) or here behind
                    # the per-subject kernel (=2)
                    evY = torch.cuda.Event()
                    evY.record(sC)
                elif balance == 2:
                    pass
                else:
                    G_Kxz = buf["G_Kxz"]                                         # c [ v (iK m)^T + V (Q - iK) ]
                    _lib.check(lib.hlvae_gp_gkxz(_lib.ptr(Y), _lib.ptr(v), _lib.ptr(iKm), _C.c_double(c), L, B, M, _lib.ptr(G_Kxz), st), "gp_gkxz")
                    _lib.check(lib.hlvae_gp_param_grad(_C.byref(k0), _lib.ptr(hyp), self.n_slots, L, Q, _lib.ptr(x), B, 0, _lib.ptr(z), M, 0,
                                                       _lib.ptr(G_Kxz), _lib.ptr(gprm), _lib.ptr(gz), None, None, _C.c_double(0.0), st),
                               "gp_param_grad(Kxz)")
                _lib.check(lib.hlvae_gp_subject_bwd(_C.byref(k0), _C.byref(k1), _lib.ptr(hyp), self.n_slots, L, Q, _lib.ptr(x), _lib.ptr(idx),
                                                    S, T, B, M, _lib.ptr(iB), _lib.ptr(K0s), _lib.ptr(V), _lib.ptr(v), _lib.ptr(Y),
                                                    _lib.ptr(log_v), _C.c_double(c), _lib.ptr(gprm), st), "gp_subject_bwd")
                if balance == 2:
                    _lib.check(lib.hlvae_gp_param_grad(_C.byref(k0), _lib.ptr(hyp), self.n_slots, L, Q, _lib.ptr(x), B, 0, _lib.ptr(z), M, 0,
                                                       _lib.ptr(Y), _lib.ptr(gprm), _lib.ptr(gz), _lib.ptr(v), _lib.ptr(iKm), _C.c_double(c), st),
                               "gp_param_grad(Kxz)")
        def chain_a():
            nonlocal evW
            with torch.cuda.stream(sA):      # chain A: the sums over subjects, the bound, natural gradient, gradient w.r.t. K0zz
                st = self._stream()
                if fused:     # W was cleared with P1 and u by prepare(): no memset node in front of the chain's first launch
                    _lib.check(lib.hlvae_gp_gemm_acc(_lib.ptr(Kxz), Kxz.shape[2], Kxz.stride(0), 1, _lib.ptr(V), V.shape[2], V.stride(0),
                                                     _lib.ptr(W), M, W.stride(0), M, M, B, L, _C.c_double(1.0), st), "gp_gemm_acc")
                else:
                    self._gemm(Kxz, True, V, W, M, M, B)                         # sum_s Ks^T iB Ks = Kxz^T V   [L,M,M]
                # P1 = V^T mu (natural-gradient term, elbo_functions.py:262-266) and u = Kxz^T v: one streaming pass per latent each
                if not fused:
                    _lib.check(lib.hlvae_gp_gemv_t_f32(_lib.ptr(V), _lib.ptr(mu), 1, L, _lib.ptr(P1), L, B, M, st), "gp_gemv_t_f32")
                    _lib.check(lib.hlvae_gp_gemv_t(_lib.ptr(Kxz), _lib.ptr(v), v.stride(0), v.stride(1), _lib.ptr(u), L, B, M, st), "gp_gemv_t")
                _lib.check(lib.hlvae_gp_bound(_lib.ptr(part), S, _lib.ptr(W), _lib.ptr(iK), _lib.ptr(N1), _lib.ptr(self.H), _lib.ptr(self.m),
                                              _lib.ptr(iKm), _lib.ptr(ldK), _lib.ptr(ldH), _lib.ptr(log_v), B, L, M, _C.c_double(c),
                                              _C.c_double(float(self.N_total)), _C.c_double(1.0 / world), _lib.ptr(self.last_kld), st),
                           "gp_bound")
                if self.dp is not None:
                    self.dp.allreduce_(self._xchg)                               # W, P1, u, bound of the GLOBAL batch
                self._grad_m, self._grad_H, self._tmp = mm["grad_m"], mm["grad_H"], mm["tmp"]
                if self._chain and M % 4 == 0:
                    # the M x M algebra behind W as ONE launch (csrc/gp.hip k_gp_chain, round 3): natural-gradient terms
                    # (elbo_functions.py:279-283) and the symmetrised K0zz gradient (G + G^T), G = -(iK R iK) + iK / 2,
                    # R = c/2 (2 u m^T - W + HiKW + HiKW^T) + 1/2 (H + m m^T) -- built from global sums only, i.e. replicated:
                    # each rank contributes 1 / world of it
                    if self._chain == 2:
                        # by 32-row blocks, two launches (k_gp_chain_rb): 2 x 128 workgroups that never wait for each other
                        _lib.check(lib.hlvae_gp_chain_rb(_lib.ptr(iK), _lib.ptr(W), _lib.ptr(HiK), _lib.ptr(self.H), _lib.ptr(iH), _lib.ptr(self.m),
                                                         _lib.ptr(P1), _lib.ptr(u), _C.c_double(self.ng_lr), _C.c_double(c),
                                                         _C.c_double(-1.0 / world), _C.c_double(1.0 / world), M, L, _lib.ptr(self._grad_m),
                                                         _lib.ptr(self._grad_H), _lib.ptr(self._tmp), _lib.ptr(Rs), _lib.ptr(G_Kzz_s), st),
                                   "gp_chain_rb")
                    else:
                        _lib.check(lib.hlvae_gp_chain(_lib.ptr(iK), _lib.ptr(W), _lib.ptr(HiK), _lib.ptr(self.H), _lib.ptr(iH), _lib.ptr(self.m),
                                                      _lib.ptr(P1), _lib.ptr(u), _C.c_double(self.ng_lr), _C.c_double(c),
                                                      _C.c_double(-1.0 / world), _C.c_double(1.0 / world), M, L, _lib.ptr(mm["T1"]),
                                                      _lib.ptr(mm["Bm"]), _lib.ptr(self._grad_m), _lib.ptr(self._grad_H), _lib.ptr(self._tmp),
                                                      _lib.ptr(mm["HiKW"]), _lib.ptr(Rs), _lib.ptr(mm["T1b"]), _lib.ptr(G_Kzz_s), st), "gp_chain")
                    _lib.check(lib.hlvae_gp_param_grad(_C.byref(k0), _lib.ptr(hyp), self.n_slots, L, Q, _lib.ptr(z), M, 1, _lib.ptr(z), M, 1,
                                                       _lib.ptr(G_Kzz_s), _lib.ptr(gprm), _lib.ptr(gz), None, None, _C.c_double(0.0), st),
                               "gp_param_grad(Kzz)")
                    if balance == 1:
                        sA.wait_event(evY)
                        _lib.check(lib.hlvae_gp_param_grad(_C.byref(k0), _lib.ptr(hyp), self.n_slots, L, Q, _lib.ptr(x), B, 0, _lib.ptr(z), M, 0,
                                                           _lib.ptr(buf["Y"]), _lib.ptr(gprm), _lib.ptr(gz), _lib.ptr(v), _lib.ptr(iKm), _C.c_double(c), st),
                                   "gp_param_grad(Kxz)")
                else:
                    evW = torch.cuda.Event()
                    evW.record(sA)                                               # W, P1, u of the global batch are final
                    T1 = self._bmm_into(iK, W, mm["T1"])
                    Bm = self._bmm_into(T1, iK, mm["Bm"], D=iK)                  # iK W iK + iK
                    _lib.check(lib.hlvae_gp_natgrad(_lib.ptr(Bm), _lib.ptr(iK), _lib.ptr(iH), _lib.ptr(self.m), _lib.ptr(P1),
                                                    _C.c_double(self.ng_lr), M, L, _lib.ptr(self._grad_m), _lib.ptr(self._grad_H),
                                                    _lib.ptr(self._tmp), st), "gp_natgrad")
        # chain A (W -> bound -> M x M algebra -> K0zz gradient -> state update) is the step's critical path; queued FIRST its first
        # launch would be the per-subject kernel's first child and stay on its hardware queue (HL_GP_A_FIRST=1) -- measured 0.592 vs
        # 0.588 ms at configs[4], three alternating pairs: chain C first stays.  HL_GP_BALANCE=1 needs chain C's event first
        if balance == 1 or not self._a_first:
            chain_c()
            chain_a()
        else:
            chain_a()
            chain_c()
        if not (self._chain and M % 4 == 0):
            # separate launches (HL_GP_CHAIN=0): K0zz's gradient needs W but none of the natural-gradient products; it goes BEHIND
            # chain C (HL_GP_SPLIT=0: behind the natural-gradient launches on chain A, the round-2 order)
            with torch.cuda.stream(sC if self._split_kzz else sA):
                st = self._stream()
                if self._split_kzz:
                    sC.wait_event(evW)
                HiKW = self._bmm_into(HiK, W, mm["HiKW"])
                _lib.check(lib.hlvae_gp_rsym(_lib.ptr(u), _lib.ptr(self.m), _lib.ptr(W), _lib.ptr(HiKW), _lib.ptr(self.H), _C.c_double(c), M, L,
                                             _lib.ptr(Rs), st), "gp_rsym")
                T1b = self._bmm_into(iK, Rs, mm["T1b"])
                self._bmm_into(T1b, iK, G_Kzz_s, D=iK, alpha=-1.0 / world, beta=1.0 / world)
                _lib.check(lib.hlvae_gp_param_grad(_C.byref(k0), _lib.ptr(hyp), self.n_slots, L, Q, _lib.ptr(z), M, 1, _lib.ptr(z), M, 1,
                                                   _lib.ptr(G_Kzz_s), _lib.ptr(gprm), _lib.ptr(gz), None, None, _C.c_double(0.0), st),
                           "gp_param_grad(Kzz)")
        self._pending = True
        if join:
            self.join()
        return g_mu, g_lv

    def _streams(self, dev):
        if self._serial:            # HL_GP_SERIAL=1 (profiling): every launch on the caller's stream, each kernel runs alone
            cur = torch.cuda.current_stream(dev)
            self._side = (cur, cur)
            return self._side
        if self._side is None:
            self._side = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        return self._side

    def join(self):
        """the caller's stream waits for the two chains kl_and_grads left running; (data parallel) the hyper-parameter /
        inducing-point gradients are then summed over the ranks"""
        self.join_tail()
        if self._pending:
            main = torch.cuda.current_stream(self.zt_list.device)
            for s_ in self._side:
                main.wait_stream(s_)
            self._pending = False
            if self.dp is not None:
                self.dp.allreduce_(self._gtheta)

    # ---- evaluation surface: posterior mean of the latent at new covariates ------------------------------------------
    @torch.no_grad()
    def batch_predict_varying_T(self, prediction_x, test_x, mu):
        """Reference utils.py:99-191 (``batch_predict_varying_T``): prediction_x [Np, Q] (fp64) with encoder means
        mu [Np, L]; test_x [Nt, Q]; returns Z_pred [Nt, L] fp64.  Kernel matrices, the per-subject blocks (B_st inverse,
        iB K0xz, iB mu: the training kernel k_gp_subject_fwd with the roles a := mu, c := 1) and the M x M inverses run in
        the HIP kernels of csrc/gp.hip; the remaining products are library GEMMs.  Evaluation path: host syncs allowed."""
        lib, st = _lib.load(), self._stream()
        L, M, Q = self.L, self.M, self.Q
        dev = prediction_x.device
        f64 = dict(dtype=torch.float64, device=dev)
        px, tx = prediction_x.to(torch.float64).contiguous(), test_x.to(torch.float64).contiguous()
        Np = px.shape[0]
        k0, k1, z = self.k0, self.k1, self.zt_list
        hyp = self._transform()
        K0xz = self.kernel_matrix(k0, px, z)                                             # utils.py:127
        K0zz = self.kernel_matrix(k0, z, z, jitter=self.eps)                             # :128,131
        K0Xz = self.kernel_matrix(k0, tx, z)                                             # :129
        idx = self._group(px)
        S, T = idx.shape
        mu64T = mu.to(torch.float64).t().contiguous()                                    # [L, Np]
        zeros32 = torch.zeros(Np, L, dtype=torch.float32, device=dev)
        iB = torch.empty(S, L, T, T, **f64); K0s = torch.empty(S, L, T, T, **f64)
        V = torch.empty(L, Np, M, **f64); v = torch.empty(L, Np, **f64); part = torch.empty(S, L, 4, **f64)
        g1, g2 = torch.empty_like(zeros32), torch.empty_like(zeros32)
        _lib.check(lib.hlvae_gp_subject_fwd(_C.byref(k0), _C.byref(k1), _lib.ptr(hyp), self.n_slots, L, Q, _lib.ptr(px),
                                            _lib.ptr(self.noise), _lib.ptr(idx), S, T, _lib.ptr(K0xz), Np, M, _lib.ptr(mu64T),
                                            _lib.ptr(zeros32), _C.c_double(1.0), _lib.ptr(iB), _lib.ptr(K0s), _lib.ptr(V), _lib.ptr(v),
                                            _lib.ptr(part), _lib.ptr(g1), _lib.ptr(g2), None, None, None, None, st), "gp_subject_fwd(predict)")
        K0zx = K0xz.transpose(1, 2)
        inv, _ = self.chol_inv(torch.cat([K0zz + K0zx @ V, K0zz]))                       # H = K0zz + sum_s Ks^T iB Ks (:156-157)
        iH, iK = inv[:L], inv[L:]
        iB_mu = v.unsqueeze(2)                                                           # :158
        t1 = (K0xz @ (iH @ (K0zx @ iB_mu))).squeeze(2)                                   # :162   [L, Np]
        valid = idx >= 0
        gi = idx.clamp(min=0).long()
        t1g = t1[:, gi] * valid[None].to(torch.float64)                                  # [L, S, T]
        t2g = torch.einsum("sltu,lsu->lst", iB, t1g)                                     # :164-166
        t2 = torch.zeros(L, Np, **f64)
        t2[:, gi[valid]] = t2g[:, valid]
        mu_tilde = iB_mu - t2.unsqueeze(2)                                               # :167
        a = K0Xz @ (iK @ (K0zx @ mu_tilde))                                              # :169
        b = torch.zeros(L, tx.shape[0], 1, **f64)
        ids_p, ids_t = px[:, self.id_covariate], tx[:, self.id_covariate]
        for s_ in torch.unique(ids_t).tolist():                                          # :175-186
            rp = torch.nonzero(ids_p == s_).flatten()
            if rp.numel() == 0:
                continue                                                                 # K1 of an unseen subject against the rest is zero
            rt = torch.nonzero(ids_t == s_).flatten()
            K1 = self.kernel_matrix(k1, tx[rt].contiguous(), px[rp].contiguous())
            b[:, rt] = K1 @ mu_tilde[:, rp]
        return (a + b).squeeze(2).t().contiguous()                                       # :188

    def optimizer_step(self, defer=False, next_batch=None):
        """next_batch = (labels, rows) of the FOLLOWING step: its K0xz is computed beside the batched inversion (compute_ahead).
        Adam on [hyper-parameters | inducing points] (HLVAE_main.py:277-278; one fused kernel, device-side step counter), then
        the natural-gradient update of (m, H), training.py:130-137: iH_new = iH + lr (gH + gH^T) in place, ONE batched inversion
        of [iH_new | K0zz of the parameters Adam has just produced] -> [H_new | iK] straight into their homes (the next step
        starts with both factorisations and log-determinants in hand), m_new = H_new (iH m - lr (grad_m - 2 gH m))."""
        if self._iH is None:
            raise RuntimeError("GPPriorHIP.optimizer_step: call kl_and_grads first (it leaves grad_m, grad_H and the update's right-hand side)")
        if defer and self.dp is None and self._pending:
            # The state update (125 us of dependent launches: Adam -> iH update -> transform -> K0zz -> the batched inversion ->
            # m_new) on the prior's own stream, behind its two chains -- the caller's stream does not wait: the next step's
            # prepare() goes behind it on the same stream, and the next step's encoder / decoder forward runs beside both.
            # join_tail() (end of a captured chain, check(), state readers) makes the caller's stream wait.
            dev = self.zt_list.device
            if self._prep_stream is None:
                self._prep_stream = torch.cuda.current_stream(dev) if self._serial else torch.cuda.Stream(device=dev)
            for s_ in self._side:
                self._prep_stream.wait_stream(s_)
            self._pending = False
            caller = torch.cuda.current_stream(dev)
            with torch.cuda.stream(self._prep_stream):
                self._state_update(next_batch, caller=caller)
            self._tail_pending = True
            return
        self.join()
        self._state_update(next_batch)

    def join_tail(self):
        """the caller's stream waits for a deferred state update (optimizer_step(defer=True))"""
        if self._tail_pending:
            torch.cuda.current_stream(self.zt_list.device).wait_stream(self._prep_stream)
            self._tail_pending = False

    def _state_update(self, next_batch=None, caller=None):
        lib, st, L, M = _lib.load(), self._stream(), self.L, self.M
        self._ahead = None
        # Adam, the transform of the hyper-parameters it produced and iH_new (in place in _KH2[:L]): one launch
        _lib.check(lib.hlvae_gp_state_head(_lib.ptr(self._theta), _lib.ptr(self._gtheta), _lib.ptr(self._adam_m), _lib.ptr(self._adam_v),
                                           self._theta.numel(), _lib.ptr(self._adam_step), _C.c_double(self.lr), _C.c_double(0.9),
                                           _C.c_double(0.999), _C.c_double(1e-8), self.n_slots, L, _lib.ptr(self._hyp),
                                           _lib.ptr(self._grad_H), _lib.ptr(self._iHb), _C.c_double(self.ng_lr), M, L, st), "gp_state_head")
        self._iH = None
        if next_batch is not None:
            if caller is not None:
                # deferred update (this runs on the preparation stream): the look-ahead stream forks from the CALLER's stream behind
                # an event of the launch above -- forked from this stream and joined back into it by the next prepare() it would be
                # the capture pattern described in kl_and_grads.  The caller's stream waits for one 8 us launch, not for the update.
                e_head = torch.cuda.Event()
                e_head.record(torch.cuda.current_stream(self.zt_list.device))
                caller.wait_event(e_head)
            self.compute_ahead(*next_batch, fork_from=caller)
        self.kernel_matrix(self.k0, self.zt_list, self.zt_list, jitter=self.eps, out=self._KH2[L:])
        self._spd_inv(self._KH2, self._HiK, self._ld2, n_neg=L, logdet_neg=self._ldH)      # log det H_new = - log det iH_new
        self._bmv(self.H, self._tmp, self.m)                                 # m_new = H_new tmp
        self._fact_key = (self._theta._version, self._KH._version)
