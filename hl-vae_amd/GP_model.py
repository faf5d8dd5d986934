"""Additive GP covariance functions of the longitudinal prior, evaluated for all latent dimensions at once.

Same parametrisation and class names as the reference's gpytorch-free statement of its kernels
(reference GP_model.py:7-116: exp(min + softplus(raw - min)) positivity transform with min = -16,
RBF lengthscale initial value 2.5, scale initial value ln 2) so that state dicts and hyper-parameter
values carry over; the executed reference builds the same kernels with gpytorch
(kernel_gen.py:199-310), which is not vendored.

Unlike the reference, one ``AdditiveKernel`` evaluates every term on a stacked tensor in a single
pass (``terms x L x n1 x n2``) instead of a Python ``sum(map(...))`` over sub-modules; inputs may carry
arbitrary leading batch dimensions (used for the padded per-subject blocks).
All tensors are float64 (the reference's GP is fp64; T x T and M x M factorizations are not bf16 work).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
from torch import nn
import torch.nn.functional as F

_MIN_LOG = -16.0


def _positive(raw: torch.Tensor) -> torch.Tensor:
    return torch.exp(_MIN_LOG + F.softplus(raw - _MIN_LOG))


def _raw(value: float) -> float:
    return math.log(value - math.exp(_MIN_LOG))


class Likelihoods(nn.Module):
    """Gaussian observation noise per latent dimension (reference GP_model.py:7-25)."""

    def __init__(self, latent_dim: int, noise: float = 1.0, constrain: bool = True):
        super().__init__()
        self.latent_dim = latent_dim
        self._log_noise = nn.Parameter(torch.full((latent_dim,), _raw(noise), dtype=torch.float64), requires_grad=not constrain)

    @property
    def noise(self) -> torch.Tensor:
        return _positive(self._log_noise)


class _Base(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim


class CatKernel(_Base):          # reference GP_model.py:35-41: 1 if equal
    def forward(self, x1, x2):
        return (x1[..., self.dim].unsqueeze(-1) == x2[..., self.dim].unsqueeze(-2)).to(torch.float64)


class BinKernel(_Base):          # reference GP_model.py:27-33: 1 if both flags are 1
    def forward(self, x1, x2):
        return ((x1[..., self.dim].unsqueeze(-1) + x2[..., self.dim].unsqueeze(-2)) == 2).to(torch.float64)


class RbfKernel(_Base):          # reference GP_model.py:43-69
    def __init__(self, dim: int, latent_dim: int = 1, lengthscale: float = 2.5):
        super().__init__(dim)
        self._log_lengthscale = nn.Parameter(torch.full((latent_dim,), _raw(lengthscale), dtype=torch.float64))

    @property
    def lengthscale(self):
        return _positive(self._log_lengthscale)

    def forward(self, x1, x2):
        d2 = (x1[..., self.dim].unsqueeze(-1) - x2[..., self.dim].unsqueeze(-2)) ** 2
        ls = self.lengthscale.view(-1, 1, 1)
        return torch.exp(-d2 / (2.0 * ls * ls))


class ProductKernel(nn.Module):  # reference GP_model.py:109-116
    def __init__(self, *kernels):
        super().__init__()
        self.factors = nn.ModuleList(kernels)

    def forward(self, x1, x2):
        out = None
        for k in self.factors:
            v = k(x1, x2)
            out = v if out is None else out * v
        return out


class ScaleKernel(nn.Module):    # reference GP_model.py:71-97
    def __init__(self, kernel: nn.Module, latent_dim: int = 1, scale: float = math.log(2)):
        super().__init__()
        self.kernel = kernel
        self._log_scale = nn.Parameter(torch.full((latent_dim,), _raw(scale), dtype=torch.float64))

    @property
    def scale(self):
        return _positive(self._log_scale)

    def forward(self, x1, x2):
        return self.scale.view(-1, 1, 1) * self.kernel(x1, x2)


class AdditiveKernel(nn.Module):  # reference GP_model.py:99-107
    def __init__(self, kernels: Sequence[nn.Module]):
        super().__init__()
        self.kernels = nn.ModuleList(kernels)

    def forward(self, x1, x2):
        """x1 [..., n1, Q], x2 [..., n2, Q] -> [..., L, n1, n2] when the inputs carry no L axis,
        or [L, n1, n2] when x2 (or x1) already has a leading L axis (inducing points zt_list [L, M, Q])."""
        out = None
        for k in self.kernels:
            v = k(x1, x2)
            out = v if out is None else out + v
        return out


def generate_kernel_batched(latent_dim, cat_kernel, bin_kernel, sqexp_kernel, cat_int_kernel, bin_int_kernel,
                            covariate_missing_val, id_covariate) -> Tuple[AdditiveKernel, AdditiveKernel]:
    """(kernel without the id covariate, kernel with it) from the reference's config lists
    (config/hlvae_config_file.txt:41-45; term order of reference GP_model.py:143-206)."""
    miss = {d["covariate"]: d["mask"] for d in covariate_missing_val}

    def masked(k, idx):
        return ProductKernel(k, BinKernel(miss[idx])) if idx in miss else k

    k0: List[nn.Module] = []
    k1: List[nn.Module] = []
    for idx in cat_kernel:
        (k1 if idx == id_covariate else k0).append(ScaleKernel(masked(CatKernel(idx), idx), latent_dim))
    for idx in sqexp_kernel:
        k0.append(ScaleKernel(masked(RbfKernel(idx, latent_dim), idx), latent_dim))
    for idx in bin_kernel:
        k0.append(ScaleKernel(masked(BinKernel(idx), idx), latent_dim))
    for d in cat_int_kernel:
        prod = ProductKernel(masked(CatKernel(d["cat_covariate"]), d["cat_covariate"]),
                             masked(RbfKernel(d["cont_covariate"], latent_dim), d["cont_covariate"]))
        (k1 if d["cat_covariate"] == id_covariate else k0).append(ScaleKernel(prod, latent_dim))
    for d in bin_int_kernel:
        prod = ProductKernel(masked(BinKernel(d["bin_covariate"]), d["bin_covariate"]),
                             masked(RbfKernel(d["cont_covariate"], latent_dim), d["cont_covariate"]))
        k0.append(ScaleKernel(prod, latent_dim))
    return AdditiveKernel(k0), AdditiveKernel(k1)
