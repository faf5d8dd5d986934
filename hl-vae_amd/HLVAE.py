"""Drop-in ``HLVAE`` for MI355X: same constructor, attributes, ``state_dict`` keys and return
tuples as the reference class (reference HLVAE.py:104-475), MLP path (``conv=False``), with the
whole forward/backward running in hand-written HIP kernels behind the C ABI of
include/hlvae_hip.h.  There is NO CPU fallback: calling the model with CPU tensors raises.

Call surface mirrored:
    HLVAE(dims, types_info, n_variables, vy_init, vy_fixed, logvar_network, conv)   HLVAE.py:109
    forward(data, mask, param_mask, types_info, do_test=False) -> 8-tuple           HLVAE.py:364-375
    encode / decode / sample_latent / loss_function / get_test_samples              HLVAE.py:284-379, 455-475
    attributes types_info, conv, logvar_network, _log_vy_real, _log_vy_pos, z_dim, num_dim, y_dim

Memory layout (MI355X-first, not the reference's): every parameter is a view into ONE flat fp32
arena (gradients and Adam moments likewise), so the optimiser is a single HBM-streaming kernel and
the data-parallel gradient all-reduce is one contiguous RCCL call; dense weights have padded bf16
shadow copies (plus transposes) that the MFMA kernels read.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import _lib
from .layout import (KIND_CAT, KIND_COUNT, KIND_ORDINAL, KIND_POS, KIND_REAL, ColumnPlan, compile_plan)


_INPUT_STAGE = ("sums", "norm", "xn", "xnT", "xt", "m8")
GRAD_SLACK = 4096        # floats behind the gradient arena (hlvae_amd.parallel.ShardPlan.pad <= world * 32)


def _ru(v, m):
    return (v + m - 1) // m * m


# --- head modules: parameter containers with the reference's names and shapes ----------------------
class Observation_Count(nn.Module):                      # reference HLVAE.py:11-22
    def __init__(self, n, y_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, y_dim, 1))
        self.bias = nn.Parameter(torch.empty(n, 1))


class Observation_Real_Pos_Beta(nn.Module):              # reference HLVAE.py:25-51
    def __init__(self, n, y_dim, logvar_network=False):
        super().__init__()
        if logvar_network:                               # registered first, as in the reference (state_dict key order)
            self.weight_logvar = nn.Parameter(torch.empty(n, y_dim, 1))
            self.bias_logvar = nn.Parameter(torch.empty(n, 1))
        self.weight_mean = nn.Parameter(torch.empty(n, y_dim, 1))
        self.bias_mean = nn.Parameter(torch.empty(n, 1))


class Observation_Cat(nn.Module):                        # reference HLVAE.py:54-68
    def __init__(self, n, y_dim, nclass):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, y_dim, nclass - 1))
        self.bias = nn.Parameter(torch.empty(n, nclass - 1))


class Observation_Ordinal(nn.Module):                    # reference HLVAE.py:70-89
    def __init__(self, n, y_dim, nclass):
        super().__init__()
        self.weight_region = nn.Parameter(torch.empty(n, y_dim, 1))
        self.bias_region = nn.Parameter(torch.empty(n, 1))
        self.weight_thresholds = nn.Parameter(torch.empty(n, nclass - 1))


class Representation_One_Hot(nn.Module):                 # reference HLVAE.py:91-102 (conv encoder input)
    def __init__(self, n, nclass):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, nclass))
        self.bias = nn.Parameter(torch.empty(n))


class _Anchor(torch.autograd.Function):
    """Connects the HIP forward/backward to autograd.  The only differentiable input is a dummy
    anchor; backward() runs the HIP backward and assigns ``p.grad`` views of the gradient arena."""

    @staticmethod
    def forward(ctx, anchor, model, data, mask, eps, B):
        ctx.model, ctx.B = model, B
        ctx.data, ctx.mask = data, mask
        ctx.set_materialize_grads(False)
        ctx.token = model._run_forward(data, mask, eps, B, want_params=True)
        ctx.eps = model._ws_t["eps"][:B].clone()      # the noise actually used (given or generated in-kernel)
        mu, lv, z, lpx, lpm = model._clone_outputs(B)
        ctx.mark_non_differentiable(lpm)
        return mu, lv, z, lpx, lpm

    @staticmethod
    def backward(ctx, g_mu, g_lv, g_z, g_lpx, g_lpm):
        model, B = ctx.model, ctx.B
        if g_z is not None:
            raise NotImplementedError("gradient through the latent sample z itself is not on the hot path")
        if model._fwd_token != ctx.token:     # another forward overwrote the workspace: recompute with the same noise
            model._run_forward(ctx.data, ctx.mask, ctx.eps, B, want_params=False)
        f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
        model._run_backward(ctx.eps, f32(g_lpx), f32(g_mu), f32(g_lv), B)
        model._assign_grads()
        return None, None, None, None, None, None


class HLVAE(nn.Module):
    """Heterogeneous longitudinal VAE, MLP or convolutional encoder/decoder, HIP hot path (see module docstring)."""

    def __init__(self, dims, types_info, n_variables, vy_init=[1., .5], vy_fixed=False, logvar_network=False,
                 conv=True, max_batch=512, materialize_samples=True, group_variables=True):
        super().__init__()
        [x_dim, h_dim_e, z_dim, h_dim_d, y_dim] = dims
        if conv and (n_variables != 36 * 36 or y_dim != 5):
            raise ValueError("conv=True views the variables as one 36 x 36 image with y_dim = 5 output channels "
                             "(reference HLVAE.py:305, 257-258)")
        # the reference's "no hidden layer" spellings (HLVAE.py:128, 233): None, [], 0, [0]
        h_dim_e = [] if h_dim_e in (None, 0, [0]) else h_dim_e
        h_dim_d = [] if h_dim_d in (None, 0, [0]) else h_dim_d
        if not (isinstance(h_dim_e, (list, tuple)) and isinstance(h_dim_d, (list, tuple))
                and len(h_dim_e) <= 1 + _lib.MAX_EXTRA and len(h_dim_d) <= 1 + _lib.MAX_EXTRA
                and all(int(w) > 0 for w in list(h_dim_e) + list(h_dim_d))):
            raise NotImplementedError(f"dims[1] / dims[3] must list 0..{1 + _lib.MAX_EXTRA} positive hidden widths per side (reference "
                                      "config: [500])")
        h_dim_e = [int(w) for w in h_dim_e]
        h_dim_d = [int(i) for i in reversed(h_dim_d)]                               # HLVAE.py:113
        # no hidden layer on a side (round 3): the kernels keep their one-hidden-layer shape with a LINEAR "hidden layer" -- encoder:
        # [mean_layer.weight; log_var_layer.weight] is the kernels' first Linear (width 2 z_dim) and their mean / log-var heads hold an
        # identity that is never trained; decoder: the kernels' decoder trunk is an identity on the latent (include/hlvae_hip.h:
        # hlvae_dims.lin_e / lin_d)
        self._lin_e, self._lin_d = len(h_dim_e) == 0, len(h_dim_d) == 0
        self.z_dim, self.num_dim, self.y_dim = z_dim, n_variables, y_dim
        self.logvar_network, self.conv = logvar_network, conv
        self.tau = 1e-3
        self.types_info = types_info
        self.materialize_samples = materialize_samples
        self._group_variables = bool(group_variables)   # extension: kernel-facing variable order grouped by kind
        self.plan: ColumnPlan = compile_plan(types_info, y_dim)
        if self.plan.logvar_network != bool(logvar_network):
            raise ValueError("types_info['param_indexes'] was built for logvar_network=%s" % self.plan.logvar_network)
        if self.plan.X != x_dim or self.plan.D != n_variables:
            raise ValueError(f"dims[0]={x_dim}/n_variables={n_variables} do not match types_info "
                             f"(X={self.plan.X}, D={self.plan.D})")
        # h_e: the LAST encoder layer (feeds mean / log-var); h_d: the LAST decoder layer (y_layer's input); h_d0: the first
        self.h_e = 2 * z_dim if self._lin_e else h_dim_e[-1]
        self.h_d, self.h_d0 = (z_dim, z_dim) if self._lin_d else (h_dim_d[-1], h_dim_d[0])
        self._h_dim_e, self._h_dim_d = h_dim_e, h_dim_d
        pl = self.plan
        # bookkeeping attributes the reference exposes (HLVAE.py:180-201)
        self.real_dim, self.pos_dim = pl.n_real, pl.n_pos

        # ---- modules with the reference's names (state_dict keys) ------------------------------
        conv_params: List[nn.Parameter] = []
        if conv:                                                                     # HLVAE.py:139-155
            self.representation_layer = nn.ModuleList()
            self._rep_of_block = {}
            for bi_, b in enumerate(pl.blocks):
                if b["type"] in ("cat", "ordinal"):
                    self._rep_of_block[bi_] = len(self.representation_layer)
                    self.representation_layer.append(Representation_One_Hot(b["n_vars"], b["nclass"]))
            self.conv1 = nn.Conv2d(1, 16, kernel_size=3, stride=1, padding=1)        # parameter containers: the compute is
            self.pool1 = nn.MaxPool2d(kernel_size=2, stride=2, padding=0)            # csrc/conv.hip
            self.conv2 = nn.Conv2d(16, 32, kernel_size=3, stride=1, padding=1)
            self.pool2 = nn.MaxPool2d(kernel_size=2, stride=2, padding=0)
            conv_params += [self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias]
            x_enc = _lib.CONV_FEATURES
        else:
            x_enc = x_dim
        e_layers, n_in = [], x_enc                                                   # HLVAE.py:128-137 / 156-165
        for w in h_dim_e:
            e_layers += [nn.Linear(n_in, w), nn.ReLU()]
            n_in = w
        self.VAE_encoder_common_layers = nn.Sequential(*e_layers)
        self.mean_layer = nn.Sequential(nn.Linear(n_in, z_dim))                     # (n_in: the last hidden width, or the input's)
        self.log_var_layer = nn.Sequential(nn.Linear(n_in, z_dim))
        if logvar_network:                                                           # HLVAE.py:219-222: plain None attributes
            self._log_vy_real = self._log_vy_pos = None
        else:
            self._log_vy_real = nn.Parameter(torch.empty(pl.n_real))
            self._log_vy_pos = nn.Parameter(torch.empty(pl.n_pos))
        self._disp_param = nn.Parameter(torch.ones(1))
        self.d_layers = nn.ModuleList()                                              # HLVAE.py:232-240
        n_in = z_dim
        for w in h_dim_d:
            self.d_layers.append(nn.Linear(n_in, w))
            self.d_layers.append(nn.ReLU())
            n_in = w
        self.hidden = nn.Sequential(*self.d_layers)                                  # alias, HLVAE.py:242
        self.y_layer = nn.Sequential(nn.Linear(self.h_d, _lib.CONV_FEATURES if conv else y_dim * n_variables))   # HLVAE.py:244-248
        if conv:                                                                     # HLVAE.py:253-259
            self.deconv_layer = nn.ModuleList([nn.ConvTranspose2d(32, 16, kernel_size=4, stride=2, padding=1), nn.ReLU(),
                                               nn.ConvTranspose2d(16, y_dim, kernel_size=4, stride=2, padding=1)])
            self.Decoder_Conv_layer = nn.Sequential(*self.deconv_layer)
            conv_params += [self.deconv_layer[0].weight, self.deconv_layer[0].bias, self.deconv_layer[2].weight,
                            self.deconv_layer[2].bias]
        self.obs_layer = nn.ModuleList()
        for b in pl.blocks:
            n, K = b["n_vars"], b["nclass"]
            if b["type"] == "count":
                self.obs_layer.append(Observation_Count(n, y_dim))
            elif b["type"] in ("real", "pos"):
                self.obs_layer.append(Observation_Real_Pos_Beta(n, y_dim, logvar_network))
            elif b["type"] == "cat":
                self.obs_layer.append(Observation_Cat(n, y_dim, K))
            else:
                self.obs_layer.append(Observation_Ordinal(n, y_dim, K))
        if conv and pl.n_real:
            self.obs_layer.append(nn.Sigmoid())                                      # HLVAE.py:271-273 ('real' sorts last)

        # ---- flat arena: [atomically accumulated grads | dense weights] -------------------------
        order: List[nn.Parameter] = ([] if logvar_network else [self._log_vy_real, self._log_vy_pos]) + [self._disp_param]
        for m in self.obs_layer:
            order += list(m.parameters())
        if conv:      # small tensors whose gradients are accumulated with atomics by the convolution kernels
            for m in self.representation_layer:
                order += [m.weight, m.bias]
            order += conv_params
        # the fused kernels' "first encoder Linear" is the LAST of the stack, their decoder trunk the FIRST decoder Linear;
        # the other hidden layers are the "extra" ones (include/hlvae_hip.h: hlvae_layer)
        # (plain attribute slots: assigning a Module to self would register it under a second state_dict key)
        object.__setattr__(self, "_extra_enc", [self.VAE_encoder_common_layers[2 * i] for i in range(len(h_dim_e) - 1)])
        object.__setattr__(self, "_extra_dec", [self.d_layers[2 * j] for j in range(1, len(h_dim_d))])

        def hidden_param(*shape):      # a tensor of the arena that is NOT a parameter of the model (no state_dict key, never trained)
            return nn.Parameter(torch.empty(*shape), requires_grad=False)
        glue = set()                   # parameters that the next one follows without padding (one matrix for the kernels)
        if self._lin_e:
            # kernels' "first Linear" = [mean ; log-var] rows, their heads = identity
            k_w1, k_b1 = [self.mean_layer[0].weight, self.log_var_layer[0].weight], [self.mean_layer[0].bias, self.log_var_layer[0].bias]
            glue |= {id(k_w1[0]), id(k_b1[0])}
            object.__setattr__(self, "_id_enc", (hidden_param(z_dim, 2 * z_dim), hidden_param(z_dim, 2 * z_dim), hidden_param(z_dim),
                                                 hidden_param(z_dim)))
            k_wmu, k_wlv, k_bmu, k_blv = self._id_enc
        else:
            enc_last = self.VAE_encoder_common_layers[2 * (len(h_dim_e) - 1)]
            k_w1, k_b1 = [enc_last.weight], [enc_last.bias]
            k_wmu, k_wlv, k_bmu, k_blv = (self.mean_layer[0].weight, self.log_var_layer[0].weight, self.mean_layer[0].bias,
                                          self.log_var_layer[0].bias)
            object.__setattr__(self, "_id_enc", ())
        if self._lin_d:
            object.__setattr__(self, "_id_dec", (hidden_param(z_dim, z_dim), hidden_param(z_dim)))
            k_wd, k_bd = self._id_dec
        else:
            k_wd, k_bd = self.d_layers[0].weight, self.d_layers[0].bias
            object.__setattr__(self, "_id_dec", ())
        object.__setattr__(self, "_kernel_params", dict(w1=k_w1[0], b1=k_b1[0], wmu=k_wmu, wlv=k_wlv, bmu=k_bmu, blv=k_blv, wd=k_wd, bd=k_bd))
        order += [self.y_layer[0].bias, k_bd, k_bmu, k_blv] + k_b1 + [m.bias for m in self._extra_enc + self._extra_dec]
        n_small = len(order)
        # y_layer's weight LAST: its gradient is final first and is all-reduced on its own while the rest of the backward
        # pass runs; everything before it is then ONE contiguous slice for the second all-reduce
        order += ([k_wd, k_wmu, k_wlv] + k_w1 + [m.weight for m in self._extra_enc + self._extra_dec] + [self.y_layer[0].weight])
        self._order = order
        offs, o = [], 0
        for i, p in enumerate(order):
            if i == n_small:
                self._atomic_region = o
            offs.append(o)
            o = o + p.numel() if id(p) in glue else _ru(o + p.numel(), 32)
        self._offsets = offs
        self._arena_size = _ru(o, 64)
        arena = torch.zeros(self._arena_size, dtype=torch.float32)
        conv_init = [p.detach().clone() for p in conv_params]       # torch's default initialisation, as in the reference
        self._bind_arena(arena)
        self._init_parameters(vy_init)
        with torch.no_grad():
            for p_, v_ in zip(conv_params, conv_init):
                p_.copy_(v_)
        if vy_fixed and not logvar_network:
            self._log_vy_real.requires_grad_(False)
            self._log_vy_pos.requires_grad_(False)
        self._anchor = torch.zeros((), requires_grad=True)
        self._plan_handle = None
        self._ws = None
        self._ws_t = {}
        self._max_batch = max_batch
        self._shadow_versions = None
        self._fwd_token = 0
        self._block_cols = None
        self._grad_region_clean = True
        self._master_sync = None

    # ------------------------------------------------------------------ arena / parameters
    def _bind_arena(self, arena: torch.Tensor):
        self._arena = arena
        self._grad_arena = None
        for p, o in zip(self._order, self._offsets):
            p.data = arena[o:o + p.numel()].view(p.shape)
            p.grad = None

    def _init_parameters(self, vy_init):
        """Row P: N(0, 0.05^2) everywhere, thresholds 1, _log_vy = log(vy - e^-8) (HLVAE.py:132-133,
        169-176, 205-216, 237-250; heads :17-18, 39-40, 60-61, 79-82)."""
        with torch.no_grad():
            for p in self._order:
                p.normal_(0.0, 0.05)
            for m in self.obs_layer:
                if isinstance(m, Observation_Ordinal):
                    m.weight_thresholds.fill_(1.0)
            min_log_vy = torch.tensor([-8.0])
            if self._log_vy_real is not None:
                self._log_vy_real.fill_(float(torch.log(vy_init[0] - torch.exp(min_log_vy))))
                self._log_vy_pos.fill_(float(torch.log(vy_init[1] - torch.exp(min_log_vy))))
            self._disp_param.fill_(1.0)
            self._fill_identities()

    def _fill_identities(self):
        """the never-trained tensors of a model without hidden layers (see __init__): [Wmu; Wlv] = I, Wd = I, their biases 0"""
        with torch.no_grad():
            if self._id_enc:
                wmu, wlv, bmu, blv = self._id_enc
                L = self.z_dim
                wmu.zero_(); wlv.zero_(); bmu.zero_(); blv.zero_()
                wmu[:, :L] = torch.eye(L, device=wmu.device)
                wlv[:, L:] = torch.eye(L, device=wlv.device)
            if self._id_dec:
                wd, bd = self._id_dec
                wd.copy_(torch.eye(self.z_dim, device=wd.device)); bd.zero_()

    def _apply(self, fn, *a, **k):
        """``.to(device)`` / ``.cuda()`` move the arena as ONE tensor and re-bind the parameter views;
        dtype casts (``.double()``, ``.to(torch.float64)`` as in reference HLVAE_main.py:156-158) are
        accepted and ignored: masters stay fp32, compute is bf16 MFMA with fp32 accumulation."""
        probe = fn(torch.empty(0, dtype=torch.float32, device=self._arena.device))
        if probe.device != self._arena.device:
            req = [p.requires_grad for p in self._order]
            self._release_device_state()
            self._bind_arena(self._arena.to(probe.device))
            for p, r in zip(self._order, req):
                p.requires_grad_(r)
            self._anchor = torch.zeros((), requires_grad=True, device=probe.device)
        return self

    def state_dict(self, *a, **k):
        """under the sharded data-parallel optimiser a rank's fp32 masters are current for its own slices only: gather
        the others first (hlvae_amd.parallel.ShardedState.sync_masters, a collective: call on every rank)"""
        if getattr(self, "_master_sync", None) is not None:
            self._master_sync()
        return super().state_dict(*a, **k)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        sd = {k: v.to(torch.float32) for k, v in state_dict.items()}
        out = super().load_state_dict(sd, strict=strict, assign=False)
        self._shadow_versions = None
        return out

    def _release_device_state(self):
        if self._plan_handle is not None:
            _lib.load().hlvae_plan_destroy(self._plan_handle)
        self._plan_handle, self._ws, self._ws_t, self._shadow_versions = None, None, {}, None
        self._grad_arena = None
        self._block_cols = None

    def __del__(self):
        try:
            self._release_device_state()
        except Exception:
            pass

    # ------------------------------------------------------------------ plan + workspace
    @property
    def device(self):
        return self._arena.device

    def arena_offset(self, p: nn.Parameter) -> int:
        for q, o in zip(self._order, self._offsets):
            if q is p:
                return o
        raise KeyError("parameter not in arena")

    def _build_dims(self) -> _lib.HlvaeDims:
        pl = self.plan
        d = _lib.HlvaeDims()
        d.D, d.X, d.y_dim, d.h_e, d.h_d, d.L = pl.D, pl.X, self.y_dim, self.h_e, self.h_d, self.z_dim
        d.Theta = pl.Theta
        d.n_real, d.n_pos = pl.n_real, pl.n_pos
        d.conv = int(bool(self.conv))
        ao = self.arena_offset
        if self.conv:
            d.o_c1w, d.o_c1b, d.o_c2w, d.o_c2b = ao(self.conv1.weight), ao(self.conv1.bias), ao(self.conv2.weight), ao(self.conv2.bias)
            d.o_t1w, d.o_t1b = ao(self.deconv_layer[0].weight), ao(self.deconv_layer[0].bias)
            d.o_t2w, d.o_t2b = ao(self.deconv_layer[2].weight), ao(self.deconv_layer[2].bias)
            first = self.representation_layer[0].weight if len(self.representation_layer) else self.conv1.weight
            d.o_cv_lo = ao(first)                         # the arena order puts these tensors back to back (see __init__)
            d.cv_n = ao(self.y_layer[0].bias) + self.y_layer[0].bias.numel() - d.o_cv_lo
        kp = self._kernel_params
        d.o_w1, d.o_b1 = ao(kp["w1"]), ao(kp["b1"])
        d.o_wmu, d.o_bmu = ao(kp["wmu"]), ao(kp["bmu"])
        d.o_wlv, d.o_blv = ao(kp["wlv"]), ao(kp["blv"])
        d.o_wd, d.o_bd = ao(kp["wd"]), ao(kp["bd"])
        d.lin_e, d.lin_d = int(self._lin_e), int(self._lin_d)
        d.o_wy, d.o_by = ao(self.y_layer[0].weight), ao(self.y_layer[0].bias)
        d.n_xe, d.n_xd, d.h_d0 = len(self._extra_enc), len(self._extra_dec), self.h_d0
        for arr, mods in ((d.xe, self._extra_enc), (d.xd, self._extra_dec)):
            for i, m in enumerate(mods):
                arr[i].n_in, arr[i].n_out, arr[i].o_w, arr[i].o_b = m.in_features, m.out_features, ao(m.weight), ao(m.bias)
        extra = self._extra_enc + self._extra_dec
        d.o_xw = ao(extra[0].weight) if extra else d.o_wy
        d.arena_size, d.atomic_region = self._arena_size, self._atomic_region
        d.frozen_lo = d.frozen_hi = 0
        if self._log_vy_real is not None and not self._log_vy_real.requires_grad and not self._log_vy_pos.requires_grad:      # vy_fixed (HLVAE.py:209-216)
            d.frozen_lo = ao(self._log_vy_real)                   # the two tensors are neighbours at the start of the arena
            d.frozen_hi = _ru(ao(self._log_vy_pos) + self._log_vy_pos.numel(), 4)
        _lib.load().hlvae_dims_fill(C.byref(d))
        return d

    def _build_vars(self):
        pl = self.plan
        arr = (_lib.HlvaeVar * pl.D)()
        ao = self.arena_offset
        lvn = self.logvar_network
        o_real, o_pos = (0, 0) if lvn else (ao(self._log_vy_real), ao(self._log_vy_pos))
        for d in range(pl.D):
            v = arr[d]
            kind, K, bi = int(pl.kind[d]), int(pl.ncls[d]), int(pl.bidx[d])
            m = self.obs_layer[int(pl.blk[d])]
            v.kind, v.ncls, v.xoff, v.sidx, v.e_off, v.pad = kind, K, int(pl.xoff[d]), -1, -1, 0
            v.r_off = v.rb_off = v.w2_off = v.b2_off = -1
            v.poff, v.poff2 = int(pl.poff[d]), int(pl.poff2[d])
            if self.conv and kind in (KIND_CAT, KIND_ORDINAL):
                rep = self.representation_layer[self._rep_of_block[int(pl.blk[d])]]
                v.r_off, v.rb_off = ao(rep.weight) + bi * K, ao(rep.bias) + bi
            if kind in (KIND_REAL, KIND_POS):
                v.w_off, v.b_off = ao(m.weight_mean) + bi * self.y_dim, ao(m.bias_mean) + bi
                si = int(pl.sidx[d])
                v.sidx = si if kind == KIND_REAL else pl.n_real + si      # reals first, then pos
                if lvn:
                    v.w2_off, v.b2_off = ao(m.weight_logvar) + bi * self.y_dim, ao(m.bias_logvar) + bi
                else:
                    v.e_off = (o_real if kind == KIND_REAL else o_pos) + si
            elif kind == KIND_COUNT:
                v.w_off, v.b_off = ao(m.weight) + bi * self.y_dim, ao(m.bias) + bi
            elif kind == KIND_CAT:
                v.w_off, v.b_off = ao(m.weight) + bi * self.y_dim * (K - 1), ao(m.bias) + bi * (K - 1)
            else:
                v.w_off, v.b_off = ao(m.weight_region) + bi * self.y_dim, ao(m.bias_region) + bi
                v.e_off = ao(m.weight_thresholds) + bi * (K - 1)
        return arr

    def _ensure_device_state(self, B: int):
        if self.device.type != "cuda":
            raise RuntimeError("hlvae_amd.HLVAE runs on MI355X through its HIP library only; move the model and the "
                               "inputs to the GPU (model.to('cuda')).  There is no CPU fallback.")
        lib = _lib.load()
        if self._plan_handle is None:
            self._dims = self._build_dims()
            h = C.c_void_p()
            order = self.kernel_var_order()
            _lib.check(lib.hlvae_plan_create(C.byref(h), C.byref(self._dims), self._build_vars(),
                                             None if order is None else order.ctypes.data_as(C.c_void_p)), "hlvae_plan_create")
            self._plan_handle = h
        Bp = _ru(max(B, 1), 128)
        grew = False
        if self._ws is None or Bp > self._ws.Bp_max:
            if self._ws is not None and getattr(self, "_grow_forbidden", None):
                raise RuntimeError(f"batch of {B} rows exceeds the workspace ({self._ws.Bp_max} rows): {self._grow_forbidden}")
            grew = self._ws is not None
            self._alloc_workspace(max(Bp, _ru(self._max_batch, 128)))
        self._sync_shadows()
        return grew          # True: every per-step buffer is new (a batch packed earlier is gone)

    def _require_capacity(self, B: int):
        """mid-step check: the workspace was sized at the top of the step (ELBOTrainer.step_rows); growing it now would swap the
        buffers between the forward and the backward pass"""
        if self._ws is None or _ru(max(B, 1), 128) > self._ws.Bp_max:
            raise RuntimeError(f"a batch of {B} rows does not fit the workspace ({0 if self._ws is None else self._ws.Bp_max} rows) "
                               "in the middle of a step: pass the prefetched batch to step_rows(prefetch_rows=...) so that it is "
                               "sized at the top of the step, or build the trainer with a larger max_batch")

    def kernel_var_order(self):
        """Order in which the head kernel walks the variables (int32 [D]; None = their own order): grouped by likelihood
        kind, then class count, stable otherwise.  The external order is the reference's (read_functions.py:142-198: data
        columns, p_params and log_p_x all follow types_info); only the bf16 shadows of y_layer's weight and dY use this one,
        so that a 16-variable tile of the kernel runs ONE likelihood body (csrc/heads.hip, include/hlvae_hip.h)."""
        if self.conv or not self._group_variables:
            return None
        key = np.asarray(self.plan.kind, dtype=np.int64) * 1024 + np.asarray(self.plan.ncls, dtype=np.int64)
        order = np.argsort(key, kind="stable").astype(np.int32)
        return None if np.array_equal(order, np.arange(len(order))) else np.ascontiguousarray(order)

    def kernel_wy_rows(self):
        """row of y_layer's weight held by each row of its bf16 shadow `wys` (tests, debugging)"""
        order = self.kernel_var_order()
        n = self.plan.D * self.y_dim
        if order is None:
            return np.arange(n)
        return (order.astype(np.int64)[:, None] * self.y_dim + np.arange(self.y_dim)[None, :]).reshape(-1)

    def _head_acc(self) -> int:
        """accumulators per variable of the head-kernel instance this plan selects (csrc/heads.hip: HeadAcc + y_dim)"""
        y, kmax = self.y_dim, max([2] + [int(k) for k, kd in zip(self.plan.ncls, self.plan.kind) if kd in (KIND_CAT, KIND_ORDINAL)])
        km = 8 if y != 5 else (3 if kmax <= 3 else 5 if kmax <= 5 else 8 if kmax <= 8 else 16)
        return max((y + 1) * (km - 1), y + km) + y

    def _alloc_workspace(self, Bp: int):
        d, dev = self._dims, self.device
        bf, f32 = torch.bfloat16, torch.float32
        z = lambda *s, dt=bf: torch.zeros(*s, dtype=dt, device=dev)
        ksteps_e, ksteps_d = d.K1p // 64, d.NYlp // 64
        # split-K: a multiple of 8 slices when K allows it (one K-slice per XCD: each slice of the operands is pulled
        # into exactly one L2), enough blocks for about two waves of the 256 CUs, and no empty split
        def pick(ksteps, tiles):
            S = max(1, min(ksteps, (640 + tiles - 1) // tiles))
            if S >= 8:
                S = (S // 8) * 8
            per = (ksteps + S - 1) // S
            return (ksteps + per - 1) // per
        S_e = pick(ksteps_e, (Bp // 64) * (d.hep // 64))
        S_d = pick(ksteps_d, (Bp // 64) * (d.hdp // 64))
        NT = (d.D + 15) // 16
        t = dict(
            G=z(self._arena_size + GRAD_SLACK, dt=f32),      # tail slack: the padded extent of the last reduce-scatter slice
            w1s=z(d.hep, d.K1p), wmls=z(2 * d.Lp, d.hep), wmlTs=z(d.hep, 2 * d.Lp), wds=z(d.hd0p, d.Lp),
            wdTs=z(d.Lp, d.hd0p), wys=z(d.NYlp if d.conv else d.NYl, d.hdp), wyTs=z(d.hdp, d.NYlp),
            sums=z(_lib.STAT_CHUNKS, 3, max(d.n_stat, 1), dt=torch.float64), norm=z(2, max(d.n_stat, 1), dt=f32),
            xn=z(Bp, d.Xep), xnT=z(d.Xep, Bp), xt=z(Bp, d.D, dt=f32), m8=z(Bp, d.D, dt=torch.uint8),
            slab=z(max(S_e, S_d), Bp, max(d.hep, d.hdp, d.hd0p), dt=f32),
            t=z(Bp, d.hep), tT=z(d.hep, Bp), mu=z(Bp, d.L, dt=f32), lv=z(Bp, d.L, dt=f32), z=z(Bp, d.L, dt=f32),
            zb=z(Bp, d.Lp), zbT=z(d.Lp, Bp), u=z(Bp, d.hdp), uT=z(d.hdp, Bp), dy=z(Bp, d.NYp), dyT=z(d.NY, Bp),
            log_p_x=z(Bp, d.D, dt=f32), log_p_x_missing=z(Bp, d.D, dt=f32), rowpart=z(NT, Bp, dt=f32), hgpart=z(Bp // 64, NT * 16, self._head_acc(), dt=f32),
            nll=z(Bp, dt=f32), scal=z(8, dt=torch.float64), klpart=z(max(Bp // 4, 1), dt=torch.float64),
            eps=z(Bp, d.L, dt=f32), rng=z(2, dt=torch.int64), pfull=z(Bp, d.Theta, dt=f32), xhat=z(Bp, d.D, dt=f32),
            metpart=z(16, 6, d.D, dt=f32),
            du=z(Bp, d.hd0p), duT=z(d.hd0p, Bp), dz=z(Bp, d.Lp, dt=f32), dml=z(Bp, 2 * d.Lp), dmlT=z(2 * d.Lp, Bp),
            dt=z(Bp, d.hep), dtT=z(d.hep, Bp))
        if d.conv:          # convolutional front / back end (csrc/conv.hip)
            t.update(w1Ts=z(d.Xep, d.hep), cpack=z(_lib.CONV_PACK_ELEMS), img=z(Bp, d.D, dt=f32), yc=z(Bp, d.NYlp),
                     a2=z(Bp, 18 * 18 * 16), yv=z(Bp, d.NY, dt=f32), da2=z(Bp, 18 * 18 * 16), dyc=z(Bp, d.NYlp),
                     dycT=z(d.NYl, Bp), dfeat=z(Bp, d.Xep, dt=f32), dimg=z(Bp, d.D, dt=f32), cvpart=z(_lib.CONV_PART_ROWS, d.cv_n, dt=f32))
        # second pair of y_layer shadows: a chain of captured training steps lets the optimiser write the updated shadows
        # beside the ones this step still reads (ELBOTrainer, include/hlvae_hip.h: wys_next)
        if not d.conv:
            t.update(wys_b=torch.zeros_like(t["wys"]), wyTs_b=torch.zeros_like(t["wyTs"]))
        # deeper trunks: shadows, activations and pre-activation gradients of the extra hidden layers (both layouts)
        if d.n_xe:
            t.update(w1Ts=z(d.K1p, d.hep))
        if d.n_xd:
            t.update(u0=z(Bp, d.hd0p), u0T=z(d.hd0p, Bp))
        for tag, arr, n in (("xe", d.xe, d.n_xe), ("xd", d.xd, d.n_xd)):
            for i in range(n):
                l = arr[i]
                last_dec = tag == "xd" and i == n - 1                   # y_layer's input: the buffers the head kernel reads
                t.update({f"{tag}{i}_w": z(l.n_out_p, l.n_in_p), f"{tag}{i}_wT": z(l.n_in_p, l.n_out_p),
                          f"{tag}{i}_a": t["u"] if last_dec else z(Bp, l.n_out_p), f"{tag}{i}_aT": t["uT"] if last_dec else z(l.n_out_p, Bp),
                          f"{tag}{i}_d": z(Bp, l.n_out_p), f"{tag}{i}_dT": z(l.n_out_p, Bp)})
        t["P"] = self._arena
        old_t = getattr(self, "_ws_t", None) or None                        # ({} before the first allocation)
        if old_t is not None and "rng" in old_t:                                         # re-allocation: the noise stream continues (seed incl. the
            t["rng"].copy_(old_t["rng"])                                    # per-rank offset of ELBOTrainer, and the step offset)
        else:
            t["rng"][0] = int(torch.randint(0, 2 ** 62, (1,)).item())      # Philox seed from torch's global RNG
        ws = _lib.HlvaeWs()
        ws.Bp_max, ws.splitk_enc, ws.splitk_dec = Bp, S_e, S_d
        for name in _lib.WS_POINTERS:
            setattr(ws, name, t[name].data_ptr() if name in t else None)
        ws.u0, ws.u0T = t.get("u0", t["u"]).data_ptr(), t.get("u0T", t["uT"]).data_ptr()
        for tag, arr, n in (("xe", ws.xe, d.n_xe), ("xd", ws.xd, d.n_xd)):
            for i in range(n):
                for f in ("w", "wT", "a", "aT", "d", "dT"):
                    setattr(arr[i], f, t[f"{tag}{i}_{f}"].data_ptr())
        # second set of the input-stage buffers (statistics + packed batch): the next batch can be normalised and packed
        # on a side stream while this one trains (ELBOTrainer.step(prefetch=...)); everything else is shared
        self._input_stage = _INPUT_STAGE + (("img",) if d.conv else ())
        alt = {n: torch.zeros_like(t[n]) for n in self._input_stage}
        ws2 = _lib.HlvaeWs()
        C.memmove(C.byref(ws2), C.byref(ws), C.sizeof(ws))
        for n in self._input_stage:
            setattr(ws2, n, alt[n].data_ptr())
        self._ws_alt, self._ws_t_alt = ws2, alt
        self._packed_key = None
        self._ws, self._ws_t = ws, t
        self._grad_arena = t["G"]
        self._shadow_versions = None

    def _set_wy_double_buffer(self, on: bool):
        """on: the next fused training step writes y_layer's updated shadows into the spare pair (ws->wys_next); off: in place"""
        t = self._ws_t
        for w in (self._ws, self._ws_alt):
            w.wys_next = t["wys_b"].data_ptr() if on else None
            w.wyTs_next = t["wyTs_b"].data_ptr() if on else None

    def _flip_wy_shadows(self):
        """after such a step: the spare pair holds the current shadows"""
        t = self._ws_t
        t["wys"], t["wys_b"] = t["wys_b"], t["wys"]
        t["wyTs"], t["wyTs_b"] = t["wyTs_b"], t["wyTs"]
        for w in (self._ws, self._ws_alt):
            w.wys, w.wyTs = t["wys"].data_ptr(), t["wyTs"].data_ptr()

    def _swap_input_buffers(self):
        """exchange the two sets of input-stage buffers (host-side pointer swap)"""
        self._ws, self._ws_alt = self._ws_alt, self._ws
        for n in self._input_stage:
            self._ws_t[n], self._ws_t_alt[n] = self._ws_t_alt[n], self._ws_t[n]

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _sync_shadows(self, force=False):
        """bf16 shadows follow the fp32 masters: refreshed when any parameter tensor was modified in
        place (optimizer step, load_state_dict) since the last refresh."""
        ver = tuple(p._version for p in self._order)
        if force or ver != self._shadow_versions:
            _lib.check(_lib.load().hlvae_refresh_shadows(self._plan_handle, C.byref(self._ws), self._stream()),
                       "hlvae_refresh_shadows")
            self._shadow_versions = ver

    def mark_shadows_fresh(self):
        """called by the fused optimiser (which rewrites the shadows itself)"""
        self._shadow_versions = tuple(p._version for p in self._order)

    # ------------------------------------------------------------------ raw stage drivers
    def _prep_inputs(self, data, mask):
        if data.device.type != "cuda" or mask.device.type != "cuda":
            raise RuntimeError("hlvae_amd.HLVAE: inputs must live on the GPU (no CPU fallback)")
        if data.dim() != 2 or data.shape[1] != self.plan.X or mask.shape != (data.shape[0], self.plan.D):
            raise ValueError(f"expected data [B,{self.plan.X}] and mask [B,{self.plan.D}], got {tuple(data.shape)} "
                             f"and {tuple(mask.shape)}")
        return data.to(torch.float64).contiguous(), mask.to(torch.float64).contiguous()

    def _run_normalize(self, data, mask, B, stats_hook=None):
        lib, ws, s = _lib.load(), C.byref(self._ws), self._stream()
        self._packed_key = None
        if stats_hook is None:      # single process: statistics and pack in one launch
            _lib.check(lib.hlvae_normalize_fused(self._plan_handle, ws, _lib.ptr(data), _lib.ptr(mask), B, s), "normalize_fused")
            return
        _lib.check(lib.hlvae_normalize_stats(self._plan_handle, ws, _lib.ptr(data), _lib.ptr(mask), B, s), "normalize_stats")
        if stats_hook is not None:
            stats_hook(self._ws_t["sums"])          # data-parallel: all-reduce the masked column sums
        _lib.check(lib.hlvae_normalize_pack(self._plan_handle, ws, _lib.ptr(data), _lib.ptr(mask), B, s), "normalize_pack")

    def _run_encoder(self, eps, sample, B):
        """eps given -> that noise; eps None and sample -> in-kernel Philox noise (host offset advances per
        call so eager calls never reuse a noise tensor); sample False -> z = mu."""
        lib, ws, s = _lib.load(), C.byref(self._ws), self._stream()
        self._rng_calls = getattr(self, "_rng_calls", 0) + 1
        _lib.check(lib.hlvae_encoder_fwd(self._plan_handle, ws, _lib.ptr(eps), int(bool(sample)),
                                         C.c_uint64(self._rng_calls << 32), B, s), "encoder_fwd")

    def _run_forward(self, data, mask, eps, B, want_params=True, g_scale=1.0, want_grad=False, stats_hook=None,
                     sample=True):
        lib, ws, s = _lib.load(), C.byref(self._ws), self._stream()
        self._run_normalize(data, mask, B, stats_hook)
        self._run_encoder(eps, sample, B)
        if want_grad:
            _lib.check(lib.hlvae_zero_grad(self._plan_handle, ws, s), "zero_grad")
        _lib.check(lib.hlvae_decoder_fwd(self._plan_handle, ws, None, C.c_float(g_scale), int(want_grad), int(want_params),
                                         0, B, s), "decoder_fwd")
        self._fwd_token += 1
        return self._fwd_token

    def _run_decoder_only(self, B, want_params=True):
        lib, ws, s = _lib.load(), C.byref(self._ws), self._stream()
        _lib.check(lib.hlvae_decoder_fwd(self._plan_handle, ws, None, C.c_float(1.0), 0, int(want_params), 1, B, s), "decoder_fwd")
        self._fwd_token += 1

    def _run_backward(self, eps, g_lpx, g_mu, g_lv, B):
        """recompute the head kernel with the real upstream gradient, then the dense backward"""
        lib, ws, s = _lib.load(), C.byref(self._ws), self._stream()
        _lib.check(lib.hlvae_zero_grad(self._plan_handle, ws, s), "zero_grad")
        _lib.check(lib.hlvae_decoder_fwd(self._plan_handle, ws, _lib.ptr(g_lpx), C.c_float(0.0 if g_lpx is None else 1.0),
                                         1, 0, 0, B, s), "decoder_fwd(grad)")
        _lib.check(lib.hlvae_backward(self._plan_handle, ws, _lib.ptr(g_mu), _lib.ptr(g_lv), C.c_float(0.0), 0, B, s), "backward")
        self._grad_region_clean = False

    def _assign_grads(self):
        G = self._grad_arena
        for p, o in zip(self._order, self._offsets):
            if p.requires_grad and p is not self._disp_param:
                p.grad = G[o:o + p.numel()].view(p.shape)

    def _clone_outputs(self, B):
        t = self._ws_t
        return (t["mu"][:B].clone(), t["lv"][:B].clone(), t["z"][:B].clone(), t["log_p_x"][:B].clone(),
                t["log_p_x_missing"][:B].clone())

    # ------------------------------------------------------------------ p_params / p_samples
    def _block_columns(self):
        if self._block_cols is None:
            pti = np.asarray(self.types_info["param_indexes"])
            self._block_cols = [torch.as_tensor(np.nonzero(pti == i)[0], device=self.device)
                                for i in range(len(self.plan.blocks))]
        return self._block_cols

    def _p_params(self, B):
        """per-type parameter tensors in ``set_of_types`` order with the reference's shapes
        (cat/ordinal [B,n,K], real/pos/count [B,n]; HLVAE.py:411-412, loglik.py 'params')."""
        pf = self._ws_t["pfull"][:B]
        out = []
        for b, cols in zip(self.plan.blocks, self._block_columns()):
            p = pf.index_select(1, cols)
            if b["type"] in ("cat", "ordinal"):
                out.append(p.reshape(B, b["n_vars"], b["nclass"]))
            elif self.logvar_network and b["type"] in ("real", "pos"):
                out.append([p[:, :b["n_vars"]], p[:, b["n_vars"]:]])        # [est_mean, est_var] (loglik.py:64-67, 112-115)
            else:
                out.append(p)
        return out

    def _norm_params(self):
        """[[mean, var]_real, [mean, var]_pos] as batch_normalization returns them (HL_VAE/utils.py:108,132)."""
        nm, pl = self._ws_t["norm"], self.plan
        real = [nm[0, :pl.n_real].clone(), nm[1, :pl.n_real].clone()] if pl.n_real else []
        pos = [nm[0, pl.n_real:pl.n_real + pl.n_pos].clone(), nm[1, pl.n_real:pl.n_real + pl.n_pos].clone()] if pl.n_pos else []
        return [real, pos]

    def _p_samples(self, p_params, B):
        """Samples from the fitted likelihoods with the reference's semantics, including its quirks
        (loglik.py:59,68; 118-119; 141-142 softmax over the VARIABLE axis; 184-186; 211).  Plain torch
        ops on the GPU: sampling is not on the training hot path."""
        import torch.distributions as td
        nm, pl = self._ws_t["norm"], self.plan
        out = []
        for b, p in zip(pl.blocks, p_params):
            K = b["nclass"]
            if b["type"] == "real" and self.logvar_network:
                out.append(p[0] + torch.sqrt(p[1]) * torch.randn_like(p[0]))
            elif b["type"] == "pos" and self.logvar_network:
                out.append(torch.clamp(torch.exp(p[0] + torch.sqrt(p[1]) * torch.randn_like(p[0])) - 1.0, 0, 1e20))
            elif b["type"] == "real":
                var_d = torch.clamp(nm[1, :pl.n_real], min=3e-4)
                lvy = -8.0 + torch.nn.functional.softplus(self._log_vy_real.detach() + 8.0)
                out.append(p + torch.sqrt(var_d * torch.exp(lvy)) * torch.randn_like(p))
            elif b["type"] == "pos":
                var_d = torch.clamp(nm[1, pl.n_real:pl.n_real + pl.n_pos], min=1e-3)
                sd = torch.sqrt(var_d * torch.exp(self._log_vy_pos.detach()))
                out.append(torch.clamp(torch.exp(p + sd * torch.randn_like(p)) - 1.0, 0, 1e20))
            elif b["type"] == "count":
                out.append(torch.poisson(p))
            elif b["type"] == "cat":
                idx = td.Categorical(probs=torch.softmax(p, dim=1)).sample()
                out.append(torch.nn.functional.one_hot(idx, K).to(torch.float64))
            else:
                idx = td.Categorical(logits=torch.log(torch.clamp(p, 1e-6, 1e20))).sample()
                out.append((torch.arange(1, K + 1, device=p.device)[None, None, :] <= (1 + idx)[:, :, None]).to(torch.float64))
        return out

    def step_metrics(self, B: int):
        """Row M on the device: per-variable (error_observed, error_missing, error_all) of the last forward pass that
        was run with parameters materialised (reference training.py:84-101 -> read_functions.error_computation with
        true_miss_mask = 1).  Also returns the imputed values x_hat [B, D] (read_functions.statistics 'mean')."""
        err = torch.empty(3, self.plan.D, dtype=torch.float32, device=self.device)
        lib = _lib.load()
        _lib.check(lib.hlvae_step_metrics(self._plan_handle, C.byref(self._ws), B, _lib.ptr(err), self._stream()), "step_metrics")
        _lib.check(lib.hlvae_join(self._plan_handle, self._stream()), "join")
        return err[0], err[1], err[2], self._ws_t["xhat"][:B]

    # ------------------------------------------------------------------ reference call surface
    def sample_latent(self, mu, log_var):                                            # HLVAE.py:351-362
        std = torch.exp(0.5 * log_var)
        return mu + torch.randn_like(std) * std

    def loss_function(self, log_px):                                                 # HLVAE.py:377-379
        return -torch.sum(log_px, 1)

    def forward(self, data, mask, param_mask, types_info, do_test=False, eps=None):
        """HLVAE.forward (HLVAE.py:364-375).  ``param_mask`` is implied by ``mask`` (each variable's bit
        repeated over its parameter slots, read_functions.py:173-176) and is not read.
        ``eps`` (optional, [B, L]) fixes the reparameterisation noise; default: Philox4x32-10 normals generated
        inside the encoder kernel (seeded once from torch's global RNG; the reference uses randn_like, HLVAE.py:361)."""
        data, mask = self._prep_inputs(data, mask)
        B = data.shape[0]
        self._ensure_device_state(B)
        if eps is not None:
            eps = eps.to(device=self.device, dtype=torch.float32).contiguous()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._order)
        if need_grad:
            mu, lv, z, lpx, lpm = _Anchor.apply(self._anchor, self, data, mask, eps, B)
        else:
            self._run_forward(data, mask, eps, B, want_params=True)
            mu, lv, z, lpx, lpm = self._clone_outputs(B)
        p_params = {"x": self._p_params(B)}
        p_samples = {"x": self._p_samples(p_params["x"], B) if self.materialize_samples else None}
        q_samples = {"s": None, "z": z}
        q_params = {"s": None, "z": [mu, lv]}
        return p_samples, mu, lv, lpx, lpm, p_params, q_samples, q_params

    def encode(self, data, mask, param_mask, types_info, norm_params=None, X_list=None):
        """HLVAE.encode (HLVAE.py:284-324), inference only (the reference calls it under no_grad:
        training.py:176, HLVAE_main.py:194).  Batch statistics are always recomputed on device."""
        data, mask = self._prep_inputs(data, mask)
        B = data.shape[0]
        self._ensure_device_state(B)
        lib, ws, s = _lib.load(), C.byref(self._ws), self._stream()
        self._run_normalize(data, mask, B)
        self._run_encoder(None, self.training, B)        # the reference samples in encode() (HLVAE.py:321)
        self._fwd_token += 1
        t = self._ws_t
        mu, lv, z = t["mu"][:B].clone(), t["lv"][:B].clone(), t["z"][:B].clone()
        return {"s": None, "z": z}, {"s": None, "z": [mu, lv]}

    def _set_latent(self, z, B):
        t, L = self._ws_t, self.z_dim
        zf = z.detach().to(device=self.device, dtype=torch.float32)
        t["z"][:B].copy_(zf)
        t["zb"].zero_()
        t["zb"][:B, :L].copy_(zf)
        Bp = _ru(B, 128)
        zbT = t["zbT"].view(-1)[: self._dims.Lp * Bp].view(self._dims.Lp, Bp)
        zbT.zero_()
        zbT[:L, :B].copy_(zf.t())

    def decode(self, z, batch_x, miss_list, param_mask, norm_params=None):
        """HLVAE.decode (HLVAE.py:326-349), inference only -> (log_p_x, log_p_x_missing, p_samples, p_params)."""
        data, mask = self._prep_inputs(batch_x, miss_list)
        B = data.shape[0]
        self._ensure_device_state(B)
        self._run_normalize(data, mask, B)
        self._set_latent(z, B)
        self._run_decoder_only(B, want_params=True)
        t = self._ws_t
        p_params = {"x": self._p_params(B)}
        p_samples = {"x": self._p_samples(p_params["x"], B) if self.materialize_samples else None}
        return t["log_p_x"][:B].clone(), t["log_p_x_missing"][:B].clone(), p_samples, p_params

    def get_test_samples(self, data, miss_list, param_mask, data_list=None, X_list=None, norm_params=None, s=None):
        """HLVAE.get_test_samples (HLVAE.py:455-475): deterministic encode, decode(mean_qz)."""
        with torch.no_grad():
            data, mask = self._prep_inputs(data, miss_list)
            B = data.shape[0]
            self._ensure_device_state(B)
            self._run_forward(data, mask, None, B, want_params=True, sample=False)      # z = mu
            mu, lv, z, lpx, lpm = self._clone_outputs(B)
            p_params = {"x": self._p_params(B)}
            p_samples = {"x": self._p_samples(p_params["x"], B) if self.materialize_samples else None}
        return {"s": None, "z": z}, {"s": None, "z": [mu, lv]}, p_samples, p_params, lpx, lpm
