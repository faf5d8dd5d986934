"""The ELBO training step -- the hot loop of the reference's ``hensman_training``
(reference training.py:70-143) as one device-resident sequence of HIP launches:

    normalise/pack -> encoder -> reparameterise -> decoder + heads + log-lik (+ their backward)
    -> KL (+ its gradient) -> dense backward -> [RCCL all-reduce of the flat gradient arena]
    -> fused Adam (+ bf16 shadow refresh) -> [natural-gradient update of (m, H)]

No autograd graph, no host synchronisation: every scalar the reference pulls to the host with
``.item()`` (training.py:139-143) stays in a device buffer that the caller reads when it wants.
The whole step can be captured into a HIP graph (``capture=True``) and replayed.

``loss = P / P_batch * sum_b nll_b + KL``  (training.py:121-124), so the upstream gradient of
every log_p_x[b, d] is the scalar  -P / P_batch, which lets the head kernel emit dY in the same
pass as the forward.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional

import torch

from . import _lib
from .HLVAE import HLVAE


class FusedAdam:
    """torch.optim.Adam(lr=1e-3) semantics (reference HLVAE_main.py:277-278) on the model's flat arena:
    one HBM-streaming kernel for all parameters, fused with the bf16 shadow refresh."""

    def __init__(self, model: HLVAE, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        dev = model.device
        self.m1 = torch.zeros(model._arena_size, dtype=torch.float32, device=dev)
        self.m2 = torch.zeros(model._arena_size, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(2, dtype=torch.int64, device=dev)

    def step(self, grad_scale: float = 1.0):
        m = self.model
        lib = _lib.load()
        _lib.check(lib.hlvae_adam_step(m._plan_handle, C.byref(m._ws), _lib.ptr(self.m1), _lib.ptr(self.m2),
                                       _lib.ptr(self.step_count), C.c_float(self.lr), C.c_float(self.betas[0]),
                                       C.c_float(self.betas[1]), C.c_float(self.eps), C.c_float(grad_scale), m._stream()),
                   "hlvae_adam_step")
        m.mark_shadows_fresh()


class ShardedAdam:
    """The same optimiser for data-parallel runs (hlvae_amd.parallel): the dense part of the arena is cut into one slice per
    rank; after the reduce-scatter of the gradients each rank runs Adam on its slices only (csrc/optim.hip k_adam_flat),
    all-gathers a bf16 copy of the updated weights and rebuilds its padded shadows from it.  The small region (head
    parameters, biases, convolution weights) is all-reduced and updated on every rank.  World size 1 runs the same kernels
    with the collectives skipped."""

    def __init__(self, model: HLVAE, dp, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        from .HLVAE import GRAD_SLACK
        from .parallel import ShardPlan, ShardedState
        self.model, self.dp, self.lr, self.betas, self.eps = model, dp, lr, betas, eps
        dev, d = model.device, model._dims
        self.m1 = torch.zeros(model._arena_size, dtype=torch.float32, device=dev)
        self.m2 = torch.zeros(model._arena_size, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(2, dtype=torch.int64, device=dev)
        a0, end, o_wy = int(d.atomic_region), int(d.arena_size), int(d.o_wy)
        # slices in the order their gradients become final.  MLP: y_layer's weight (the last tensor of the arena) first --
        # its reduce-scatter overlaps the rest of the backward pass; the convolutional model's y_layer gradient is final only
        # after the transposed convolutions' backward, so its (smaller) dense region is one slice
        rest = 0x1e | sum(1 << (5 + i) for i in range(int(d.n_xe))) | sum(1 << (5 + _lib.MAX_EXTRA + j) for j in range(int(d.n_xd)))
        ranges = [(a0, end, 0x01 | rest)] if model.conv else [(o_wy, end, 0x01), (a0, o_wy, rest)]
        self.plan = ShardPlan(ranges, dp.world, dp.rank)
        if self.plan.pad > GRAD_SLACK:
            raise ValueError(f"world size {dp.world}: slice padding {self.plan.pad} exceeds the gradient arena's slack")
        # The gathered bf16 copy of the first encoder Linear [h_e][X] IS its row-major shadow when X needs no column padding
        # (D4: 5184 = 81 x 64): the MFMA kernels then read it where the all-gather puts it, and only the small matrices of that
        # slice are rebuilt.  The padding rows h_e .. hep-1 fall into the (zero) slack behind the copy.
        # (not with extra encoder layers: their backward pass also reads the TRANSPOSED shadow of that Linear, which is rebuilt
        # together with the row-major one)
        # (nor with extra decoder layers: the arena then continues [... W1 | extra decoder weights | Wy], and the padding rows of the
        #  aliased shadow would land on the bf16 copy of those weights instead of on zero slack)
        self._w1_alias = (not model.conv) and int(d.n_xe) == 0 and int(d.n_xd) == 0 and int(d.Xe) == int(d.Xep)
        slack = (int(d.hep) - int(d.h_e)) * int(d.Xep) + 64 if self._w1_alias else 0
        self.state = ShardedState(dp, self.plan, dev, slack=slack)
        if self._w1_alias:
            assert int(d.o_w1) + int(d.h_e) * int(d.Xe) == self.plan.slices[-1].hi, "the first encoder Linear must end its slice"
            self.plan.slices[-1].which &= ~0x02
            self._alias_w1(first=True)
        model._master_sync = self.sync_masters
        self._masters_stale = False
        self._pending = None            # (work handle of y_layer's all-gather, slice index) left running by the last step

    def _alias_w1(self, first=False):
        """point the workspace's first-Linear shadow at its place inside the gathered copy (again after a workspace re-allocation)"""
        m, d, k = self.model, self.model._dims, len(self.plan.slices) - 1
        off = int(d.o_w1) - self.plan.slices[k].lo
        alias = self.state.pb[k][off:off + int(d.hep) * int(d.Xep)]
        if m._ws.w1s == alias.data_ptr() and m._ws_alt.w1s == alias.data_ptr():
            return
        alias.copy_(m._ws_t["w1s"].reshape(-1))                           # current shadow (incl. its zero padding rows)
        m._ws.w1s = m._ws_alt.w1s = alias.data_ptr()
        m._ws_t["w1s"] = alias.view(int(d.hep), int(d.Xep))

    def _args(self):
        return (C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps), C.c_float(1.0))

    def reduce_scatter(self, k: int, async_op: bool = False):
        if self.plan.world == 1:          # the gradient arena itself is the "summed slice": no copy
            from .parallel import _Done
            return _Done()
        return self.state.reduce_scatter_slice(k, self.model._grad_arena, async_op=async_op)

    def step_slice(self, k: int):
        """Adam on this rank's part of slice k (its summed gradients are in the reduce-scatter's output buffer)"""
        m, lib = self.model, _lib.load()
        lo, hi = self.plan.slices[k].own(self.plan.rank)
        if hi > lo:
            g = self.state.gsh[k] if self.plan.world > 1 else m._grad_arena[lo:hi]
            _lib.check(lib.hlvae_adam_shard(m._plan_handle, C.byref(m._ws), _lib.ptr(g), _lib.ptr(self.m1),
                                            _lib.ptr(self.m2), _lib.ptr(self.state.own_copy_view(k)), _lib.ptr(self.step_count),
                                            lo, hi - lo, *self._args(), m._stream()), "hlvae_adam_shard")

    def step_small(self):
        """replicated Adam on the (all-reduced) small region; commits the step number: after every step_slice"""
        m = self.model
        _lib.check(_lib.load().hlvae_adam_small(m._plan_handle, C.byref(m._ws), _lib.ptr(self.m1), _lib.ptr(self.m2),
                                                _lib.ptr(self.step_count), *self._args(), m._stream()), "hlvae_adam_small")

    def gather(self, k: int, async_op: bool = False):
        return self.state.all_gather_slice(k, async_op=async_op)

    def rebuild_shadows(self, k: int):
        m, s = self.model, self.plan.slices[k]
        if self._w1_alias and k == len(self.plan.slices) - 1:
            self._alias_w1()
        if s.which:
            _lib.check(_lib.load().hlvae_shadows_from_bf16(m._plan_handle, C.byref(m._ws), _lib.ptr(self.state.pb[k]), s.lo, s.which,
                                                           m._stream()), "hlvae_shadows_from_bf16")
        self._masters_stale = self.plan.world > 1
        m.mark_shadows_fresh()

    def finish_pending(self):
        """y_layer's all-gather + shadow rebuild that the last step left running (they overlap the next step's first GEMM and
        fused middle: its shadows are first read by the head kernel): the current stream waits for them"""
        if self._pending is not None:
            h, k = self._pending
            self._pending = None
            if isinstance(h, torch.cuda.Stream):     # gather + rebuild already queued on that stream: wait for it
                torch.cuda.current_stream(self.model.device).wait_stream(h)
            else:
                h.wait()
                self.rebuild_shadows(k)

    def sync_masters(self):
        """collective: brings the fp32 masters of the other ranks' slices up to date (state_dict, checkpoints)"""
        self.finish_pending()
        if self._masters_stale:
            self.state.sync_masters(self.model._arena)
            self._masters_stale = False
            self.model.mark_shadows_fresh()


class ELBOTrainer:
    """One object per model: owns the optimiser state and (optionally) captured HIP graphs.

    kl = "normal": closed-form KL(q(z|x) || N(0, I)) -- NOT in the reference (its only KL is the GP prior,
                   SURVEY.md 0.3); this is the GP-free configuration of BASELINE.json configs 2-4.
    kl = "gp":     the reference's GP-prior KL; ``gp`` must be a hlvae_amd.elbo_functions.GPPrior.
    kl = None:     no KL term (reconstruction only).
    """

    def __init__(self, model: HLVAE, P_total: int, kl: Optional[str] = "normal", gp=None, lr: float = 1e-3,
                 max_batch: int = 512, dp=None, metrics: bool = False):
        if model.device.type != "cuda":
            raise RuntimeError("ELBOTrainer needs the model on the GPU (no CPU fallback)")
        self.model, self.P_total, self.kl, self.gp, self.dp, self.metrics = model, P_total, kl, gp, dp, metrics
        model._max_batch = max(model._max_batch, max_batch)
        model._ensure_device_state(max_batch)
        hidden_ids = {id(t) for t in tuple(getattr(model, "_id_enc", ())) + tuple(getattr(model, "_id_dec", ()))}   # identity layers of h_dim = []
        frozen = [p for p in model._order if not p.requires_grad and p is not model._log_vy_real and p is not model._log_vy_pos
                  and id(p) not in hidden_ids]
        if frozen or (model._log_vy_real is not None and model._log_vy_real.requires_grad != model._log_vy_pos.requires_grad):
            raise ValueError("the fused optimiser step freezes _log_vy_real / _log_vy_pos together (vy_fixed) and trains every other "
                             "parameter, as the reference does (HLVAE.py:209-216)")
        self.opt = FusedAdam(model, lr=lr) if dp is None else ShardedAdam(model, dp, lr=lr)
        if dp is not None and dp.world > 1:
            # a re-allocation rebuilds the bf16 shadows from the fp32 masters, which on this rank are current for its OWN slices
            # only; bringing them up to date is a collective, and ranks do not see the same batch sizes -- so the workspace must be
            # large enough from the start (max_batch >= the sampler's largest batch, folded tail included)
            model._grow_forbidden = ("data-parallel run: the workspace cannot grow after the trainer is built; construct ELBOTrainer "
                                     "with max_batch >= the largest batch of the sampler (datafeed.SubjectBatchSampler.max_rows)")
        dev, L = model.device, model.z_dim
        self._graphs = {}
        self._wy_dbuf = False        # y_layer shadows double-buffered (only inside an even chain of captured steps)
        self._gp_defer_noahead = os.environ.get("HL_GP_DEFER_AHEAD", "1") == "0"
        self._gp_defer = False       # GP state update deferred onto the prior's stream (only inside a captured chain, prepare() in use)
        if dp is not None and dp.world > 1:
            # every rank draws its own reparameterisation noise: the in-kernel Philox stream is indexed by the LOCAL row, so
            # the seed carries the rank (the reference's single process draws one randn_like for the whole batch, HLVAE.py:361)
            model._ws_t["rng"][0] = (int(model._ws_t["rng"][0].item()) + (dp.rank + 1) * 0x9E3779B97F4A7C15) % (2 ** 62)
        self._pf_stream = torch.cuda.Stream(device=dev)      # input stage of the NEXT batch (prefetch)
        self._wy_stream = torch.cuda.Stream(device=dev)      # data parallel: y_layer's weight gradient + the start of its reduce-scatter
        self.err = torch.zeros(3, model.plan.D, dtype=torch.float32, device=dev)     # error_observed / missing / all

    # -- the step, eager ------------------------------------------------------------------------
    @staticmethod
    def _batch_key(data, mask):
        return (data.data_ptr(), mask.data_ptr(), data._version, mask._version, tuple(data.shape))

    def prime(self, data: torch.Tensor, mask: torch.Tensor):
        """Run the input stage (statistics + normalise + pack, row A) of a batch now, so that the next
        ``step(data, mask, ...)`` (or a graph captured with ``prepacked=True``) finds it done."""
        m = self.model
        B = data.shape[0]
        m._ensure_device_state(B)
        m._run_normalize(data, mask, B, self.dp.allreduce_stats if self.dp is not None else None)
        m._packed_key = self._batch_key(data, mask)
        self._pf_ref = (data, mask)

    def step(self, data: torch.Tensor, mask: torch.Tensor, P_batch: int, eps: Optional[torch.Tensor] = None,
             train_x: Optional[torch.Tensor] = None, prefetch=None, prepacked: Optional[bool] = None):
        """data [B, X] fp64, mask [B, D] fp64 (already resident on the GPU), P_batch = subjects in the batch.

        prefetch = (next_data, next_mask): the input stage of the NEXT batch (it depends on the data only, not on the
        weights) runs on a side stream into the second buffer set while this batch trains; the next ``step`` on those
        tensors skips its own input stage.  prepacked: None = detect (same tensors, unmodified), True = trust the
        caller (HIP-graph capture: the decision is baked into the graph)."""
        m = self.model
        lib = _lib.load()
        B = data.shape[0]
        m._ensure_device_state(B)
        scale = float(self.P_total) / float(P_batch)
        hook = self.dp.allreduce_stats if self.dp is not None else None
        if prepacked is None:
            prepacked = m._packed_key is not None and m._packed_key == self._batch_key(data, mask)
        if not prepacked:
            m._run_normalize(data, mask, B, hook)
        self._step_core(B, scale, eps, train_x, P_batch, prefetch, hook)

    def prime_rows(self, ds, rows: torch.Tensor):
        """input stage of a batch of the compact dataset now, for a following ``step_rows(..., prepacked=True)``"""
        m = self.model
        B = rows.shape[0]
        m._ensure_device_state(B)
        self._feed_stage(ds, rows, B)
        if self._gp_ahead():
            self.gp.prime_ahead(ds.labels, rows)

    def _gp_ahead(self):
        """GP prior: the next batch's K0xz computed beside the state update's inversion (GPPriorHIP.compute_ahead; HL_GP_AHEAD=0
        switches it off)"""
        return self.kl == "gp" and hasattr(self.gp, "compute_ahead") and os.environ.get("HL_GP_AHEAD", "1") != "0" \
            and os.environ.get("HL_GP_PREPARE", "1") != "0"

    def step_rows(self, ds, rows: torch.Tensor, P_batch: int, eps: Optional[torch.Tensor] = None, groups=None,
                  prefetch_rows: Optional[torch.Tensor] = None, prepacked: bool = False):
        """One step on the rows ``rows`` (int32 device tensor) of a device-resident ``datafeed.DeviceDataset``: the input
        stage gathers from the compact form inside its kernels (csrc/feed.hip); nothing else crosses PCIe.  Capturable:
        refill ``rows`` in place and replay.

        prefetch_rows: the NEXT batch's rows.  Its input stage (statistics, normalise, pack: data only, no weights) is queued
        by the library on the side stream of this step's backward pass, into the second buffer set; the next call passes
        ``prepacked=True`` and starts at the first GEMM.  Single process, MLP model (the convolutional input stage reads
        the weights; data-parallel runs all-reduce the statistics between the two kernels)."""
        m = self.model
        lib = _lib.load()
        B = rows.shape[0]
        # the workspace is sized HERE for this batch and for the prefetched one (the sampler folds a short tail into the batch before
        # it, so a larger-than-nominal batch is a normal event): growing it in the middle of the step would hand the backward pass
        # the buffers the forward pass did not write
        if m._ensure_device_state(max(B, prefetch_rows.shape[0] if prefetch_rows is not None else 0)):
            prepacked = False            # the re-allocation dropped the batch a previous step had packed: its input stage runs again, here
        m._packed_key = None
        ws, s = C.byref(m._ws), m._stream()
        if prefetch_rows is not None and m.conv:
            raise ValueError("step_rows(prefetch_rows=...): the convolutional input stage reads the weights, it cannot run ahead")
        if not prepacked:
            self._feed_stage(ds, rows, B)
        train_x = None
        if self.kl == "gp":
            if hasattr(self.gp, "prepare") and os.environ.get("HL_GP_PREPARE", "1") != "0":
                # the prior's state-only launches (and the covariate gather) run on a stream of its own under the VAE's forward pass
                train_x = self.gp.prepare(ds.labels, rows, groups=groups, ahead=prepacked and self._gp_ahead() and not (self._gp_defer and self._gp_defer_noahead))
            else:
                train_x = ds.labels.index_select(0, rows.long())
        self._step_core(B, float(self.P_total) / float(P_batch), eps, train_x, P_batch, None, None,
                        feed_next=None if prefetch_rows is None else (ds, prefetch_rows), groups=groups)

    def _feed_stage(self, ds, rows, B):
        """statistics -> [all-reduce over ranks] -> normalise + pack of the rows ``rows`` on the current stream, into the
        front buffer set"""
        m, lib = self.model, _lib.load()
        ws, s = C.byref(m._ws), m._stream()
        if self.dp is None or self.dp.world == 1:
            _lib.check(lib.hlvae_feed_fused(m._plan_handle, ws, _lib.ptr(ds.values), _lib.ptr(ds.mask), _lib.ptr(rows), B, s), "feed_fused")
        else:
            _lib.check(lib.hlvae_feed_stats(m._plan_handle, ws, _lib.ptr(ds.values), _lib.ptr(ds.mask), _lib.ptr(rows), B, s), "feed_stats")
            self.dp.allreduce_stats(m._ws_t["sums"])
            _lib.check(lib.hlvae_feed_pack(m._plan_handle, ws, _lib.ptr(ds.values), _lib.ptr(ds.mask), _lib.ptr(rows), B, s), "feed_pack")

    def _step_core(self, B, scale, eps, train_x, P_batch, prefetch, hook, feed_next=None, groups=None):
        m = self.model
        lib = _lib.load()
        ws, s = C.byref(m._ws), m._stream()
        # forward (+ head backward in the same pass: upstream gradient of log_p_x is -scale).
        # eps None -> reparameterisation noise from the in-kernel Philox stream (device-side offset: graph safe)
        _lib.check(lib.hlvae_encoder_fwd(m._plan_handle, ws, _lib.ptr(eps), 1, C.c_uint64(0), B, s), "encoder_fwd")
        ev_enc = None
        if self.kl == "gp" and getattr(self.gp, "_early", False):
            # mu / log_var are final: the GP prior's per-subject kernel forks HERE (beside the head kernel), see kl_and_grads(after=)
            ev_enc = torch.cuda.Event()
            ev_enc.record(torch.cuda.current_stream(m.device))
        if prefetch is not None:
            # forked HERE: the HBM-streaming input stage of the next batch shares the machine with the compute-heavy head
            # kernel instead of the bandwidth/latency-bound first GEMM (measured: forking at the top of the step costs more
            # than it saves)
            main = torch.cuda.current_stream(m.device)
            self._pf_stream.wait_stream(main)
            m._swap_input_buffers()
            with torch.cuda.stream(self._pf_stream):
                m._run_normalize(prefetch[0], prefetch[1], prefetch[0].shape[0], hook)
            m._swap_input_buffers()
        if not m._grad_region_clean:     # normally the fused Adam leaves the atomically-accumulated region zeroed
            _lib.check(lib.hlvae_zero_grad(m._plan_handle, ws, s), "zero_grad")
        if self.dp is not None:
            self.opt.finish_pending()    # y_layer's shadows of the previous optimiser step (left gathering beside the code above)
        _lib.check(lib.hlvae_decoder_fwd(m._plan_handle, ws, None, C.c_float(-scale), 2, 2 if self.metrics else 0, 0, B, s), "decoder_fwd")     # want_grad = 2: ELBO scalars deferred to the backward's side stream
        if self.metrics:     # row M: imputed values + per-variable errors (training.py:84-101), device resident
            _lib.check(lib.hlvae_step_metrics(m._plan_handle, ws, B, _lib.ptr(self.err), s), "step_metrics")
        multi = self.dp is not None and self.dp.world > 1
        if feed_next is not None and not multi:   # deferred: the backward pass queues it on its side stream (hlvae_feed_prefetch)
            nds, nrows = feed_next
            m._require_capacity(nrows.shape[0])
            _lib.check(lib.hlvae_feed_prefetch(m._plan_handle, C.byref(m._ws_alt), _lib.ptr(nds.values), _lib.ptr(nds.mask),
                                               _lib.ptr(nrows), nrows.shape[0], s), "feed_prefetch")
            self._pf_ref = feed_next
        elif feed_next is not None:
            # several ranks: the next batch's statistics cross the ranks between its two kernels.  Forked from here on a stream
            # of ours, so that this small all-reduce is the FIRST collective of the step on RCCL's queue and the next step starts
            # at its first GEMM instead of behind a blocking exchange
            nds, nrows = feed_next
            m._require_capacity(nrows.shape[0])
            main = torch.cuda.current_stream(m.device)
            self._pf_stream.wait_stream(main)
            m._swap_input_buffers()
            with torch.cuda.stream(self._pf_stream):
                self._feed_stage(nds, nrows, nrows.shape[0])
            m._swap_input_buffers()
            self._pf_ref = feed_next
        g_mu = g_lv = None
        kl_w = 1.0 if self.kl == "normal" else 0.0
        if self.kl == "gp":
            # the GP prior's own chains (bound, natural gradient, hyper-parameter gradients) keep running on its streams beside
            # the VAE's backward pass; gp.optimizer_step() below joins them
            kw = {"join": False} if hasattr(self.gp, "join") else {}
            if ev_enc is not None:
                kw["after"] = ev_enc
            g_mu, g_lv = self.gp.kl_and_grads(m._ws_t["mu"][:B], m._ws_t["lv"][:B], train_x, self.P_total, P_batch, groups=groups, **kw)
        fused_opt = self.dp is None
        late_join = False
        if fused_opt:
            # backward and optimiser in one call: y_layer's Adam update runs under the rest of the backward pass
            o = self.opt
            dbuf = self._wy_dbuf and not m.conv and bool(lib.hlvae_backward_adam_fused(m._plan_handle, B))
            if dbuf:
                m._set_wy_double_buffer(True)
                ws = C.byref(m._ws)
            # GP prior: its state update (Adam, kernel matrix, the 85 us batched inversion) does not depend on the VAE's deferred side
            # chain (metrics, the next batch's input stage) -- that chain is joined BEHIND it (round-3 trace: the side chain shared a
            # hardware queue with the prior's chain C and ended 45 us after both GP chains; the state update waited for it)
            late_join = self.kl == "gp" and hasattr(self.gp, "join") and os.environ.get("HL_GP_LATE_JOIN", "1") != "0"
            if late_join:
                _lib.check(lib.hlvae_set_defer_join(m._plan_handle, 2 if self._gp_defer else 1), "set_defer_join")
            try:
                _lib.check(lib.hlvae_backward_adam(m._plan_handle, ws, _lib.ptr(g_mu), _lib.ptr(g_lv), C.c_float(kl_w), B, _lib.ptr(o.m1),
                                                   _lib.ptr(o.m2), _lib.ptr(o.step_count), C.c_float(o.lr), C.c_float(o.betas[0]),
                                                   C.c_float(o.betas[1]), C.c_float(o.eps), C.c_float(1.0), s), "backward_adam")
            finally:
                if late_join:
                    lib.hlvae_set_defer_join(m._plan_handle, 0)
                if dbuf:
                    m._set_wy_double_buffer(False)       # (also when the call failed: later eager steps must not write the spare pair)
            if dbuf:
                m._flip_wy_shadows()
            m.mark_shadows_fresh()
        else:
            # data parallel (hlvae_amd.parallel): reduce-scatter of the dense gradient slices -> Adam on this rank's slices ->
            # all-gather of the bf16 copies -> shadows; the small region is all-reduced and updated on every rank
            o = self.opt
            G, d = m._grad_arena, m._dims
            if m.conv:
                _lib.check(lib.hlvae_backward(m._plan_handle, ws, _lib.ptr(g_mu), _lib.ptr(g_lv), C.c_float(kl_w), 0, B, s), "backward")
                pend = [o.reduce_scatter(0, async_op=True)]
            else:
                # y_layer's gradient (the largest slice) is final first: its reduce-scatter runs on RCCL's stream while the
                # rest of the backward pass is still computing
                # (dWy itself runs on a stream of its own beside dU / the fused middle: on the caller's stream it was 21 us in front of them)
                main = torch.cuda.current_stream(m.device)
                self._wy_stream.wait_stream(main)
                with torch.cuda.stream(self._wy_stream):
                    _lib.check(lib.hlvae_backward_wy(m._plan_handle, ws, B, m._stream()), "backward_wy")
                    pend = [o.reduce_scatter(0, async_op=True)]
                _lib.check(lib.hlvae_backward(m._plan_handle, ws, _lib.ptr(g_mu), _lib.ptr(g_lv), C.c_float(kl_w), 1, B, s), "backward")
                pend.append(o.reduce_scatter(1, async_op=True))
                main.wait_stream(self._wy_stream)
            small = self.dp.allreduce_async(G[:int(d.atomic_region)])
            for k, h in enumerate(pend):        # slice k: wait for its sums, Adam on the owned part
                h.wait()
                o.step_slice(k)
            small.wait()
            o.step_small()
            # all-gathers of the bf16 copies, LAST slice first: it holds the first encoder Linear, which the next step reads at
            # once; y_layer's (slice 0, MLP) is first read by the head kernel, so its gather and shadow rebuild are left running
            # and the next step waits for them right before that kernel
            order = list(range(len(pend)))[::-1]
            gath = [(k, o.gather(k, async_op=True)) for k in order]
            for k, h in gath:
                if k == 0 and len(pend) > 1:
                    # the shadow rebuild too goes on the side stream (8 us in front of the next head kernel when the caller's
                    # stream did it in finish_pending)
                    side = self._wy_stream
                    side.wait_stream(torch.cuda.current_stream(m.device))
                    with torch.cuda.stream(side):
                        h.wait()
                        o.rebuild_shadows(k)
                    o._pending = (side, k)
                else:
                    h.wait()
                    o.rebuild_shadows(k)
            if not m.conv:      # hlvae_backward(skip_wy = 1) left the deferred side chain (metrics, next batch's input stage) running
                _lib.check(lib.hlvae_join(m._plan_handle, s), "join")
        m._fwd_token += 1
        m._grad_region_clean = True
        if self.kl == "gp":
            # (not together with the deferred state update, an experiment that stays off: HL_GP_DEFER)
            nb = {"next_batch": (feed_next[0].labels, feed_next[1])} if (feed_next is not None and self._gp_ahead() and not (self._gp_defer and self._gp_defer_noahead)) else {}
            if self._gp_defer and hasattr(self.gp, "join_tail"):
                self.gp.optimizer_step(defer=True, **nb)      # (inside a captured chain: the state update runs beside the next step's forward pass)
            else:
                self.gp.optimizer_step(**nb)
            if late_join:
                _lib.check(lib.hlvae_join(m._plan_handle, s), "join")
        if prefetch is not None:                 # join; the prefetched batch's buffers become the front set
            torch.cuda.current_stream(m.device).wait_stream(self._pf_stream)
            m._swap_input_buffers()
            m._packed_key = self._batch_key(prefetch[0], prefetch[1])
            self._pf_ref = prefetch          # keep the tensors alive: their addresses identify the packed batch
        elif feed_next is not None:              # joined by the backward call (or here); its buffers become the front set
            if multi:
                torch.cuda.current_stream(m.device).wait_stream(self._pf_stream)
            m._swap_input_buffers()
            m._packed_key = None
        else:
            m._packed_key = None
            self._pf_ref = None

    # -- captured -------------------------------------------------------------------------------
    def capture(self, key, data: torch.Tensor, mask: torch.Tensor, P_batch: int, train_x=None, prefetch=None):
        """Capture one step reading the given (static) input tensors into a HIP graph.

        With ``prefetch=(next_data, next_mask)`` the graph does NOT contain this batch's own input stage: it expects it
        done (by the previous graph of the chain, or by ``prime`` before the first replay) and runs the next batch's
        input stage as a parallel branch.  Chains must have an even number of graphs (two buffer sets)."""
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up on the side stream (allocations, lazy init)
                self.step(data, mask, P_batch, train_x=train_x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if prefetch is not None:
            self.prime(data, mask)
            torch.cuda.synchronize()
        with torch.cuda.graph(g, **self._capture_kw()):
            self.step(data, mask, P_batch, train_x=train_x, prefetch=prefetch, prepacked=(prefetch is not None) or None)
            if self.dp is not None:
                self.opt.finish_pending()
        self._graphs[key] = g
        return g

    def capture_rows(self, key, ds, rows, P_batch, next_rows=None, groups=None):
        """Capture ``step_rows`` reading the STATIC index tensor ``rows``: refill it in place (``rows.copy_(...)``) and
        replay -- one graph serves every batch of that size and subject count.

        groups: the batch's subject structure for the GP prior (datafeed.subject_index; a list for a chain), static like ``rows``.

        ``rows`` / ``P_batch`` may be LISTS: that many consecutive steps (one per index tensor) go into ONE graph.  Inside a
        graph the last kernel of a step and the first of the next are neighbours on one hardware queue (no gap); between
        two graph launches the executor joins and re-forks its queues (~6 us on MI355X).

        next_rows (a tensor, or a list for a chain): PIPELINED input stage -- the graph does not contain its batch's own
        input stage (``prime_rows`` before the first replay, the previous step of the chain afterwards) and every step runs
        the input stage of ``next_rows`` beside its backward pass.  The steps alternate between the two buffer sets, so a
        ring of such graphs must hold an even number of steps and be replayed in capture order."""
        many = isinstance(rows, (list, tuple))
        chain = list(zip(rows, P_batch)) if many else [(rows, P_batch)]
        nxt = [None] * len(chain) if next_rows is None else (list(next_rows) if many else [next_rows])
        grp = [None] * len(chain) if groups is None else (list(groups) if many else [groups])
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                       # warm-up (an even number of steps: the buffer sets end where they began)
                self.step_rows(ds, chain[0][0], chain[0][1], groups=grp[0])
            if next_rows is not None:
                self.prime_rows(ds, chain[0][0])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # y_layer's shadows alternate between two buffer pairs inside a chain with an even number of steps (the graph ends where
        # it began): the optimiser launch of a step need not wait for the step's last reader of the shadows.  Never in eager
        # steps: a graph captured earlier has the buffer it starts from baked in.
        # (below 2048 rows the library starts y_layer's launch behind dU_splitk anyway -- the fused middle must be resident first,
        #  csrc/cabi.hip -- and the second pair changes nothing; large batches use it)
        self._wy_dbuf = len(chain) % 2 == 0
        self._gp_defer = (self.kl == "gp" and self.dp is None and hasattr(self.gp, "join_tail") and os.environ.get("HL_GP_PREPARE", "1") != "0"
                          and os.environ.get("HL_GP_DEFER", "0") != "0")      # (measured: 0.834 vs 0.796 ms -- off by default)
        if self.kl == "gp" and hasattr(self.gp, "_defer_capture"):
            self.gp._defer_capture = self._gp_defer
        try:
            with torch.cuda.graph(g, **self._capture_kw()):
                for (r, pb), nr, gr in zip(chain, nxt, grp):
                    self.step_rows(ds, r, pb, prefetch_rows=nr, prepacked=nr is not None, groups=gr)
                if self.dp is not None:
                    self.opt.finish_pending()          # nothing may stay in flight across the end of a graph
                if self._gp_defer:
                    self.gp.join_tail()
                if self._gp_ahead():
                    self.gp.join_ahead()
        finally:
            self._wy_dbuf = False
            self._gp_defer = False
            if self.kl == "gp" and hasattr(self.gp, "_defer_capture"):
                self.gp._defer_capture = False
        self._graphs[key] = g
        return g

    def reset_after_failed_capture(self):
        """A capture that raised leaves host-side bookkeeping mid-step: y_layer's double-buffer pointers set, deferred side work
        recorded against events of the dead capture, an all-gather handle pending.  Bring everything back to the eager steady
        state (the next eager step then runs as if nothing had been captured)."""
        m = self.model
        self._wy_dbuf = False
        if not m.conv:
            m._set_wy_double_buffer(False)
        if hasattr(self.opt, "_pending"):
            self.opt._pending = None
        torch.cuda.synchronize()
        _lib.load().hlvae_reset_pending(m._plan_handle)
        m._packed_key = None
        m._grad_region_clean = False                 # a half-captured backward may or may not have consumed the atomic region
        m._sync_shadows(force=True)                  # shadows from the fp32 masters (whichever shadow pair is current)
        torch.cuda.synchronize()

    def _capture_kw(self):
        """several ranks: the process group's watchdog thread queries events while this thread captures -- only THIS thread's
        calls may be checked against the capture (torch's default mode would invalidate it)"""
        return {"capture_error_mode": "thread_local"} if (self.dp is not None and self.dp.world > 1) else {}

    def replay(self, key):
        self._graphs[key].replay()

    # -- device-resident scalars (the reference's .item() values, training.py:139-143) ----------
    def scalars(self):
        """device-resident scalars of the last step (no host sync: ``.item()`` them where the reference logs, training.py:139-143).
        With the GP prior, ``gp_not_spd`` is the device flag of its SPD inversions (non-zero: a covariance stopped being positive
        definite, where the reference's torch.cholesky raises at once); ``check()`` turns it into an exception."""
        sc = self.model._ws_t["scal"]
        out = {"nll_sum": sc[0], "kl": sc[1]}
        if self.gp is not None and hasattr(self.gp, "fail"):
            out["gp_not_spd"] = self.gp.fail
        return out

    def check(self):
        """host-side check at a logging / epoch boundary (one device read): raises if the GP prior met a non-positive pivot"""
        if self.gp is not None and hasattr(self.gp, "check"):
            self.gp.check()
