"""Roofline accounting for bench.py: algorithmic bytes / flops of every kernel of the step and a live
HIP-event measurement of each kernel's average duration (library facility hlvae_prof_enable /
hlvae_prof_report: events are recorded on the stream the kernel is launched on).

Per-unit figures, B = rows per step, P_n = parameters:
  F_fwd   = 2 (X h + h 2L + L h + h D y)  flop/row ;  F_train = 3 F_fwd - 2 X h
  Adam    = 28 B / parameter  (4 grad read + 3 x (4 read + 4 write) for master, m, v)
Peaks from MI355X_MICROARCH.md: HBM 8 TB/s, dense bf16 MFMA 2.5 PFLOP/s.
"""
from __future__ import annotations

import ctypes as C

from . import _lib

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_FP64_PEAK_TFLOPS = 78.6


def _ru(v, m):
    return (v + m - 1) // m * m


def gp_algorithmic_work(gp, B, S, T):
    """fp64 GP-prior kernels (csrc/gp.hip): bytes per launch; their arithmetic is fp64 vector work, not MFMA, so only
    the HBM side is priced.  spd_inv / kernel_matrix / param_grad are launched twice per step with different sizes:
    the figure is the mean of the two launches."""
    L, M, Q, n = gp.L, gp.M, gp.Q, gp.n_slots
    f = 8
    w = {}
    w["gp_transform"] = (4 * n * L * f, 0)
    w["gp_kernel_matrix"] = ((L * M * M + L * B * M + 2 * L * M * Q + B * Q) * f // 2, 0)
    w["gp_spd_inv"] = ((2 * L + L) * M * M * 2 * f // 2, 0)
    w["gp_subject_fwd"] = ((L * B * M * 2 + 2 * S * L * T * T + 3 * L * B) * f + 3 * B * L * 4, 0)
    w["gp_subject_bwd"] = ((2 * L * B * M + 2 * S * L * T * T + L * B) * f + B * L * 4, 0)
    w["gp_param_grad"] = ((L * B * M + L * M * M + 3 * L * M * Q + B * Q) * f // 2, 0)
    w["gp_bound"] = ((4 * L * M * M + 2 * L * M + S * L * 4) * f + B * L * 4, 0)
    w["gp_adam"] = ((n * L + L * M * Q) * 7 * f, 0)
    # fp64 matrix-core products of the M x M algebra: priced against the dense fp64 MFMA peak (MI355X: 78.6 TFLOP/s)
    w["gp_bmm"] = (4 * L * M * M * f, 2 * L * M * M * M, MFMA_FP64_PEAK_TFLOPS)
    # the rectangular products Y = V (iK - Q) and W = K0xz^T V (k_gp_gemm, two launches per step of the same size)
    w["gp_gemm"] = ((2 * L * B * M + L * M * M) * f, 2 * L * B * M * M, MFMA_FP64_PEAK_TFLOPS)
    # the M x M algebra behind W by row blocks (k_gp_chain_rb, both launches under one label): six 32 x M x M product passes per
    # row block; iK, W, H iK, H, iH in, grad_H, Rs (written, read back), G out
    w["gp_chain"] = (10 * L * M * M * f, 12 * L * M * M * M, MFMA_FP64_PEAK_TFLOPS)
    w["gp_rsym"] = (4 * L * M * M * f, 0)
    w["gp_gemv_t"] = (L * B * M * f, 0)
    w["gp_gkxz"] = (2 * L * B * M * f, 0)
    return w


def algorithmic_work(model, B, trainer_world=1):
    """label -> (bytes, flops) per launch (algorithmic: every operand read once, every result written once)."""
    d = model._dims
    Bp = _ru(B, 128)
    X, Xp, D, he, hep, hd, hdp, L, Lp, NY = d.X, d.Xp, d.D, d.h_e, d.hep, d.h_d, d.hdp, d.L, d.Lp, d.NY
    Xe, Xep, NYl, NYlp = d.Xe, d.Xep, d.NYl, d.NYlp          # first encoder Linear input / y_layer output (conv: 2592)
    S_e, S_d = model._ws.splitk_enc, model._ws.splitk_dec
    Pn = model._arena_size
    w = {}
    w["colstats"] = (B * d.n_stat * 16, 0)
    w["normalize_pack"] = (B * (X + D) * 8 + 2 * B * Xp * 2 + B * D * 5, 0)
    w["enc1_splitk"] = ((B * Xep + hep * Xep) * 2 + S_e * B * hep * 4, 2 * B * Xe * he)
    w["mid_fwd_fused"] = (S_e * B * hep * 4 + 2 * B * hep * 2 + 2 * Lp * hep * 2 + B * L * 16 + hdp * Lp * 2 + 2 * B * hdp * 2,
                          2 * B * he * 2 * L + 2 * B * L * hd)
    if not d.conv and d.K1p <= 256:      # narrow input: the fused middle computes Xn W1^T itself (csrc/mid.hip), no split-K slabs
        w["mid_fwd_fused"] = ((B * d.K1p + hep * d.K1p) * 2 + 2 * B * hep * 2 + 2 * Lp * hep * 2 + B * L * 16 + hdp * Lp * 2 + 2 * B * hdp * 2,
                              2 * B * d.K1 * he + 2 * B * he * 2 * L + 2 * B * L * hd)
    w["dec1_relu"] = ((B * Lp + hdp * Lp) * 2 + 2 * B * hdp * 2, 2 * B * L * hd)
    # U + Wy panels in, targets + mask in (5 B / entry), dY out in both layouts, log_p_x + log_p_x_missing + x_hat out
    w["y_heads_loglik"] = ((B * hdp + NY * hdp) * 2 + B * D * 5 + 2 * B * NY * 2 + 3 * B * D * 4, 2 * B * NY * hd + 150 * B * D)
    w["metrics_partial"] = (B * D * 9, 0)
    w["metrics_finish"] = (16 * 6 * D * 4, 0)
    w["elbo_finalize"] = (((D + 15) // 16) * B * 4, 0)
    w["dWy"] = ((NYl * Bp + hdp * Bp) * 2 + NYl * hd * 4, 2 * B * NYl * hd)
    w["dU_splitk"] = ((B * NYlp + hdp * NYlp) * 2 + S_d * B * hdp * 4, 2 * B * NYl * hd)
    w["mid_bwd_fused"] = (S_d * B * hdp * 4 + 2 * B * hdp * 2 + B * L * 16 + 2 * B * 2 * Lp * 2 + 2 * B * hep * 2,
                          2 * B * hd * L + 2 * B * 2 * L * he)
    if not d.conv and d.n_xd == 0 and NYlp <= 512:      # narrow y_layer: dY Wy inside the fused middle
        w["mid_bwd_fused"] = ((B * NYlp + hdp * NYlp) * 2 + 2 * B * hdp * 2 + B * L * 16 + 2 * B * 2 * Lp * 2 + 2 * B * hep * 2,
                              2 * B * NYl * hd + 2 * B * hd * L + 2 * B * 2 * L * he)
    w["dWd"] = ((hdp * Bp + Lp * Bp) * 2 + hd * L * 4, 2 * B * hd * L)
    w["dWmu_dWlv"] = ((2 * Lp * Bp + hep * Bp) * 2 + 2 * L * he * 4, 2 * B * 2 * L * he)
    w["dW1"] = ((hep * Bp + Xep * Bp) * 2 + he * Xe * 4, 2 * B * Xe * he)
    w["dW1_dWd_dWmu"] = tuple(w["dW1"][i] + w["dWd"][i] + w["dWmu_dWlv"][i] for i in range(2))      # one grouped launch
    # optimiser: 28 B per parameter (grad read, master / m / v read + write) plus the bf16 shadows it writes; since
    # hlvae_backward_adam y_layer's weight has its own early launch, the rest (and the small flat region) the final one
    n_wy, n_rest = NYl * hd, he * Xe + 2 * L * he + hd * L
    sh_wy = 2 * n_wy * 2
    sh_rest = ((2 if d.conv else 1) * he * Xe + 2 * (2 * L * he + hd * L)) * 2
    w["adam_wy_early"] = (n_wy * 28 + sh_wy, 0)
    # y_layer's weight gradient and optimiser step as one kernel: dY^T + U^T in, master / m / v in and out, both shadows out
    w["dWy_adam"] = ((NYl * Bp + hdp * Bp) * 2 + n_wy * 24 + sh_wy, 2 * B * NYl * hd)
    # the other three weight gradients + their optimiser step (one launch): operands, 24 B per parameter, shadows, the four biases
    w["dW1_dWd_dWmu_adam"] = (w["dW1_dWd_dWmu"][0] - (he * Xe + hd * L + 2 * L * he) * 4 + n_rest * 24 + sh_rest + (2 * L + he + hd) * 32,
                              w["dW1_dWd_dWmu"][1])
    w["adam_weights_shadows"] = (n_rest * 28 + sh_rest + model._atomic_region * 32, 0)
    w["adam_all_in_one"] = ((n_wy + n_rest) * 28 + sh_wy + sh_rest + model._atomic_region * 32, 0)     # data-parallel path
    w["shadow_cast"] = ((n_wy + n_rest) * 4 + sh_wy + sh_rest, 0)
    # sharded optimiser (data parallel; world 1 here unless the trainer says otherwise): 28 B + 2 B (bf16 copy) per owned
    # parameter, then the shadows from the flat bf16 copy
    world = trainer_world
    w["adam_flat_shard"] = ((n_wy + n_rest) * 30 // (2 * world), 0)          # two launches per step (one per slice): the mean
    w["adam_small"] = (model._atomic_region * 32, 0)
    w["shadows_wy"] = (n_wy * 2 + sh_wy, 0)
    w["shadows_rest"] = (n_rest * 2 + sh_rest, 0)
    if d.conv:      # csrc/conv.hip, per launch over the whole batch: activations in / out once; MACs x 2
        px = 36 * 36
        w["conv_enc_fwd"] = (B * (X + D) * 8 + B * px * 4 + 2 * B * Xe * 2 + B * D * 5, 2 * B * (px * 16 * 9 + 324 * 32 * 144))
        w["y_layer_conv"] = ((B * hdp + NYl * hdp) * 2 + B * NYl * 2, 2 * B * NYl * hd)
        w["convT1_fwd"] = (B * NYl * 2 + B * 5184 * 2, 2 * B * 324 * 16 * 128)
        w["convT2_fwd"] = (B * 5184 * 2 + B * NY * 4, 2 * B * px * 5 * 64)
        w["y_heads_loglik"] = (B * NY * 4 + B * D * 5 + B * NY * 2 + 2 * B * D * 4, 150 * B * D)
        w["convT2_bwd"] = (B * NY * 2 + 2 * B * 5184 * 2, 2 * 2 * B * px * 5 * 64)
        w["convT1_bwd"] = (B * 5184 * 2 + B * NYl * 2 + 2 * B * NYl * 2, 2 * 2 * B * 324 * 16 * 128)
        w["dfeat"] = ((B * hep + Xe * hep) * 2 + B * Xe * 4, 2 * B * Xe * he)
        w["conv_enc_bwd"] = (B * px * 4 + B * Xe * 4 + B * D * 5, 2 * B * (3 * 324 * 32 * 144 + 2 * px * 16 * 9))
        w["xn_transpose"] = (2 * B * Xe * 2, 0)
        w["dyc_transpose"] = (2 * B * NYl * 2, 0)
        w["conv_grad_finish"] = (512 * 21800 * 4 + B * D * 9, 0)
        w["adam_dense_early"] = (n_rest * 28 + sh_rest, 0)
        w["adam_small"] = (model._atomic_region * 32, 0)
        w["conv_pack_weights"] = (32256 * 2 + 15000 * 4, 0)
    return w


def measure_dominant_kernel(trainer, batch, steps, eager_steps=None, ds=None, P_batch=None):
    """Run `steps` eager steps with the library's per-kernel HIP events on, find the kernel with the
    largest total time and price it against its roofline.  Returns the bench.py 'roofline' object
    (plus the per-kernel table for DESIGN.md / profiles)."""
    lib = _lib.load()
    m = trainer.model
    n = steps if eager_steps is None else eager_steps
    n = max(5, min(n, 100))
    B = len(batch["rows"])
    P_batch = batch["P_batch"] if P_batch is None else P_batch
    import torch
    def one():
        if ds is not None:
            trainer.step_rows(ds, batch["rows_dev"], P_batch, groups=batch.get("groups_dev"))
        else:
            trainer.step(batch["data"], batch["mask"], P_batch, train_x=batch.get("labels"))

    # the GP prior's chains are Python-side streams: for this pass they collapse onto the caller's stream, like the library's own
    # side streams under hlvae_prof_enable -- every kernel is then timed ALONE (with the chains side by side the "alone" column
    # of a GP configuration was the contended duration)
    gp = getattr(trainer, "gp", None)
    serial_gp = gp is not None and hasattr(gp, "_serial") and not gp._serial

    def gp_streams(serial):
        torch.cuda.synchronize()
        gp._serial = serial
        gp._side = gp._prep_stream = gp._ahead_stream = None
        gp._prep = None

    if serial_gp:
        gp_streams(True)
    for _ in range(3):
        one()
    torch.cuda.synchronize()
    lib.hlvae_prof_enable(1)
    for _ in range(n):
        one()
    lib.hlvae_prof_enable(0)
    if serial_gp:
        gp_streams(False)
    buf = C.create_string_buffer(1 << 16)
    _lib.check(lib.hlvae_prof_report(buf, len(buf)), "hlvae_prof_report")
    work = algorithmic_work(m, B, trainer.dp.world if trainer.dp is not None else 1)
    if ds is not None:      # compact feed: 5 B per entry in instead of the expanded fp64 matrices
        d_ = m._dims
        work["normalize_pack"] = (B * d_.D * 5 + 2 * B * d_.Xp * 2 + B * d_.D * 5, 0)
        work["colstats"] = (B * d_.n_stat * 5, 0)
    if getattr(trainer, "gp", None) is not None:
        S, T = batch["groups"].shape
        work.update(gp_algorithmic_work(trainer.gp, B, S, T))
    table = {}
    for line in buf.value.decode().splitlines():
        name, cnt, tot = line.split()
        cnt, tot = int(cnt), float(tot)
        per_step_launches = cnt / n
        avg_us = 1e3 * tot / cnt
        wk = work.get(name, (0, 0))
        by, fl = wk[0], wk[1]
        table[name] = dict(launches_per_step=per_step_launches, avg_us=avg_us, us_per_step=avg_us * per_step_launches,
                           bytes=by, flops=fl, peak_tflops=wk[2] if len(wk) > 2 else MFMA_BF16_PEAK_TFLOPS)
    if not table:
        return None
    dom = max(table, key=lambda k: table[k]["us_per_step"])
    t = table[dom]
    t_hbm = t["bytes"] / (HBM_PEAK_GBS * 1e9)
    pk = t["peak_tflops"]
    t_mfma = t["flops"] / (pk * 1e12)
    if t_hbm >= t_mfma:
        ach = t["bytes"] / (t["avg_us"] * 1e-6) / 1e9
        roof = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None)
    else:
        ach = t["flops"] / (t["avg_us"] * 1e-6) / 1e12
        roof = dict(bound="mfma", achieved=ach, peak=pk, unit="TFLOP/s", frac=ach / pk, traffic=None)
    roof["kernel"] = dom
    roof["avg_us"] = t["avg_us"]
    roof["algorithmic_per_launch"] = t["bytes"] if roof["bound"] == "hbm" else t["flops"]
    roof["kernels_us_per_step"] = {k: round(v["us_per_step"], 2) for k, v in sorted(table.items(), key=lambda kv: -kv[1]["us_per_step"])}
    roof["sum_kernels_us_per_step"] = round(sum(v["us_per_step"] for v in table.values()), 2)
    return roof


# labels of the stamp slots (csrc/common.h HL_ST_*, include/hlvae_hip.h hlvae_stamp_buffer)
STAMP_LABELS = ("enc1_splitk", "mid_fwd_fused", "y_heads_loglik", "dU_splitk", "mid_bwd_fused", "dW1_dWd_dWmu_adam", "dWy_adam")


def measure_in_step(stamps, one_step, n=24):
    """Durations of the stamped kernels INSIDE the step as it is timed (the replayed HIP graph, side queues busy): arm the
    slots, run one step (`one_step()`: a single-step graph replay or an eager step), read {first workgroup start, last workgroup
    end} of every stamped kernel (10 ns ticks of s_memrealtime).  Returns label -> {"start_us" (from the step's first stamped
    kernel), "dur_us"} medians over n steps, or None when nothing was stamped."""
    import numpy as np
    import torch
    nk = len(STAMP_LABELS)
    sub = stamps.numel() // (8 * nk)                 # sub-slots per kernel (csrc/common.h HL_STAMP_SUB), 8 words each
    arm = torch.zeros_like(stamps).view(nk, sub, 8)
    arm[:, :, 0] = -1                                # ~0 as uint64: armed; word 1 (max end) 0
    arm = arm.view(-1)
    rec = []
    for _ in range(n):
        stamps.copy_(arm)
        torch.cuda.synchronize()
        one_step()
        torch.cuda.synchronize()
        v = stamps.cpu().numpy().astype(np.uint64).reshape(nk, sub, 8)
        armed = np.uint64(0xFFFFFFFFFFFFFFFF)
        row = []
        for k in range(nk):
            st = v[k, :, 0]
            en = v[k, :, 1]
            st = st[st != armed]
            en = en[en != 0]
            row.append((int(st.min()), int(en.max())) if len(st) and len(en) else None)
        rec.append(row)
    stamps.zero_()                                   # disarmed again
    torch.cuda.synchronize()
    out = {}
    for k in range(nk):
        st, du = [], []
        for row in rec:
            live = [r[0] for r in row if r is not None]
            if row[k] is None or not live:
                continue
            t0 = min(live)
            st.append((row[k][0] - t0) * 0.01)
            du.append((row[k][1] - row[k][0]) * 0.01)
        if du:
            out[STAMP_LABELS[k]] = {"start_us": round(float(np.median(st)), 2), "dur_us": round(float(np.median(du)), 2)}
    return out or None


def add_in_step(roof, in_step, model, ms_per_step):
    """roofline.in_step: the dominant kernel priced on its duration inside the timed step (beside the kernel-alone figure of the
    eager pass); roofline.step: the whole step on SURVEY.md 8(d)'s algorithmic bytes (32 B per parameter: bf16 weights read by
    the forward and the backward pass, fp32 gradient, master / m / v read and written)."""
    pn = int(model._arena_size)
    step_bytes = 32 * pn
    ach = step_bytes / (ms_per_step * 1e-3) / 1e9
    roof["step"] = {"algorithmic_bytes": step_bytes, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "note": "SURVEY.md 8(d): 32 B x parameters per step, independent of the batch"}
    if not in_step:
        return
    roof["in_step_timeline_us"] = in_step
    k = roof.get("kernel")
    if k in in_step and in_step[k]["dur_us"] > 0:
        dur = in_step[k]["dur_us"]
        if roof["bound"] == "hbm":
            a = roof["algorithmic_per_launch"] / (dur * 1e-6) / 1e9
        else:
            a = roof["algorithmic_per_launch"] / (dur * 1e-6) / 1e12
        roof["in_step"] = {"kernel": k, "avg_us": dur, "achieved": a, "frac": a / roof["peak"],
                           "how": "first-workgroup start to last-workgroup end (s_memrealtime stamps) inside the replayed step"}
        roof["frac_in_step"] = a / roof["peak"]
