"""GPU parity tests: the HIP path (through the C ABI, via the drop-in HLVAE class) against
  (1) golden fixtures produced by the reference itself (tests/golden/*.npz),
  (2) the fp64 CPU oracle on the same seeded inputs,
  (3) size-independent properties at BASELINE sizes.

Tolerances (bf16 MFMA inputs, fp32 accumulation, fp32 heads/ELBO; reference is fp64):
  * ELBO / NLL totals:        1e-4 relative   (BASELINE.json north_star)
  * per-element log_p_x:      3e-2 absolute + 2e-2 relative (bf16 rounding of y, amplified by 1/var in the real head)
  * mu / log_var:             2e-2 absolute
  * gradients, per tensor:    2.5e-2 relative L2 (bf16 activations and weights in both backward GEMMs: the first encoder
                              Linear of the D4 model measures 2.2e-2 at 512 rows -- ReLU gates that flip under bf16 rounding --,
                              every other tensor <= 1e-2)
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import hlvae_amd                      # noqa: E402
from hlvae_amd import synthetic       # noqa: E402
from tests_common import MIX_SPEC, load_mix_case, max_abs_err, rel_err   # noqa: E402

ELBO_RTOL = 1e-4
GRAD_RTOL = 2.5e-2
FLIP_TOL = 1.2e-2       # share of a tensor's cells whose first Adam updates differ by more than lr (a tiny gradient changed sign under
                        # bf16 rounding): 3 x the 4e-3 measured on the dense tensors (gpurun_out/parity_report*.json)
GP_STATE_TOL = 1e-2     # m_new / H_new after one natural-gradient step from the device's (bf16-product) mu / log_var
REPORT = {}


def _report(key, **kv):
    REPORT.setdefault(key, {}).update({k: float(v) for k, v in kv.items()})
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _model_from_state(src, dims, state, max_batch=128):
    from hlvae_amd.HLVAE import HLVAE
    m = HLVAE(dims, src.types_info, src.n_variables, vy_init=[1.0, 0.5], conv=False, max_batch=max_batch,
              materialize_samples=False)
    m.load_state_dict({k: v for k, v in state.items()})
    return m.to(_dev())


def test_library_loaded_and_fails_loudly_on_cpu():
    from hlvae_amd import _lib
    from hlvae_amd.HLVAE import HLVAE
    assert os.path.exists(_lib.LIB_PATH)
    _lib.load()
    src = synthetic.make_tabular(n_rows=8, T=4, seed=1, spec=MIX_SPEC)
    m = HLVAE([src.cov_dim_ext, [16], 4, [16], 5], src.types_info, src.n_variables, conv=False)
    with pytest.raises(RuntimeError):
        m(torch.tensor(src.data), torch.tensor(src.mask), torch.tensor(src.param_mask), src.types_info)


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (128, 64, 128), (70, 50, 96), (500, 5184 // 8, 128), (6480 // 4, 500, 256),
                                   (512, 32, 512), (64, 500, 128)])
def test_gemm_nt_against_torch(M, N, K):
    """asymmetric random operands; fp32 torch matmul of the SAME bf16-rounded values is the reference"""
    from hlvae_amd import _lib
    import ctypes as C
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    B = (torch.randn(N, K, generator=g) + 0.3).to(dev).to(torch.bfloat16)
    Cc = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32)
    lib = _lib.load()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.hlvae_gemm_nt_f32(_lib.ptr(A), K, _lib.ptr(B), K, _lib.ptr(Cc), N, M, N, K, s), "gemm")
    torch.cuda.synchronize()
    ref = A.float() @ B.float().t()
    err = (Cc - ref).abs().max().item()
    assert err < 1e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("name", ["mix_init", "mix_trained"])
def test_forward_backward_against_reference_fixture(golden_dir, name):
    g, src, dims, state = load_mix_case(golden_dir, name)
    dev = _dev()
    model = _model_from_state(src, dims, state)
    data, mask = torch.tensor(g["data"], device=dev), torch.tensor(g["mask"], device=dev)
    pmask = torch.tensor(src.param_mask, device=dev)
    eps = torch.tensor(g["eps"], device=dev)
    p_samples, mu, lv, lpx, lpm, p_params, q_samples, q_params = model(data, mask, pmask, src.types_info, eps=eps)
    torch.cuda.synchronize()
    e_mu, e_lv = max_abs_err(mu.cpu(), g["mu"]), max_abs_err(lv.cpu(), g["log_var"])
    e_lpx = np.abs(lpx.detach().double().cpu().numpy() - g["log_p_x"])
    e_lpm = np.abs(lpm.detach().double().cpu().numpy() - g["log_p_x_missing"])
    elbo = float(lpx.double().sum())
    elbo_ref = float(g["log_p_x"].sum())
    _report(name, mu=e_mu, lv=e_lv, lpx_max=e_lpx.max(), lpm_max=e_lpm.max(), elbo_rel=abs(elbo - elbo_ref) / abs(elbo_ref))
    assert e_mu < 2e-2 and e_lv < 2e-2
    assert np.all(e_lpx <= 3e-2 + 2e-2 * np.abs(g["log_p_x"]))
    assert np.all(e_lpm <= 3e-2 + 2e-2 * np.abs(g["log_p_x_missing"]))
    assert abs(elbo - elbo_ref) <= ELBO_RTOL * abs(elbo_ref)          # measured 3e-7 / 7e-6 on the two fixtures
    # p_params per type block, reference shapes
    for i, p in enumerate(p_params["x"]):
        ref = g[f"p_params_{i}"]
        assert tuple(p.shape) == ref.shape
        assert max_abs_err(p.cpu(), ref) <= 3e-2 + 2e-2 * np.abs(ref).max(), i
    # same loss as the fixture: scale * sum(nll) + KL_std(mu, lv)
    nll = model.loss_function(lpx)
    kl = -0.5 * torch.sum(1.0 + lv - mu ** 2 - torch.exp(lv))
    loss = float(g["nll_scale"][0]) * nll.sum() + kl
    loss.backward()
    torch.cuda.synchronize()
    kl_ref = -0.5 * float(np.sum(1.0 + g["log_var"] - g["mu"] ** 2 - np.exp(g["log_var"])))
    _report(name + "_loss", loss_rel=abs(float(loss) - float(g["loss"][0])) / abs(float(g["loss"][0])), kl_abs=abs(float(kl) - kl_ref),
            kl_ref=kl_ref, nll_abs=abs(float(g["nll_scale"][0]) * (float(nll.sum()) + elbo_ref)))
    # the KL term is computed from mu / log_var, which carry the bf16 operand rounding of the encoder (6e-3 / 8e-3 abs on the
    # trained fixture): |d KL| ~ sum |mu| |d mu| = 0.076 of 81.5 there, 1.8e-4 of the loss (4.8e-7 on the initial-weights fixture)
    assert abs(float(loss) - float(g["loss"][0])) <= 5e-4 * abs(float(g["loss"][0]))
    worst = 0.0
    sd = dict(model.named_parameters())
    for k in g.files:
        if not k.startswith("grad__"):
            continue
        pname = k[len("grad__"):]
        gr = sd[pname].grad
        assert gr is not None, pname
        e = rel_err(gr.double().cpu().numpy(), g[k])
        _report(name + "_grads", **{pname: e})
        worst = max(worst, e)
        assert e < GRAD_RTOL, (pname, e)
    assert worst > 0


def test_test_samples_and_metrics_against_fixture(golden_dir):
    g, src, dims, state = load_mix_case(golden_dir, "mix_trained")
    dev = _dev()
    model = _model_from_state(src, dims, state)
    data, mask = torch.tensor(g["data"], device=dev), torch.tensor(g["mask"], device=dev)
    qs, qp, ps, pp, lpt, lpmt = model.get_test_samples(data, mask, None)
    assert max_abs_err(qp["z"][0].cpu(), g["test_mu"]) < 2e-2
    assert np.all(np.abs(lpt.detach().double().cpu().numpy() - g["test_log_p_x"]) <= 3e-2 + 2e-2 * np.abs(g["test_log_p_x"]))
    assert torch.equal(qs["z"], qp["z"][0])          # decode(mean_qz): no noise (HLVAE.py:472)
    # imputed values (row M): x_hat of the deterministic pass agrees with statistics() on the oracle params
    xhat = model._ws_t["xhat"][:24].double().cpu().numpy()
    import hlvae_oracle as orc
    import metrics_oracle as mo
    st = {k: v for k, v in state.items()}
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    ts = om.test_samples(torch.tensor(g["data"]), torch.tensor(g["mask"]))
    xh_ref, _, _, _ = mo.step_metrics(ts["p_params"], torch.tensor(g["data"]), torch.tensor(g["mask"]), src.types_info,
                                      st["_log_vy_pos"])
    xh_ref = xh_ref.numpy()
    disc = np.isin(model.plan.kind, [3, 4])
    assert np.mean(xhat[:, disc] == xh_ref[:, disc]) > 0.97       # argmax may flip on near-ties
    assert np.allclose(xhat[:, ~disc], xh_ref[:, ~disc], rtol=3e-2, atol=3e-2)


def _oracle_step(src, rows, dims, state, eps, scale):
    import hlvae_oracle as orc
    st = {k: v.double().clone().requires_grad_(True) for k, v in state.items() if not k.startswith("hidden.")}
    for k in list(st):
        if k.startswith("d_layers."):
            st["hidden." + k[len("d_layers."):]] = st[k]
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    out = om.forward(torch.tensor(src.data[rows]), torch.tensor(src.mask[rows]), eps.double().cpu())
    loss = scale * om.loss_function(out["log_p_x"]).sum() + orc.standard_normal_kl(out["mu"], out["log_var"])
    loss.backward()
    return out, loss, st


def test_d4_batch512_against_oracle():
    """BASELINE config 2 shape: D4 layout (324 real + 972 cat5), hidden 500, latent 32, batch 512."""
    dev = _dev()
    src = synthetic.make_d4(n_subjects=26, T=20, seed=100)
    rows = np.arange(512)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    from hlvae_amd.HLVAE import HLVAE
    torch.manual_seed(1234)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=512, materialize_samples=False).to(dev)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    eps = torch.randn(512, 32, generator=torch.Generator().manual_seed(5))
    data = torch.tensor(src.data[rows], device=dev)
    mask = torch.tensor(src.mask[rows], device=dev)
    out = model(data, mask, None, src.types_info, eps=eps.to(dev))
    mu, lv, lpx = out[1], out[2], out[3]
    scale = 200.0 / 26.0
    loss = scale * model.loss_function(lpx).sum() - 0.5 * torch.sum(1.0 + lv - mu ** 2 - torch.exp(lv))
    loss.backward()
    torch.cuda.synchronize()
    ref, ref_loss, st = _oracle_step(src, rows, dims, state, eps, scale)
    elbo, elbo_ref = float(lpx.double().sum()), float(ref["log_p_x"].sum())
    rel = abs(elbo - elbo_ref) / abs(elbo_ref)
    _report("d4_b512", elbo_rel=rel, loss_rel=abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)),
            mu=max_abs_err(mu.cpu(), ref["mu"].detach()), lv=max_abs_err(lv.cpu(), ref["log_var"].detach()),
            lpx_max=max_abs_err(lpx.cpu(), ref["log_p_x"].detach()))
    assert rel <= ELBO_RTOL, rel
    assert abs(float(loss) - float(ref_loss)) <= ELBO_RTOL * abs(float(ref_loss))
    sd = dict(model.named_parameters())
    for k, p in sd.items():
        if p.grad is None or st[k].grad is None:
            assert k == "_disp_param" or p.numel() == 0, k      # unused dispersion parameter / empty type block
            continue
        e = rel_err(p.grad.double().cpu().numpy(), st[k].grad.numpy())
        _report("d4_b512_grads", **{k: e})
        assert e < GRAD_RTOL, (k, e)


def test_properties_at_full_size():
    """size-independent properties (no oracle needed): masked-out entries carry no gradient, the
    gradient is linear in the upstream scale, rows are independent given the batch statistics."""
    dev = _dev()
    src = synthetic.make_d4(n_subjects=26, T=20, seed=7)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    from hlvae_amd.HLVAE import HLVAE
    torch.manual_seed(3)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=512, materialize_samples=False).to(dev)
    data = torch.tensor(src.data[:512], device=dev)
    mask = torch.tensor(src.mask[:512], device=dev)
    eps = torch.randn(512, 32, device=dev)

    def grads(scale):
        out = model(data, mask, None, src.types_info, eps=eps)
        (scale * model.loss_function(out[3]).sum()).backward()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, out

    g1, out1 = grads(1.0)
    g3, _ = grads(3.0)
    for k in g1:
        assert rel_err(g3[k].cpu().numpy(), 3.0 * g1[k].cpu().numpy()) < 2e-2, k
    lpx, lpm = out1[3], out1[4]
    m = mask.bool()
    assert float(lpx[~m].abs().max()) == 0.0 and float(lpm[m].abs().max()) == 0.0     # HLVAE.py:409-410
    assert torch.isfinite(lpx).all() and torch.isfinite(lpm).all()
    # all-missing row block: zero observed log-likelihood and no NaN (mask = 0 for a whole row)
    mask2 = mask.clone()
    mask2[5] = 0
    out2 = model(data, mask2, None, src.types_info, eps=eps)
    assert float(out2[3][5].abs().max()) == 0.0 and torch.isfinite(out2[3]).all()


def test_training_steps_against_oracle(golden_dir):
    """three fused training steps (forward + backward + KL(q||N(0,I)) + Adam, no autograd) against the oracle running the
    same sequence in fp64 with the same noise: the NLL trajectory and the parameters after the steps."""
    import hlvae_oracle as orc
    from hlvae_amd.training import ELBOTrainer
    g, src, dims, state = load_mix_case(golden_dir, "mix_trained")
    dev = _dev()
    model = _model_from_state(src, dims, state)
    P_total, P_batch = 40, 4
    tr = ELBOTrainer(model, P_total=P_total, kl="normal", max_batch=128)
    data, mask = torch.tensor(g["data"], device=dev), torch.tensor(g["mask"], device=dev)
    gen = torch.Generator().manual_seed(77)
    eps_seq = [torch.randn(24, dims[2], generator=gen) for _ in range(3)]
    nll_gpu = []
    for e in eps_seq:
        tr.step(data, mask, P_batch, eps=e.to(dev))
        nll_gpu.append(float(tr.scalars()["nll_sum"]))
        kl_gpu = float(tr.scalars()["kl"])
    # oracle
    st = {k: v.double().clone().requires_grad_(True) for k, v in state.items() if not k.startswith("hidden.")}
    for k in list(st):
        if k.startswith("d_layers."):
            st["hidden." + k[len("d_layers."):]] = st[k]
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    names = [k for k in st if not k.startswith("hidden.") and k != "_disp_param"]
    params = [st[k] for k in names]
    m1, m2 = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    nll_ref = []
    for it, e in enumerate(eps_seq):
        for p in params:
            p.grad = None
        out = om.forward(torch.tensor(g["data"]), torch.tensor(g["mask"]), e.double())
        nll = om.loss_function(out["log_p_x"]).sum()
        kl = orc.standard_normal_kl(out["mu"], out["log_var"])
        (nll * P_total / P_batch + kl).backward()
        orc.adam_step(params, [p.grad for p in params], m1, m2, it + 1)
        nll_ref.append(float(nll))
        kl_ref = float(kl)
    _report("training_steps", **{f"nll_rel_{i}": abs(a - b) / abs(b) for i, (a, b) in enumerate(zip(nll_gpu, nll_ref))},
            kl_rel=abs(kl_gpu - kl_ref) / abs(kl_ref))
    for a, b in zip(nll_gpu, nll_ref):
        assert abs(a - b) <= ELBO_RTOL * abs(b), (nll_gpu, nll_ref)
    # named exception: the KL of the TRAINED fixture (81.5) is a function of mu / log_var, which leave the encoder through two bf16
    # products (measured 1e-3 of its value)
    assert abs(kl_gpu - kl_ref) <= 5e-3 * abs(kl_ref) + 1e-3
    sd = dict(model.named_parameters())
    for k, p in zip(names, params):
        delta_ref = (p.detach() - state[k].double()).numpy()
        delta = (sd[k].detach().double().cpu() - state[k].double()).numpy()
        if delta_ref.size == 0:
            continue
        # Adam's first steps move every parameter by about lr * sign(g): compare the updates, tolerating the
        # few entries whose tiny gradient changes sign under bf16 rounding
        bad = np.abs(delta - delta_ref) > 1e-3
        _report("training_steps", **{"flip__" + k: float(bad.mean())})
        assert bad.sum() <= max(1, int(FLIP_TOL * bad.size)), (k, bad.mean())       # (one cell of a 64-cell tensor is 1.6 %)


def test_input_stage_prefetch_matches_serial(golden_dir):
    """ELBOTrainer.step(prefetch=next batch) -- the next batch's statistics/normalise/pack on a side stream into the second
    buffer set -- must give the same trajectory as running every input stage inside its own step: eager chain and a chain
    of captured HIP graphs, two alternating batches (only the atomically accumulated head gradients may differ in the
    last bit -> 1e-6)."""
    from hlvae_amd.training import ELBOTrainer
    g, src, dims, state = load_mix_case(golden_dir, "mix_trained")
    dev = _dev()
    data0, mask0 = torch.tensor(g["data"], device=dev), torch.tensor(g["mask"], device=dev)
    perm = torch.randperm(24, generator=torch.Generator().manual_seed(5)).to(dev)
    keep = (torch.rand(24, mask0.shape[1], generator=torch.Generator().manual_seed(6)) > 0.1).to(dev)
    data1, mask1 = data0[perm].contiguous(), (mask0[perm] * keep).contiguous()
    batches = [(data0, mask0), (data1, mask1)]
    eps = [torch.randn(24, dims[2], generator=torch.Generator().manual_seed(11 + i)).to(dev) for i in range(6)]

    def run(prefetch):
        model = _model_from_state(src, dims, state)
        tr = ELBOTrainer(model, P_total=40, kl="normal", max_batch=128)
        nll = []
        for i in range(6):
            d, m = batches[i % 2]
            tr.step(d, m, 4, eps=eps[i], prefetch=batches[(i + 1) % 2] if prefetch else None)
            nll.append(float(tr.scalars()["nll_sum"]))
        return nll, model._arena.clone()

    nll_a, P_a = run(False)
    nll_b, P_b = run(True)
    assert rel_err(np.array(nll_a), np.array(nll_b)) < 1e-6, (nll_a, nll_b)
    assert rel_err(P_a, P_b) < 1e-6
    # captured chain with prefetch against the same steps run eagerly without it, from the same parameters, optimiser
    # state and noise stream (the in-kernel Philox offset lives on the device)
    model = _model_from_state(src, dims, state)
    tr = ELBOTrainer(model, P_total=40, kl="normal", max_batch=128)
    P0 = model._arena.clone()
    for i, (d, m) in enumerate(batches):
        tr.capture(i, d, m, 4, prefetch=batches[(i + 1) % 2])

    def reset():
        model._arena.copy_(P0)
        model._sync_shadows(force=True)
        tr.opt.m1.zero_(); tr.opt.m2.zero_(); tr.opt.step_count.zero_()

    reset()
    rng0 = model._ws_t["rng"].clone()
    tr.prime(*batches[0])
    nll_g = []
    for i in range(4):
        tr.replay(i % 2)
        nll_g.append(float(tr.scalars()["nll_sum"]))
    P_g = model._arena.clone()
    reset()
    model._ws_t["rng"].copy_(rng0)
    nll_e = []
    for i in range(4):
        tr.step(*batches[i % 2], 4)
        nll_e.append(float(tr.scalars()["nll_sum"]))
    assert rel_err(np.array(nll_g), np.array(nll_e)) < 1e-6, (nll_g, nll_e)
    assert rel_err(P_g, model._arena) < 1e-6


def test_inkernel_noise_statistics():
    """Philox normals generated in the encoder kernel: mean 0, variance 1, different every step."""
    from hlvae_amd.training import ELBOTrainer
    dev = _dev()
    src = synthetic.make_tabular(n_rows=512, T=16, seed=3)
    dims = [src.cov_dim_ext, [64], 32, [64], 5]
    from hlvae_amd.HLVAE import HLVAE
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=512, materialize_samples=False).to(dev)
    tr = ELBOTrainer(model, P_total=32, kl="normal", max_batch=512)
    data, mask = torch.tensor(src.data, device=dev), torch.tensor(src.mask, device=dev)
    tr.step(data, mask, 32)
    e1 = model._ws_t["eps"][:512].clone()
    tr.step(data, mask, 32)
    e2 = model._ws_t["eps"][:512].clone()
    assert abs(float(e1.mean())) < 0.03 and abs(float(e1.var()) - 1.0) < 0.05
    assert float((e1 - e2).abs().max()) > 0.5
    assert abs(float((e1 * e2).mean())) < 0.03


@pytest.mark.parametrize("impl", ["torch", "hip"])
def test_gp_prior_training_step_against_oracle(golden_dir, impl):
    """row K on the device: one fused step with the GP-prior KL (batched fp64 torch-ROCm ops feeding g_mu / g_lv into the
    HIP backward) against the oracles: NLL, KL value, and the direction of the first Adam update."""
    import hlvae_oracle as orc
    import gp_oracle as gpo
    from hlvae_amd.elbo_functions import GPPrior, GPPriorHIP
    from hlvae_amd.training import ELBOTrainer
    dev = _dev()
    src = synthetic.make_tabular(n_rows=48, T=6, seed=7, spec=MIX_SPEC)
    dims = [src.cov_dim_ext, [16], 4, [16], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=5, std=0.2)
    model = _model_from_state(src, dims, state)
    labels = torch.tensor(src.labels, device=dev)
    gp = (GPPriorHIP if impl == "hip" else GPPrior)(dims[2], labels, M=10, id_covariate=2, N_total=480, seed=3)
    m0, H0, z0 = gp.m.clone(), gp.H.clone(), gp.zt_list.detach().clone()
    tr = ELBOTrainer(model, P_total=80, kl="gp", gp=gp, max_batch=128)
    eps = torch.randn(48, dims[2], generator=torch.Generator().manual_seed(9))
    data, mask = torch.tensor(src.data, device=dev), torch.tensor(src.mask, device=dev)
    tr.step(data, mask, 8, eps=eps.to(dev), train_x=labels)
    torch.cuda.synchronize()
    nll_gpu, kld_gpu = float(tr.scalars()["nll_sum"]), float(gp.last_kld)
    # oracle
    st = {k: v.double().clone().requires_grad_(True) for k, v in state.items() if not k.startswith("hidden.")}
    for k in list(st):
        if k.startswith("d_layers."):
            st["hidden." + k[len("d_layers."):]] = st[k]
    om = orc.OracleHLVAE(dims, src.types_info, src.n_variables, st)
    out = om.forward(torch.tensor(src.data), torch.tensor(src.mask), eps.double())
    spec = gpo.spec_from_config([2], [], [0], [{"cont_covariate": 0, "cat_covariate": 2}, {"cont_covariate": 0, "cat_covariate": 3},
                                               {"cont_covariate": 1, "cat_covariate": 4}], [], 2)
    kprm = gpo.init_kernel_params(spec, dims[2])
    kld, gm, gH = gpo.minibatch_kld_upper_bound_iter(spec, kprm, torch.ones(dims[2], dtype=torch.float64), dims[2], m0.cpu(), H0.cpu(),
                                                     torch.tensor(src.labels), out["mu"], out["log_var"], z0.cpu(), 80, 8, 480,
                                                     True, 2, 1e-6)
    nll = om.loss_function(out["log_p_x"]).sum()
    (nll * 80 / 8 + kld.sum()).backward()
    _report("gp_prior_training_step", nll_rel=abs(nll_gpu - float(nll)) / abs(float(nll)), kld_rel=abs(kld_gpu - float(kld)) / abs(float(kld)))
    assert abs(nll_gpu - float(nll)) <= ELBO_RTOL * abs(float(nll))
    # named exception: the bound is evaluated at the device's mu / log_var (bf16 products behind them)
    assert abs(kld_gpu - float(kld)) <= 1e-3 * abs(float(kld)) + 1e-2
    # Adam's first update is -lr * sign(g): the encoder weights feel the GP gradient through mu / log_var
    w = dict(model.named_parameters())["mean_layer.0.weight"].detach().double().cpu()
    delta = w - state["mean_layer.0.weight"].double()
    gref = st["mean_layer.0.weight"].grad
    big = gref.abs() > 0.05 * gref.abs().max()
    assert (torch.sign(delta[big]) == -torch.sign(gref[big])).double().mean() > 0.97
    m_ref, H_ref = gpo.natural_gradient_update(m0.cpu(), H0.cpu(), gm.detach(), gH.detach(), 0.01)
    _report("gp_prior_training_step", m_new=rel_err(gp.m.cpu(), m_ref), H_new=rel_err(gp.H.cpu(), H_ref))
    assert rel_err(gp.m.cpu(), m_ref) < GP_STATE_TOL and rel_err(gp.H.cpu(), H_ref) < GP_STATE_TOL


@pytest.mark.parametrize("B", [1, 17, 130, 401])
def test_ragged_batch_sizes_against_oracle(B):
    """batch sizes that are not multiples of any tile (padding rows must not leak into sums, statistics or gradients);
    one model / workspace serves all sizes in turn."""
    import hlvae_oracle as orc
    dev = _dev()
    src = synthetic.make_tabular(n_rows=401, T=7, seed=23, spec=MIX_SPEC)
    dims = [src.cov_dim_ext, [32], 8, [32], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=8, std=0.15)
    model = _model_from_state(src, dims, state, max_batch=512)
    rows = np.arange(B)
    if B == 1:          # a single row has zero batch variance: use observed-everything so that the reference is finite
        pass
    eps = torch.randn(B, dims[2], generator=torch.Generator().manual_seed(B))
    data, mask = torch.tensor(src.data[rows], device=dev), torch.tensor(src.mask[rows], device=dev)
    out = model(data, mask, None, src.types_info, eps=eps.to(dev))
    loss = 1.3 * model.loss_function(out[3]).sum() - 0.5 * torch.sum(1.0 + out[2] - out[1] ** 2 - torch.exp(out[2]))
    loss.backward()
    torch.cuda.synchronize()
    ref, ref_loss, st = _oracle_step(src, rows, dims, state, eps, 1.3)
    fin = torch.isfinite(ref["log_p_x"]).all() and np.isfinite(float(ref_loss))
    if not fin:
        # A real/pos column with no observed entry in the batch: the reference divides 0/0 (HL_VAE/utils.py:105) and its
        # NaN*0 products poison every output.  The HIP path drops unobserved entries with selects, so log_p_x stays
        # finite; the undefined statistics still surface as NaN in log_p_x_missing of that column.  Documented difference.
        assert torch.isnan(out[4]).any()
        return
    _report(f"ragged_{B}", loss_rel=abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)), mu=max_abs_err(out[1], ref["mu"]))
    assert abs(float(loss) - float(ref_loss)) <= ELBO_RTOL * abs(float(ref_loss)), (float(loss), float(ref_loss))
    assert max_abs_err(out[1], ref["mu"]) < 3e-2
    # the device-side KL(q || N(0, I)) scalar (one partial per 16-row tile) against the same sum over the device mu / log_var
    kl_dev = float(model._ws_t["scal"][1])
    kl_sum = float(-0.5 * torch.sum(1.0 + out[2].double() - out[1].double() ** 2 - torch.exp(out[2].double())))
    assert abs(kl_dev - kl_sum) <= 1e-5 * abs(kl_sum) + 1e-6, (kl_dev, kl_sum)
    e = np.abs(out[3].detach().double().cpu().numpy() - ref["log_p_x"].detach().numpy())
    assert np.all(e <= 5e-2 + 3e-2 * np.abs(ref["log_p_x"].detach().numpy()))
    sd = dict(model.named_parameters())
    for k in ("y_layer.0.weight", "VAE_encoder_common_layers.0.weight", "d_layers.0.bias", "obs_layer.1.weight"):
        _report(f"ragged_{B}_grads", **{k: rel_err(sd[k].grad, st[k].grad)})
        assert rel_err(sd[k].grad, st[k].grad) < GRAD_RTOL, k


def test_tabular_config4_batch4096_against_oracle():
    """BASELINE config 4 shape: 64 mixed-type features (16 real / 16 pos / 8 count / 16 cat5 / 8 ordinal5, interleaved),
    hidden 500, latent 32, 4096 rows."""
    dev = _dev()
    src = synthetic.make_tabular(n_rows=4096, T=16, seed=41)
    dims = [src.cov_dim_ext, [500], 32, [500], 5]
    from hlvae_amd.HLVAE import HLVAE
    torch.manual_seed(7)
    model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=4096, materialize_samples=False).to(dev)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    eps = torch.randn(4096, 32, generator=torch.Generator().manual_seed(2))
    out = model(torch.tensor(src.data, device=dev), torch.tensor(src.mask, device=dev), None, src.types_info, eps=eps.to(dev))
    loss = model.loss_function(out[3]).sum() - 0.5 * torch.sum(1.0 + out[2] - out[1] ** 2 - torch.exp(out[2]))
    loss.backward()
    torch.cuda.synchronize()
    ref, ref_loss, st = _oracle_step(src, np.arange(4096), dims, state, eps, 1.0)
    rel = abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))
    _report("tabular_b4096", loss_rel=rel)
    assert rel <= ELBO_RTOL, rel
    sd = dict(model.named_parameters())
    for k, p in sd.items():
        if p.grad is None or st[k].grad is None:
            continue
        e = rel_err(p.grad, st[k].grad)
        _report("tabular_b4096_grads", **{k: e})
        assert e < 1.5e-2, (k, e)                  # measured <= 5.6e-3


def test_data_parallel_code_path_single_rank_rccl(golden_dir):
    """the data-parallel step (statistics all-reduce, split backward with the overlapped y_layer all-reduce, the two
    remaining arena slices) on a ONE-rank RCCL group must reproduce the plain step bit for bit (same kernels, the
    collectives are identities) and replay as a captured HIP graph: exercises the real nccl/RCCL code path that the multi-GPU
    bench uses.  Body: tests/rccl_single_rank_case.py, in a process of its own (see there)."""
    import subprocess
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_single_rank_case.py")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_step_metrics_against_reference_fixture(golden_dir):
    """row M: error_observed / error_missing per variable from the device kernel against the values the reference's
    read_functions.error_computation produced for the same batch (fixture), up to bf16-induced argmax flips."""
    g, src, dims, state = load_mix_case(golden_dir, "mix_trained")
    dev = _dev()
    model = _model_from_state(src, dims, state)
    data, mask = torch.tensor(g["data"], device=dev), torch.tensor(g["mask"], device=dev)
    with torch.no_grad():
        model(data, mask, None, src.types_info, eps=torch.tensor(g["eps"], device=dev))
    e_obs, e_mis, e_all, xhat = model.step_metrics(24)
    disc = np.isin(model.plan.kind, [3, 4])
    xh, xr = xhat.cpu().numpy(), g["x_hat_mean"]
    assert np.mean(xh[:, disc] == xr[:, disc]) > 0.97
    assert np.allclose(xh[:, ~disc], xr[:, ~disc], rtol=3e-2, atol=3e-2)
    # continuous variables: RMSE agrees to bf16 accuracy; discrete ones: at most one flipped row out of 24
    eo, em = e_obs.cpu().numpy(), e_mis.cpu().numpy()
    assert np.allclose(eo[~disc], g["err_observed"][~disc], rtol=5e-2, atol=5e-3)
    assert np.allclose(em[~disc], g["err_missing"][~disc], rtol=5e-2, atol=5e-3)
    assert np.all(np.abs(eo[disc] - g["err_observed"][disc]) <= 1.0 / 8) and np.all(np.abs(em[disc] - g["err_missing"][disc]) <= 1.0 / 4)


def _copy_hip_params_to_torch_gp(hip, ref):
    """hyper-parameter rows of GPPriorHIP -> the GP_model kernel modules of GPPrior (same term order)"""
    from hlvae_amd import GP_model
    with torch.no_grad():
        for row, (which, t, f) in enumerate(hip.slot_names):
            sk = (ref.k0 if which == "k0" else ref.k1).kernels[t]
            if f is None:
                sk._log_scale.copy_(hip.prm[row].to(sk._log_scale.device))
            else:
                facs = list(sk.kernel.factors) if isinstance(sk.kernel, GP_model.ProductKernel) else [sk.kernel]
                facs[f]._log_lengthscale.copy_(hip.prm[row].to(ref.zt_list.device))
        ref.zt_list.copy_(hip.zt_list.to(ref.zt_list.device))
        ref.m, ref.H = hip.m.clone().to(ref.zt_list.device), hip.H.clone().to(ref.zt_list.device)


@pytest.mark.parametrize("varying_T", [False, True])
def test_gp_prior_hip_against_autograd_statement(varying_T):
    """row K, hand-written path: kernel matrices, LDS Cholesky/inverse, the per-(subject, latent) block kernel and the
    analytic chain rule into hyper-parameters / inducing points (csrc/gp.hip) against the batched torch + autograd
    statement of the same bound (which is pinned to the reference fixture in tests/test_gp_prior.py).  fp64: 1e-9."""
    from hlvae_amd.elbo_functions import GPPrior, GPPriorHIP
    dev = _dev()
    torch.manual_seed(0)
    L, Q, M = 6, 6, 20
    Ts = [5, 7, 3, 6, 7, 4, 7, 2] if varying_T else [6] * 8
    rows = []
    for s_, T in enumerate(Ts):
        for t in range(T):
            rows.append([float(t), float(t - 2) if s_ % 2 else 0.0, float(s_ + 3), float(s_ % 2), float(s_ % 2), float((s_ // 2) % 2)])
    x = torch.tensor(rows, dtype=torch.float64)
    x = x[torch.randperm(x.shape[0])].to(dev)          # rows of a subject are not contiguous
    B = x.shape[0]
    hip = GPPriorHIP(L, x, M, 2, N_total=777, seed=4)
    # the autograd statement runs on the CPU (LAPACK): the batched Cholesky of torch-ROCm proved order-dependent here
    ref = GPPrior(L, x.cpu(), M, 2, N_total=777, seed=4)
    with torch.no_grad():
        hip.prm.add_(0.3 * torch.randn_like(hip.prm))
        hip.zt_list.add_(0.05 * torch.randn_like(hip.zt_list))
    _copy_hip_params_to_torch_gp(hip, ref)
    mu = torch.randn(B, L, device=dev)
    lv = (0.5 * torch.randn(B, L, device=dev) - 1.0)
    g_mu_r, g_lv_r = ref.kl_and_grads(mu.cpu(), lv.cpu(), x.cpu(), 40, len(Ts))
    assert bool(torch.isfinite(ref.last_kld).all())
    g_mu_h, g_lv_h = hip.kl_and_grads(mu, lv, x, 40, len(Ts))
    torch.cuda.synchronize()
    assert int(hip.fail.item()) == 0
    assert rel_err(hip.last_kld, ref.last_kld) < 1e-10
    assert rel_err(g_mu_h, g_mu_r) < 1e-6 and rel_err(g_lv_h, g_lv_r) < 1e-6          # fp32 outputs
    assert rel_err(hip._grad_m, ref._grad_m) < 1e-9 and rel_err(hip._grad_H, ref._grad_H) < 1e-9
    assert rel_err(hip.zt_list.grad, ref.zt_list.grad) < 1e-8
    from hlvae_amd import GP_model
    for row, (which, t, f) in enumerate(hip.slot_names):
        sk = (ref.k0 if which == "k0" else ref.k1).kernels[t]
        if f is None:
            gr = sk._log_scale.grad
        else:
            facs = list(sk.kernel.factors) if isinstance(sk.kernel, GP_model.ProductKernel) else [sk.kernel]
            gr = facs[f]._log_lengthscale.grad
        assert rel_err(hip.prm.grad[row], gr) < 1e-8, (row, which, t, f)
    hip.optimizer_step()
    ref.optimizer_step()
    assert rel_err(hip.m, ref.m) < 1e-8 and rel_err(hip.H, ref.H) < 1e-8
    assert rel_err(hip.zt_list, ref.zt_list) < 1e-9


@pytest.mark.parametrize("N,batch", [(120, 32), (128, 3), (37, 5), (66, 2)])
def test_gp_bmm_and_rsym_against_torch(N, batch):
    """csrc/gp.hip k_gp_bmm (fp64 MFMA, C = alpha A B + beta D with D aliasing nothing / an operand) and k_gp_rsym against
    torch.float64 on the same operands: 1e-13 (different summation order only)."""
    import ctypes as C
    from hlvae_amd import _lib
    lib = _lib.load()
    dev = _dev()
    g = torch.Generator().manual_seed(N)
    A, B, D = (torch.randn(batch, N, N, generator=g, dtype=torch.float64).to(dev) for _ in range(3))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.full_like(A, float("nan"))
    _lib.check(lib.hlvae_gp_bmm(_lib.ptr(A), _lib.ptr(B), None, _lib.ptr(out), N, batch, C.c_double(1.0), C.c_double(0.0), st), "bmm")
    assert rel_err(out, A @ B) < 1e-13
    _lib.check(lib.hlvae_gp_bmm(_lib.ptr(A), _lib.ptr(B), _lib.ptr(D), _lib.ptr(out), N, batch, C.c_double(-0.5), C.c_double(2.0), st), "bmm")
    assert rel_err(out, -0.5 * (A @ B) + 2.0 * D) < 1e-13
    u, m = (torch.randn(batch, N, 1, generator=g, dtype=torch.float64).to(dev) for _ in range(2))
    _lib.check(lib.hlvae_gp_rsym(_lib.ptr(u), _lib.ptr(m), _lib.ptr(A), _lib.ptr(B), _lib.ptr(D), C.c_double(1.7), N, batch,
                                 _lib.ptr(out), st), "rsym")
    mT = m.transpose(1, 2)
    ref = 1.7 * (u @ mT + m @ u.transpose(1, 2) - A + B + B.transpose(1, 2)) + D + m @ mT
    assert rel_err(out, ref) < 1e-13
    # matrix^T-vector product with a strided vector operand (k_gp_gemv_t)
    rows = 203
    R = torch.randn(batch, rows, N, generator=g, dtype=torch.float64).to(dev)
    xv = torch.randn(rows, batch, generator=g, dtype=torch.float64).to(dev)             # element (l, b) at xv[b, l]
    o2 = torch.full((batch, N), float("nan"), dtype=torch.float64, device=dev)
    _lib.check(lib.hlvae_gp_gemv_t(_lib.ptr(R), _lib.ptr(xv), xv.stride(1), xv.stride(0), _lib.ptr(o2), batch, rows, N, st), "gemv_t")
    assert rel_err(o2, (R.transpose(1, 2) @ xv.t().unsqueeze(2)).squeeze(2)) < 1e-13


@pytest.mark.parametrize("M,N,K,batch,transA", [(1024, 120, 120, 4, 0), (120, 120, 1024, 4, 1), (203, 100, 36, 2, 0), (100, 128, 300, 3, 1),
                                                 (64, 130, 64, 2, 0), (33, 7, 21, 2, 1)])
def test_gp_gemm_against_torch(M, N, K, batch, transA):
    """hlvae_gp_gemm, C = alpha op(A) B + beta D on the fp64 matrix cores (both tile heights, with and without split-K, odd sizes)
    against torch.float64."""
    import ctypes as C
    from hlvae_amd import _lib
    lib = _lib.load()
    dev = _dev()
    g = torch.Generator().manual_seed(M + 7 * N + 13 * K)
    A = torch.randn(batch, *((K, M) if transA else (M, K)), generator=g, dtype=torch.float64).to(dev)
    B = torch.randn(batch, K, N, generator=g, dtype=torch.float64).to(dev)
    D = torch.randn(batch, M, N, generator=g, dtype=torch.float64).to(dev)
    out = torch.full((batch, M, N), float("nan"), dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    opA = A.transpose(1, 2) if transA else A
    for alpha, beta, d in ((1.0, 0.0, None), (-0.5, 2.0, D)):
        _lib.check(lib.hlvae_gp_gemm(_lib.ptr(A), A.shape[2], A.stride(0), transA, _lib.ptr(B), N, B.stride(0), _lib.ptr(d), N if d is not None else 0,
                                     d.stride(0) if d is not None else 0, _lib.ptr(out), N, out.stride(0), M, N, K, batch, C.c_double(alpha),
                                     C.c_double(beta), st), "gp_gemm")
        ref = alpha * (opA @ B) + (beta * d if d is not None else 0.0)
        assert rel_err(out, ref) < 1e-13, (alpha, beta, rel_err(out, ref))


@pytest.mark.parametrize("N,batch", [(120, 32), (128, 3), (24, 5), (100, 2), (4, 1)])
def test_gp_chain_kernels_against_torch(N, batch):
    """The M x M algebra of a GP step behind W as one call (round 3): hlvae_gp_chain (one workgroup per (latent, chain)) and
    hlvae_gp_chain_rb (two launches of independent 32-row blocks) against the torch.float64 statement of elbo_functions.py:279-283
    and of the K0zz gradient, on symmetric positive definite operands -- different summation order only."""
    import ctypes as C
    from hlvae_amd import _lib
    lib = _lib.load()
    dev = _dev()
    g = torch.Generator().manual_seed(1000 + N)

    def spd():
        a = torch.randn(batch, N, N, generator=g, dtype=torch.float64)
        return (a @ a.transpose(1, 2) / N + torch.eye(N, dtype=torch.float64)).to(dev)

    iK, W, H, iH = spd(), spd(), spd(), spd()
    HiK = (H @ iK).contiguous()
    m, P1, u = (torch.randn(batch, N, 1, generator=g, dtype=torch.float64).to(dev) for _ in range(3))
    lr, c, ga, gb = 0.01, 1.7, -0.5, 0.5
    # the statement
    Bm = iK @ W @ iK + iK
    grad_m = Bm @ m - iK @ P1
    grad_H = 0.5 * (Bm - iH)
    tmp = iH @ m - lr * (grad_m - 2.0 * grad_H @ m)
    X = HiK @ W
    mT = m.transpose(1, 2)
    Rs = c * (u @ mT + m @ u.transpose(1, 2) - W + X + X.transpose(1, 2)) + H + m @ mT
    G = ga * (iK @ Rs @ iK) + gb * iK
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = _lib.ptr
    e = lambda *sh: torch.full(sh, float("nan"), dtype=torch.float64, device=dev)
    for which in ("chain", "chain_rb"):
        o = dict(T1=e(batch, N, N), Bm=e(batch, N, N), grad_m=e(batch, N, 1), grad_H=e(batch, N, N), tmp=e(batch, N, 1), HiKW=e(batch, N, N),
                 Rs=e(batch, N, N), T1b=e(batch, N, N), G=e(batch, N, N))
        if which == "chain":
            _lib.check(lib.hlvae_gp_chain(p(iK), p(W), p(HiK), p(H), p(iH), p(m), p(P1), p(u), C.c_double(lr), C.c_double(c), C.c_double(ga),
                                          C.c_double(gb), N, batch, p(o["T1"]), p(o["Bm"]), p(o["grad_m"]), p(o["grad_H"]), p(o["tmp"]),
                                          p(o["HiKW"]), p(o["Rs"]), p(o["T1b"]), p(o["G"]), st), which)
        else:
            _lib.check(lib.hlvae_gp_chain_rb(p(iK), p(W), p(HiK), p(H), p(iH), p(m), p(P1), p(u), C.c_double(lr), C.c_double(c), C.c_double(ga),
                                             C.c_double(gb), N, batch, p(o["grad_m"]), p(o["grad_H"]), p(o["tmp"]), p(o["Rs"]), p(o["G"]), st), which)
        torch.cuda.synchronize()
        for k, ref in (("grad_m", grad_m), ("grad_H", grad_H), ("tmp", tmp), ("Rs", Rs), ("G", G)):
            assert rel_err(o[k], ref) < 1e-12, (which, k, rel_err(o[k], ref))


def test_gp_prior_config5_size_against_autograd_statement():
    """BASELINE configs[4] (GP variant) at its full size: 32 latent GPs, 120 inducing points, 6 covariates, a 1024-row batch
    of 51 whole subjects x 20 rows + 4 rows of a 52nd -- bound, gradients w.r.t. mu / log-variance, natural-gradient
    statistics and hyper-parameter gradients of the HIP path against the torch-autograd statement on the CPU (LAPACK)."""
    from hlvae_amd.elbo_functions import GPPrior, GPPriorHIP
    dev = _dev()
    torch.manual_seed(1)
    L, M = 32, 120
    Ts = [20] * 51 + [4]
    rows = []
    for s_, T in enumerate(Ts):
        for t in range(T):
            sick = s_ % 2
            rows.append([float(t), float(t - 9) if sick else 0.0, float(s_), float(s_ % 2), float(sick), float((s_ // 2) % 2)])
    x = torch.tensor(rows, dtype=torch.float64)
    x = x[torch.randperm(x.shape[0])].to(dev)
    B = x.shape[0]
    assert B == 1024
    hip = GPPriorHIP(L, x, M, 2, N_total=50000, seed=4)
    ref = GPPrior(L, x.cpu(), M, 2, N_total=50000, seed=4)
    with torch.no_grad():
        hip.prm.add_(0.2 * torch.randn_like(hip.prm))
        hip.zt_list.add_(0.05 * torch.randn_like(hip.zt_list))
    _copy_hip_params_to_torch_gp(hip, ref)
    mu = torch.randn(B, L, device=dev)
    lv = (0.5 * torch.randn(B, L, device=dev) - 1.0)
    g_mu_r, g_lv_r = ref.kl_and_grads(mu.cpu(), lv.cpu(), x.cpu(), 2500, len(Ts))
    assert bool(torch.isfinite(ref.last_kld).all())
    g_mu_h, g_lv_h = hip.kl_and_grads(mu, lv, x, 2500, len(Ts))
    torch.cuda.synchronize()
    assert int(hip.fail.item()) == 0
    assert rel_err(hip.last_kld, ref.last_kld) < 1e-9
    assert rel_err(g_mu_h, g_mu_r) < 1e-5 and rel_err(g_lv_h, g_lv_r) < 1e-5          # fp32 outputs
    assert rel_err(hip._grad_m, ref._grad_m) < 1e-7 and rel_err(hip._grad_H, ref._grad_H) < 1e-7
    assert rel_err(hip.zt_list.grad, ref.zt_list.grad) < 1e-6
    hip.optimizer_step()
    ref.optimizer_step()
    assert rel_err(hip.m, ref.m) < 1e-7 and rel_err(hip.H, ref.H) < 1e-7


@pytest.mark.parametrize("conv", [False, True])
def test_compact_feed_matches_expanded_inputs(conv):
    """the input stage gathering from the device-resident compact dataset (csrc/feed.hip) leaves exactly the
    buffers the fp64 path leaves for the same rows -- packed encoder input (both layouts), likelihood targets, mask,
    statistics -- and a training step from it follows the same trajectory."""
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset
    dev = _dev()
    if conv:
        src = synthetic.make_d4(n_subjects=6, T=5, seed=2)
        dims = [src.cov_dim_ext, [32], 8, [32], 5]
    else:
        src = synthetic.make_tabular(n_rows=60, T=6, seed=3, spec=MIX_SPEC)
        dims = [src.cov_dim_ext, [16], 4, [16], 5]
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    rows = np.array([7, 3, 22, 23, 24, 11, 0, 29, 18, 19, 20, 5, 6, 1, 28, 2, 9], dtype=np.int32)
    eps = torch.randn(len(rows), dims[2], generator=torch.Generator().manual_seed(4)).to(dev)

    def run(compact):
        torch.manual_seed(0)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=conv, max_batch=128, materialize_samples=False).to(dev)
        tr = ELBOTrainer(model, P_total=10, kl="normal", max_batch=128)
        for _ in range(2):
            if compact:
                tr.step_rows(ds, torch.tensor(rows, device=dev), 5, eps=eps)
            else:
                tr.step(torch.tensor(src.data[rows], device=dev), torch.tensor(src.mask[rows], device=dev), 5, eps=eps)
        torch.cuda.synchronize()
        t = model._ws_t
        B = len(rows)
        return dict(xn=t["xn"][:B].float().cpu(), xnT=t["xnT"][:, :B].float().cpu(), xt=t["xt"][:B].cpu(), m8=t["m8"][:B].cpu(),
                    norm=t["norm"].cpu(), nll=float(tr.scalars()["nll_sum"]), P=model._arena.cpu())

    a, b = run(False), run(True)
    assert torch.equal(a["m8"], b["m8"]) and torch.equal(a["xt"], b["xt"])
    assert rel_err(b["norm"], a["norm"]) < 1e-6
    assert rel_err(b["xn"], a["xn"]) < 1e-6 and rel_err(b["xnT"], a["xnT"]) < 1e-6
    assert abs(a["nll"] - b["nll"]) <= 1e-6 * abs(a["nll"])
    assert rel_err(b["P"], a["P"]) < 1e-6


def test_compact_feed_pipelined_matches_serial():
    """step_rows(prefetch_rows=next) -- the next batch's input stage deferred to the side stream of this step's backward pass
    (hlvae_feed_prefetch), into the second buffer set -- follows the trajectory of steps that run their own input stage:
    eagerly, and as ONE captured HIP graph of 4 chained steps replayed twice (ragged batch sizes, two alternating batches of
    different length are not allowed in one static graph: same length, different rows)."""
    from hlvae_amd.HLVAE import HLVAE
    from hlvae_amd.training import ELBOTrainer
    from hlvae_amd.datafeed import CompactDataset
    dev = _dev()
    src = synthetic.make_tabular(n_rows=90, T=6, seed=8, spec=MIX_SPEC)
    dims = [src.cov_dim_ext, [16], 4, [16], 5]
    ds = CompactDataset.from_expanded(src.data, src.mask, src.labels, src.types_info, src.id_covariate).to(dev)
    rg = np.random.default_rng(3)
    R = [torch.tensor(rg.permutation(90)[:37].astype(np.int32), device=dev) for _ in range(4)]
    eps = [torch.randn(37, dims[2], generator=torch.Generator().manual_seed(20 + i)).to(dev) for i in range(8)]

    def fresh():
        torch.manual_seed(0)
        model = HLVAE(dims, src.types_info, src.n_variables, conv=False, max_batch=128, materialize_samples=False).to(dev)
        return model, ELBOTrainer(model, P_total=15, kl="normal", max_batch=128)

    def run(pipelined):
        model, tr = fresh()
        if pipelined:
            tr.prime_rows(ds, R[0])
        nll, err = [], []
        for i in range(8):
            tr.step_rows(ds, R[i % 4], 7, eps=eps[i], prefetch_rows=R[(i + 1) % 4] if pipelined else None, prepacked=pipelined)
            nll.append(float(tr.scalars()["nll_sum"]))
            err.append(tr.err.clone())
        return nll, model._arena.clone(), torch.stack(err)

    nll_a, P_a, e_a = run(False)
    nll_b, P_b, e_b = run(True)
    assert rel_err(np.array(nll_b), np.array(nll_a)) < 1e-6, (nll_a, nll_b)
    assert rel_err(P_b, P_a) < 1e-6 and rel_err(e_b, e_a) < 1e-5
    # one graph of 4 chained pipelined steps, replayed twice, against 8 eager serial steps with the same in-kernel noise
    model, tr = fresh()
    P0 = model._arena.clone()
    tr.capture_rows("ring", ds, R, [7] * 4, next_rows=[R[(i + 1) % 4] for i in range(4)])

    def reset():
        model._arena.copy_(P0)
        model._sync_shadows(force=True)
        tr.opt.m1.zero_(); tr.opt.m2.zero_(); tr.opt.step_count.zero_()

    reset()
    rng0 = model._ws_t["rng"].clone()
    tr.prime_rows(ds, R[0])
    tr.replay("ring")
    tr.replay("ring")
    nll_g, P_g = float(tr.scalars()["nll_sum"]), model._arena.clone()
    assert int(tr.opt.step_count[0]) == 8
    reset()
    model._ws_t["rng"].copy_(rng0)
    for i in range(8):
        tr.step_rows(ds, R[i % 4], 7)
    assert abs(nll_g - float(tr.scalars()["nll_sum"])) <= 1e-6 * abs(nll_g)
    assert rel_err(P_g, model._arena) < 1e-6


def test_gp_posterior_prediction_against_reference_fixture(golden_dir):
    """GPPriorHIP.batch_predict_varying_T (HIP kernel matrices, per-subject blocks, M x M inverses) against the
    reference's utils.batch_predict_varying_T output stored in tests/golden/gp_predict.npz.  fp64: 1e-8."""
    from hlvae_amd.elbo_functions import GPPriorHIP
    g = np.load(os.path.join(golden_dir, "gp_predict.npz"))
    dev = _dev()
    x, tx, mu = (torch.tensor(g[k], device=dev) for k in ("x", "test_x", "mu"))
    L, M = mu.shape[1], g["z"].shape[1]
    gp = GPPriorHIP(L, x, M, int(g["scalars"][1]), N_total=1, eps=float(g["scalars"][0]))
    with torch.no_grad():
        gp.zt_list.copy_(torch.tensor(g["z"], device=dev))
        for row, (which, t, f) in enumerate(gp.slot_names):
            key = f"kp__{which}.{t}.scale" if f is None else f"kp__{which}.{t}.{f}.ls"
            gp.prm[row].copy_(torch.tensor(g[key], device=dev))
    Zp = gp.batch_predict_varying_T(x, tx, mu)
    assert tuple(Zp.shape) == g["Z_pred"].shape
    assert rel_err(Zp, g["Z_pred"]) < 1e-8


@pytest.mark.parametrize("y_dim", [3, 8])
def test_other_y_dim_against_oracle(y_dim):
    """y_dim is a configuration value of the reference (config/hlvae_config_file.txt: y_dim, default 5): the head kernel is
    also instantiated for 3 and 8.  Forward + backward on the five-type mix (class counts up to 5) against the oracle."""
    import hlvae_oracle as orc
    dev = _dev()
    src = synthetic.make_tabular(n_rows=48, T=6, seed=7, spec=MIX_SPEC)
    dims = [src.cov_dim_ext, [16], 4, [16], y_dim]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=21, std=0.2)
    model = _model_from_state(src, dims, state)
    rows = np.arange(48)
    eps = torch.randn(48, dims[2], generator=torch.Generator().manual_seed(3))
    data, mask = torch.tensor(src.data[rows], device=dev), torch.tensor(src.mask[rows], device=dev)
    out = model(data, mask, None, src.types_info, eps=eps.to(dev))
    loss = 1.3 * model.loss_function(out[3]).sum() - 0.5 * torch.sum(1.0 + out[2] - out[1] ** 2 - torch.exp(out[2]))
    loss.backward()
    torch.cuda.synchronize()
    ref, ref_loss, st = _oracle_step(src, rows, dims, state, eps, 1.3)
    _report(f"y_dim_{y_dim}", loss_rel=abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)))
    assert abs(float(loss) - float(ref_loss)) <= ELBO_RTOL * abs(float(ref_loss)), (float(loss), float(ref_loss))
    e = np.abs(out[3].detach().double().cpu().numpy() - ref["log_p_x"].detach().numpy())
    assert np.all(e <= 5e-2 + 3e-2 * np.abs(ref["log_p_x"].detach().numpy()))
    sd = dict(model.named_parameters())
    for k in ("y_layer.0.weight", "y_layer.0.bias", "d_layers.0.bias", "obs_layer.0.weight", "obs_layer.3.weight_mean",
              "obs_layer.2.weight_region", "obs_layer.2.weight_thresholds"):
        if k in sd and st[k].grad is not None:
            _report(f"y_dim_{y_dim}_grads", **{k: rel_err(sd[k].grad, st[k].grad)})
            assert rel_err(sd[k].grad, st[k].grad) < GRAD_RTOL, k


def test_wide_categorical_and_ordinal_against_oracle():
    """class counts 9..16 (one extra head-kernel instance): a 12-class categorical and a 10-class ordinal variable among the
    usual mix, forward + backward against the oracle."""
    import hlvae_oracle as orc
    dev = _dev()
    spec = [("real", 1), ("cat", 12), ("pos", 1), ("ordinal", 10), ("count", 1), ("cat", 5), ("ordinal", 4), ("cat", 16)]
    # X = 50: not a multiple of 4 -> the optimiser's element-wise path for the first Linear's rows
    src = synthetic.make_tabular(n_rows=60, T=6, seed=17, spec=spec)
    dims = [src.cov_dim_ext, [16], 4, [16], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=31, std=0.2)
    model = _model_from_state(src, dims, state)
    rows = np.arange(60)
    eps = torch.randn(60, dims[2], generator=torch.Generator().manual_seed(8))
    data, mask = torch.tensor(src.data[rows], device=dev), torch.tensor(src.mask[rows], device=dev)
    out = model(data, mask, None, src.types_info, eps=eps.to(dev))
    loss = 0.7 * model.loss_function(out[3]).sum() - 0.5 * torch.sum(1.0 + out[2] - out[1] ** 2 - torch.exp(out[2]))
    loss.backward()
    torch.cuda.synchronize()
    ref, ref_loss, st = _oracle_step(src, rows, dims, state, eps, 0.7)
    _report("wide_k", loss_rel=abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)))
    assert abs(float(loss) - float(ref_loss)) <= ELBO_RTOL * abs(float(ref_loss)), (float(loss), float(ref_loss))
    e = np.abs(out[3].detach().double().cpu().numpy() - ref["log_p_x"].detach().numpy())
    assert np.all(e <= 5e-2 + 3e-2 * np.abs(ref["log_p_x"].detach().numpy()))
    sd = dict(model.named_parameters())
    n = 0
    for k, p in sd.items():
        if k.startswith("obs_layer.") and st[k].grad is not None and st[k].grad.numel() and float(st[k].grad.abs().max()) > 0:
            _report("wide_k_grads", **{k: rel_err(p.grad, st[k].grad)})
            assert rel_err(p.grad, st[k].grad) < GRAD_RTOL, k
            n += 1
    assert n >= 8
    assert rel_err(sd["VAE_encoder_common_layers.0.weight"].grad, st["VAE_encoder_common_layers.0.weight"].grad) < GRAD_RTOL
    # two fused optimiser steps on the odd-width model against torch.optim.Adam on the same gradients
    from hlvae_amd.training import ELBOTrainer
    m2 = _model_from_state(src, dims, state)
    tr = ELBOTrainer(m2, P_total=10, kl="normal", max_batch=128)
    w0 = m2.VAE_encoder_common_layers[0].weight.detach().clone()
    tr.step(data, mask, 10, eps=eps.to(dev))
    torch.cuda.synchronize()
    dw = (m2.VAE_encoder_common_layers[0].weight.detach() - w0).abs()
    assert float(dw.max()) <= 1.001e-3 and float(dw.mean()) > 5e-4          # first Adam step: |delta| = lr wherever g != 0
    assert bool(torch.isfinite(m2._arena).all())


def test_gp_factorisation_carried_across_steps():
    """GPPriorHIP inverts K0zz of the UPDATED hyper-parameters together with iH_new at the end of a step; the next step must
    be identical to recomputing both factorizations from scratch."""
    from hlvae_amd.elbo_functions import GPPriorHIP
    dev = _dev()
    torch.manual_seed(0)
    L, M = 6, 20
    rows = []
    for s_ in range(8):
        for t in range(6):
            rows.append([float(t), float(t - 2) if s_ % 2 else 0.0, float(s_ + 3), float(s_ % 2), float(s_ % 2), float((s_ // 2) % 2)])
    x = torch.tensor(rows, dtype=torch.float64, device=dev)
    mus = [torch.randn(48, L, device=dev) for _ in range(3)]
    lvs = [0.5 * torch.randn(48, L, device=dev) - 1.0 for _ in range(3)]

    def run(carry):
        gp = GPPriorHIP(L, x, M, 2, N_total=777, seed=4)
        out = []
        for i in range(3):
            if not carry:
                gp._fact_key = None
            g_mu, g_lv = gp.kl_and_grads(mus[i], lvs[i], x, 40, 8)
            out.append((float(gp.last_kld), g_mu.clone()))
            gp.optimizer_step()
        torch.cuda.synchronize()
        assert int(gp.fail.item()) == 0
        return out, gp.m.clone(), gp.H.clone(), gp.prm.clone()

    a, b = run(True), run(False)
    # (the carried iH is iH_old + lr * (...) exactly; the recomputed one is inv(inv(.)) of a matrix with condition ~1e6)
    for (ka, ga), (kb, gb) in zip(a[0], b[0]):
        assert abs(ka - kb) <= 1e-7 * abs(kb) and rel_err(ga, gb) < 1e-6
    # (hyper-parameters: 3e-8 measured since round 3 -- the per-subject sums P1, u are accumulated with fp64 atomics, whose order
    #  differs from run to run in the last bits, and Adam's first steps are lr * g / |g|)
    assert rel_err(a[1], b[1]) < 1e-6 and rel_err(a[2], b[2]) < 1e-6 and rel_err(a[3], b[3]) < 2e-7


def test_odd_layer_widths_against_oracle():
    """hidden 30 / 26, latent 6, X = 38: nothing in the path needs widths that are multiples of 4 or of a tile."""
    import hlvae_oracle as orc
    from hlvae_amd.training import ELBOTrainer
    dev = _dev()
    spec = synthetic.tabular_type_spec(n_real=3, n_pos=3, n_count=2, n_cat=4, n_ord=2, K=5)
    src = synthetic.make_tabular(n_rows=80, T=8, seed=12, spec=spec)
    dims = [src.cov_dim_ext, [30], 6, [26], 5]
    state = orc.init_state(dims, src.types_info, src.n_variables, seed=4, std=0.2)
    model = _model_from_state(src, dims, state)
    rows = np.arange(80)
    eps = torch.randn(80, 6, generator=torch.Generator().manual_seed(2))
    data, mask = torch.tensor(src.data[rows], device=dev), torch.tensor(src.mask[rows], device=dev)
    out = model(data, mask, None, src.types_info, eps=eps.to(dev))
    loss = 1.1 * model.loss_function(out[3]).sum() - 0.5 * torch.sum(1.0 + out[2] - out[1] ** 2 - torch.exp(out[2]))
    loss.backward()
    torch.cuda.synchronize()
    ref, ref_loss, st = _oracle_step(src, rows, dims, state, eps, 1.1)
    _report("odd_widths", loss_rel=abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)))
    assert abs(float(loss) - float(ref_loss)) <= ELBO_RTOL * abs(float(ref_loss)), (float(loss), float(ref_loss))
    sd = dict(model.named_parameters())
    for k in ("y_layer.0.weight", "VAE_encoder_common_layers.0.weight", "d_layers.0.weight", "mean_layer.0.weight",
              "log_var_layer.0.weight", "d_layers.0.bias", "mean_layer.0.bias"):
        _report("odd_widths_grads", **{k: rel_err(sd[k].grad, st[k].grad)})
        assert rel_err(sd[k].grad, st[k].grad) < GRAD_RTOL, k
    tr = ELBOTrainer(_model_from_state(src, dims, state), P_total=10, kl="normal", max_batch=128)
    for _ in range(3):
        tr.step(data, mask, 10, eps=eps.to(dev))
    assert np.isfinite(float(tr.scalars()["nll_sum"])) and bool(torch.isfinite(tr.model._arena).all())
