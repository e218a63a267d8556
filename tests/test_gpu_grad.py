"""-m gpu: gradients through the solve (BASELINE.json configs[4]) -- the HIP backward sweep + record reduction against the
committed checker fixtures (tests/golden/grad_fixtures.npz, made by make_grad_fixtures.py: fp64 autograd through a torch
replay of the oracle's accepted steps).  Parity is UNPINNED by the reference (it never differentiates through odeint);
the stated tolerance is against the checker:

    GRAD_REL_TOL = 1e-4   relative L2 per gradient block (dL/dp, dL/dy0, sampled dL/dW, per-tensor |dL/dW|)

The kernel recomputes the fp32 MLP in a different summation order than the checker's CPU GEMV and accumulates dW in fp32
on the MFMA; measured errors are printed by the test (1e-16 .. 3e-6 on MI355X).
"""
import importlib
import os

import numpy as np
import pytest
import torch

import kat_cases as K
import ref_style_modules as M

pytestmark = pytest.mark.gpu
GRAD_REL_TOL = 1e-4

_spec = importlib.util.spec_from_file_location("make_grad_fixtures", os.path.join(K.GOLDEN, "make_grad_fixtures.py"))
F = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(F)
FIX = np.load(os.path.join(K.GOLDEN, "grad_fixtures.npz"))


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def _run(ion, dev, name, f32, **kw):
    model = F.CASES[name]
    pv, te, params, pot, y0, coef = F.problem(name)
    sdt = torch.float32 if f32 else torch.float64
    w = torch.from_numpy(K.load_weights(name).copy()).to(dev).requires_grad_(True)
    p = torch.from_numpy(params).to(dev).requires_grad_(True)
    y0t = torch.from_numpy(y0).to(dev).to(sdt).requires_grad_(True)
    y, status = ion.grad.solve(model, w, p, torch.from_numpy(pv).to(dev), y0t, torch.from_numpy(te).to(dev),
                               mlp_layers=K.MLP_L, mlp_width=K.MLP_N, prot_t0=0.0, prot_dt=1.0,
                               prot_of_traj=torch.from_numpy(pot).to(dev), **kw)
    assert (status == 0).all()
    loss = (y.double() * torch.from_numpy(coef).to(dev)).sum()
    loss.backward()
    return y, w.grad.double().cpu().numpy(), p.grad.cpu().numpy(), y0t.grad.double().cpu().numpy()


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("name", ["s1", "d2"])
def test_backward_sweep_against_checker_fixtures(ion, gpu, oracle, name, f32):
    tag = f"{name}_{'f32' if f32 else 'f64'}"
    y, gw, gp, gy0 = _run(ion, gpu, name, f32)
    # the differentiable forward is the ordinary forward: bit-identical to the oracle
    pv, te, params, pot, y0, _ = F.problem(name)
    o = oracle.solve(F.CASES[name], params, pv, np.float32(y0).astype(np.float64) if f32 else y0, te,
                     weights=K.load_weights(name), mlp_layers=5, mlp_width=200, prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot,
                     state_f32=f32)
    assert np.array_equal(y.detach().double().cpu().numpy(), o["y"])
    cols = slice(4, 8) if name == "s1" else slice(0, 8)  # NN-f has no p1..p4
    errs = {"dL/dp": _rel(gp[:, cols], FIX[tag + "_gp"][:, cols]), "dL/dy0": _rel(gy0, FIX[tag + "_gy0"]),
            "dL/dW sampled": _rel(gw[FIX["gw_idx"]], FIX[tag + "_gw_val"]),
            "|dL/dW| per tensor": _rel([np.linalg.norm(gw[a:b]) for a, b in F.tensor_slices()], FIX[tag + "_gw_norm"])}
    print(tag, {k: f"{v:.2e}" for k, v in errs.items()})
    if errs["dL/dp"] > GRAD_REL_TOL:
        print("gp (HIP)\n", gp, "\ngp (checker)\n", FIX[tag + "_gp"])
    if name == "s1":
        assert np.abs(gp[:, :4]).max() == 0.0
    assert max(errs.values()) <= GRAD_REL_TOL, errs


def test_chunked_sweep_equals_single_launch(ion, gpu):
    """record_budget_bytes forces the sweep + reduce into several chunks (adjoint state carried in HBM between launches)."""
    _, gw1, gp1, gy1 = _run(ion, gpu, "d2", False)
    _, gw2, gp2, gy2 = _run(ion, gpu, "d2", False, record_budget_bytes=40 * 6 * 40000 * 4)  # 40 iterations per chunk
    assert np.array_equal(gp1, gp2) and np.array_equal(gy1, gy2)      # the sweep itself is chunk-invariant
    assert _rel(gw2, gw1) < 1e-5                                      # slab boundaries move: fp32 summation order only


def test_small_checkpoint_buffer_is_regrown(ion, gpu):
    _, gw1, gp1, _ = _run(ion, gpu, "s1", False)
    _, gw2, gp2, _ = _run(ion, gpu, "s1", False, ckpt_cap=16)
    assert np.array_equal(gp1, gp2) and np.array_equal(gw1, gw2)


def _module(name):
    nm, npar = K.MODELS[name][3], K.MODELS[name][4]
    func = (M.NNf if nm == K.MODEL_NNF else M.NNd)(npar)
    M.load_flat_weights(func.net, K.load_weights(name))
    return func


@pytest.mark.parametrize("adjoint", [False, True, "capped"])
def test_dropin_odeint_gives_module_gradients(ion, gpu, adjoint):
    """`from torchdiffeq import odeint_adjoint as odeint` (train-s1.py:29-32) on a reference-style module: parameters of
    func.net receive gradients equal to the batched path's -- odeint and odeint_adjoint: the exact derivative of the reference step
    sequence (round 5: the adjoint name no longer changes the forward); adjoint_options={"max_step": "auto"}: the stabilised sweep
    (step sequence capped at grad.stable_step_cap, opt-in)."""
    import functools
    import torchdiffeq
    odeint = torchdiffeq.odeint_adjoint if adjoint else torchdiffeq.odeint
    if adjoint == "capped":
        odeint = functools.partial(torchdiffeq.odeint_adjoint, adjoint_options={"max_step": "auto"})
    func = _module("d2")
    pv, te, params, pot, y0, coef = F.problem("d2")
    func.set_fixed_form_voltage_protocol(np.arange(pv.shape[1], dtype=np.float64), pv[0])
    for i in range(8):
        setattr(func, f"p{i + 1}", float(params[0, i]))
    y0t = torch.tensor(y0[:1], requires_grad=True)              # CPU tensors, as the reference's scripts hold them
    y = odeint(func, y0t, torch.from_numpy(te))
    assert y.shape == (te.size, 1, 2) and y.requires_grad and y.device == y0t.device
    (y[:, 0, :] * torch.from_numpy(coef[0])).sum().backward()
    got = np.concatenate([np.concatenate([m.weight.grad.numpy().ravel(), m.bias.grad.numpy().ravel()])
                          for m in func.net if isinstance(m, torch.nn.Linear)])
    # same trajectory alone through the batched API
    w = torch.from_numpy(K.load_weights("d2").copy()).to(gpu).requires_grad_(True)
    yb, _ = ion.grad.solve(K.MODEL_NND, w, torch.from_numpy(params[:1]).to(gpu), torch.from_numpy(pv[:1]).to(gpu),
                           torch.from_numpy(y0[:1]).to(gpu), torch.from_numpy(te).to(gpu), mlp_layers=5, mlp_width=200,
                           prot_t0=0.0, prot_dt=1.0, max_step=("auto" if adjoint == "capped" else 0.0))
    (yb[0] * torch.from_numpy(coef[0]).to(gpu)).sum().backward()
    assert np.array_equal(y.detach().numpy()[:, 0, :], yb.detach().cpu().numpy()[0])
    assert _rel(got, w.grad.cpu().numpy()) < 1e-6 and y0t.grad is not None
    with torch.no_grad():                                        # the reference's own usage: no graph, plain path
        assert not odeint(func, y0t, torch.from_numpy(te)).requires_grad


@pytest.mark.parametrize("which", ["hh", "markov6"])
def test_dropin_odeint_differentiates_the_closed_form_modules(ion, gpu, which):
    """`odeint(Lambda(), y0, t)` with y0 requiring grad (train-s1.py:161-177 / train-d1.py:165-187 modules): d(weighted sum of
    the trace)/dy0 equals central finite differences of the same call (fp64 state; the step sequence is frozen in the
    derivative, free in the differences: agreement to the integrator's tolerance)."""
    from torchdiffeq import odeint
    if which == "hh":
        truth, y0v = M.HodgkinHuxley(K.P_HH), [[0.05, 0.9]]
    else:
        truth, y0v = M.Markov6(K.P_M6), [[0.05, 0.8, 0.02, 0.03, 0.05, 0.05]]
    tp, vp, _ = K.activation(20)
    truth.set_fixed_form_voltage_protocol(tp, vp)
    t = torch.linspace(0.0, 600.0, 61, dtype=torch.float64)
    y0 = torch.tensor(y0v, dtype=torch.float64, requires_grad=True)
    D = y0.shape[1]
    cw = torch.linspace(0.5, 1.5, D, dtype=torch.float64)
    out = odeint(truth, y0, t)
    assert out.requires_grad and out.shape == (61, 1, D)
    (out * cw).sum().backward()
    g = y0.grad.clone()
    with torch.no_grad():
        for d in range(D):
            e = torch.zeros_like(y0); e[0, d] = 1e-5
            fd = ((odeint(truth, (y0 + e).detach(), t) * cw).sum() - (odeint(truth, (y0 - e).detach(), t) * cw).sum()) / 2e-5
            assert abs(float(fd) - float(g[0, d])) <= 2e-4 * max(1.0, abs(float(fd))), (d, float(fd), float(g[0, d]))


def _rand_weights(L, N, seed):
    rng = np.random.default_rng(seed)  # gain ~1 per layer so that deep stacks stay O(1): sigma = 1 / sqrt(N), capped
    return rng.normal(0, min(0.3, 1.0 / np.sqrt(N)), 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)


@pytest.mark.parametrize("L,N,model", [(2, 10, K.MODEL_NNF), (10, 10, K.MODEL_NND), (3, 100, K.MODEL_NND), (10, 100, K.MODEL_NNF),
                                       (1, 200, K.MODEL_NNF), (10, 200, K.MODEL_NND), (15, 100, K.MODEL_NNF), (1, 500, K.MODEL_NNF), (5, 500, K.MODEL_NND)])
def test_other_widths_and_depths_against_the_checker(ion, gpu, oracle, L, N, model):
    """The NT = 1 (N = 10) and NT = 7 (N = 100) instantiations of the sweep / reduce kernels and other depths of the N = 200
    one, and the N = 500 one with its short weight ring and the column-blocked reduce (architectures s01-s11; 15 hidden layers
    is the sweep's limit), ragged batch (19 trajectories = 2 tiles), explicit protocol time grid, one
    trajectory that fails (NaN start: zero gradient) -- the checker (tests/grad_check.py) is evaluated in the test."""
    import grad_check as G
    w = _rand_weights(L, N, 11 * L + N)
    rng = np.random.default_rng(L + N)
    B = 19
    pv = np.stack([K.atau(30)[1][900:1300], K.atau(100)[1][900:1300]])
    pt = np.arange(400, dtype=np.float64) * 1.0
    pt[1:] += rng.uniform(-1e-7, 1e-7, 399)                  # not uniform in bits: the explicit-grid lookup
    te = np.arange(0.0, 140.0, 1.0)
    params = np.tile(K.P_HH, (B, 1)) * rng.uniform(0.9, 1.1, (B, 8))
    pot = (np.arange(B) % 2).astype(np.int32)
    y0 = np.stack([rng.uniform(0.0, 0.3, B), rng.uniform(0.6, 1.0, B)], 1)
    y0[5, 0] = np.nan
    coef = rng.normal(size=(B, te.size, 2))
    wt = torch.from_numpy(w.copy()).to(gpu).requires_grad_(True)
    p = torch.from_numpy(params).to(gpu).requires_grad_(True)
    y0t = torch.from_numpy(y0).to(gpu).requires_grad_(True)
    y, status = ion.grad.solve(model, wt, p, torch.from_numpy(pv).to(gpu), y0t, torch.from_numpy(te).to(gpu), mlp_layers=L,
                               mlp_width=N, prot_t=torch.from_numpy(pt).to(gpu), prot_of_traj=torch.from_numpy(pot).to(gpu))
    st = status.cpu().numpy()
    assert st[5] != 0 and (np.delete(st, 5) == 0).all()
    ok = torch.from_numpy(st == 0).to(gpu)
    (torch.nan_to_num(y) * torch.from_numpy(coef).to(gpu) * ok[:, None, None]).sum().backward()
    gw, gp, gy0 = wt.grad.double().cpu().numpy(), p.grad.cpu().numpy(), y0t.grad.cpu().numpy()
    assert np.all(gp[5] == 0) and np.all(gy0[5] == 0)         # the failed trajectory contributes nothing
    # checker: three of the healthy trajectories (dL/dp, dL/dy0 per trajectory; dL/dW needs all of them)
    flat = torch.from_numpy(w.copy()).requires_grad_(True)
    torch.set_num_threads(8)
    for b in [b for b in range(B) if b != 5]:
        o = oracle.solve(model, params[b], pv[pot[b]], y0[b], te, weights=w, mlp_layers=L, mlp_width=N, prot_t=pt, step_log_cap=8192)
        assert np.array_equal(y[b].detach().cpu().numpy(), o["y"][0])
        pb = torch.tensor(params[b], dtype=torch.float64, requires_grad=True)
        yb = torch.tensor(y0[b], dtype=torch.float64, requires_grad=True)
        yr = G.replay(model, flat, L, N, pb, yb, pt, pv[pot[b]], te, G.accepted_steps(o["step_log"]))
        (yr * torch.from_numpy(coef[b])).sum().backward()
        cols = slice(4, 8) if model == K.MODEL_NNF else slice(0, 8)
        assert _rel(gp[b, cols], pb.grad.numpy()[cols]) <= GRAD_REL_TOL and _rel(gy0[b], yb.grad.numpy()) <= GRAD_REL_TOL
    e = _rel(gw, flat.grad.double().numpy())
    print(f"L={L} N={N}: dL/dW rel-L2 vs checker {e:.2e}")
    assert e <= GRAD_REL_TOL


def test_gradient_of_unsupported_shapes_is_refused(ion, gpu):
    pv = torch.zeros((1, 100), dtype=torch.float64, device=gpu) - 80.0
    for L, N in ((2, 300), (16, 100)):    # a width outside architectures s00-s11; more than 15 hidden layers
        w = torch.from_numpy(_rand_weights(L, N, 1)).to(gpu).requires_grad_(True)
        with pytest.raises(ion.IonodeError, match="variants|width"):
            y, _ = ion.grad.solve(K.MODEL_NNF, w, torch.from_numpy(K.P_HH[None]).to(gpu), pv,
                                  torch.tensor([[0.0, 1.0]], dtype=torch.float64, device=gpu),
                                  torch.arange(10, dtype=torch.float64, device=gpu), mlp_layers=L, mlp_width=N)
            y.sum().backward()


def test_config5_share_at_full_size_gradient_is_additive(ion, gpu):
    """BASELINE configs[4], one GPU's share at full size (1024 of the 8192 trajectories, fp32 state, 100 001 samples): no
    checker runs at this size, but the loss is a sum over trajectories, so dL/dW of the whole share must equal the sum of the
    two half-shares' gradients (different chunking, different split-K order: fp32 accumulation tolerance 2e-5), the forward
    states must be the ordinary forward's bits, and the gradient must be finite and non-trivial."""
    P = ion.protocols
    B, Nt = 1024, 100001
    pv = P.sinewave(P.sinewave_scales(0, B), n_samples=Nt, dt=0.1, xp=torch, device=gpu)
    te = torch.arange(Nt, dtype=torch.float64, device=gpu) * 0.1
    vobs = pv + 86.0
    weights = K.load_weights("s1")
    params = torch.from_numpy(np.tile(K.P_HH, (B, 1))).to(gpu)
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float32, device=gpu).repeat(B, 1)

    def grad_of(lo, hi):
        w = torch.from_numpy(weights.copy()).to(gpu).requires_grad_(True)
        y, status = ion.grad.solve(K.MODEL_NNF, w, params[lo:hi], pv[lo:hi], y0[lo:hi], te, mlp_layers=5, mlp_width=200,
                                   prot_t0=0.0, prot_dt=0.1, t_eval_hint=(0.0, 0.1))
        assert bool((status == 0).all())
        ((y[..., 0] * y[..., 1]).double() * vobs[lo:hi]).abs().sum().backward()      # sum_b sum_k |i_bk|  (train-s1.py:329 x N)
        probe = y[:4, ::1000].detach().clone()
        del y
        torch.cuda.empty_cache()
        return w.grad.double(), probe

    g_all, probe = grad_of(0, B)
    g_a, _ = grad_of(0, B // 2)
    g_b, _ = grad_of(B // 2, B)
    assert bool(torch.isfinite(g_all).all()) and float(g_all.norm()) > 0
    assert float((g_all - (g_a + g_b)).norm() / g_all.norm()) < 2e-5
    plain = ion.solve(K.MODEL_NNF, params[:4], pv[:4], y0[:4], te, weights=weights, mlp_layers=5, mlp_width=200, prot_t0=0.0,
                      prot_dt=0.1, t_eval_hint=(0.0, 0.1))
    assert torch.equal(plain.y[:, ::1000], probe)


def test_launch_order_leaves_gradients_in_the_callers_order(ion, gpu):
    """grad.solve(order=perm): states come back in launch order, gradients at params / y0 / weights in the caller's order
    (bit-identical per trajectory for dL/dp and dL/dy0, which are accumulated per trajectory; dL/dW sums over trajectories in
    a different order: 1e-5)."""
    pv, te, params, pot, y0, coef = F.problem("s1")
    B = params.shape[0]
    order = torch.from_numpy(np.random.default_rng(4).permutation(B)).to(gpu)

    def run(order_):
        w = torch.from_numpy(K.load_weights("s1").copy()).to(gpu).requires_grad_(True)
        p = torch.from_numpy(params).to(gpu).requires_grad_(True)
        y0t = torch.from_numpy(y0).to(gpu).requires_grad_(True)
        y, status = ion.grad.solve(F.CASES["s1"], w, p, torch.from_numpy(pv).to(gpu), y0t, torch.from_numpy(te).to(gpu),
                                   mlp_layers=K.MLP_L, mlp_width=K.MLP_N, prot_t0=0.0, prot_dt=1.0,
                                   prot_of_traj=torch.from_numpy(pot).to(gpu), order=order_)
        c = torch.from_numpy(coef).to(gpu)
        (y.double() * (c if order_ is None else c.index_select(0, order_))).sum().backward()
        return y.detach(), w.grad.double(), p.grad, y0t.grad

    ya, gwa, gpa, gya = run(None)
    yb, gwb, gpb, gyb = run(order)
    assert torch.equal(yb, ya.index_select(0, order))
    assert torch.equal(gpa, gpb) and torch.equal(gya, gyb)
    assert float((gwa - gwb).norm() / gwa.norm()) < 1e-5


@pytest.mark.parametrize("model", [K.MODEL_HH2, K.MODEL_MARKOV6])
@pytest.mark.parametrize("f32", [False, True])
def test_closed_form_gradients_against_the_checker(ion, gpu, oracle, f32, model):
    """The backward sweep of the closed-form models (HH 2-state, train-s1.py:161-177, and 6-state, train-d1.py:165-187; the same
    kernel without the MLP collective): dL/dp and dL/dy0 of every trajectory against autograd through the torch replay of the oracle's accepted steps (evaluated here);
    ragged batch, uniform and explicit protocol grids, one failing trajectory, chunked sweep.  fp32 state: the replay is
    anchored on the forward's own states (as for the fixtures), times formed in fp32."""
    import grad_check as G
    rng = np.random.default_rng(41 + int(f32))
    B = 37
    pv = np.stack([K.atau(30)[1][900:1300], K.atau(100)[1][900:1300], K.activation(20)[1][:400]])
    te = np.arange(0.0, 140.0, 1.0)
    m6 = model == K.MODEL_MARKOV6
    D = 6 if m6 else 2
    params = np.tile(K.P_M6 if m6 else K.P_HH, (B, 1)) * rng.uniform(0.8, 1.25, (B, 12 if m6 else 8))
    pot = rng.integers(0, 3, B).astype(np.int32)
    y0 = np.stack([rng.uniform(0.0, 0.3, B), rng.uniform(0.6, 1.0, B)], 1)
    if m6:
        y0 = np.concatenate([y0, rng.uniform(0.0, 0.1, (B, 4))], 1)
    if f32:
        y0 = y0.astype(np.float32).astype(np.float64)
    y0[11, 1] = np.nan
    coef = rng.normal(size=(B, te.size, D))
    sdt = torch.float32 if f32 else torch.float64
    for pt in (None, np.arange(400, dtype=np.float64) + np.concatenate([[0.0], rng.uniform(-1e-7, 1e-7, 399)])):
        p = torch.from_numpy(params).to(gpu).requires_grad_(True)
        y0t = torch.from_numpy(y0).to(gpu).to(sdt).requires_grad_(True)
        y, status = ion.grad.solve(model, None, p, torch.from_numpy(pv).to(gpu), y0t, torch.from_numpy(te).to(gpu),
                                   prot_t=None if pt is None else torch.from_numpy(pt).to(gpu), prot_t0=0.0, prot_dt=1.0,
                                   prot_of_traj=torch.from_numpy(pot).to(gpu), ckpt_cap=16)        # too small: regrown
        st = status.cpu().numpy()
        assert st[11] != 0 and (np.delete(st, 11) == 0).all()
        ok = torch.from_numpy(st == 0).to(gpu)
        (torch.nan_to_num(y.double()) * torch.from_numpy(coef).to(gpu) * ok[:, None, None]).sum().backward()
        gp, gy0 = p.grad.cpu().numpy(), y0t.grad.double().cpu().numpy()
        assert np.all(gp[11] == 0) and np.all(gy0[11] == 0)
        ptx = np.arange(400, dtype=np.float64) if pt is None else pt
        worst = 0.0
        for b in [b for b in range(B) if b != 11][::3]:
            o = oracle.solve(model, params[b], pv[pot[b]], y0[b], te, prot_t=pt, prot_t0=0.0, prot_dt=1.0, state_f32=f32,
                             step_log_cap=8192)
            assert np.array_equal(y[b].detach().double().cpu().numpy(), o["y"][0])      # the differentiable forward is the forward
            steps = G.accepted_steps(o["step_log"])
            pb = torch.tensor(params[b], dtype=torch.float64, requires_grad=True)
            yb = torch.tensor(y0[b], dtype=torch.float64, requires_grad=True)
            anchors = None
            if f32:   # end states of the accepted steps, from a second oracle run that reports every step end
                ends = np.array([t0 + dt for t0, dt in steps])
                anchors = oracle.solve(model, params[b], pv[pot[b]], y0[b], np.concatenate([[te[0]], ends]), prot_t=pt,
                                       prot_t0=0.0, prot_dt=1.0, state_f32=True)["y"][0][1:]
            yr = G.replay(model, None, 0, 0, pb, yb, ptx, pv[pot[b]], te, steps, f32_times=f32, anchors=anchors)
            (yr * torch.from_numpy(coef[b])).sum().backward()
            worst = max(worst, _rel(gp[b], pb.grad.numpy()), _rel(gy0[b], yb.grad.numpy()))
        print(f"model {model} {'f32' if f32 else 'f64'} {'explicit' if pt is not None else 'uniform'} grid: worst rel-L2 vs checker {worst:.2e}")
        assert worst <= GRAD_REL_TOL


@pytest.mark.parametrize("model,L,N", [(K.MODEL_HH2, 0, 0), (K.MODEL_MARKOV6, 0, 0), (K.MODEL_NNF, 5, 10)])
def test_checkpoints_of_the_64_per_wavefront_kernels(ion, gpu, model, L, N):
    """The forward variants that integrate one trajectory per lane (closed-form models from ~40-80 k trajectories, N <= 16 nets from
    73 728; forced here with tile_waves = 64) write the same accepted-step checkpoints as the 16-per-wavefront ones: states and
    gradients of a ragged batch are bit-identical between the two geometries."""
    rng = np.random.default_rng(3 + model)
    B = 150
    m6 = model == K.MODEL_MARKOV6
    pv = np.stack([K.atau(30)[1][900:1300], K.atau(100)[1][900:1300]])
    te = np.arange(0.0, 140.0, 1.0)
    params = np.tile(K.P_M6 if m6 else K.P_HH, (B, 1)) * rng.uniform(0.9, 1.1, (B, 12 if m6 else 8))
    y0 = np.tile([0.05, 0.9] + ([0.01] * 4 if m6 else []), (B, 1))
    w = None if N == 0 else _rand_weights(L, N, 5)
    coef = torch.from_numpy(rng.normal(size=(B, te.size, y0.shape[1]))).to(gpu)
    res = []
    for tw in (64, 16 if N == 0 else 1):
        p = torch.from_numpy(params).to(gpu).requires_grad_(True)
        y0t = torch.from_numpy(y0).to(gpu).requires_grad_(True)
        wt = None if w is None else torch.from_numpy(w.copy()).to(gpu).requires_grad_(True)
        y, st = ion.grad.solve(model, wt, p, torch.from_numpy(pv).to(gpu), y0t, torch.from_numpy(te).to(gpu), mlp_layers=L, mlp_width=N,
                               prot_t0=0.0, prot_dt=1.0, tile_waves=tw)
        assert bool((st == 0).all())
        (y * coef).sum().backward()
        res.append((y.detach(), p.grad.clone(), y0t.grad.clone(), None if wt is None else wt.grad.clone()))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    if a[3] is not None:
        assert float((a[3] - b[3]).double().norm() / b[3].double().norm()) < 1e-6
