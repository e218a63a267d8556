"""CPU: the gradient checker (tests/grad_check.py) is self-consistent, and the committed gradient fixtures are what it
produces.  The backward sweep's scalar algebra (`manual_adjoint`, the hand-written reverse of the dopri5 step with FSAL
carry and dense-output adjoint that the HIP kernel implements) must equal autograd through the replayed discretisation."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import grad_check as G
import kat_cases as K


def _rand_weights(L, N, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(0, 0.3, 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)


@pytest.mark.parametrize("model,f32", [(K.MODEL_NNF, False), (K.MODEL_NND, False), (K.MODEL_NND, True), (K.MODEL_HH2, False)])
def test_manual_adjoint_equals_autograd(oracle, model, f32):
    L, N = 2, 10
    w = _rand_weights(L, N, 3)
    pv = K.atau(30)[1][900:1400]
    pt = np.arange(pv.size, dtype=np.float64)
    te = np.arange(0.0, 300.0, 3.0)
    p = K.P_HH * np.random.default_rng(1).uniform(0.9, 1.1, 8)
    kw = dict(weights=w, mlp_layers=L, mlp_width=N) if model != K.MODEL_HH2 else {}
    o = oracle.solve(model, p, pv, [0.1, 0.8], te, prot_t0=0.0, prot_dt=1.0, state_f32=f32, step_log_cap=4096, **kw)
    steps = G.accepted_steps(o["step_log"])
    assert len(steps) == o["stats"][0, 0] > 30
    flat = torch.from_numpy(w.copy()).double().requires_grad_(True)
    pp = torch.tensor(p, dtype=torch.float64, requires_grad=True)
    y0 = torch.tensor([0.1, 0.8], dtype=torch.float64, requires_grad=True)
    y = G.replay(model, flat, L, N, pp, y0, pt, pv, te, steps, f32_times=f32, net_dtype=torch.float64)
    assert np.abs(y.detach().numpy() - o["y"][0]).max() < (1e-4 if f32 else 1e-6)  # the replay is the oracle's solve
    gy = torch.from_numpy(np.random.default_rng(2).normal(size=tuple(y.shape)))
    (y * gy).sum().backward()
    gf, gp, gy0 = G.manual_adjoint(model, flat, L, N, pp, y0, pt, pv, te, steps, gy, f32_times=f32, net_dtype=torch.float64)

    def rel(a, b):
        return float((a - b).norm() / b.norm())
    assert rel(gp, pp.grad) < 1e-10 and rel(gy0, y0.grad) < 1e-10
    if model != K.MODEL_HH2:
        assert rel(gf, flat.grad) < 1e-10


@pytest.mark.slow
def test_committed_gradient_fixture_is_the_checkers_output(oracle):
    spec = importlib.util.spec_from_file_location("make_grad_fixtures", os.path.join(K.GOLDEN, "make_grad_fixtures.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    fix = np.load(os.path.join(K.GOLDEN, "grad_fixtures.npz"))
    torch.set_num_threads(4)
    gw, gp, gy0 = M.checker_gradients("d2", False)
    assert np.allclose(gp, fix["d2_f64_gp"], rtol=1e-9, atol=0) and np.allclose(gy0, fix["d2_f64_gy0"], rtol=1e-9, atol=0)
    assert np.allclose(gw[fix["gw_idx"]], fix["d2_f64_gw_val"], rtol=1e-6, atol=1e-9)
