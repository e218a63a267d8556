"""-m gpu: seeded random sweep of the gradient path against the checker (tests/grad_check.py, evaluated here).

Cases come from the forward sweep's generator (tests/test_gpu_fuzz.py: model, state dtype, MLP shape, batch size, uniform or
explicit protocol grids, uniform / inexact / irregular / two-point / beyond-the-protocol output grids, tolerances, step limits
that trip sometimes, dt cap, a NaN start), restricted to batches the CPU replay finishes in seconds.  Loss = sum(coef * y) over
the trajectories that succeeded; dL/dp and dL/dy0 of every checked trajectory and dL/dW (NN models: all trajectories) must
agree with autograd through the replay of the oracle's accepted steps to GRAD_REL_TOL (fp32 state: anchored replay)."""
import os

import numpy as np
import pytest
import torch

import grad_check as G
import kat_cases as K
from test_gpu_fuzz import _case

pytestmark = pytest.mark.gpu
GRAD_REL_TOL = 1e-4


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# A contiguous seed range, no curation (round-2 review): every seed of range(20) runs; a case may skip itself only for the two
# reasons written in the body (a single output time, every trajectory failing).  The round-3 log of all twenty, with per-seed
# agreement and durations, is profiles/r03_grad_fuzz.log.
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("IONODE_GRAD_FUZZ_SEED0", "0")), int(os.environ.get("IONODE_GRAD_FUZZ_SEED0", "0")) + 20)))   # (env: another block of twenty, for one-off wider sweeps)
def test_random_gradients_match_the_checker(ion, gpu, oracle, seed):
    model, f32, params, pv, y0, te, kw, mlp, obs, rng = _case(seed)
    B = min(params.shape[0], 6 if (f32 and mlp) else 17)   # (the capped fp32 replays of an MLP are the slow ones on the CPU)
    params, y0 = params[:B], y0[:B]
    if f32:
        # fp32 state: the exact derivative of accepted-but-unstable steps at an equilibrium (h * lambda >> 1) turns the
        # forward's own rounding noise into percent-level differences of dL/dp between any two evaluations; the dt cap is the remedy
        kw["max_step"] = 1.0   # < 3.3 / lambda_max for rates up to 1.4 x nominal at -120 mV
        kw.pop("max_steps", None); kw.pop("max_total_steps", None)   # (the capped steps would trip the sweep's small step limits)
    pot = kw.pop("prot_of_traj", None)
    pot = (np.arange(B) % pv.shape[0]).astype(np.int32) if pot is None else pot[:B]
    if te.size < 2:
        pytest.skip("a single output time has no step to differentiate")
    L, N = mlp.get("mlp_layers", 0), mlp.get("mlp_width", 0)
    w = mlp.get("weights")
    coef = rng.normal(size=(B, te.size, y0.shape[1]))
    sdt = torch.float32 if f32 else torch.float64
    dev_kw = dict(kw)
    pt = dev_kw.pop("prot_t")
    p = torch.from_numpy(params).to(gpu).requires_grad_(True)
    y0t = torch.from_numpy(y0).to(gpu).to(sdt).requires_grad_(True)
    wt = None if w is None else torch.from_numpy(w.copy()).to(gpu).requires_grad_(True)
    y, status = ion.grad.solve(model, wt, p, torch.from_numpy(pv).to(gpu), y0t, torch.from_numpy(te).to(gpu), mlp_layers=L, mlp_width=N,
                               prot_t=None if pt is None else torch.from_numpy(pt).to(gpu),
                               prot_of_traj=torch.from_numpy(pot).to(gpu), **dev_kw)
    st = status.cpu().numpy()
    ok = st == 0
    if not ok.any():
        pytest.skip("every trajectory of this case fails")
    okt = torch.from_numpy(ok).to(gpu)
    (torch.nan_to_num(y.double()) * torch.from_numpy(coef).to(gpu) * okt[:, None, None]).sum().backward()
    gp, gy0 = p.grad.cpu().numpy(), y0t.grad.double().cpu().numpy()
    assert np.all(gp[~ok] == 0) and np.all(gy0[~ok] == 0)
    flat = None if w is None else torch.from_numpy(w.copy()).requires_grad_(True)
    ptx = pt if pt is not None else kw["prot_t0"] + np.arange(pv.shape[1]) * kw["prot_dt"]
    okw = {k: v for k, v in kw.items() if k != "prot_t"}
    torch.set_num_threads(8)
    worst = 0.0
    for b in np.nonzero(ok)[0]:
        o = oracle.solve(model, params[b], pv[pot[b]], y0[b], te, prot_t=pt, state_f32=f32, step_log_cap=1 << 16, **okw, **mlp)
        assert o["status"][0] == 0 and np.array_equal(y[b].detach().double().cpu().numpy(), o["y"][0])
        steps = G.accepted_steps(o["step_log"])
        anchors = None
        if f32:
            ends = np.array([t0 + dt for t0, dt in steps])
            anchors = oracle.solve(model, params[b], pv[pot[b]], y0[b], np.concatenate([[te[0]], ends]), prot_t=pt, state_f32=True,
                                   **okw, **mlp)["y"][0][1:]
        pb = torch.tensor(params[b], dtype=torch.float64, requires_grad=True)
        yb = torch.tensor(y0[b], dtype=torch.float64, requires_grad=True)
        yr = G.replay(model, flat, L, N, pb, yb, ptx, pv[pot[b]], te, steps, f32_times=f32, anchors=anchors)
        (yr * torch.from_numpy(coef[b])).sum().backward()
        cols = slice(4, 8) if model == K.MODEL_NNF else slice(0, params.shape[1])
        e1, e2 = _rel(gp[b, cols], pb.grad.numpy()[cols]), _rel(gy0[b], yb.grad.numpy())
        if max(e1, e2) > GRAD_REL_TOL:
            print(f"  trajectory {b}: dL/dp {e1:.2e} |{np.linalg.norm(pb.grad.numpy()[cols]):.3e}|  dL/dy0 {e2:.2e} |{np.linalg.norm(yb.grad.numpy()):.3e}|"
                  f"  steps {len(steps)}  nfe {int(o['stats'][0, 2])}  max|y| {np.abs(o['y']).max():.3g}")
        worst = max(worst, e1, e2)
    if flat is not None:
        worst = max(worst, _rel(wt.grad.double().cpu().numpy(), flat.grad.double().numpy()))
    print(f"seed {seed}: model {model} {'f32' if f32 else 'f64'} L={L} N={N} B={B} ok={int(ok.sum())} worst rel-L2 {worst:.2e}")
    assert worst <= GRAD_REL_TOL
