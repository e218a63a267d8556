#!/usr/bin/env python3
"""Gradient fixtures for the backward sweep (BASELINE config 5).

**Parity unpinned**: the reference never differentiates through `odeint` (SURVEY.md finding 3), so these numbers come
from the checker, not from the reference: autograd (fp64) through tests/grad_check.replay() -- a torch restatement of the
discretisation replaying the CPU oracle's accepted-step log, which the HIP forward reproduces bit for bit.

    python tests/golden/make_grad_fixtures.py      ->  tests/golden/grad_fixtures.npz

fp32-state cases: the replay is anchored on the fp32 forward's actual step-end states (grad_check.replay `anchors`), so
the Jacobians are evaluated along the trajectory the forward really took, as the HIP sweep does from its checkpoints.

Per case (s1 = NN-f, d2 = NN-d with the reference's trained weights; fp64 and fp32 solver state; 3 trajectories with
their own rate parameters and protocols): loss = sum_b sum_k c_bk . y_b(t_k) with seeded coefficients c, and
  gp  [3, 8]  dL/dp1..p8        gy0 [3, 2]  dL/dy0
  gw_idx / gw_val  4096 seeded entries of dL/dW (flat state-dict order),  gw_norm  per-tensor L2 norms of dL/dW
The full dL/dW (807 KB per case) is not stored; the sampled entries and the norms pin it.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import grad_check as G  # noqa: E402
import kat_cases as K  # noqa: E402
from oracle import oracle  # noqa: E402

L, N = K.MLP_L, K.MLP_N
CASES = {"s1": K.MODEL_NNF, "d2": K.MODEL_NND}


def problem(name):
    """Inputs of one case: 3 trajectories, 2 protocols (windows around the voltage steps of two tau protocols)."""
    rng = np.random.default_rng({"s1": 21, "d2": 22}[name])
    pv = np.stack([K.atau(30)[1][900:1400], K.atau(100)[1][900:1400]])
    # Output window 0..149 ms (voltage steps at 100 and 130 ms).  It deliberately stops before the long -120 mV hold has
    # equilibrated: there dopri5 grows dt until h*lambda ~ 20 (the error estimate of a state sitting AT equilibrium is ~0),
    # far outside the stability region, and the derivative of such an accepted-but-unstable step multiplies fp32-level
    # differences of the state by ~1e6 -- dL/dp7 of one trajectory then reads -0.2, -7 or +80 depending on which
    # 1e-7-different fp32 state it is linearised at (measured with the first version of this fixture).  That is a
    # property of the discretisation in fp32 state, not of either implementation, and makes a useless test vector.
    te = np.arange(0.0, 150.0, 1.0)
    params = np.tile(K.MODELS[name][4], (3, 1)) * rng.uniform(0.9, 1.1, (3, 8))
    pot = np.array([0, 1, 0], dtype=np.int32)
    y0 = np.array([[0.0, 1.0], [0.05, 0.9], [0.2, 0.7]])
    coef = rng.normal(size=(3, te.size, 2))
    return pv, te, params, pot, y0, coef


def tensor_slices():
    sl, off = [], 0
    for (o, i) in [(N, 2)] + [(N, N)] * L + [(1, N)]:
        sl += [(off, off + o * i), (off + o * i, off + o * i + o)]
        off += o * i + o
    return sl


def checker_gradients(name, f32):
    model = CASES[name]
    w = K.load_weights(name)
    pv, te, params, pot, y0, coef = problem(name)
    prot_t = np.arange(pv.shape[1], dtype=np.float64)
    flat = torch.from_numpy(w.copy()).requires_grad_(True)
    gp, gy0 = [], []
    for b in range(3):
        y0b = np.float32(y0[b]).astype(np.float64) if f32 else y0[b]
        o = oracle.solve(model, params[b], pv[pot[b]], y0b, te, weights=w, mlp_layers=L, mlp_width=N, prot_t0=0.0,
                         prot_dt=1.0, state_f32=f32, step_log_cap=8192)
        assert o["status"][0] == 0
        steps = G.accepted_steps(o["step_log"])
        anchors = None
        if f32:
            # the fp32 forward's actual state at the end of every accepted step: the same solve with the step ends as
            # output times (steps are never clipped to output times, so the step sequence is unchanged)
            ends = steps[:, 0] + steps[:, 1]
            keep = ends <= te[-1] + (ends[-1] - te[-1]) + 0.0
            oa = oracle.solve(model, params[b], pv[pot[b]], y0b, np.concatenate([[te[0]], ends[keep]]), weights=w,
                              mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=1.0, state_f32=True)
            assert np.array_equal(oa["stats"][0], o["stats"][0])
            anchors = oa["y"][0][1:]
        p = torch.tensor(params[b], dtype=torch.float64, requires_grad=True)
        y0t = torch.tensor(y0b, dtype=torch.float64, requires_grad=True)
        y = G.replay(model, flat, L, N, p, y0t, prot_t, pv[pot[b]], te, steps, f32_times=f32, anchors=anchors)
        dmax = np.abs(y.detach().numpy() - o["y"][0]).max(); print("replay vs oracle max abs", dmax)
        assert dmax < (2e-4 if f32 else 1e-6)  # the replay IS the oracle's solve (fp32 state: up to its rounding noise)
        (y * torch.from_numpy(coef[b])).sum().backward()
        gp.append(p.grad.numpy().copy())
        gy0.append(y0t.grad.numpy().copy())
    return flat.grad.double().numpy(), np.stack(gp), np.stack(gy0)


def main():
    oracle.build()
    torch.set_num_threads(4)
    out = {}
    n = 2 * N + N + L * (N * N + N) + N + 1
    idx = np.sort(np.random.default_rng(5).choice(n, 4096, replace=False))
    out["gw_idx"] = idx
    for name in CASES:
        for f32 in (False, True):
            gw, gp, gy0 = checker_gradients(name, f32)
            tag = f"{name}_{'f32' if f32 else 'f64'}"
            out[tag + "_gp"], out[tag + "_gy0"] = gp, gy0
            out[tag + "_gw_val"] = gw[idx]
            out[tag + "_gw_norm"] = np.array([np.linalg.norm(gw[a:b]) for a, b in tensor_slices()])
            print(tag, "|dW|", np.linalg.norm(gw), "gp[0]", gp[0][:4], "gy0", gy0[0])
    np.savez_compressed(os.path.join(HERE, "grad_fixtures.npz"), **out)


if __name__ == "__main__":
    main()
