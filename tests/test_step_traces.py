"""Per-step trace fixtures (tests/golden/step_traces.npz): every step attempt's (t0, dt, error ratio,
accepted) and the dense output for three short protocols, fp64 and fp32 state.

CPU: the oracle still reproduces them bit for bit (a changed canonical order / transcendental / controller must be a
deliberate regeneration, followed by tests/test_oracle_kats.py).  GPU: the HIP kernel's `step_log` equals them bit
for bit -- the step SEQUENCE, not only the end result, is identical."""
import importlib.util
import os

import numpy as np
import pytest

import kat_cases as K

_spec = importlib.util.spec_from_file_location("make_step_traces", os.path.join(K.GOLDEN, "make_step_traces.py"))
T = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(T)

FIX = np.load(os.path.join(K.GOLDEN, "step_traces.npz"))
NAMES = sorted(T.cases())


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_step_traces(oracle, name, f32):
    tag = f"{name}_{'f32' if f32 else 'f64'}"
    r = T.run(name, f32)
    assert np.array_equal(r["step_log"], FIX[tag + "_steps"])
    assert np.array_equal(r["y"][0], FIX[tag + "_y"].astype(np.float64))
    steps = FIX[tag + "_steps"]
    # a trace is a valid dopri5 history: accepted steps tile the time axis, rejected ones repeat their t0
    acc = steps[:, 3] == 1.0
    t_next = np.where(acc, steps[:, 0] + steps[:, 1], steps[:, 0])
    assert np.array_equal(steps[1:, 0], t_next[:-1]) and (steps[acc, 2] <= 1.0).all() and (steps[~acc, 2] > 1.0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("name", NAMES)
def test_hip_kernel_step_log_equals_fixture(ion, gpu, name, f32):
    import torch
    from gpu_util import run_gpu
    tag = f"{name}_{'f32' if f32 else 'f64'}"
    model, p, y0, wname, pk, pv, te = T.cases()[name]
    kw = dict(pk)
    if wname:
        kw.update(weights=K.load_weights(wname), L=K.MLP_L, N=K.MLP_N)
    slog = torch.full((4096, 4), float("nan"), dtype=torch.float64, device=gpu)
    g = run_gpu(ion, gpu, model, p, pv, y0, te, f32=f32, step_log=slog, **kw)
    n = int(g["stats"][0, 0] + g["stats"][0, 1])
    want = FIX[tag + "_steps"]
    assert n == want.shape[0]
    assert np.array_equal(slog[:n].cpu().numpy(), want)
    assert np.array_equal(g["y"][0], FIX[tag + "_y"].astype(np.float64))
