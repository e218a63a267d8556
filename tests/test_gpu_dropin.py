"""-m gpu: the drop-in boundary.  `from torchdiffeq import odeint` + reference-style RHS modules, driven the way
train-s1.py:319-330 drives them, reproduce the reference's logged prediction losses and the oracle bit for bit."""
import numpy as np
import pytest
import torch

import kat_cases as K
import ref_style_modules as M

pytestmark = pytest.mark.gpu


def _modules(name):
    tm, tp, ty0, nm, npar = K.MODELS[name]
    truth = M.HodgkinHuxley(tp) if tm == K.MODEL_HH2 else M.Markov6(tp)
    func = (M.NNf if nm == K.MODEL_NNF else M.NNd)(npar)
    M.load_flat_weights(func.net, K.load_weights(name))
    func.eval()
    return truth, torch.tensor([ty0]), func, torch.tensor([K.NN_Y0])


@pytest.mark.parametrize("name", ["s1", "d2"])
def test_reference_prediction_flow_reproduces_logged_losses(ion, gpu, oracle, name):
    from torchdiffeq import odeint  # the shim at the repository root

    kats = K.load_kats()
    truth, ty0, func, y0 = _modules(name)
    tm, tp, _, nm, npar = K.MODELS[name]
    cases = [K.all_cases()[i] for i in (0, 4, 12, 20)]  # AP 2 Hz, act 0 mV, deact -80 mV, tau 30 ms
    with torch.no_grad():
        for sec, key, (pt, pv, te) in cases:
            t = torch.from_numpy(te).float()  # the reference's fp32 linspace grids
            truth.set_fixed_form_voltage_protocol(pt, pv)
            yt = odeint(truth, ty0, t, method="dopri5")
            func.set_fixed_form_voltage_protocol(pt, pv)
            yp = odeint(func, y0, t).to("cpu")
            assert yp.shape == (t.numel(), 1, 2) and yp.dtype == torch.float32 and yt.shape[1:] == ty0.shape
            v = func._v(t) + 86
            gate_t = yt[:, 0, -1] if tm == K.MODEL_MARKOV6 else yt[:, 0, 0] * yt[:, 0, 1]
            loss = torch.mean(torch.abs(yp[:, 0, 0] * yp[:, 0, 1] * v - gate_t.cpu() * v)).item()
            assert abs(loss - K.expected(kats, name, sec, key)) <= K.KAT_ABS_TOL, (name, sec, key, loss)
            # and the same call is the oracle's fp32-state solve, bit for bit
            o = oracle.solve(nm, npar, pv, K.NN_Y0, te, prot_t=None if sec != "AP 2Hz" else None, prot_t0=float(pt[0]),
                             prot_dt=float((pt[-1] - pt[0]) / (pt.size - 1)), weights=K.load_weights(name),
                             mlp_layers=5, mlp_width=200, state_f32=True)
            assert np.array_equal(yp[:, 0, :].double().numpy(), o["y"][0])


def test_protocol_is_read_at_call_time_and_weights_updates_are_seen(ion, gpu):
    from torchdiffeq import odeint
    _, _, func, y0 = _modules("s1")
    t = torch.linspace(0.0, 2000.0, 201)
    pt, pv, _ = K.activation(40)
    func.set_fixed_form_voltage_protocol(pt, pv)
    a = odeint(func, y0, t)
    func.set_fixed_form_voltage_protocol(*K.activation(-40)[:2])
    b = odeint(func, y0, t)
    assert not torch.equal(a, b)
    func.set_fixed_form_voltage_protocol(pt, pv)
    assert torch.equal(odeint(func, y0, t), a)
    with torch.no_grad():
        func.net[12].bias += 0.5  # in-place update of a cached weight tensor must invalidate the packed image
    assert not torch.equal(odeint(func, y0, t), a)


def test_failures_raise_torchdiffeq_assertions(ion, gpu):
    from torchdiffeq import odeint
    truth, ty0, _, _ = _modules("s1")
    truth.set_fixed_form_voltage_protocol(*K.activation(20)[:2])
    t = torch.linspace(0.0, 8000.0, 801)
    # torchdiffeq counts max_num_steps per output interval (its _advance restarts the counter): the first 10 ms interval
    # needs more than 3 attempts (dt grows from ~1e-4 by at most 10x per step), none needs 500
    with pytest.raises(AssertionError, match="max_num_steps exceeded"):
        odeint(truth, ty0, t, options={"max_num_steps": 3})
    assert torch.equal(odeint(truth, ty0, t, options={"max_num_steps": 500}), odeint(truth, ty0, t))
    with pytest.raises(AssertionError, match="max_num_steps exceeded"):  # the library's whole-solve runaway bound
        odeint(truth, ty0, t, options={"max_total_steps": 20})
    with pytest.raises(AssertionError, match="underflow in dt|non-finite"):
        odeint(truth, torch.tensor([[float("nan"), 1.0]]), t)


def test_fp64_state_and_device_placement(ion, gpu):
    from torchdiffeq import odeint
    truth, ty0, _, _ = _modules("s1")
    truth.set_fixed_form_voltage_protocol(*K.activation(20)[:2])
    t = torch.linspace(0.0, 8000.0, 801)
    y32 = odeint(truth, ty0, t)
    y64 = odeint(truth, ty0.double(), t)
    assert y64.dtype == torch.float64 and y64.device == ty0.device
    assert torch.allclose(y32.double(), y64, atol=5e-5)
