"""Recognise the reference's right-hand-side modules and turn them into kernel inputs.

The reference passes duck-typed `nn.Module`s to `odeint(func, y0, t)`; the integrator fuses the four
families it defines:

  HH 2-state     `Lambda`    train-s1.py:134-177   p1..p8,  no `net`
  candidate HH   `ODEFunc`   train-d0.py:321-374   p1..p8,  no `net`, settable p1..p4
  6-state        `Lambda`    train-d1.py:134-187   p1..p12, no `net`
  NN-f           `ODEFunc`   train-s1.py:181-247   `net` (Sequential of Linear/LeakyReLU), p5..p8
  NN-d           `ODEFunc`   train-d2.py:191-272   `net`, p1..p8, `_dadt`

and the protocol those modules carry (`set_fixed_form_voltage_protocol(t, v)` stores `_t_regular`,
`_v_regular`, train-s1.py:218-222).  Recognition is by attributes, then CONFIRMED by evaluating the module's own
`forward` at a few probe points against the formula the kernel implements: a look-alike module with different
maths is not silently mis-integrated, it is reported as unrecognised.
"""
import hashlib
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import capi


@dataclass
class RhsSpec:
    model: int                      # capi.MODEL_*
    n_state: int
    params: np.ndarray              # [8] or [12] fp64
    weights: Optional[np.ndarray]   # flat fp32 state dict (reference order) or None
    mlp_layers: int
    mlp_width: int
    prot_t: Optional[np.ndarray]    # explicit protocol times (ms) or None when uniform
    prot_v: np.ndarray              # [Np] fp64 mV
    prot_t0: float
    prot_dt: float
    weights_key: Optional[tuple] = None  # content digest of the flat weights (packed-image cache key)
    prot_key: Optional[bytes] = None     # content digest of the protocol arrays (device-resident copy cache key)


class UnrecognisedRhs(Exception):
    pass


def digest(*arrays):
    """Content digest of numpy arrays (cache keys must follow the VALUES: `.data.copy_()` / `load_state_dict` change
    neither `data_ptr()` nor `_version`, so tensor identity is not a safe key)."""
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str((a.dtype.str, a.shape)).encode())
        h.update(a.view(np.uint8).reshape(-1))
    return h.digest()


def _scalar(x):
    if isinstance(x, torch.Tensor):
        if x.numel() != 1:
            raise UnrecognisedRhs("parameter tensor is not a scalar")
        return float(x.detach().cpu().double().item())
    return float(x)


def _params(func, names):
    out = []
    for n in names:
        if not hasattr(func, n):
            raise UnrecognisedRhs(f"missing attribute {n}")
        out.append(_scalar(getattr(func, n)))
    return np.array(out, dtype=np.float64)


def mlp_shape_and_weights(net):
    """nn.Sequential(Linear(2,N), [LeakyReLU, Linear(N,N)] x L, LeakyReLU, Linear(N,1)) -> (L, N, flat fp32)."""
    if not isinstance(net, nn.Sequential):
        raise UnrecognisedRhs("net is not nn.Sequential")
    mods = list(net)
    if len(mods) < 3 or len(mods) % 2 == 0:
        raise UnrecognisedRhs("net is not Linear/LeakyReLU alternating")
    lin = mods[0::2]
    act = mods[1::2]
    if not all(isinstance(m, nn.Linear) and m.bias is not None for m in lin):
        raise UnrecognisedRhs("net layers are not biased nn.Linear")
    if not all(isinstance(m, nn.LeakyReLU) and abs(m.negative_slope - 0.01) < 1e-12 for m in act):
        raise UnrecognisedRhs("activations are not nn.LeakyReLU(0.01)")
    N = lin[0].out_features
    if lin[0].in_features != 2 or lin[-1].out_features != 1 or lin[-1].in_features != N:
        raise UnrecognisedRhs("net is not 2 -> N -> ... -> 1")
    for m in lin[1:-1]:
        if m.in_features != N or m.out_features != N:
            raise UnrecognisedRhs("hidden layers are not N x N")
    if any(m.weight.dtype != torch.float32 for m in lin):
        raise UnrecognisedRhs("net is not float32 (the reference hard-casts with .float())")
    L = len(lin) - 2
    flat = np.concatenate([np.concatenate([m.weight.detach().cpu().numpy().reshape(-1),
                                           m.bias.detach().cpu().numpy().reshape(-1)]) for m in lin]).astype(np.float32)
    return L, N, flat, ("w", digest(flat))


def _protocol(func, force_explicit=False):
    t = getattr(func, "_t_regular", None)
    v = getattr(func, "_v_regular", None)
    if t is None or v is None:
        raise UnrecognisedRhs("no voltage protocol set (call set_fixed_form_voltage_protocol first)")
    t = np.ascontiguousarray(np.asarray(t, dtype=np.float64).reshape(-1))
    v = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    if t.shape != v.shape or t.size < 2 or not np.all(np.diff(t) > 0):
        raise UnrecognisedRhs("protocol times must be strictly increasing and match the voltages")
    dt = (t[-1] - t[0]) / (t.size - 1)
    grid = t[0] + np.arange(t.size) * dt
    uniform = np.max(np.abs(t - grid)) <= 1e-9 * max(abs(t[-1]), abs(t[0]), dt)
    # the kernel rebuilds the last sample time as t0 + (n-1)*dt: if that rounds below t[-1], a query at exactly t[-1]
    # (normally the last output sample) would be judged out of range -> keep the explicit grid for such protocols
    uniform = uniform and (t[0] + (t.size - 1) * dt == t[-1])
    key = digest(t, v)
    if uniform and not force_explicit:
        return None, v, float(t[0]), float(dt), key
    return t, v, float(t[0]), float(dt), key


def _formula(spec, net, t_s, y):
    """The RHS the kernel implements, in torch (fp64 maths, fp32 net), for the recognition probe only."""
    tq = float(t_s)
    if spec.prot_t is not None:
        x = spec.prot_t
    else:
        x = spec.prot_t0 + np.arange(spec.prot_v.size) * spec.prot_dt
    v = float(np.interp(tq, x, spec.prot_v)) if x[0] <= tq <= x[-1] else -80.0
    p = spec.params
    yd = [float(c) for c in y.reshape(-1)]
    if spec.model == capi.MODEL_MARKOV6:
        a1, b1 = p[0] * np.exp(p[1] * v), p[2] * np.exp(-p[3] * v)
        bh, ah = p[4] * np.exp(p[5] * v), p[6] * np.exp(-p[7] * v)
        a2, b2 = p[8] * np.exp(p[9] * v), p[10] * np.exp(-p[11] * v)
        c1, c2, i_, ic1, ic2, o = yd
        return np.array([a1 * c2 + ah * ic1 + b2 * o - (b1 + bh + a2) * c1,
                         b1 * c1 + ah * ic2 - (a1 + bh) * c2,
                         a2 * ic1 + bh * o - (b2 + ah) * i_,
                         a1 * ic2 + bh * c1 + b2 * i_ - (b1 + ah + a2) * ic1,
                         b1 * ic1 + bh * c2 - (ah + a1) * ic2,
                         a2 * c1 + ah * i_ - (b2 + bh) * o])
    a, r = yd
    drdt = -p[4] * np.exp(p[5] * v) * r + p[6] * np.exp(-p[7] * v) * (1 - r)
    dadt = 0.0
    if spec.model in (capi.MODEL_HH2, capi.MODEL_NND):
        dadt = p[0] * np.exp(p[1] * v) * (1 - a) - p[2] * np.exp(-p[3] * v) * a
    if spec.model in (capi.MODEL_NNF, capi.MODEL_NND):
        with torch.no_grad():
            dev = next(net.parameters()).device
            dadt += float(net(torch.tensor([v / 100.0, a], dtype=torch.float32, device=dev)).item()) / 1000.0
    return np.array([dadt, drdt])


def recognise(func, y0, *, force_explicit_protocol=False, probe=True) -> RhsSpec:
    """RhsSpec for a reference-style module, or raise UnrecognisedRhs."""
    D = int(y0.reshape(-1).numel())
    has_net = hasattr(func, "net")
    if has_net:
        L, N, flat, key = mlp_shape_and_weights(func.net)
        for name, want in (("vrange", 100.0), ("netscale", 1000.0)):
            if hasattr(func, name) and abs(_scalar(getattr(func, name)) - want) > 0:
                raise UnrecognisedRhs(f"{name} != {want}")
        if D != 2:
            raise UnrecognisedRhs("NN models have two states (a, r)")
        nnd = hasattr(func, "_dadt") and all(hasattr(func, f"p{i}") for i in range(1, 5))
        if nnd:
            model, params = capi.MODEL_NND, _params(func, [f"p{i}" for i in range(1, 9)])
        else:
            model = capi.MODEL_NNF
            params = np.concatenate([np.zeros(4), _params(func, [f"p{i}" for i in range(5, 9)])])
    else:
        L, N, flat, key = 0, 0, None, None
        if hasattr(func, "p12"):
            if D != 6:
                raise UnrecognisedRhs("12-parameter model expects six states")
            model, params = capi.MODEL_MARKOV6, _params(func, [f"p{i}" for i in range(1, 13)])
        elif hasattr(func, "p8") and not hasattr(func, "p9"):
            if D != 2:
                raise UnrecognisedRhs("HH model has two states (a, r)")
            model, params = capi.MODEL_HH2, _params(func, [f"p{i}" for i in range(1, 9)])
        else:
            raise UnrecognisedRhs("no p1..p8 / p1..p12 rate parameters")
    pt, pv, t0, dt, pkey = _protocol(func, force_explicit_protocol)
    spec = RhsSpec(model=model, n_state=D, params=params, weights=flat, mlp_layers=L, mlp_width=N,
                   prot_t=pt, prot_v=pv, prot_t0=t0, prot_dt=dt, weights_key=key, prot_key=pkey)
    if probe:
        # the three probe forward() calls run once per (module, weights, parameters, protocol): the reference calls
        # odeint in loops on the same module (train-s1.py:441-457), and the confirmation cannot change between them
        ck = (id(func), type(func).__qualname__, model, D, params.tobytes(), key, pkey, pt is None, str(y0.device))
        if ck not in _confirmed:
            _confirm(func, spec, y0)
            if len(_confirmed) > 256:
                _confirmed.clear()
            _confirmed.add(ck)
    return spec


_confirmed = set()


def _confirm(func, spec, y0):
    """Evaluate func.forward at three in-range points and compare with the kernel's formula."""
    x0 = spec.prot_t[0] if spec.prot_t is not None else spec.prot_t0
    x1 = spec.prot_t[-1] if spec.prot_t is not None else spec.prot_t0 + (spec.prot_v.size - 1) * spec.prot_dt
    rng = np.random.default_rng(0)
    net = func.net if hasattr(func, "net") else None
    was_training = getattr(func, "training", False)
    with torch.no_grad():
        for frac in (0.137, 0.52, 0.93):
            tq = x0 + frac * (x1 - x0)
            yv = rng.uniform(0.05, 0.9, spec.n_state)
            if spec.n_state == 6:
                yv = yv / yv.sum()
            y = torch.tensor(yv, dtype=torch.float64, device=y0.device).reshape(y0.shape)
            try:
                got = func(torch.tensor(tq, dtype=torch.float64, device=y0.device), y)
            except Exception as e:  # the module does not accept what the reference modules accept
                raise UnrecognisedRhs(f"probe call failed: {e!r}")
            got = np.asarray(got.detach().cpu().double().numpy()).reshape(-1)
            want = _formula(spec, net, tq, y.cpu().numpy())
            if got.shape != want.shape or not np.allclose(got, want, rtol=1e-5, atol=1e-9):
                raise UnrecognisedRhs("forward() does not match the recognised model's formula")
    if was_training and hasattr(func, "train"):
        func.train(was_training)
