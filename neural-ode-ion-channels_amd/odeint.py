"""Drop-in `odeint(func, y0, t)` with torchdiffeq's call signature (the reference's hot-path boundary).

    from torchdiffeq import odeint            # train-s1.py:29-32 -> resolves to the shim at the repo root
    pred_y = odeint(func, true_y0, prediction_t)                       # train-s1.py:327
    o = odeint(self._ode, self._y0_torch, t, method='dopri5')          # train-d0.py:428

Returns a tensor of shape (len(t), *y0.shape), dtype y0.dtype, on y0.device, as torchdiffeq does.  The reference's
RHS modules (rhs.py) are integrated by the fused HIP kernel.  `func` is read at call time: the protocol set by
`set_fixed_form_voltage_protocol` between calls and the current weights are picked up on every call.

Gradients.  Called with autograd enabled and a `func.net` parameter, a tensor-valued rate parameter or `y0` requiring
grad, the result carries a graph: the backward sweep of grad.py (exact derivative of the executed discretisation, accepted
steps as constants).  `odeint_adjoint` returns the same values and the same gradients (forward bit-identical to odeint, as
torchdiffeq's); with `adjoint_options={"max_step": "auto"}` it differentiates a step sequence capped at grad.stable_step_cap()
(3 / lambda_max of the rate constants): like torchdiffeq's continuous adjoint, that derivative stays bounded on long holds and
converges to the continuous adjoint as rtol -> 0, whereas the exact derivative of the UNcapped sequence amplifies rounding noise
at equilibria (grad.py warns).
The reference's --adjoint flag only switches the import (train-s1.py:29-32) and never differentiates, so there is no
reference behaviour to mirror beyond the forward values.  All RHS families of the reference are covered (NN-f / NN-d for the widths of architectures
s00-s11 with at most 15 hidden layers; HH 2-state and 6-state in closed form); other shapes raise instead of silently returning a
graph-less tensor.

Failures raise AssertionError with torchdiffeq's messages ('underflow in dt', 'non-finite values in state `y`',
'max_num_steps exceeded').  There is no CPU fallback for recognised modules: without a HIP device the call raises.
Modules that are not one of the reference's families raise UnrecognisedRhs unless the caller opts into the
generic torch stepper with options={'allow_generic': True} (same algorithm, `func.forward` called from Python).
"""
import warnings

import numpy as np
import torch

from . import batched, capi, grad, rhs
from .generic import generic_dopri5

_KNOWN_OPTIONS = {"allow_generic", "explicit_protocol", "max_num_steps", "max_total_steps", "max_step", "tile_waves",
                  # torchdiffeq 0.1.x-era keys the reference passes in train-d0.py:436; 0.2.x warns and ignores them
                  "grid_points", "eps"}


def odeint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, event_fn=None):
    if event_fn is not None:
        raise NotImplementedError("event handling is not part of the reference's path")
    if method not in (None, "dopri5"):
        raise NotImplementedError(f"method={method!r}: the reference only ever runs dopri5 (SURVEY.md section 5)")
    if not isinstance(y0, torch.Tensor) or not isinstance(t, torch.Tensor):
        raise TypeError("y0 and t must be torch tensors")
    if t.dim() != 1 or t.numel() < 1:
        raise ValueError("t must be a 1-D tensor")
    options = dict(options or {})
    unknown = set(options) - _KNOWN_OPTIONS
    if unknown:
        warnings.warn(f"odeint: unexpected options {sorted(unknown)} ignored")
    if "grid_points" in options or "eps" in options:
        warnings.warn("odeint: 'grid_points'/'eps' are torchdiffeq 0.1.x options; ignored (0.2.1 ignores them too)")
    if y0.dtype not in (torch.float32, torch.float64):
        raise TypeError("y0 must be float32 or float64")

    try:
        spec = rhs.recognise(func, y0, force_explicit_protocol=bool(options.get("explicit_protocol", False)))
    except rhs.UnrecognisedRhs as e:
        if options.get("allow_generic", False):
            return generic_dopri5(func, y0, t, rtol=rtol, atol=atol, max_steps=options.get("max_num_steps", 2**31 - 1))
        raise rhs.UnrecognisedRhs(
            f"odeint: func is not one of the reference's RHS families ({e}); pass options={{'allow_generic': True}} "
            "to integrate it with the generic torch stepper") from None

    if torch.is_grad_enabled() and _wants_grad(func, y0):
        return _odeint_with_grad(func, y0, t, spec, rtol, atol, options)

    max_step = options.get("max_step", 0.0)
    if isinstance(max_step, str):  # "auto": the same cap the differentiable call would use (grad.stable_step_cap)
        if max_step != "auto":
            raise ValueError(f"odeint: options['max_step'] must be a number of milliseconds or 'auto', not {max_step!r}")
        max_step = grad.stable_step_cap(spec.model, torch.from_numpy(np.asarray(spec.params, dtype=np.float64)[None, :]),
                                        torch.from_numpy(np.asarray(spec.prot_v, dtype=np.float64)))
    t64 = t.detach().to(torch.float64)
    t_key = ("t", rhs.digest(t64.cpu().numpy()))  # the output grid stays device-resident while its values do not change
    sol = batched.solve(spec.model, spec.params, spec.prot_v, y0.reshape(1, -1), t64,
                        weights=spec.weights, mlp_layers=spec.mlp_layers, mlp_width=spec.mlp_width,
                        weights_key=spec.weights_key, prot_t=spec.prot_t, prot_t0=spec.prot_t0, prot_dt=spec.prot_dt,
                        state_dtype=y0.dtype, rtol=float(rtol), atol=float(atol),
                        max_steps=int(options.get("max_num_steps", 0)),
                        max_total_steps=int(options.get("max_total_steps", 0)), max_step=float(max_step),
                        prot_key=spec.prot_key, t_eval_key=t_key,
                        tile_waves=int(options.get("tile_waves", 0)))
    sol.raise_on_failure()
    out = sol.y[0].reshape((t.numel(),) + tuple(y0.shape))
    return out.to(y0.device)


def _rate_tensors(func, n=12):
    return [getattr(func, f"p{i}", None) for i in range(1, n + 1)]


def _wants_grad(func, y0):
    if y0.requires_grad:
        return True
    if isinstance(func, torch.nn.Module) and any(p.requires_grad for p in func.parameters()):
        return True
    return any(isinstance(p, torch.Tensor) and p.requires_grad for p in _rate_tensors(func))


def _odeint_with_grad(func, y0, t, spec, rtol, atol, options):
    """Differentiable call: same forward kernel (plus accepted-step checkpoints), backward sweep on demand."""
    dev = batched._dev()
    flat = None
    if spec.model in (capi.MODEL_NNF, capi.MODEL_NND):
        lin = [m for m in func.net if isinstance(m, torch.nn.Linear)]
        flat = torch.cat([x.reshape(-1) for m in lin for x in (m.weight, m.bias)]).to(dev)
    rates = _rate_tensors(func, 12 if spec.model == capi.MODEL_MARKOV6 else 8)
    if any(isinstance(p, torch.Tensor) and p.requires_grad for p in rates):
        cols = [(p.reshape(()).to(device=dev, dtype=torch.float64) if isinstance(p, torch.Tensor)
                 else torch.tensor(float(0.0 if p is None else p), dtype=torch.float64, device=dev)) for p in rates]
        if spec.model == capi.MODEL_NNF:
            cols[:4] = [torch.zeros((), dtype=torch.float64, device=dev)] * 4
        params = torch.stack(cols)[None, :]
    else:
        params = torch.from_numpy(spec.params[None, :]).to(dev)
    t64 = t.detach().to(torch.float64)
    te = batched._to(t64, torch.float64, dev, key=("t", rhs.digest(t64.cpu().numpy())))
    y, status = grad.solve(spec.model, flat, params, batched._to(spec.prot_v[None, :], torch.float64, dev, key=(spec.prot_key, "v2")),
                           y0.reshape(1, -1).to(dev), te, mlp_layers=spec.mlp_layers, mlp_width=spec.mlp_width,
                           prot_t=batched._to(spec.prot_t, torch.float64, dev, key=(spec.prot_key, "t")),
                           prot_t0=spec.prot_t0, prot_dt=spec.prot_dt, rtol=float(rtol), atol=float(atol),
                           max_steps=int(options.get("max_num_steps", 0)), max_total_steps=int(options.get("max_total_steps", 0)),
                           max_step=options.get("max_step", 0.0), weights_key=spec.weights_key)
    st = int(status[0].item())
    if st != 0:
        raise AssertionError(capi.STATUS_TEXT[st])
    return y[0].reshape((t.numel(),) + tuple(y0.shape)).to(y0.device)


class StepCapNotice(UserWarning):
    """Kept for callers that filter on it (round 4 issued it when odeint_adjoint capped the step size by default; since round 5 the
    cap is opt-in and nothing is capped silently, so it is no longer raised)."""


def odeint_adjoint(func, y0, t, *, rtol=1e-7, atol=1e-9, method=None, options=None, event_fn=None,
                   adjoint_rtol=None, adjoint_atol=None, adjoint_method=None, adjoint_options=None, adjoint_params=None):
    """`from torchdiffeq import odeint_adjoint as odeint` (train-s1.py:29-32).  Forward values are odeint's, bit for bit, with or
    without a gradient requested -- as torchdiffeq's adjoint returns exactly odeint's forward.  Under torch.no_grad() -- every call
    site of the reference (SURVEY.md finding 3) -- this IS odeint: same kernel, same bits.

    With a gradient requested the backward is grad.py's sweep: the exact derivative of the executed step sequence.  For gradients
    with respect to the RATE parameters through long holds that derivative amplifies rounding noise (dopri5 coasts through an
    equilibrium with h * lambda >> 1; grad.solve raises a RuntimeWarning).  The STABILISED sweep is opt-in:
    `adjoint_options={"max_step": "auto"}` (or a number of milliseconds; `options={"max_step": ...}` does the same for both names)
    runs the forward with dt capped at grad.stable_step_cap() (3 / lambda_max of the gating rates at the protocol's extreme
    voltages) and differentiates that step sequence.  What torchdiffeq's continuous adjoint buys -- a gradient that stays bounded
    at equilibria -- is obtained this way without re-integrating the state backwards (unstable for these dissipative gating
    equations); the capped discrete gradient converges to the continuous adjoint as rtol -> 0
    (tests/test_gpu_round3.py::test_odeint_adjoint_is_the_stabilised_sweep); its forward values then follow the capped sequence and
    differ from odeint's at the rtol level -- which is why it is never the default.  torchdiffeq's other adjoint_* knobs tune its
    adjoint ODE solve and have no counterpart here: accepted and ignored."""
    options = dict(options or {})
    cap = (adjoint_options or {}).get("max_step")
    if cap is not None and "max_step" not in options and torch.is_grad_enabled() and _wants_grad(func, y0):
        options["max_step"] = cap
    return odeint(func, y0, t, rtol=rtol, atol=atol, method=method, options=options, event_fn=event_fn)
