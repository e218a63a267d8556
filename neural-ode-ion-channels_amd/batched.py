"""Batched host API over the C ABI: numpy / torch in, torch (device) out.

`solve()` is what the drop-in `odeint` wraps for one trajectory and what sweeps / benchmarks call directly for
B trajectories (per-trajectory rate parameters, per-trajectory or shared protocols, shared MLP weights).
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import capi

_packed_cache = {}
_dev_cache = {}  # (content digest, dtype, device) -> device tensor: protocols / output grids that do not change between calls


def _dev(device=None):
    if not torch.cuda.is_available():
        raise capi.IonodeError("no HIP device visible: the integrator has no CPU fallback")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(device)


def _to(x, dtype, dev, key=None):
    """Device-resident contiguous copy; with `key` (a content digest) the copy is cached and re-used across calls."""
    if x is None:
        return None
    ck = (key, dtype, str(dev)) if key is not None else None
    if ck is not None and ck in _dev_cache:
        return _dev_cache[ck]
    if isinstance(x, torch.Tensor):
        out = x.to(device=dev, dtype=dtype).contiguous()
    else:
        out = torch.from_numpy(np.ascontiguousarray(np.asarray(x))).to(device=dev, dtype=dtype).contiguous()
    if ck is not None:
        if len(_dev_cache) > 32:
            _dev_cache.clear()
        _dev_cache[ck] = out
    return out


def packed_weights(weights, mlp_layers, mlp_width, dev, key=None):
    """Device-resident MFMA-order image of a flat fp32 state dict; cached per (key, device)."""
    ck = (key, mlp_layers, mlp_width, str(dev)) if key is not None else None
    if ck is not None and ck in _packed_cache:
        return _packed_cache[ck]
    if isinstance(weights, torch.Tensor):
        weights = weights.detach().cpu().numpy()
    weights = np.asarray(weights)
    if weights.ndim == 2:   # several weight sets (solve(traj_per_image=...)): one packed image per row
        img = torch.from_numpy(np.stack([capi.mlp_pack(w, mlp_layers, mlp_width) for w in weights])).to(dev).contiguous()
    else:
        img = torch.from_numpy(capi.mlp_pack(weights, mlp_layers, mlp_width)).to(dev)
    if ck is not None:
        if len(_packed_cache) > 16:
            _packed_cache.clear()
        _packed_cache[ck] = img
    return img


@dataclass
class Solution:
    y: torch.Tensor                 # [B, Nt, D] state dtype, on device
    i: Optional[torch.Tensor]       # [B, Nt] fp64 current trace or None
    status: torch.Tensor            # [B] int32
    stats: Optional[torch.Tensor]   # [B, 4] int64: accepted, rejected, nfe, status
    kernel: str
    sse: Optional[torch.Tensor] = None  # [B] fp64 fused sum of squared current residuals (sse_ref given), inf where failed
    order: Optional[torch.Tensor] = None  # [B] int64 launch order (solve(order=...)): row k of every field is trajectory order[k]

    def to_original(self, x):
        """Scatter a per-trajectory tensor (leading dimension B, launch order) back to the caller's trajectory order."""
        if self.order is None:
            return x
        out = torch.empty_like(x)
        out.index_copy_(0, self.order, x)
        return out

    def raise_on_failure(self):
        st = self.status.cpu().numpy()
        bad = np.nonzero(st)[0]
        if bad.size:
            b = int(bad[0])
            # torchdiffeq raises AssertionError with these messages
            raise AssertionError(f"{capi.STATUS_TEXT[int(st[b])]} (trajectory {b}; {bad.size} of {st.size} failed)")


def solve(model, params, prot_v, y0, t_eval, *, weights=None, mlp_layers=0, mlp_width=0, weights_key=None,
          prot_t=None, prot_t0=0.0, prot_dt=1.0, prot_of_traj=None, state_dtype=None, rtol=1e-7, atol=1e-9,
          v_oob=-80.0, max_steps=0, max_total_steps=0, max_step=0.0, current=False, obs_g=1.0, obs_e=-86.0, obs_open_state_only=False,
          tile_waves=0, device=None, step_log=None, t_eval_hint="auto", prot_key=None, t_eval_key=None, sse_ref=None,
          states=True, order=None, traj_per_image=0, launch_order="auto") -> Solution:
    """Integrate B trajectories on the GPU (asynchronous on the current stream).

    params [B, 8|12] (or [8|12] -> B = 1), prot_v [P, Np] (or [Np]), y0 [B, D] / [D] (broadcast over B),
    t_eval [Nt].  state_dtype: torch.float32 (reference-compatible) or torch.float64; default = y0's dtype if it
    is a floating torch tensor, else float64.
    order: optional permutation of range(B) (schedule.lpt_order), or 'pilot' (a closed-form pilot solve ranks the trajectories
    first: schedule.pilot_cost): launch slot k integrates trajectory order[k], so 16
    consecutive entries share an MFMA tile and earlier tiles start first.  Every field of the Solution is then in LAUNCH
    order (un-permuting [B, Nt, D] traces would cost a second pass over them; Solution.to_original() does it on request).
    Each trajectory's values do not depend on its tile-mates, so the ordering changes the time, never the results.
    launch_order: the same schedule WITHOUT moving any data (ionode_desc.launch_order, ABI 6): a permutation of range(B), the
    kernel maps launch slots to trajectories itself and every field of the Solution stays in the caller's order -- what a training
    loop wants (`launch_order=schedule.lpt_order(previous.stats[:, 2])`).  "auto" (default): protocol-major order for the
    one-trajectory-per-lane kernels, index order otherwise; None: index order.  Not combined with `order` or traj_per_image.
    traj_per_image: with `weights` [n_sets, n] (an ensemble of trained nets, a population of initialisations): trajectory b is
    integrated with weight set b // traj_per_image (a multiple of 16); not combined with `order`.
    """
    dev = _dev(device)
    t_eval_exact = None
    params_t = _to(params, torch.float64, dev)
    if params_t.dim() == 1:
        params_t = params_t[None, :].contiguous()
    B = params_t.shape[0]
    prot_v_t = _to(prot_v, torch.float64, dev, key=None if prot_key is None else (prot_key, "v"))
    if prot_v_t.dim() == 1:
        prot_v_t = prot_v_t[None, :].contiguous()
    if state_dtype is None:
        state_dtype = y0.dtype if isinstance(y0, torch.Tensor) and y0.dtype in (torch.float32, torch.float64) else torch.float64
    y0_t = _to(y0, state_dtype, dev)
    if y0_t.dim() == 1:
        y0_t = y0_t[None, :]
    if y0_t.shape[0] != B:
        y0_t = y0_t.expand(B, y0_t.shape[1])
    y0_t = y0_t.contiguous()
    pot_t = _to(prot_of_traj, torch.int32, dev)
    order_t = None
    if isinstance(order, str):
        if order != "pilot":
            raise capi.IonodeError("order: a permutation, None or 'pilot'")
        # cost-sorted launch order from a closed-form pilot solve of the same protocols (schedule.pilot_cost)
        from . import schedule
        te_h = t_eval.detach().cpu().numpy() if isinstance(t_eval, torch.Tensor) else np.asarray(t_eval)
        order = schedule.lpt_order(schedule.pilot_cost(params_t[:, :8], prot_v_t, float(te_h[0]), float(te_h[-1]), prot_t0=prot_t0,
                                                       prot_dt=prot_dt, prot_t=prot_t, prot_of_traj=pot_t, rtol=rtol, atol=atol,
                                                       v_oob=v_oob, device=dev))
    if order is not None:
        order_t = _to(order, torch.int64, dev)
        if order_t.shape != (B,) or int(order_t.min()) < 0 or int(order_t.max()) >= B or \
                not bool(torch.bincount(order_t, minlength=B).eq(1).all()):
            raise capi.IonodeError(f"order must be a permutation of range({B})")
        params_t = params_t.index_select(0, order_t)
        y0_t = y0_t.index_select(0, order_t)
        # the kernel's default protocol of trajectory b is b % P: make it explicit before permuting
        pot_t = (order_t % prot_v_t.shape[0]).to(torch.int32) if pot_t is None else pot_t.index_select(0, order_t)
    if isinstance(t_eval_hint, str) and t_eval_hint == "auto" and not (isinstance(t_eval, torch.Tensor) and t_eval.is_cuda):
        # host-side grid: derive the output-cursor hint without touching the device
        te = t_eval.detach().double().numpy() if isinstance(t_eval, torch.Tensor) else np.asarray(t_eval, dtype=np.float64)
        t_eval_hint = None
        if te.size > 1:
            dth = (te[-1] - te[0]) / (te.size - 1)
            if dth > 0 and np.max(np.abs(te - (te[0] + np.arange(te.size) * dth))) <= 0.5 * dth:
                t_eval_hint = (float(te[0]), float(dth))
                t_eval_exact = bool(np.array_equal(te, te[0] + np.arange(te.size, dtype=np.float64) * dth))
    t_eval_t = _to(t_eval, torch.float64, dev, key=t_eval_key)
    packed = None
    if model in (capi.MODEL_NNF, capi.MODEL_NND):
        if weights is None:
            raise capi.IonodeError("NN models need `weights` (flat fp32 state dict)")
        packed = packed_weights(weights, mlp_layers, mlp_width, dev, key=weights_key)
    if launch_order is not None and not isinstance(launch_order, str):
        if order is not None or traj_per_image:
            raise capi.IonodeError("launch_order cannot be combined with order / traj_per_image")
        lo = _to(launch_order, torch.int64, dev)
        if lo.shape != (B,) or int(lo.min()) < 0 or int(lo.max()) >= B or not bool(torch.bincount(lo, minlength=B).eq(1).all()):
            raise capi.IonodeError(f"launch_order must be a permutation of range({B})")
        launch_order = lo.to(torch.int32).contiguous()
    elif order is not None or traj_per_image:
        launch_order = None
    if traj_per_image and order is not None:
        raise capi.IonodeError("traj_per_image ties trajectories to weight sets by position: it cannot be combined with order")
    r = capi.dopri5(model, params_t, prot_v_t, y0_t, t_eval_t, mlp_packed=packed, traj_per_image=traj_per_image, mlp_layers=mlp_layers,
                    mlp_width=mlp_width, prot_t=_to(prot_t, torch.float64, dev, key=None if prot_key is None else (prot_key, "t")),
                    prot_t0=prot_t0, prot_dt=prot_dt,
                    prot_of_traj=pot_t, rtol=rtol, atol=atol, v_oob=v_oob,
                    max_steps=max_steps, max_total_steps=max_total_steps, max_step=max_step, current=current, obs_g=obs_g, obs_e=obs_e,
                    obs_open_state_only=obs_open_state_only, tile_waves=tile_waves, step_log=step_log,
                    t_eval_hint=t_eval_hint, t_eval_exact=t_eval_exact, sse_ref=_to(sse_ref, torch.float64, dev), states=states,
                    launch_order=launch_order)
    return Solution(y=r["y"], i=r["i"], status=r["status"], stats=r["stats"], kernel=r["kernel"], sse=r["sse"],
                    order=order_t)
