"""Gradients through the batched solve (BASELINE.json configs[4]).

Reference interface: `from torchdiffeq import odeint_adjoint as odeint` (train-s1.py:29-32).  The reference only switches
that import -- every call site runs under torch.no_grad() and nothing is ever differentiated (SURVEY.md finding 3) -- so
there is no reference gradient to reproduce: **parity is unpinned**.  What this module computes is the exact reverse-mode
derivative of the discretisation the forward launch executed, with the accepted steps (t0, dt) as constants
(discretise-then-optimise, controller frozen); the checker is autograd through a torch restatement replaying the same
steps (tests/grad_check.py).  `odeint` and `odeint_adjoint` share this backward: both names give the same values and the
same gradients.

Three launches per backward (csrc/ionode_grad.hpp, ionode_grad_reduce.hpp), all through the C ABI:
  forward   ionode_dopri5 with accepted-step checkpoints (160 B per step and trajectory)
  sweep     adjoints of y0 and p1..p8, and the (d_l, h_l) record stream of every MLP vector-Jacobian product (160 KB per
            16-trajectory tile evaluation for s00 -- sized for 288 GB of HBM3E and chunked over iterations above
            `record_budget_bytes`).  NN models: two phases -- ionode_dopri5_backward_recompute (unit-seed products of every
            (tile, step) at once, whole chip) one chunk ahead of ionode_dopri5_backward_sweep (the sequential walk: adjoint
            algebra only); closed-form models and two_phase=False: ionode_dopri5_backward (everything inside the walk)
  reduce    ionode_grad_reduce[_unit]: split-K fp32 MFMA GEMM of the records into per-slab partial weight gradients, summed
            here in fp64.
There is no CPU fallback: without libionode.so or a HIP device every call raises.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi

_image_cache = {}
DEFAULT_CKPT_CAP = 4096
DEFAULT_RECORD_BUDGET = 64 << 30   # of 288 GB HBM3E per GPU: two record buffers of half of it (24 GB: +1.5 % sweep time; 96 GB: -0.5 %)
DEFAULT_CKPT_BUDGET = 96 << 30   # bytes of accepted-step checkpoints one forward may allocate (a third of the 288 GB of HBM3E)
MAX_RECOMPUTE_ITERS = 65535 * 4  # phase A launches dim3(tiles, ceil(iterations / GRAD_RECOMPUTE_IB = 4)): HIP caps grid.y at 65535


def _bounded_budget(requested, default, dev, share):
    """An explicit budget is taken as given; the default (tuned for an empty 288 GB MI355X) is capped at `share` of the memory
    that is free on `dev` right now, so that a smaller or partly occupied GPU gets smaller chunks instead of an allocator failure."""
    if requested:
        return int(requested)
    try:
        free, _total = torch.cuda.mem_get_info(dev)
        # memory the caching allocator holds but has handed out to nobody is reusable by the next allocation: without it the previous
        # iteration's own freed checkpoint / record buffers would shrink the budget iteration by iteration
        free += max(0, torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev))
    except Exception:  # no device query available: keep the default
        return int(default)
    return int(max(1 << 28, min(default, share * free)))


def stable_step_cap(model, params, prot_v, v_oob=-80.0, safety=3.0):
    """Largest dt (ms) that keeps dopri5 inside its real-axis stability interval (|h lambda| < ~3.3) for every trajectory
    of the batch: safety / lambda_max, lambda_max = the largest relaxation rate of the gating equations at the protocol's
    extreme voltages (the rates p exp(+-q v) are monotone in v, so the maximum sits at an end of the voltage range).
    2-state models: lambda = k_open + k_close of the a and r gates (train-s1.py:161-177); 6-state model: Gershgorin bound
    2 x (sum of the six rates) (train-d1.py:165-187).  This is what `max_step="auto"` passes to the solve."""
    p = params.detach().to(torch.float64)
    v = torch.stack([prot_v.min().to(torch.float64), prot_v.max().to(torch.float64),
                     torch.as_tensor(float(v_oob), dtype=torch.float64, device=prot_v.device)]).to(p.device)
    n = 6 if model == capi.MODEL_MARKOV6 else 4
    sign = torch.tensor([1.0 if i % 2 == 0 else -1.0 for i in range(n)], dtype=torch.float64, device=p.device)
    amp, exp = p[:, 0:2 * n:2], p[:, 1:2 * n:2] * sign            # [B, n]
    rates = amp[:, :, None] * torch.exp(exp[:, :, None] * v[None, None, :])   # [B, n, 3]
    if model == capi.MODEL_MARKOV6:
        lam = 2.0 * rates.abs().sum(1)
    else:
        pairs = rates.abs().reshape(p.shape[0], n // 2, 2, 3).sum(2)          # (k1 + k2), (k3 + k4)
        if model == capi.MODEL_NNF:
            pairs = pairs[:, 1:]                                              # the a gate is the MLP: only the r gate is stiff
        lam = pairs.amax(1)
    lam_max = float(lam.max().item())
    return safety / lam_max if lam_max > 0 and np.isfinite(lam_max) else 0.0


def grad_image(weights_flat, L, N, dev, key=None):
    """Device-resident grad image (forward + transposed MFMA fragments) of a flat fp32 state dict."""
    ck = (key, L, N, str(dev)) if key is not None else None
    if ck is not None and ck in _image_cache:
        return _image_cache[ck]
    w = np.ascontiguousarray(np.asarray(weights_flat, dtype=np.float32).reshape(-1))
    expect = 2 * N + N + L * (N * N + N) + N + 1
    if w.size != expect:
        raise capi.IonodeError(f"state dict has {w.size} floats, (L={L}, N={N}) needs {expect}")
    n = capi.lib().ionode_grad_image_floats(L, N)
    if n == 0:
        raise capi.IonodeError("gradient path needs at least one hidden layer")
    out = np.empty(n, dtype=np.float32)
    if capi.lib().ionode_grad_pack(w.ctypes.data, L, N, out.ctypes.data) != 0:
        raise capi.IonodeError(capi.lib().ionode_grad_last_error().decode())
    img = torch.from_numpy(out).to(dev)
    if ck is not None:
        if len(_image_cache) > 8:
            _image_cache.clear()
        _image_cache[ck] = img
    return img


def unpack_partial(part, L, N):
    """Padded partial-gradient layout (include/ionode.h: ionode_grad_partial_floats) -> flat state-dict order."""
    NP = 16 * ((N + 15) // 16)
    out = []
    j0 = part[:4 * NP].reshape(NP, 4)
    out += [j0[:N, 1:3].reshape(-1), j0[:N, 0]]
    off = 4 * NP
    for _ in range(L):
        W = part[off:off + NP * NP].reshape(NP, NP)
        b = part[off + NP * NP:off + NP * NP + NP]
        out += [W[:N, :N].reshape(-1), b[:N]]
        off += NP * NP + NP
    out += [part[off:off + N], part[off + NP:off + NP + 1]]
    return torch.cat(out)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _Solve(torch.autograd.Function):
    """y[B, Nt, 2] = dopri5 solve; differentiable in (weights_flat, params, y0)."""

    @staticmethod
    def forward(ctx, weights_flat, params, y0, cfg):
        dev = y0.device
        L, N = cfg["mlp_layers"], cfg["mlp_width"]
        B = y0.shape[0]
        cap = int(cfg.get("ckpt_cap") or DEFAULT_CKPT_CAP)
        w_np, packed = None, None
        if weights_flat is not None:   # (None: the closed-form HH 2-state model)
            w_np = weights_flat.detach().to(torch.float32).cpu().numpy()
            from . import batched  # packed forward image: shared cache with the plain solve
            packed = batched.packed_weights(w_np, L, N, dev, key=cfg.get("weights_key"))
        limit = _bounded_budget(cfg.get("ckpt_budget_bytes"), DEFAULT_CKPT_BUDGET, dev, 0.6)
        row_bytes = B * (4 + 8 * y0.shape[1]) * 8
        if cap * row_bytes > limit:   # the FIRST allocation obeys the budget too (a huge batch on a small or occupied GPU)
            cap = max(1, limit // row_bytes)
        while True:
            ckpt = torch.empty((B, cap, 4 + 8 * y0.shape[1]), dtype=torch.float64, device=dev)
            r = capi.dopri5(cfg["model"], params.detach(), cfg["prot_v"], y0.detach(), cfg["t_eval"], mlp_packed=packed,
                            mlp_layers=L, mlp_width=N, prot_t=cfg.get("prot_t"), prot_t0=cfg["prot_t0"], prot_dt=cfg["prot_dt"],
                            prot_of_traj=cfg.get("prot_of_traj"), rtol=cfg["rtol"], atol=cfg["atol"], v_oob=cfg["v_oob"],
                            max_steps=cfg["max_steps"], max_total_steps=cfg["max_total_steps"], max_step=cfg.get("max_step", 0.0), ckpt=ckpt,
                            tile_waves=cfg.get("tile_waves", 0),
                            t_eval_hint=cfg.get("t_eval_hint", "auto"))
            # the checkpoint buffer is sized for the trajectories that SUCCEEDED: a failed one (status != 0: step budget spent,
            # dt underflow) can have 10^5..10^6 accepted steps, contributes no gradient (its n_acc is zeroed in backward) and
            # must not grow a [B, cap, 4 + 8 D] buffer to hundreds of GB
            nacc = torch.where(r["status"] == 0, r["stats"][:, 0], torch.zeros_like(r["stats"][:, 0]))
            most = int(nacc.max().item())
            if most <= cap:
                break
            cap = 1 << int(np.ceil(np.log2(most + 1)))  # the buffer was too small: run the forward again with room
            if cap * row_bytes > limit and most * row_bytes <= limit:
                cap = limit // row_bytes   # the power of two does not fit the budget, the steps themselves do
            need = cap * row_bytes
            if need > limit:
                raise capi.IonodeError(f"checkpoints of {most} accepted steps x {B} trajectories need {need / 2**30:.1f} GiB "
                                       f"(> ckpt_budget_bytes = {limit / 2**30:.1f} GiB): split the batch or raise the budget")
        ctx.cfg, ctx.desc = cfg, r["desc"]
        ctx.w_np = w_np
        ctx.save_for_backward(params.detach(), ckpt, r["stats"], r["status"])
        ctx.mark_non_differentiable(r["status"])
        ctx.n_acc_max = most
        return r["y"], r["status"]

    @staticmethod
    def backward(ctx, gy, _gstatus):
        cfg, desc = ctx.cfg, ctx.desc
        params, ckpt, stats, status = ctx.saved_tensors
        dev = params.device
        L, N = cfg["mlp_layers"], cfg["mlp_width"]
        B, Nt = desc.n_traj, desc.n_out
        need_w = ctx.needs_input_grad[0] and ctx.w_np is not None
        lib = capi.lib()
        sdt = torch.float32 if desc.state_f32 else torch.float64
        # failed trajectories (status != 0): their rows of y are NaN-filled and carry no gradient -- zero upstream rows (a caller's
        # unmasked loss would otherwise feed NaN into the sweep) and, below, zero dL/dy0 / dL/dp rows
        failed = status != 0
        gy = torch.where(failed[:, None, None], torch.zeros((), dtype=gy.dtype, device=gy.device), gy).to(sdt).contiguous()
        n_acc = torch.where(status == 0, stats[:, 0], torch.zeros_like(stats[:, 0])).to(torch.int32).contiguous()
        n_iter = int(n_acc.max().item()) + 1
        image = grad_image(ctx.w_np, L, N, dev, key=cfg.get("weights_key")) if ctx.w_np is not None else None
        D, npar = desc.n_state, (12 if desc.model == capi.MODEL_MARKOV6 else 8)
        state = torch.empty((B, 2 * D + npar), dtype=torch.float64, device=dev)
        g_params = torch.zeros((B, npar), dtype=torch.float64, device=dev)
        g_y0 = torch.zeros((B, D), dtype=torch.float64, device=dev)
        tiles = (B + 15) // 16
        recf = lib.ionode_grad_record_floats(L, N) if need_w else 0
        partf = lib.ionode_grad_partial_floats(L, N) if need_w else 0
        budget = _bounded_budget(cfg.get("record_budget_bytes"), DEFAULT_RECORD_BUDGET, dev, 0.5)
        chunk = n_iter if not need_w else max(1, min(n_iter, budget // (tiles * 6 * recf * 4)))
        acc = torch.zeros(partf, dtype=torch.float64, device=dev) if need_w else None
        main = torch.cuda.current_stream(dev)
        # Two-phase sweep (NN models; csrc/ionode_grad.hpp).  A stage's vector-Jacobian product is linear in its seed
        # (a scalar per trajectory) and everything else it needs comes from the step's checkpoint: phase A
        # (ionode_dopri5_backward_recompute) computes the UNIT-SEED products of every (tile, step) of a chunk at once on the whole
        # chip, on its own stream, one chunk AHEAD of phase B (ionode_dopri5_backward_sweep: the sequential walk, adjoint algebra
        # only, one wavefront per tile); the reduction of a finished chunk (ionode_grad_reduce_unit: records scaled by the seeds
        # the walk wrote) runs on a third stream.  Records and packets are double-buffered.
        two_phase = bool(cfg.get("two_phase", os.environ.get("IONODE_GRAD_ONE_PHASE", "0") != "1")) and image is not None
        if two_phase:   # small batch, small net, very long solve: every phase-A launch stays inside HIP's grid.y limit,
            chunk = min(chunk, MAX_RECOMPUTE_ITERS)   # decided BEFORE the chunk count and the buffering that follow from it
        n_chunks = (n_iter + chunk - 1) // chunk
        n_buf = 2 if (n_chunks > 1 and (need_w or two_phase)) else 1
        if need_w and n_buf == 2:
            chunk = max(1, min(n_iter, (budget // 2) // (tiles * 6 * recf * 4)))
            if two_phase:
                chunk = min(chunk, MAX_RECOMPUTE_ITERS)
            n_chunks = (n_iter + chunk - 1) // chunk
        elif two_phase and not need_w:
            # no record stream: the packets are what a chunk holds (two buffers); bounded chunks also keep phase A one chunk ahead
            per_it = tiles * int(lib.ionode_grad_packet_doubles()) * 8
            chunk = max(1, min(n_iter, 256, (budget // 2) // per_it))
            n_chunks = (n_iter + chunk - 1) // chunk
            n_buf = 2 if n_chunks > 1 else 1
        records = [torch.empty(tiles * chunk * 6 * recf, dtype=torch.float32, device=dev) for _ in range(n_buf)] if need_w else [None] * n_buf
        pkd = int(lib.ionode_grad_packet_doubles()) if two_phase else 0
        packets = [torch.empty(tiles * chunk * pkd, dtype=torch.float64, device=dev) for _ in range(n_buf)] if two_phase else [None] * n_buf
        side = torch.cuda.Stream(dev) if (need_w and n_buf == 2) else main          # reductions
        pre = torch.cuda.Stream(dev) if (two_phase and n_buf == 2) else main         # phase A
        if pre is not main:
            pre.wait_stream(main)    # gy / state / inputs were produced on the caller's stream
        free = [None] * n_buf   # event: the reduce (or, without weight gradients, the walk) that last used this buffer pair has finished
        ready = [None] * n_buf  # event: phase A has filled this buffer pair
        desc.ckpt, desc.ckpt_cap = ckpt.data_ptr(), ckpt.shape[1]
        common = (_ptr(image), _ptr(params), _ptr(cfg["prot_v"]), _ptr(cfg.get("prot_t")), _ptr(cfg.get("prot_of_traj")),
                  _ptr(cfg["t_eval"]), _ptr(n_acc))
        bounds = [(it0, min(n_iter, it0 + chunk)) for it0 in range(0, n_iter, chunk)]

        def phase_a(k):
            it0, it1 = bounds[k]
            b = k % n_buf
            if free[b] is not None:
                pre.wait_event(free[b])
            rc = lib.ionode_dopri5_backward_recompute(C.byref(desc), it0, it1, n_iter, *common, _ptr(gy), _ptr(records[b]),
                                                      _ptr(packets[b]), C.c_void_p(pre.cuda_stream))
            if rc != 0:
                raise capi.IonodeError(f"ionode_dopri5_backward_recompute failed ({rc}): {lib.ionode_grad_last_error().decode()}")
            ev = torch.cuda.Event()
            ev.record(pre)
            ready[b] = ev

        if two_phase:
            phase_a(0)
        for k, (it0, it1) in enumerate(bounds):
            b = k % n_buf
            rec = records[b]
            if two_phase:
                if k + 1 < len(bounds) and n_buf == 2:
                    phase_a(k + 1)                      # one chunk ahead, beside this chunk's walk
                main.wait_event(ready[b])
                rc = lib.ionode_dopri5_backward_sweep(C.byref(desc), it0, it1, n_iter, *common, _ptr(gy), _ptr(state), _ptr(rec),
                                                      _ptr(packets[b]), _ptr(g_params), _ptr(g_y0),
                                                      C.c_void_p(main.cuda_stream))
            else:
                if free[b] is not None:
                    main.wait_event(free[b])
                rc = lib.ionode_dopri5_backward(C.byref(desc), it0, it1, n_iter, *common, _ptr(gy), _ptr(state), _ptr(rec),
                                                _ptr(g_params), _ptr(g_y0), C.c_void_p(main.cuda_stream))
            if rc != 0:
                raise capi.IonodeError(f"backward sweep failed ({rc}): {lib.ionode_grad_last_error().decode()}")
            swept = torch.cuda.Event()
            swept.record(main)
            if need_w:
                n_rec = tiles * (it1 - it0) * 6
                n_slabs = int(os.environ.get("IONODE_GRAD_SLABS", 0)) or int(lib.ionode_grad_reduce_slabs(L, N, n_rec))   # one round of workgroups on this device (env: dev override for A/B runs)
                side.wait_event(swept)
                with torch.cuda.stream(side):
                    partials = torch.empty((n_slabs, partf), dtype=torch.float32, device=dev)
                    reduce = lib.ionode_grad_reduce_unit if two_phase else lib.ionode_grad_reduce   # unit-seed records: scaled while staged
                    rc = reduce(L, N, _ptr(rec), n_rec, n_slabs, _ptr(partials), C.c_void_p(side.cuda_stream))
                    if rc != 0:
                        raise capi.IonodeError(f"ionode_grad_reduce failed ({rc}): {lib.ionode_grad_last_error().decode()}")
                    acc += partials.double().sum(0)
                    done = torch.cuda.Event()
                    done.record(side)
                free[b] = done
            else:
                free[b] = swept
            if two_phase and n_buf == 1 and k + 1 < len(bounds):
                phase_a(k + 1)                          # single buffer: strictly alternate
        if need_w and side is not main:
            main.wait_stream(side)
        if pre is not main:
            main.wait_stream(pre)
        g_params[failed] = 0.0
        g_y0[failed] = 0.0
        g_w = unpack_partial(acc, L, N).to(torch.float32) if need_w else None
        return g_w, (g_params if ctx.needs_input_grad[1] else None), (g_y0.to(sdt) if ctx.needs_input_grad[2] else None), None


def solve(model, weights_flat, params, prot_v, y0, t_eval, *, mlp_layers=0, mlp_width=0, prot_t=None, prot_t0=0.0,
          prot_dt=1.0, prot_of_traj=None, rtol=1e-7, atol=1e-9, v_oob=-80.0, max_steps=0, max_total_steps=0, max_step=0.0,
          ckpt_cap=None, record_budget_bytes=None, weights_key=None, t_eval_hint="auto", order=None, tile_waves=0,
          ckpt_budget_bytes=None, two_phase=None):
    """Differentiable batched solve.  weights_flat [n] fp32 (reference state-dict order; None for the closed-form HH 2-state
    and 6-state models, whose params are [B, 8] / [B, 12] and y0 [B, 2] / [B, 6]), params [B, 8] fp64, y0 [B, 2]
    fp32 | fp64 (the state dtype) -- device tensors, any of which may require grad; prot_v [P, Np], t_eval [Nt] fp64 device
    tensors.  Returns (y [B, Nt, 2], status [B]): gradients of failed trajectories (status != 0) are zero -- their rows of the
    upstream gradient are ignored (y is NaN-filled there), and dL/dy0, dL/dp rows are zero whether or not the caller masks its loss.
    A failed trajectory never grows the checkpoint buffer; a batch whose successful trajectories need more than
    ckpt_budget_bytes (default 96 GiB) of checkpoints raises IonodeError instead of running out of memory.

    max_step (ms | "auto", extension; 0 = off = the reference's dopri5; "auto" = stable_step_cap(); a RuntimeWarning is issued
    when params requires grad and no cap is set): at an equilibrium dopri5 lets dt grow until h*lambda is far
    outside its stability region (the error estimate of a state AT equilibrium is ~0); the forward solve copes through
    rejections, but the exact derivative of those accepted-but-unstable steps multiplies the adjoint by |R(h*lambda)| >> 1
    per step (measured on the 10 s sine-wave protocol, fp32 state: |dL/dp| ~ 1e36 while dL/dW, whose state `a` is not
    stiff, stays O(100)).  max_step < 3.3 / lambda_max -- 10 ms for the reference's rate constants -- keeps it bounded.

    order (optional permutation of range(B), schedule.lpt_order): launch slot k integrates trajectory order[k] (homogeneous
    tiles, expensive tiles first -- pays from two tiles per compute unit, B > 4096); y and status are then in LAUNCH order
    (row k = trajectory order[k]), and the gradients still arrive at params / y0 in the caller's order (the gather is part of
    the autograd graph)."""
    if isinstance(max_step, str):
        if max_step != "auto":
            raise capi.IonodeError("max_step must be a number (ms) or 'auto'")
        max_step = stable_step_cap(model, params, prot_v, v_oob)
    elif float(max_step) == 0.0 and isinstance(params, torch.Tensor) and params.requires_grad:
        import warnings
        warnings.warn("gradients w.r.t. the rate parameters through an UNCAPPED dopri5 solve: at equilibria dopri5 accepts steps with "
                      "h*lambda >> 1 whose exact derivative amplifies rounding noise without bound (|dL/dp| ~ 1e36 on long holds, "
                      "DESIGN.md 5.4).  Pass max_step='auto' (= 3 / lambda_max of the rate constants, grad.stable_step_cap) or a "
                      "value in ms; the forward values then follow the capped step sequence.", RuntimeWarning, stacklevel=2)
    if model in (capi.MODEL_HH2, capi.MODEL_MARKOV6):
        # closed-form models (train-s1.py:161-177, train-d1.py:165-187): gradients w.r.t. the rate parameters and y0; no weights
        if weights_flat is not None:
            raise capi.IonodeError("the closed-form models have no MLP: pass weights_flat=None")
        mlp_layers = mlp_width = 0
    elif model not in (capi.MODEL_NNF, capi.MODEL_NND):
        raise NotImplementedError("unknown model")
    if not (isinstance(y0, torch.Tensor) and y0.is_cuda):
        raise capi.IonodeError("no HIP tensors: the integrator and its backward sweep have no CPU path")
    if order is not None:
        order = torch.as_tensor(order, dtype=torch.int64, device=y0.device)
        B = y0.shape[0]
        if order.shape != (B,) or int(order.min()) < 0 or int(order.max()) >= B or \
                not bool(torch.bincount(order, minlength=B).eq(1).all()):
            raise capi.IonodeError(f"order must be a permutation of range({B})")
        params, y0 = params.index_select(0, order), y0.index_select(0, order)
        pot = prot_of_traj if prot_of_traj is not None else (torch.arange(B, device=y0.device) % prot_v.shape[0]).to(torch.int32)
        prot_of_traj = torch.as_tensor(pot, device=y0.device).index_select(0, order).to(torch.int32).contiguous()
    cfg = dict(model=model, mlp_layers=int(mlp_layers), mlp_width=int(mlp_width), prot_v=prot_v, prot_t=prot_t,
               prot_t0=float(prot_t0), prot_dt=float(prot_dt), prot_of_traj=prot_of_traj, t_eval=t_eval, rtol=float(rtol),
               atol=float(atol), v_oob=float(v_oob), max_steps=int(max_steps), max_total_steps=int(max_total_steps),
               max_step=float(max_step), ckpt_cap=ckpt_cap, record_budget_bytes=record_budget_bytes, weights_key=weights_key, t_eval_hint=t_eval_hint,
               tile_waves=int(tile_waves), ckpt_budget_bytes=ckpt_budget_bytes)
    if two_phase is not None:   # None: the two-phase sweep unless IONODE_GRAD_ONE_PHASE=1; results agree to fp32 rounding (1e-5), not bit for bit: the seed multiplies at the end of the product
        cfg["two_phase"] = bool(two_phase)
    return _Solve.apply(weights_flat, params, y0.contiguous(), cfg)


def allreduce_gradients(tensors, group=None):
    """Sum gradient tensors over the ranks with ONE collective (flattened bucket): RCCL over xGMI under `nccl`.
    For s00 that is 201 801 fp32 = 807 KB -- latency-bound next to a >= 100 ms sweep."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    flat = torch.cat([t.reshape(-1).to(torch.float64) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    out, off = [], 0
    for t in tensors:
        out.append(flat[off:off + t.numel()].reshape(t.shape).to(t.dtype))
        off += t.numel()
    return out
