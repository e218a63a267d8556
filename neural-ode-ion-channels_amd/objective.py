"""Batched candidate-model objective: the consumer of the batched solve that replaces the reference's
PINTS `ForwardModel.simulate` + `SumOfSquaresError` + process pool (train-d0.py:377-439, :508-540).

The reference evaluates ONE candidate at a time: for each protocol, `odeint(ODEFunc(p1..p4 = x), y0, t)`, current
`o[:,0,0]*o[:,0,1]*(V+86)`, and a SIGALRM time limit that turns a stuck solve into an `inf` error vector
(train-d0.py:426-438).  Here a whole CMA-ES population is one launch: C candidates x P protocols = C*P trajectories,
per-trajectory rate parameters, the current trace fused into the dense output, failed solves -> inf.  With several GPUs
the candidates are sharded contiguously and the per-candidate errors are all-gathered (the optimiser needs every
candidate's value on every rank); a scalar loss needs only `distributed.allreduce_sum_count`.
"""
import numpy as np
import torch

from . import batched, capi, distributed


def population_sum_of_squares(candidates, protocols_v, data_i, t_eval, *, base_params, free=(0, 1, 2, 3), prot_t0=0.0,
                              prot_dt=0.1, y0=(0.0, 1.0), state_dtype=torch.float32, obs_g=1.0, obs_e=-86.0,
                              max_total_steps=1_000_000, group=None, device=None, solver=None, fused=True, cost=None,
                              model=capi.MODEL_HH2, weights=None, mlp_layers=0, mlp_width=0, weights_key=None):
    """Sum-of-squares error of every candidate over all protocols (PINTS SumOfSquaresError on a multi-output problem).

    candidates  [C, len(free)]  values of the free rate parameters (train-d0.py: p1..p4 -> free = (0, 1, 2, 3))
    protocols_v [P, Np] mV;  data_i [P, Nt] measured currents;  t_eval [Nt] ms;  base_params [8]
    Returns a [C] fp64 tensor on the device: inf where any of a candidate's solves failed (the reference's time-limit
    rule).  Under torch.distributed the candidates are sharded over the ranks and the result is all-gathered.
    fused (default): the squared residuals are accumulated inside the kernel (ionode_desc.sse_ref / sse_out) and neither states
    nor current traces are written -- the only way BASELINE configs[3] fits (65 536 candidates x 32 sweeps x 1e5 samples would be
    4.5 TB of traces); fused=False stores the traces and reduces them with torch (same values to ~1e-13, for tests).
    model / weights / mlp_layers / mlp_width: the candidate model -- HH 2-state by default (train-d0.py), or NN-f / NN-d with one
    shared set of MLP weights and per-candidate rate parameters (free = (4, 5, 6, 7) for NN-f's p5..p8); weights [C, n]: a
    population of nets, candidate c integrated with its own weight set (free may then be empty: candidates [C, 0]).
    cost (optional, [C]): predicted cost per candidate (e.g. the previous generation's step counts): the candidates are then
    cut into contiguous shards of equal cost rather than equal count (distributed.shard_bounds_by_cost).
    `solver` (tests only): a stand-in with batched.solve's signature, so the sharding / all-gather logic can run under
    gloo on a box without a GPU; the product default is the HIP solve and there is no CPU fallback.
    """
    import torch.distributed as dist
    cand = np.asarray(candidates, dtype=np.float64)
    C, P = cand.shape[0], np.asarray(protocols_v).shape[0]
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if on else (0, 1)
    if cost is not None and len(cost) != C:
        raise capi.IonodeError(f"cost has {len(cost)} entries for {C} candidates: the shards would not cover the population")
    bounds = [distributed.shard_bounds(C, r, world) for r in range(world)] if cost is None \
        else distributed.shard_bounds_by_cost(cost, world)
    lo, hi = bounds[rank]
    dev = batched._dev(device) if solver is None else torch.device(device or "cpu")
    solve = batched.solve if solver is None else solver
    mlp = {} if weights is None else dict(weights=weights, mlp_layers=mlp_layers, mlp_width=mlp_width, weights_key=weights_key)
    # a population of NETS: weights [C, n] -- candidate c has its own MLP weights (and its rate parameters).  Every candidate then
    # owns whole 16-trajectory tiles (ionode_desc.traj_per_image): its P sweeps are padded to a multiple of 16 with repeats of
    # sweep 0 whose scores are dropped.
    per_cand = weights is not None and np.asarray(weights).ndim == 2
    S = P
    if per_cand:
        if np.asarray(weights).shape[0] != C:
            raise capi.IonodeError("weights [C, n]: one weight set per candidate")
        S = 16 * ((P + 15) // 16)
        mlp.update(weights=np.asarray(weights)[lo:hi], traj_per_image=S, weights_key=None)
    sse = torch.full((hi - lo,), float("inf"), dtype=torch.float64, device=dev)
    if hi > lo:
        params = np.tile(np.asarray(base_params, dtype=np.float64), (hi - lo, 1))
        params[:, list(free)] = cand[lo:hi]
        params = np.repeat(params, S, axis=0)                       # candidate-major: trajectory = c*S + p
        pot = np.tile(np.where(np.arange(S) < P, np.arange(S), 0).astype(np.int32), hi - lo)
        ref = torch.as_tensor(np.asarray(data_i), dtype=torch.float64, device=dev)     # [P, Nt]
        if fused and solver is None:
            sol = solve(model, params, protocols_v, torch.tensor([list(y0)], dtype=state_dtype), t_eval,
                        prot_t0=prot_t0, prot_dt=prot_dt, prot_of_traj=pot, obs_g=obs_g, obs_e=obs_e,
                        max_total_steps=max_total_steps, device=dev, sse_ref=ref.contiguous(), states=False, **mlp)
            err = sol.sse.reshape(hi - lo, S)[:, :P].sum(dim=1)      # failed solves are inf already
        else:
            sol = solve(model, params, protocols_v, torch.tensor([list(y0)], dtype=state_dtype), t_eval,
                        prot_t0=prot_t0, prot_dt=prot_dt, prot_of_traj=pot, current=True, obs_g=obs_g, obs_e=obs_e,
                        max_total_steps=max_total_steps, device=dev, **mlp)
            err = ((sol.i.reshape(hi - lo, S, -1)[:, :P] - ref[None]) ** 2).sum(dim=(1, 2))
        ok = (sol.status.reshape(hi - lo, S)[:, :P] == 0).all(dim=1)
        sse = torch.where(ok, err, torch.full_like(err, float("inf")))
    if world == 1:
        return sse
    # all-gather of unequal shards: pad to the largest shard
    m = max(b[1] - b[0] for b in bounds)
    pad = torch.full((m,), float("inf"), dtype=torch.float64, device=dev)
    pad[: hi - lo] = sse
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: bounds[r][1] - bounds[r][0]] for r in range(world)])
