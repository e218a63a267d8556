"""Launch-order planning for the MLP kernels: which trajectories share a 16-wide MFMA tile, and which tiles start first.

A tile integrates its 16 trajectories in lock step until the slowest one finishes, and the hardware hands
workgroups to compute units in grid order as units free up.  So the launch order decides two things: how much of a tile's
time is spent on lanes that have already finished (tile-max over tile-mean NFE, ~9 % with an arbitrary order), and how long
the last tiles run alone at the end of the launch.  Sorting the trajectories by predicted cost, most expensive first,
makes tiles homogeneous and turns the dispatcher into a longest-processing-time-first list scheduler.  With one tile per
compute unit (B = 16 x 256 = 4096) the launch is as long as its slowest trajectory whatever the order; the order pays
from two tiles per unit upwards (BASELINE configs[2], [3]).

The reference has no counterpart: it solves one trajectory per `odeint` call (train-s1.py:566-580).  SURVEY.md 8(e)
asks for protocol-binned sharding; `lpt_order` of a per-protocol cost gives exactly that as a special case.
"""
import numpy as np
import torch

from . import batched, capi


def lpt_order(cost):
    """Permutation of range(B), most expensive trajectory first (stable), for `batched.solve(order=...)`.

    cost: [B] predicted cost -- the `stats[:, 2]` (RHS evaluations) of an earlier solve of the same trajectories (training
    loops re-solve the same protocols with slowly changing weights), `pilot_cost()`, or any per-protocol activity measure.
    """
    c = cost if isinstance(cost, torch.Tensor) else torch.from_numpy(np.asarray(cost))
    return torch.argsort(c.to(torch.float64), descending=True, stable=True)


def pilot_cost(params, prot_v, t0, t1, *, prot_t0=0.0, prot_dt=1.0, prot_t=None, prot_of_traj=None, rtol=1e-7, atol=1e-9,
               v_oob=-80.0, y0=(0.0, 1.0), device=None):
    """Predicted step count of every trajectory from a pilot solve with the closed-form Hodgkin-Huxley right-hand side.

    The step-size controller follows the voltage protocol far more than the exact form of the activation rate, so the
    number of RHS evaluations of the HH 2-state model (train-s1.py:161-177) over [t0, t1] on the same protocol, with the
    same tolerances and the trajectory's own p1..p8, ranks the trajectories almost as the MLP models' own counts do.
    Two output times, no dense output: the pilot costs well under 1 % of an s00 solve.  Returns [B] fp64 on the device.
    """
    sol = batched.solve(capi.MODEL_HH2, params, prot_v, torch.tensor([list(y0)], dtype=torch.float64),
                        torch.tensor([float(t0), float(t1)], dtype=torch.float64), prot_t=prot_t, prot_t0=prot_t0,
                        prot_dt=prot_dt, prot_of_traj=prot_of_traj, rtol=rtol, atol=atol, v_oob=v_oob, device=device,
                        t_eval_hint=None)
    return sol.stats[:, 2].to(torch.float64)


def protocol_order(prot_of_traj):
    """Launch order that puts the trajectories of one protocol next to each other (stable: ties keep their index order).
    The lanes of a closed-form wavefront then interpolate ONE protocol instead of up to 64 different ones: the sample loads of a
    stage lookup coalesce, and the protocols' footprint per XCD L2 shrinks -- HH 2-state, 393 216 x 20 001 on 64 protocols
    (10 MB against 4 MB of L2 per XCD): 44.7 -> 42.5 ms.  Use as `order=` of batched.solve / grad.solve (results are unchanged;
    rows come back in launch order, Solution.to_original() undoes it) or apply it to the inputs directly."""
    import torch
    p = torch.as_tensor(prot_of_traj)
    return torch.sort(p.to(torch.int64), stable=True).indices
