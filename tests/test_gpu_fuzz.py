"""-m gpu: seeded random sweep over the descriptor space against the oracle, bit for bit.

Each case draws: the model (HH 2-state, 6-state, NN-f, NN-d), the state dtype, the MLP shape (from the compiled widths), the batch
size (around the tile sizes), uniform or explicit protocol time grids with random step protocols, per-trajectory or shared
protocols, an output grid that is uniform / uniform-but-inexact / irregular / a single time / reaching beyond the protocol,
tolerances, per-interval and whole-solve step limits small enough to trip sometimes, the optional dt cap, the fused current
trace (with and without the protocol-at-outputs table) and a launch order.  Whatever comes out -- states, current, status,
step counters -- must equal the oracle's output exactly (NaN patterns included)."""
import numpy as np
import pytest
import torch

import kat_cases as K

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    model = [K.MODEL_HH2, K.MODEL_MARKOV6, K.MODEL_NNF, K.MODEL_NND][seed % 4]
    f32 = bool(rng.integers(0, 2))
    B = int(rng.choice([1, 3, 15, 16, 17, 33, 64, 65, 70]))
    P = int(rng.choice([1, 2, 5]))
    Np = int(rng.integers(150, 400))
    dt = float(rng.choice([0.5, 1.0, 2.0]))
    # random step protocols: 3-7 plateaus between -120 and +60 mV
    pv = np.empty((P, Np))
    for p in range(P):
        edges = np.sort(rng.choice(np.arange(5, Np - 5), size=int(rng.integers(2, 6)), replace=False))
        levels = rng.uniform(-120, 60, edges.size + 1)
        pv[p] = levels[np.searchsorted(edges, np.arange(Np), side="right")]
    explicit = bool(rng.integers(0, 3) == 0)
    t0 = float(rng.choice([0.0, 10.0]))
    pt = t0 + np.cumsum(rng.uniform(0.5, 1.5, Np) * dt) if explicit else None
    t_first = pt[0] if explicit else t0
    t_last = pt[-1] if explicit else t0 + (Np - 1) * dt
    kind = int(rng.integers(0, 5))
    if kind == 0:      # exact uniform grid on the protocol grid
        n = int(rng.integers(50, 300))
        te = t_first + np.arange(n) * ((t_last - t_first) / (Np - 1))
    elif kind == 1:    # uniform, not bit-exact
        te = np.linspace(t_first, t_last * 0.97, int(rng.integers(40, 500)))
    elif kind == 2:    # irregular increasing
        te = t_first + np.sort(rng.uniform(0, t_last - t_first, int(rng.integers(5, 200))))
        te[0] = t_first
        te = np.unique(te)
    elif kind == 3:    # two output times only
        te = np.array([t_first, t_first + 0.8 * (t_last - t_first)])
    else:              # reaches beyond the protocol: hold voltage (-80 mV) there
        te = np.linspace(t_first, t_last * 1.1, int(rng.integers(40, 300)))
    n_par = 12 if model == K.MODEL_MARKOV6 else 8
    base = K.P_M6 if model == K.MODEL_MARKOV6 else (K.P_NN_D if model == K.MODEL_NND else K.P_HH)
    params = np.tile(base, (B, 1)) * rng.uniform(0.7, 1.4, (B, n_par))
    if model == K.MODEL_MARKOV6:
        y0 = np.tile([0.0, 1.0, 0, 0, 0, 0], (B, 1)) + 0.0
    else:
        y0 = np.tile([0.0, 1.0], (B, 1)) + rng.uniform(0, 0.1, (B, 2)) * [1, -1]
    kw = dict(prot_t0=t0, prot_dt=dt, prot_t=pt, rtol=float(rng.choice([1e-7, 1e-5, 1e-9])), atol=float(rng.choice([1e-9, 1e-7])))
    if P > 1 and rng.integers(0, 2):
        kw["prot_of_traj"] = rng.integers(0, P, B).astype(np.int32)
    lim = int(rng.integers(0, 4))
    if lim == 1:
        kw["max_steps"] = int(rng.integers(3, 40))
    elif lim == 2:
        kw["max_total_steps"] = int(rng.integers(20, 400))
    if rng.integers(0, 4) == 0:
        kw["max_step"] = float(rng.uniform(2.0, 20.0))
    mlp = {}
    if model in (K.MODEL_NNF, K.MODEL_NND):
        L, N = [(1, 10), (5, 10), (2, 100), (5, 200), (1, 200), (1, 500)][int(rng.integers(0, 6))]
        w = rng.normal(0, min(0.3, 1.0 / np.sqrt(N)), 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)
        mlp = dict(weights=w, mlp_layers=L, mlp_width=N)
        if N == 500:
            B = min(B, 17)
            params, y0 = params[:B], y0[:B]
            if "prot_of_traj" in kw:
                kw["prot_of_traj"] = kw["prot_of_traj"][:B]
    if rng.integers(0, 8) == 0:
        y0[rng.integers(0, B), 0] = np.nan          # a trajectory that fails at once inside a healthy tile
    if f32:
        y0 = y0.astype(np.float32).astype(np.float64)   # the state dtype of the caller's y0
    obs = dict(obs_g=float(rng.choice([1.0, 0.7])), obs_e=float(rng.choice([-86.0, -80.0])),
               obs_open_state_only=bool(model == K.MODEL_MARKOV6))
    return model, f32, params, pv, y0, te, kw, mlp, obs, rng


import os


@pytest.mark.parametrize("seed", range(int(os.environ.get("IONODE_FUZZ_SEEDS", "56"))))
def test_random_descriptors_match_the_oracle(ion, gpu, oracle, seed):
    model, f32, params, pv, y0, te, kw, mlp, obs, rng = _case(seed)
    B = params.shape[0]
    o = oracle.solve(model, params, pv, y0, te, state_f32=f32, nthreads=4, **kw, **mlp)
    sdt = torch.float32 if f32 else torch.float64
    for variant in range(3):
        skw = dict(kw)
        if variant == 1:
            skw["order"] = torch.from_numpy(rng.permutation(B))
        if variant == 2 and (model in (K.MODEL_HH2, K.MODEL_MARKOV6) or mlp.get("mlp_width", 99) <= 16):
            skw["tile_waves"] = 64          # closed-form: 64 per wavefront; N <= 16 nets: the 64-per-wavefront MLP kernel
        sol = ion.solve(model, params, pv, torch.from_numpy(y0).to(sdt), te, current=True, **obs, **skw, **mlp)
        y = sol.to_original(sol.y).double().cpu().numpy()
        assert np.array_equal(sol.to_original(sol.status).cpu().numpy(), o["status"]), (seed, variant)
        assert np.array_equal(sol.to_original(sol.stats).cpu().numpy(), o["stats"]), (seed, variant)
        assert np.array_equal(y, o["y"], equal_nan=True), (seed, variant)
        # current trace: g * gate * (V - E) from the returned states and the oracle's protocol lookup
        i = sol.to_original(sol.i).cpu().numpy()
        pot = kw.get("prot_of_traj")
        for b in range(0, B, max(1, B // 5)):
            p = int(pot[b]) if pot is not None else b % pv.shape[0]
            v, _ = oracle.protocol_v(pv[p], te, prot_t=kw["prot_t"], prot_t0=kw["prot_t0"], prot_dt=kw["prot_dt"])
            want = oracle.current(o["y"][b], v, g=obs["obs_g"], e_rev=obs["obs_e"], state_f32=f32,
                                  open_state_only=obs["obs_open_state_only"])
            ok = o["status"][b] == 0
            if ok:
                assert np.array_equal(i[b], want), (seed, variant, b)
