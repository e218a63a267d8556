"""-m gpu, round 4: the 4-trajectory tile of the N = 200 nets (MlpTile4: small batches, single odeint calls), the lean variants and the
per-lane vector-ALU net -- every new kernel form returns the oracle's bits."""
import numpy as np
import pytest
import torch

import kat_cases as K
from gpu_util import run_gpu

pytestmark = pytest.mark.gpu


def _kernel(ion, gpu, model, params, pv, y0, te, **kw):
    """run_gpu + the name of the kernel that ran"""
    g = run_gpu(ion, gpu, model, params, pv, y0, te, **kw)
    g["kernel"] = ion.capi.lib().ionode_last_kernel_name().decode()
    return g


@pytest.mark.parametrize("name,model", [("s1", K.MODEL_NNF), ("d2", K.MODEL_NND)])
@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("B", [1, 5, 37])
def test_four_trajectory_tile_is_bit_identical(ion, gpu, oracle, name, model, f32, B):
    """tile_waves = 2: 4 trajectories per tile on v_mfma_f32_4x4x1 (16 blocks of 4 rows x 4 trajectories x 1 k), same canonical chains as
    the 16-column tile: states, step counters and the fused current trace equal the oracle's and the 16-column kernel's bit for bit --
    single call, ragged tiles, per-trajectory protocols, one trajectory that fails."""
    rng = np.random.default_rng(100 * B + f32)
    w = K.load_weights(name)
    base = K.P_NN_D if model == K.MODEL_NND else K.P_HH
    params = np.tile(base, (B, 1)) * rng.uniform(0.9, 1.1, (B, 8))
    pv = np.stack([K.activation(v)[1] for v in (-20, 20, 40)])
    te = K.activation(0)[2][:1501]
    pot = rng.integers(0, 3, B).astype(np.int32)
    y0 = np.tile(K.NN_Y0, (B, 1)).astype(np.float64)
    if B > 4:
        y0[3, 1] = np.nan   # a failing trajectory inside a tile
    kw = dict(prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot, max_total_steps=20000)
    o = oracle.solve(model, params, pv, y0, te, weights=w, mlp_layers=5, mlp_width=200, state_f32=f32, nthreads=4, **kw)
    g4 = _kernel(ion, gpu, model, params, pv, y0, te, weights=w, L=5, N=200, f32=f32, current=True, tile_waves=2, **kw)
    g16 = _kernel(ion, gpu, model, params, pv, y0, te, weights=w, L=5, N=200, f32=f32, current=True, tile_waves=4, **kw)
    assert ", 4, 4, 13, 13, 24>" in g4["kernel"] and ", 4, 4, 13, 13, 8>" in g16["kernel"], (g4["kernel"], g16["kernel"])
    for g in (g4, g16):
        assert np.array_equal(g["status"], o["status"]) and np.array_equal(g["stats"], o["stats"])
        assert np.array_equal(g["y"], o["y"], equal_nan=True)
    assert np.array_equal(g4["i"], g16["i"], equal_nan=True)
    # small batches choose a small tile by themselves (round 5: up to 256 trajectories the one-trajectory tile, then this one)
    auto = _kernel(ion, gpu, model, params, pv, y0, te, weights=w, L=5, N=200, f32=f32, **kw)
    assert ", 4, 4, 13, 13, 40>" in auto["kernel"] and np.array_equal(auto["y"], o["y"], equal_nan=True)


@pytest.mark.parametrize("L", [1, 2, 4])
def test_four_trajectory_tile_other_depths_and_general_variant(ion, gpu, oracle, L):
    """Odd / even hidden-layer counts (the activation buffers ping-pong), an explicit protocol time grid (takes the GENERAL variant,
    TAIL slot 16) and a step log."""
    rng = np.random.default_rng(L)
    N, B = 200, 6
    w = rng.normal(0, 0.1, 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)
    params = np.tile(K.P_HH, (B, 1)) * rng.uniform(0.9, 1.1, (B, 8))
    pt, pv, te = K.atau(30)
    te = te[:801]
    o = oracle.solve(K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, mlp_layers=L, mlp_width=N, prot_t=pt)
    g = _kernel(ion, gpu, K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, prot_t=pt, tile_waves=2)
    assert ", 4, 4, 13, 13, 16>" in g["kernel"], g["kernel"]
    assert np.array_equal(g["y"], o["y"]) and np.array_equal(g["stats"], o["stats"])
    o2 = oracle.solve(K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, mlp_layers=L, mlp_width=N, prot_t0=float(pt[0]), prot_dt=float(pt[1] - pt[0]))
    g2 = _kernel(ion, gpu, K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, prot_t0=float(pt[0]), prot_dt=float(pt[1] - pt[0]), tile_waves=2)
    assert ", 4, 4, 13, 13, 24>" in g2["kernel"] and np.array_equal(g2["y"], o2["y"]) and np.array_equal(g2["stats"], o2["stats"])


def test_single_odeint_call_takes_the_small_tile(ion, gpu, oracle):
    """The reference's own call shape -- odeint(func, y0, t) with one trajectory (train-s1.py:319-330) -- runs on the 4-trajectory tile."""
    import ref_style_modules as M
    from torchdiffeq import odeint
    func = M.NNf(K.MODELS["s1"][4])
    M.load_flat_weights(func.net, K.load_weights("s1"))
    func.eval()
    pt, pv, te = K.activation(20)
    func.set_fixed_form_voltage_protocol(pt, pv)
    with torch.no_grad():
        y = odeint(func, torch.tensor([K.NN_Y0]), torch.from_numpy(te).float())
    assert ", 4, 4, 13, 13, " in ion.capi.lib().ionode_last_kernel_name().decode()
    o = oracle.solve(K.MODEL_NNF, K.MODELS["s1"][4], pv, K.NN_Y0, te, weights=K.load_weights("s1"), mlp_layers=5, mlp_width=200,
                     prot_t0=float(pt[0]), prot_dt=float(pt[1] - pt[0]), state_f32=True)
    assert np.array_equal(y[:, 0, :].double().cpu().numpy(), o["y"][0])


@pytest.mark.parametrize("model,f32", [(K.MODEL_HH2, False), (K.MODEL_HH2, True), (K.MODEL_MARKOV6, False), (K.MODEL_MARKOV6, True)])
@pytest.mark.parametrize("tpw", [64, 16])
def test_work_list_emission_with_short_and_very_long_steps(ion, gpu, oracle, model, f32, tpw):
    """The dense output of the lane-wise kernels is driven by a work list of 8-sample chunks; steps of more than 64 samples are emitted
    by the whole wavefront.  Step protocols with long holds on a 1/16 ms output grid give both in one launch -- steps of 1 .. 3
    samples right behind a voltage jump, steps of several hundred to > 1000 samples on the holds (two and more whole-wavefront
    passes), trajectories that finish early -- in the lean variant (exact grid, states only), bit for bit against the oracle."""
    rng = np.random.default_rng(17 + tpw + f32)
    B = 150
    D = 6 if model == K.MODEL_MARKOV6 else 2
    base = K.P_M6 if model == K.MODEL_MARKOV6 else K.P_HH
    params = np.tile(base, (B, 1)) * rng.uniform(0.7, 1.4, (B, base.size))
    pv = np.stack([K.activation(v)[1][:601] for v in (-40, 20, 60)])
    te = np.arange(0, 9601) * 0.0625   # (a binary fraction: the grid is exactly uniform, which the lean variant requires)
    y0 = [0.0, 1.0] + [0.0] * (D - 2)
    pot = rng.integers(0, 3, B).astype(np.int32)
    kw = dict(prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot)
    o = oracle.solve(model, params, pv, y0, te, state_f32=f32, nthreads=8, **kw)
    g = _kernel(ion, gpu, model, params, pv, y0, te, f32=f32, tile_waves=tpw, **kw)
    assert ", 1, %d, 0, 0, 1>" % (0 if tpw == 64 else 16) in g["kernel"], g["kernel"]
    steps = o["stats"][:, 0]
    assert steps.min() < 9600 / 64          # some trajectory's average accepted step covers more than 64 output samples
    assert np.array_equal(g["stats"], o["stats"]) and np.array_equal(g["y"], o["y"])
    # ... and the same solve with the current trace (general / table variants) returns the same states
    gi = _kernel(ion, gpu, model, params, pv, y0, te, f32=f32, tile_waves=tpw, current=True, obs_open_state_only=(D == 6), **kw)
    assert np.array_equal(gi["y"], o["y"])


def test_per_lane_net_with_several_weight_sets(ion, gpu, oracle):
    """N = 10 at 64 trajectories per wavefront (MlpLane: weights as scalar operands) with traj_per_image = 64: every wavefront reads
    its own image through the scalar cache; bit for bit one oracle run per weight set."""
    rng = np.random.default_rng(5)
    L, N, n_sets = 5, 10, 3
    B = 64 * n_sets
    ws = [rng.normal(0, 0.3, 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32) for _ in range(n_sets)]
    params = np.tile(K.P_HH, (B, 1)) * rng.uniform(0.9, 1.1, (B, 8))
    pv = np.stack([K.atau(30)[1], K.atau(300)[1]])
    te = K.atau(30)[2][:901]
    pot = rng.integers(0, 2, B).astype(np.int32)
    host = np.stack([ion.capi.mlp_pack(w, L, N) for w in ws])
    # the images sit at the very END of a device allocation of their own: the launch is rounded up to 32 tiles for 3 tiles of work,
    # and a tile past the batch must not touch "its" image (image 3 .. 31 would lie past the array -- a GPU memory fault when the
    # next page is unmapped, as it was on one box in round 4)
    arena = torch.empty(64 << 20, dtype=torch.uint8, device=gpu)
    packed = arena[arena.numel() - host.nbytes:].view(torch.float32).view(host.shape)
    packed.copy_(torch.from_numpy(host))
    r = ion.capi.dopri5(K.MODEL_NNF, torch.from_numpy(params).to(gpu), torch.from_numpy(pv).to(gpu),
                        torch.tensor([K.NN_Y0], dtype=torch.float64, device=gpu).repeat(B, 1).contiguous(), torch.from_numpy(te).to(gpu),
                        mlp_packed=packed, mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=1.0,
                        prot_of_traj=torch.from_numpy(pot).to(gpu), traj_per_image=64, tile_waves=64)
    assert ", 1, 64, 1, 10, " in r["kernel"]
    for k, w in enumerate(ws):
        sl = slice(64 * k, 64 * (k + 1))
        o = oracle.solve(K.MODEL_NNF, params[sl], pv, K.NN_Y0, te, weights=w, mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot[sl])
        assert np.array_equal(r["y"][sl].cpu().numpy(), o["y"]) and np.array_equal(r["stats"][sl].cpu().numpy(), o["stats"])


@pytest.mark.parametrize("L,N", [(5, 10), (2, 100), (1, 500)])
def test_two_phase_sweep_equals_the_one_phase_sweep_at_other_widths(ion, gpu, L, N):
    """the factored two-phase sweep (the default backward path) was only compared with the one-phase sweep at N = 200.
    Widths 10, 100 and 500 (their own forward / backward tile kernels and reduce geometry), random nets, fp32 state:
    dL/dW, dL/dp, dL/dy0 agree to fp32 rounding (1e-5; the checker's tolerance is 1e-4)."""
    import importlib
    import warnings
    grad = importlib.import_module("neural-ode-ion-channels_amd.grad")
    capi = ion.capi
    B, Nt = 21, 2001
    P = ion.protocols
    pv = P.sinewave(P.sinewave_scales(0, B), n_samples=Nt, xp=torch, device=gpu)
    te = torch.arange(0, Nt, 4, dtype=torch.float64, device=gpu) * 0.1
    w0 = np.random.default_rng(N).normal(0, 0.1, 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)
    got = {}
    for tag, kw in (("one", dict(two_phase=False)), ("two", dict(two_phase=True))):
        w = torch.from_numpy(w0.copy()).to(gpu).requires_grad_(True)
        params = torch.from_numpy(np.tile(K.P_HH, (B, 1)) * np.random.default_rng(5).uniform(0.9, 1.1, (B, 8))).to(gpu).requires_grad_(True)
        y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float32, device=gpu).repeat(B, 1).requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            y, status = grad.solve(capi.MODEL_NNF, w, params, pv, y0, te, mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=0.1, **kw)
        assert bool((status == 0).all())
        (y[..., 0] * y[..., 1]).double().sum().backward()
        got[tag] = (w.grad.clone(), params.grad.clone(), y0.grad.clone())
    for a, b in zip(got["two"], got["one"]):
        assert float((a.double() - b.double()).norm() / b.double().norm()) < 1e-5


def test_two_phase_sweep_without_weight_gradients_over_several_packet_chunks(ion, gpu):
    """weights that do NOT require a gradient and more than 256 accepted steps -- the branch of the two-phase sweep that keeps
    no record stream, double-buffers the packets in chunks of <= 256 iterations and frees a buffer when the WALK (not a reduction) is
    done with it.  dL/dp and dL/dy0 equal the one-phase sweep's bit for bit (the adjoint algebra does not depend on the chunking) and
    the two-phase run WITH weight gradients."""
    import importlib
    import warnings
    grad = importlib.import_module("neural-ode-ion-channels_amd.grad")
    capi = ion.capi
    B, Nt = 19, 20001
    P = ion.protocols
    pv = P.sinewave(P.sinewave_scales(3, B), n_samples=Nt, xp=torch, device=gpu)
    te = torch.arange(0, Nt, 8, dtype=torch.float64, device=gpu) * 0.1
    got = {}
    for tag, wgrad, kw in (("two_nograd", False, dict(two_phase=True)), ("two_grad", True, dict(two_phase=True)), ("one_nograd", False, dict(two_phase=False))):
        w = torch.from_numpy(K.load_weights("s1").copy()).to(gpu).requires_grad_(wgrad)
        params = torch.from_numpy(np.tile(K.P_HH, (B, 1)) * np.random.default_rng(9).uniform(0.9, 1.1, (B, 8))).to(gpu).requires_grad_(True)
        y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64, device=gpu).repeat(B, 1).requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            y, status = grad.solve(capi.MODEL_NNF, w, params, pv, y0, te, mlp_layers=5, mlp_width=200, prot_t0=0.0, prot_dt=0.1, **kw)
        assert bool((status == 0).all())
        (y[..., 0] * y[..., 1]).double().sum().backward()
        got[tag] = (params.grad.clone(), y0.grad.clone())
    rel = lambda a, b: float((a - b).norm() / b.norm())
    for k in (0, 1):
        assert rel(got["two_nograd"][k], got["one_nograd"][k]) < 1e-5, (k, rel(got["two_nograd"][k], got["one_nograd"][k]))
        assert torch.equal(got["two_nograd"][k], got["two_grad"][k])   # same walk, with and without the record stream
