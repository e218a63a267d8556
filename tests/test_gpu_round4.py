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
    # chosen by itself for small batches
    auto = _kernel(ion, gpu, model, params, pv, y0, te, weights=w, L=5, N=200, f32=f32, **kw)
    assert ", 4, 4, 13, 13, 24>" in auto["kernel"] and np.array_equal(auto["y"], o["y"], equal_nan=True)


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
