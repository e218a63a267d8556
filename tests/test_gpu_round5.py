"""-m gpu, round 5: the run-time-width MLP tile (any N <= 512) and the one-trajectory tile of the N = 200 nets (a lane owns a row:
the reference's own odeint(func, y0, t) call shape), every result against the oracle bit for bit."""
import numpy as np
import pytest
import torch

import kat_cases as K
from gpu_util import run_gpu

pytestmark = pytest.mark.gpu


def _rand_weights(L, N, seed):
    n = 2 * N + N + L * (N * N + N) + N + 1
    return np.random.default_rng(seed).normal(0, 0.1, n).astype(np.float32)


def _same(g, o):
    assert np.array_equal(g["status"], o["status"])
    assert np.array_equal(g["stats"], o["stats"])
    assert np.array_equal(g["y"], o["y"], equal_nan=True)


@pytest.mark.parametrize("L,N", [(3, 17), (2, 32), (5, 48), (4, 64), (1, 80), (3, 150), (2, 300), (0, 64), (12, 40)])
@pytest.mark.parametrize("f32", [False, True])
def test_any_width_integrates_through_the_run_time_width_tile(ion, gpu, oracle, L, N, f32):
    """table-s1.py:145-153 builds Linear(2, N) ... Linear(N, 1) for ANY (n_layers, n_nodes); widths without a tuned tile
    (everything but N <= 16, 100, 200, 500) used to return IONODE_ERR_UNSUPPORTED.  NT = ceil(N / 16) from 2 (all remainder tiles)
    over 3, 4 (no remainder), 5, 10 to 19; an odd and an even depth, no hidden layer at all, a deep stack; ragged last tile;
    both state dtypes, general and lean variants, NN-f and NN-d."""
    if f32 and (L, N) not in ((3, 17), (4, 64), (3, 150)):
        pytest.skip("fp32 state shares the MLP path; three shapes suffice")
    capi = ion.capi
    w = _rand_weights(L, N, 7 * L + N)
    pv = np.stack([K.atau(30)[1], K.atau(300)[1]])
    te = K.atau(30)[2][:801]
    B = 21
    params = np.tile(K.P_HH, (B, 1)) * np.random.default_rng(5).uniform(0.9, 1.1, (B, 8))
    kw = dict(prot_t0=0.0, prot_dt=1.0, prot_of_traj=(np.arange(B) % 2).astype(np.int32), max_total_steps=3000)
    for model in ((K.MODEL_NNF, K.MODEL_NND) if (L, N) in ((4, 64), (3, 150)) else (K.MODEL_NNF,)):
        o = oracle.solve(model, params, pv, K.NN_Y0, te, weights=w, mlp_layers=L, mlp_width=N, state_f32=f32, **kw)
        g = run_gpu(ion, gpu, model, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, f32=f32, **kw)            # lean variant (uniform grids)
        _same(g, o)
        slog = torch.zeros((64, 4), dtype=torch.float64, device=gpu)
        g2 = run_gpu(ion, gpu, model, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, f32=f32, step_log=slog, current=True, **kw)   # general
        _same(g2, o)
    d = capi.make_desc(model=capi.MODEL_NNF, n_state=2, n_out=te.size, n_traj=B, n_prot=2, prot_n=pv.shape[1], mlp_layers=L, mlp_width=N,
                       n_params=8, prot_dt=1.0, rtol=1e-7, atol=1e-9)
    assert ", 4, 1, 0, 1, " in capi.kernel_name(d)


def test_run_time_width_tile_with_several_weight_sets_and_a_launch_order(ion, gpu, oracle):
    """The generic tile honours traj_per_image (16 trajectories per image) and launch_order like the tuned tiles."""
    capi = ion.capi
    L, N, B = 2, 72, 48
    ws = [_rand_weights(L, N, s) for s in (1, 2, 3)]
    packed = torch.from_numpy(np.stack([capi.mlp_pack(w, L, N) for w in ws])).to(gpu)
    pv = K.atau(100)[1][None, :]
    te = K.atau(100)[2][:601]
    params = np.tile(K.P_HH, (B, 1)) * np.random.default_rng(9).uniform(0.9, 1.1, (B, 8))
    r = capi.dopri5(capi.MODEL_NNF, torch.from_numpy(params).to(gpu), torch.from_numpy(pv).to(gpu),
                    torch.tensor([K.NN_Y0], dtype=torch.float64, device=gpu).repeat(B, 1).contiguous(), torch.from_numpy(te).to(gpu),
                    mlp_packed=packed, mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=1.0, traj_per_image=16, max_total_steps=3000)
    torch.cuda.synchronize()
    for k, w in enumerate(ws):
        o = oracle.solve(K.MODEL_NNF, params[16 * k:16 * k + 16], pv, K.NN_Y0, te, weights=w, mlp_layers=L, mlp_width=N,
                         prot_t0=0.0, prot_dt=1.0, max_total_steps=3000)
        assert np.array_equal(r["y"][16 * k:16 * k + 16].cpu().numpy(), o["y"]) and np.array_equal(r["stats"][16 * k:16 * k + 16].cpu().numpy(), o["stats"])
    order = torch.from_numpy(np.random.default_rng(0).permutation(B).astype(np.int32)).to(gpu)
    one = torch.from_numpy(capi.mlp_pack(ws[0], L, N)).to(gpu)
    args = (capi.MODEL_NNF, torch.from_numpy(params).to(gpu), torch.from_numpy(pv).to(gpu),
            torch.tensor([K.NN_Y0], dtype=torch.float64, device=gpu).repeat(B, 1).contiguous(), torch.from_numpy(te).to(gpu))
    kw = dict(mlp_packed=one, mlp_layers=L, mlp_width=N, prot_t0=0.0, prot_dt=1.0, max_total_steps=3000)
    a, b = capi.dopri5(*args, **kw), capi.dopri5(*args, launch_order=order, **kw)
    torch.cuda.synchronize()
    assert torch.equal(a["y"], b["y"]) and torch.equal(a["stats"], b["stats"])


def _kernel(ion, gpu, model, params, pv, y0, te, **kw):
    g = run_gpu(ion, gpu, model, params, pv, y0, te, **kw)
    g["kernel"] = ion.capi.lib().ionode_last_kernel_name().decode()
    return g


@pytest.mark.parametrize("name,model", [("s1", K.MODEL_NNF), ("d2", K.MODEL_NND)])
@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("B", [1, 3, 7])
def test_one_trajectory_tile_is_bit_identical(ion, gpu, oracle, name, model, f32, B):
    """tile_waves = 16: ONE trajectory per workgroup, a lane owns a row (v_fmac_f32_dpp, activations broadcast inside quads), the same
    canonical chains as the 16-column and the 4-trajectory tiles: states, step counters and the fused current trace equal the oracle's
    and both other kernels' bit for bit -- single call, several trajectories, per-trajectory protocols, one trajectory that fails."""
    rng = np.random.default_rng(200 * B + f32)
    w = K.load_weights(name)
    base = K.P_NN_D if model == K.MODEL_NND else K.P_HH
    params = np.tile(base, (B, 1)) * rng.uniform(0.9, 1.1, (B, 8))
    pv = np.stack([K.activation(v)[1] for v in (-20, 20, 40)])
    te = K.activation(0)[2][:1501]
    pot = rng.integers(0, 3, B).astype(np.int32)
    y0 = np.tile(K.NN_Y0, (B, 1)).astype(np.float64)
    if B > 4:
        y0[3, 1] = np.nan   # a failing trajectory
    kw = dict(prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot, max_total_steps=20000)
    o = oracle.solve(model, params, pv, y0, te, weights=w, mlp_layers=5, mlp_width=200, state_f32=f32, nthreads=4, **kw)
    g1 = _kernel(ion, gpu, model, params, pv, y0, te, weights=w, L=5, N=200, f32=f32, current=True, tile_waves=16, **kw)
    g4 = _kernel(ion, gpu, model, params, pv, y0, te, weights=w, L=5, N=200, f32=f32, current=True, tile_waves=2, **kw)
    g16 = _kernel(ion, gpu, model, params, pv, y0, te, weights=w, L=5, N=200, f32=f32, current=True, tile_waves=4, **kw)
    assert ", 4, 4, 13, 13, 40>" in g1["kernel"] and ", 4, 4, 13, 13, 24>" in g4["kernel"] and ", 4, 4, 13, 13, 8>" in g16["kernel"]
    for g in (g1, g4, g16):
        assert np.array_equal(g["status"], o["status"]) and np.array_equal(g["stats"], o["stats"])
        assert np.array_equal(g["y"], o["y"], equal_nan=True)
    assert np.array_equal(g1["i"], g16["i"], equal_nan=True) and np.array_equal(g1["i"], g4["i"], equal_nan=True)


@pytest.mark.parametrize("L", [1, 2, 4, 6])
def test_one_trajectory_tile_other_depths_general_variant_objective_and_images(ion, gpu, oracle, L):
    """Odd / even hidden-layer counts (the activation buffers ping-pong), an explicit protocol time grid (the GENERAL variant, TAIL slot
    32) with a step log, the fused objective, no output-grid hint (cooperative scan), and several weight sets (one per trajectory)."""
    capi = ion.capi
    rng = np.random.default_rng(L)
    N, B = 200, 3
    w = rng.normal(0, 0.1, 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)
    params = np.tile(K.P_HH, (B, 1)) * rng.uniform(0.9, 1.1, (B, 8))
    pt, pv, te = K.atau(30)
    te = te[:801]
    o = oracle.solve(K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, mlp_layers=L, mlp_width=N, prot_t=pt)
    slog = torch.zeros((4000, 4), dtype=torch.float64, device=gpu)
    g = _kernel(ion, gpu, K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, prot_t=pt, tile_waves=16, step_log=slog)
    assert ", 4, 4, 13, 13, 32>" in g["kernel"], g["kernel"]
    assert np.array_equal(g["y"], o["y"]) and np.array_equal(g["stats"], o["stats"])
    n_att = int(o["stats"][0, 0] + o["stats"][0, 1])
    assert float(slog[:n_att, 3].sum()) == float(o["stats"][0, 0]) and float(slog[n_att:, 1].abs().sum()) == 0.0
    g_nohint = _kernel(ion, gpu, K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, prot_t=pt, tile_waves=16, t_eval_hint=None)
    assert np.array_equal(g_nohint["y"], o["y"]) and np.array_equal(g_nohint["stats"], o["stats"])
    # fused objective against the 16-column tile (same sum per trajectory: one wavefront reduces 64-sample chunks in both)
    kwu = dict(prot_t0=float(pt[0]), prot_dt=float(pt[1] - pt[0]))
    ref = torch.from_numpy(rng.normal(0, 0.3, (1, te.size))).to(gpu)
    args = (capi.MODEL_NNF, torch.from_numpy(params).to(gpu), torch.from_numpy(pv[None, :]).to(gpu),
            torch.tensor([K.NN_Y0], dtype=torch.float64, device=gpu).repeat(B, 1).contiguous(), torch.from_numpy(te).to(gpu))
    kwo = dict(mlp_packed=torch.from_numpy(capi.mlp_pack(w, L, N)).to(gpu), mlp_layers=L, mlp_width=N, sse_ref=ref, states=False, **kwu)
    s1, s16 = capi.dopri5(*args, tile_waves=16, **kwo), capi.dopri5(*args, tile_waves=4, **kwo)
    torch.cuda.synchronize()
    assert ", 13, 13, 40>" in s1["kernel"] and torch.equal(s1["sse"], s16["sse"]) and bool(torch.isfinite(s1["sse"]).all())
    # one weight set per trajectory
    ws = [rng.normal(0, 0.1, w.size).astype(np.float32) for _ in range(B)]
    packed = torch.from_numpy(np.stack([capi.mlp_pack(x, L, N) for x in ws])).to(gpu)
    r = capi.dopri5(*args, mlp_packed=packed, mlp_layers=L, mlp_width=N, traj_per_image=1, tile_waves=16, **kwu)
    torch.cuda.synchronize()
    for k in range(B):
        ok = oracle.solve(K.MODEL_NNF, params[k:k + 1], pv, K.NN_Y0, te, weights=ws[k], mlp_layers=L, mlp_width=N, **kwu)
        assert np.array_equal(r["y"][k:k + 1].cpu().numpy(), ok["y"]) and np.array_equal(r["stats"][k:k + 1].cpu().numpy(), ok["stats"])


def test_single_odeint_call_takes_the_one_trajectory_tile(ion, gpu, oracle):
    """The reference's own call shape -- odeint(func, y0, t) with one trajectory (train-s1.py:319-330) -- runs on the one-trajectory tile,
    with and without an exactly uniform output grid (linspace(0, 8000, 80001) in fp32 is not: the general variant)."""
    import ref_style_modules as M
    from torchdiffeq import odeint
    func = M.NNf(K.MODELS["s1"][4])
    M.load_flat_weights(func.net, K.load_weights("s1"))
    func.eval()
    pt, pv, te = K.activation(20)
    func.set_fixed_form_voltage_protocol(pt, pv)
    with torch.no_grad():
        y = odeint(func, torch.tensor([K.NN_Y0]), torch.from_numpy(te).float())
    name = ion.capi.lib().ionode_last_kernel_name().decode()
    assert ", 4, 4, 13, 13, 32>" in name or ", 4, 4, 13, 13, 40>" in name, name
    o = oracle.solve(K.MODEL_NNF, K.MODELS["s1"][4], pv, K.NN_Y0, te, weights=K.load_weights("s1"), mlp_layers=5, mlp_width=200,
                     prot_t0=float(pt[0]), prot_dt=float(pt[1] - pt[0]), state_f32=True)
    assert np.array_equal(y[:, 0, :].double().cpu().numpy(), o["y"][0])
    with torch.no_grad():
        y2 = odeint(func, torch.tensor([K.NN_Y0]), torch.arange(0, 1501, dtype=torch.float32) * 5.0)
    assert ", 4, 4, 13, 13, 40>" in ion.capi.lib().ionode_last_kernel_name().decode() and bool(torch.isfinite(y2).all())


@pytest.mark.parametrize("L", [7, 10, 15])
def test_deep_stacks_take_the_one_trajectory_tile_without_resident_steps(ion, gpu, oracle, L):
    """The one-trajectory tile keeps two steps of every hidden layer's weights in LDS (24 KB per layer): stacks of more than six hidden
    layers do not fit beside them and take the variant that streams every step (TAIL & 64; s02 is 10 x 200) -- same bits as the oracle and
    as the 4-trajectory tile, which a 16-layer stack falls back to."""
    rng = np.random.default_rng(3 + L)
    N = 200
    w = rng.normal(0, 0.1 if L < 12 else 0.07, 2 * N + N + L * (N * N + N) + N + 1).astype(np.float32)
    pt, pv, te = K.atau(30)
    te = te[:401]
    kw = dict(prot_t0=float(pt[0]), prot_dt=float(pt[1] - pt[0]))
    B = 3
    params = np.tile(K.P_HH, (B, 1)) * rng.uniform(0.9, 1.1, (B, 8))
    o = oracle.solve(K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, mlp_layers=L, mlp_width=N, **kw)
    g = _kernel(ion, gpu, K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, **kw)
    assert ", 4, 4, 13, 13, 104>" in g["kernel"], g["kernel"]
    assert np.array_equal(g["y"], o["y"]) and np.array_equal(g["stats"], o["stats"])
    g4 = _kernel(ion, gpu, K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, tile_waves=2, **kw)
    assert ", 4, 4, 13, 13, 24>" in g4["kernel"] and np.array_equal(g4["y"], o["y"])
    slog = torch.zeros((64, 4), dtype=torch.float64, device=gpu)
    gg = _kernel(ion, gpu, K.MODEL_NNF, params, pv, K.NN_Y0, te, weights=w, L=L, N=N, step_log=slog, **kw)      # general variant
    assert ", 4, 4, 13, 13, 96>" in gg["kernel"] and np.array_equal(gg["y"], o["y"])


@pytest.mark.parametrize("f32", [False, True])
def test_one_trajectory_tile_general_variant_nnd_and_checkpoints(ion, gpu, oracle, f32):
    """The GENERAL variant (TAIL slot 32) of the one-trajectory tile for NN-d in both state dtypes, with what takes a launch off the
    lean contract all at once: a step log, accepted-step checkpoints (the gradient path's forward) and the fused current trace."""
    capi = ion.capi
    w = K.load_weights("d2")
    B = 2
    params = np.tile(K.P_NN_D, (B, 1)) * np.random.default_rng(11).uniform(0.9, 1.1, (B, 8))
    pv = np.stack([K.activation(v)[1] for v in (-20, 40)])
    te = K.activation(0)[2][:1201]
    kw = dict(prot_t0=0.0, prot_dt=1.0, prot_of_traj=np.arange(B, dtype=np.int32))
    o = oracle.solve(K.MODEL_NND, params, pv, K.NN_Y0, te, weights=w, mlp_layers=5, mlp_width=200, state_f32=f32, **kw)
    cap = int(o["stats"][:, 0].max()) + 4
    ck = torch.zeros((B, cap, 4 + 16), dtype=torch.float64, device=gpu)
    slog = torch.zeros((3000, 4), dtype=torch.float64, device=gpu)
    g = _kernel(ion, gpu, K.MODEL_NND, params, pv, K.NN_Y0, te, weights=w, L=5, N=200, f32=f32, tile_waves=16, step_log=slog, ckpt=ck, current=True, **kw)
    g16 = _kernel(ion, gpu, K.MODEL_NND, params, pv, K.NN_Y0, te, weights=w, L=5, N=200, f32=f32, tile_waves=4, current=True, **kw)
    assert ", 4, 4, 13, 13, 32>" in g["kernel"], g["kernel"]
    assert np.array_equal(g["y"], o["y"]) and np.array_equal(g["stats"], o["stats"]) and np.array_equal(g["i"], g16["i"])
    ckn = ck.cpu().numpy()
    for b in range(B):
        n = int(o["stats"][b, 0])
        t0, dt = ckn[b, :n, 0], ckn[b, :n, 1]
        assert np.all(dt > 0) and np.allclose(t0[1:], t0[:-1] + dt[:-1], rtol=0, atol=1e-9) and float(ckn[b, n:].sum()) == 0.0
        assert int(ckn[b, :n, 3].sum()) == te.size - 1          # every output sample but the first belongs to exactly one accepted step
