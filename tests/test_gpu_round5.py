"""-m gpu, round 5: the run-time-width MLP tile (any N <= 512), every result against the oracle bit for bit."""
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
