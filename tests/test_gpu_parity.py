"""-m gpu: libionode (HIP, through the C ABI) against the CPU oracle on identical inputs.

Tolerance (DESIGN.md "Parity"): ZERO.  Both sides execute the same IEEE operation sequence (no contraction; the
MLP is the same fmaf chain as the fp32 MFMA; exp and the fifth root are the same deterministic operation sequences),
so the accept/reject sequences (step counters) and every output bit must be identical, in fp64 state and in the
reference-compatible fp32 state.  The north star's floating-point tolerance is 1e-6 relative L2 on current traces;
bit equality is the stronger statement and is what is asserted.
"""
import numpy as np
import pytest

import kat_cases as K
from gpu_util import rel_l2, run_gpu

pytestmark = pytest.mark.gpu

TOL_F64 = 0.0
TOL_F32 = 0.0


def _check(g, o, tol):
    assert (g["status"] == o["status"]).all(), (g["status"], o["status"])
    assert np.array_equal(g["stats"], o["stats"]), (g["stats"][:4], o["stats"][:4])
    ok = o["status"] == 0
    assert rel_l2(g["y"][ok], o["y"][ok]) <= tol


@pytest.mark.parametrize("f32", [False, True])
def test_hh2_step_protocols(ion, gpu, oracle, f32):
    """HH ground-truth model on the seven Pr3 activation protocols, per-trajectory parameters."""
    rng = np.random.default_rng(0)
    pv = np.stack([K.activation(v)[1] for v in (-60, -40, -20, 0, 20, 40, 60)])
    te = K.activation(0)[2]
    B = 70
    params = K.P_HH[None, :] * rng.uniform(0.7, 1.4, (B, 8))
    pot = (np.arange(B) % 7).astype(np.int32)
    kw = dict(prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot)
    g = run_gpu(ion, gpu, K.MODEL_HH2, params, pv, [0.0, 1.0], te, f32=f32, **kw)
    o = oracle.solve(K.MODEL_HH2, params, pv, [0.0, 1.0], te, state_f32=f32, **kw)
    _check(g, o, TOL_F32 if f32 else TOL_F64)


@pytest.mark.parametrize("f32", [False, True])
def test_markov6_ap2hz(ion, gpu, oracle, f32):
    """6-state ground truth on the AP 2 Hz protocol with its explicit (non-uniform-in-bits) time grid."""
    pt, pv, te = K.ap2hz()
    rng = np.random.default_rng(1)
    params = K.P_M6[None, :] * rng.uniform(0.8, 1.25, (5, 12))
    y0 = [0.0, 1.0, 0.0, 0.0, 0.0, 0.0]
    g = run_gpu(ion, gpu, K.MODEL_MARKOV6, params, pv, y0, te, f32=f32, prot_t=pt)
    o = oracle.solve(K.MODEL_MARKOV6, params, pv, y0, te, state_f32=f32, prot_t=pt)
    _check(g, o, TOL_F32 if f32 else TOL_F64)


@pytest.mark.parametrize("name,model", [("s1", K.MODEL_NNF), ("d2", K.MODEL_NND)])
@pytest.mark.parametrize("f32", [False, True])
def test_nn_shipped_weights(ion, gpu, oracle, name, model, f32):
    """NN-f (s1) and NN-d (d2) with the reference's trained s00 weights; 19 trajectories = 2 tiles, one ragged."""
    w = K.load_weights(name)
    pv = np.stack([K.deactivation(v)[1] for v in (-120, -90, -60, -40)])
    te = K.deactivation(0)[2][:4001]
    B = 19
    rng = np.random.default_rng(2)
    params = K.MODELS[name][4][None, :] * rng.uniform(0.9, 1.1, (B, 8))
    pot = (np.arange(B) % 4).astype(np.int32)
    kw = dict(prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot)
    g = run_gpu(ion, gpu, model, params, pv, K.NN_Y0, te, weights=w, L=K.MLP_L, N=K.MLP_N, f32=f32, **kw)
    o = oracle.solve(model, params, pv, K.NN_Y0, te, weights=w, mlp_layers=K.MLP_L, mlp_width=K.MLP_N,
                     state_f32=f32, **kw)
    _check(g, o, TOL_F32 if f32 else TOL_F64)
