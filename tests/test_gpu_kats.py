"""-m gpu: every reproducible known-answer value of the reference (92 logged '--pred' losses)
through the HIP path in the reference's fp32 state, batched by protocol family; plus the figure-0-s golden trace."""
import numpy as np
import pytest

import kat_cases as K
from gpu_util import run_gpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", ["s1", "s2", "d1", "d2"])
def test_all_logged_losses_on_gpu(ion, gpu, oracle, m):
    kats = K.load_kats()
    tm, tp, ty0, nm, npar = K.MODELS[m]
    w = K.load_weights(m)
    cases = K.all_cases()
    groups = {}
    for c in cases:  # one launch per protocol family (same grid): AP 2 Hz, activation x7, deactivation x9, tau x6
        groups.setdefault(c[0], []).append(c)
    diffs = []
    for sec, cs in groups.items():
        pt, _, te = cs[0][2]
        pv = np.stack([c[2][1] for c in cs])
        n = len(cs)
        pot = np.arange(n, dtype=np.int32)
        kw = dict(prot_t=pt, prot_of_traj=pot, f32=True, current=True)
        gt = run_gpu(ion, gpu, tm, np.tile(tp, (n, 1)), pv, ty0, te, obs_open_state_only=(tm == K.MODEL_MARKOV6), **kw)
        gn = run_gpu(ion, gpu, nm, np.tile(npar, (n, 1)), pv, K.NN_Y0, te, weights=w, L=K.MLP_L, N=K.MLP_N, **kw)
        assert (gt["status"] == 0).all() and (gn["status"] == 0).all()
        for k, (_, key, _) in enumerate(cs):
            loss = float(np.mean(np.abs(gn["i"][k] - gt["i"][k])))  # fused current traces: i = gate * (V + 86)
            exp = K.expected(kats, m, sec, key)
            diffs.append(abs(loss - exp))
            case = cs[k][2]

            def converged(case=case):  # rtol 1e-10 fp64-state value of the same KAT (kat_cases.kat_within_tolerance)
                kwc = dict(prot_t=case[0], state_f32=False, rtol=1e-10, atol=1e-12)
                a = oracle.solve(tm, tp, case[1], ty0, case[2], **kwc)
                b = oracle.solve(nm, npar, case[1], K.NN_Y0, case[2], weights=w, mlp_layers=K.MLP_L, mlp_width=K.MLP_N, **kwc)
                vv, _ = oracle.protocol_v(case[1], case[2], prot_t=case[0])
                return float(np.mean(np.abs(oracle.current(b["y"][0], vv)
                                            - oracle.current(a["y"][0], vv, open_state_only=(tm == K.MODEL_MARKOV6)))))
            assert K.kat_within_tolerance(loss, exp, converged), (m, sec, key, loss, exp)
        # and the batch is the oracle's fp32-state solve bit for bit (first and last protocol of the family)
        for k in (0, n - 1):
            o = oracle.solve(nm, npar, pv[k], K.NN_Y0, te, prot_t=pt, weights=w, mlp_layers=K.MLP_L, mlp_width=K.MLP_N,
                             state_f32=True)
            assert np.array_equal(gn["y"][k], o["y"][0])
    assert len(diffs) == 23 and np.median(diffs) <= 5e-6


def test_figure0s_golden_trace_on_gpu(ion, gpu):
    gold = np.fromfile(K.GOLDEN + "/fig0s_hh_current.f64", dtype="<f8")
    v = np.zeros(80001)
    v[:10000] = -80; v[10000:60000] = 40; v[60000:70000] = -40; v[70000:75000] = -120; v[75000:] = -80
    te = K.f32_linspace(0, 8000, 80001)
    g = run_gpu(ion, gpu, K.MODEL_HH2, K.P_HH, v, [0.0, 1.0], te, f32=True, prot_t0=0.0, prot_dt=0.1, current=True)
    i = g["i"][0][::10]
    assert np.linalg.norm(i - gold) / np.linalg.norm(gold) <= 5e-6
    # state-level golden: the reference's own r(t) (tests/golden/make_fixtures.py fig0s_hh_r.f32; figure-0-s.py:147-153,196-200)
    gr = np.fromfile(K.GOLDEN + "/fig0s_hh_r.f32", dtype="<f4").astype(np.float64)
    r = g["y"][0][::10, 1].astype(np.float64)
    assert np.linalg.norm(r - gr) / np.linalg.norm(gr) <= 1e-6 and np.abs(r - gr).max() <= 2e-5   # north_star's 1e-6 relative L2
