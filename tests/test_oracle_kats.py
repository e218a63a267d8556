"""Pin the CPU oracle against the reference's own recorded results.

torchdiffeq (the integrator the reference calls) is absent from the reference tree and from this
container, so these artefacts are what pins parity at the odeint boundary:
  * 92 '--pred' losses recorded in {s1,s2,d1,d2}/log2, each = mean|i_model - i_truth| over two
    dopri5 solves in fp32 state (NN RHS + ground-truth RHS) -> tests/golden/kat_losses.json
  * the 80 001-sample HH current trace cached in figure-0-s/i_n.pt (seeded noise removed)

Tolerances: SURVEY.md 8d's contract, |loss - logged| <= 2e-5 for every KAT and a median <= 5e-6.
The reference's solves run with rtol = 1e-7 ~ fp32 epsilon, so its accept/reject sequence is
rounding-noise driven; two of the 92 logged numbers are themselves > 1.7e-5 from the converged
solution of their ODE -- kat_cases.kat_within_tolerance states the rule and names them.
"""
import numpy as np
import pytest

import kat_cases as K

ABS_TOL = K.KAT_ABS_TOL
MEDIAN_TOL = K.KAT_MEDIAN_TOL


def test_tableau_identities(oracle):
    se, ss, sm, mb = oracle.selfcheck()
    assert abs(se) < 1e-16 and abs(ss - 1) < 1e-15 and abs(sm - 0.5) < 1e-15 and mb < 1e-15


def _kat_loss(oracle, m, case, state_f32=True, rtol=1e-7, atol=1e-9):
    tm, tp, ty0, nm, npar = K.MODELS[m]
    pt, pv, te = case
    w = K.load_weights(m)
    tr = oracle.solve(tm, tp, pv, ty0, te, prot_t=pt, state_f32=state_f32, rtol=rtol, atol=atol)
    nn = oracle.solve(nm, npar, pv, K.NN_Y0, te, prot_t=pt, weights=w, mlp_layers=K.MLP_L,
                      mlp_width=K.MLP_N, state_f32=state_f32, rtol=rtol, atol=atol)
    assert tr["status"][0] == 0 and nn["status"][0] == 0
    v, inr = oracle.protocol_v(pv, te, prot_t=pt)
    assert inr.all()
    i_t = oracle.current(tr["y"][0], v, open_state_only=(tm == K.MODEL_MARKOV6), state_f32=state_f32)
    i_m = oracle.current(nn["y"][0], v, state_f32=state_f32)
    return float(np.mean(np.abs(i_m - i_t)))


@pytest.mark.parametrize("m", ["s1", "s2", "d1", "d2"])
def test_logged_prediction_losses(oracle, m):
    kats = K.load_kats()
    diffs = []
    for sec, key, case in K.all_cases():
        got = _kat_loss(oracle, m, case)
        exp = K.expected(kats, m, sec, key)
        diffs.append(abs(got - exp))
        assert K.kat_within_tolerance(got, exp, lambda: _kat_loss(oracle, m, case, False, 1e-10, 1e-12)), \
            f"{m} {sec} {key}: {got:.6f} vs logged {exp:.6f}"
    assert len(diffs) == 23
    assert np.median(diffs) <= MEDIAN_TOL
    assert sum(d > ABS_TOL for d in diffs) <= (2 if m == "s1" else 0)  # the two named cases, no others


def test_fp64_state_agrees_with_logged_losses(oracle):
    """fp64 state (BASELINE configs 2-3) is the tighter solve of the same problem: same KATs, same tolerance."""
    kats = K.load_kats()
    for sec, key, case in K.all_cases()[:8]:
        got = _kat_loss(oracle, "s1", case, state_f32=False)
        assert abs(got - K.expected(kats, "s1", sec, key)) <= ABS_TOL


def test_outliers_are_the_references_own_rounding_noise(oracle):
    """The two s1 values that miss 2e-5: the converged solution (fp64 state, rtol 1e-10) is itself 1.7e-5 / 2.6e-5 from
    the logged number, fp32 and fp64 state at the reference's tolerance scatter around it by ~1e-5, and a one-notch
    change of rtol moves the value by more than the residual -- accept/reject noise, not a modelling difference."""
    kats = K.load_kats()
    for key, case in (("-70.0", K.deactivation(-70)), ("-50.0", K.deactivation(-50))):
        logged = K.expected(kats, "s1", "deact", key)
        conv = _kat_loss(oracle, "s1", case, False, 1e-10, 1e-12)
        conv2 = _kat_loss(oracle, "s1", case, False, 1e-11, 1e-13)
        assert abs(conv - conv2) <= 1e-6                   # converged: tightening further changes nothing
        assert 1e-5 < abs(logged - conv) < 3e-5            # the logged value is off the converged answer
        f32 = _kat_loss(oracle, "s1", case, True)
        f64 = _kat_loss(oracle, "s1", case, False)
        assert abs(f32 - conv) <= 2e-5 and abs(f64 - conv) <= 2e-5
        nudged = _kat_loss(oracle, "s1", case, True, rtol=1.5e-7)
        assert abs(nudged - f32) > 2e-6                    # the value moves at the 1e-5 level with the step sequence


def _fig0s_protocol():
    # figure-0-s.py:45-56 (pr3 with a +40 mV step, 0.1 ms grid), t1 = linspace(0, 8000, 80001) fp32 (:36)
    pt = np.linspace(0.0, 8000.0, 80001)
    v = np.zeros(80001)
    v[:10000] = -80
    v[10000:60000] = 40
    v[60000:70000] = -40
    v[70000:75000] = -120
    v[75000:] = -80
    return pt, v, K.f32_linspace(0, 8000, 80001)


def test_figure0s_golden_trace(oracle):
    gold = np.fromfile(K.GOLDEN + "/fig0s_hh_current.f64", dtype="<f8")
    pt, v, te = _fig0s_protocol()
    r = oracle.solve(K.MODEL_HH2, K.P_HH, v, [0.0, 1.0], te, prot_t=pt, state_f32=True)
    vv, _ = oracle.protocol_v(v, te, prot_t=pt)
    i = oracle.current(r["y"][0], vv, state_f32=True)[::10]
    rel = np.linalg.norm(i - gold) / np.linalg.norm(gold)
    assert rel <= 5e-6, rel  # <= 5e-6 rel-L2; the reference itself sits 2.6e-6 from truth
    # away from the four voltage steps the agreement is ~1e-6 relative per segment
    mask = np.ones(gold.size, bool)
    for s in (1000, 6000, 7000, 7500):
        mask[s - 2:s + 6] = False
    assert np.abs(i - gold)[mask].max() <= 2e-4


def test_figure0s_reference_r_state_trace(oracle):
    """STATE-level golden (round 5): the reference's own r(t) = odeint(m, y0, t1)[:, 0, 1] of figure-0-s (figure-0-s.py:147-153),
    recovered exactly from the cached a_n / i_n / v tensors (tests/golden/make_fixtures.py: r = i_n / (a_n (v + 86)), on the fp32
    grid to 2e-16).  Unlike the current trace this is not a product with a: it pins one state component of torchdiffeq's output."""
    gold = np.fromfile(K.GOLDEN + "/fig0s_hh_r.f32", dtype="<f4").astype(np.float64)
    pt, v, te = _fig0s_protocol()
    r = oracle.solve(K.MODEL_HH2, K.P_HH, v, [0.0, 1.0], te, prot_t=pt, state_f32=True)
    mine = r["y"][0][::10, 1].astype(np.float64)
    assert gold.size == mine.size == 8001 and gold[0] == 1.0
    rel = np.linalg.norm(mine - gold) / np.linalg.norm(gold)
    assert rel <= 1e-6, rel          # BASELINE.json north_star's tolerance (1e-6 relative L2 of torchdiffeq); measured 8.1e-7
    assert np.abs(mine - gold).max() <= 2e-5    # worst sample: behind the -40 -> -120 mV step (1.1e-5: the step sequences differ there)
    mask = np.ones(gold.size, bool)
    for s in (1000, 6000, 7000, 7500):
        mask[s - 2:s + 40] = False
    assert np.abs(mine - gold)[mask].max() <= 4e-6   # elsewhere a few fp32 ulps of r <= 1 (measured 3.2e-6)


def test_uniform_grid_lookup_equals_explicit_times(oracle):
    """The arithmetic index rule used by the HIP kernels reproduces interp1d's searchsorted rule."""
    rng = np.random.default_rng(0)
    v = rng.uniform(-120, 60, 5001)
    pt = np.arange(5001) * 1.0
    t = np.concatenate([rng.uniform(-5, 5005, 20000), pt, [-1e-9, 5000 + 1e-9, 0.0, 5000.0]])
    a, ia = oracle.protocol_v(v, t, prot_t=pt)
    b, ib = oracle.protocol_v(v, t, prot_t0=0.0, prot_dt=1.0)
    assert (ia == ib).all() and np.array_equal(a, b)
    assert a[~ia].tolist() == [-80.0] * int((~ia).sum())


def test_failure_status_codes(oracle):
    """max-steps and non-finite states end a trajectory with a status code and NaN-filled tail."""
    pt, v, te = K.activation(20)
    r = oracle.solve(K.MODEL_HH2, K.P_HH, v, [0.0, 1.0], te, prot_t=pt, max_total_steps=50)
    assert r["status"][0] == 3 and np.isnan(r["y"][0, -1]).all() and np.isfinite(r["y"][0, 0]).all()
    # max_steps is torchdiffeq's max_num_steps: per output interval (1 ms grid here).  Some interval (a voltage step)
    # needs more than 3 attempts; no interval needs 200; the whole solve needs far more than 200
    r3 = oracle.solve(K.MODEL_HH2, K.P_HH, v, [0.0, 1.0], te, prot_t=pt, max_steps=3)
    assert r3["status"][0] == 3 and r3["stats"][0, 0] + r3["stats"][0, 1] > 3
    r200 = oracle.solve(K.MODEL_HH2, K.P_HH, v, [0.0, 1.0], te, prot_t=pt, max_steps=200)
    assert r200["status"][0] == 0 and r200["stats"][0, 0] + r200["stats"][0, 1] > 200
    # a non-finite start makes dt NaN: torchdiffeq's first assertion ('underflow in dt') fires before
    # the 'non-finite values in state' one, so status is 1 here; both are failures with a NaN tail
    r = oracle.solve(K.MODEL_HH2, K.P_HH, v, [np.inf, 1.0], te, prot_t=pt)
    assert r["status"][0] in (1, 2) and np.isnan(r["y"][0, 1:]).all()


def test_batch_is_trajectory_independent(oracle):
    """B trajectories in one call == B single calls (per-trajectory params and protocols)."""
    rng = np.random.default_rng(1)
    cases = [K.activation(v)[1] for v in (-40, 0, 40)]
    pv = np.stack(cases)
    te = K.activation(0)[2][:2001]
    params = K.P_HH[None, :] * rng.uniform(0.5, 2.0, (6, 8))
    pot = np.array([0, 1, 2, 2, 1, 0], dtype=np.int32)
    rb = oracle.solve(K.MODEL_HH2, params, pv, [0.0, 1.0], te, prot_t0=0.0, prot_dt=1.0, prot_of_traj=pot, nthreads=3)
    for b in range(6):
        r1 = oracle.solve(K.MODEL_HH2, params[b], pv[pot[b]], [0.0, 1.0], te, prot_t0=0.0, prot_dt=1.0)
        assert np.array_equal(r1["y"][0], rb["y"][b]) and np.array_equal(r1["stats"][0], rb["stats"][b])


def test_deterministic_exp_and_fifth_root_accuracy(oracle):
    """det_exp / det_root5 (shared operation sequences of oracle and kernels): within 1 ulp of
    libm's exp and within 2 ulp of the exact fifth root over the ranges the path uses (and well beyond)."""
    import ctypes
    import math
    from decimal import Decimal, getcontext

    lib = oracle.lib()
    lib.det_exp.restype = ctypes.c_double
    lib.det_exp.argtypes = [ctypes.c_double]
    lib.det_root5.restype = ctypes.c_double
    lib.det_root5.argtypes = [ctypes.c_double]
    rng = np.random.default_rng(0)
    for x in np.concatenate([rng.uniform(-20, 20, 20000), rng.uniform(-700, 700, 2000), [0.0, -0.0, 1e-300]]):
        a, b = lib.det_exp(float(x)), math.exp(float(x))
        assert abs(a - b) <= np.spacing(b), x
    assert lib.det_exp(710.0) == math.inf and lib.det_exp(-746.0) == 0.0 and math.isnan(lib.det_exp(math.nan))
    getcontext().prec = 40
    for x in 10.0 ** rng.uniform(-14, 14, 1500):
        a = lib.det_root5(float(x))
        exact = Decimal(float(x)) ** (Decimal(1) / Decimal(5))
        assert abs(Decimal(a) - exact) <= 2 * Decimal(float(np.spacing(a))), x
