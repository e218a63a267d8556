"""Preprocessing + checkpoint compatibility, CPU.

End-to-end pin: train-s1.py's data path -- seeded noise on the HH current of 7 activation + 9 deactivation sweeps
(:556-580), r(t) of the NN-f RHS (:616-626), Hanning smoothing + cubic splines per voltage segment (:668-690),
a = i / (g r (V - E)) and da/dt (:733-746), masks / skip / stride (:52-63, :783-803) -- restated with the package's
preprocess.py and the CPU oracle for the solves, reproduces the reference's cached s1/{v,a,dadt}.pt rows: V bit for bit,
a to 1e-6 relative, da/dt to 2e-3 relative (a spline derivative of a noisy trace at 0.1 ms spacing amplifies the 1e-6
fp32-solver noise by ~1e3; with a different noise stream `a` itself would already differ at the 1e-2 level)."""
import importlib
import os

import numpy as np
import pytest
import torch

import kat_cases as K

pp = importlib.import_module("neural-ode-ion-channels_amd.preprocess")


def _sweeps():
    pt1, pt2 = np.linspace(0., 8000., 80001), np.linspace(0., 10000., 100001)
    t1, t2 = K.f32_linspace(0, 8000, 80001), K.f32_linspace(0, 10000, 100001)
    p1 = []
    for v_i in (-60, -40, -20, 0, 20, 40, 60):           # train-s1.py:69-80
        v = np.zeros(80001); v[:10000] = -80; v[10000:60000] = v_i; v[60000:70000] = -40; v[70000:75000] = -120; v[75000:] = -80
        p1.append(v)
    p2 = []
    for v_i in (-120, -110, -100, -90, -80, -70, -60, -50, -40):  # train-s1.py:84-95
        v = np.zeros(100001); v[:10000] = -80; v[10000:30000] = 50; v[30000:90000] = v_i; v[90000:95000] = -120; v[95000:] = -80
        p2.append(v)
    m1 = pp.step_mask(80001, [10000, 60000, 70000, 75000])
    m2 = pp.step_mask(100001, [10000, 30000, 90000, 95000])
    return [(pt1, t1, v, m1) for v in p1] + [(pt2, t2, v, m2) for v in p2]


def test_reference_regression_samples_are_reproduced(oracle):
    np.random.seed(0)                                     # train-s1.py:37; the 16 noise draws follow in sweep order
    V, A, D, masks = [], [], [], []
    for pt, te, pv, mask in _sweeps():
        y = oracle.solve(K.MODEL_HH2, K.P_HH, pv, [0.0, 1.0], te, prot_t=pt, state_f32=True)["y"][0]
        vt, _ = oracle.protocol_v(pv, te, prot_t=pt)
        i = oracle.current(y, vt, state_f32=True) + np.random.normal(0, 0.1, te.shape)   # noise_sigma = 0.1 (:40)
        r = y[:, 1]                                        # the NN-f RHS's r equation is the HH one with the same p5..p8
        k3, k4 = K.P_HH[4] * np.exp(K.P_HH[5] * vt), K.P_HH[6] * np.exp(-K.P_HH[7] * vt)
        i_fit, didt = pp.fit_current(te, i, pt, pv)
        a, dadt = pp.state_space_samples(i_fit, didt, r, -k3 * r + k4 * (1. - r), vt)
        V.append(vt); A.append(a); D.append(dadt); masks.append(mask)
    v, a, d = pp.training_rows(V, A, D, masks)
    fix = np.load(os.path.join(K.GOLDEN, "preproc_s1_rows.npz"))
    s = int(fix["stride"])
    assert v.size == int(fix["n_rows"]) == 132410
    assert np.array_equal(v[::s], fix["v"])
    rel = lambda x, y: float(np.linalg.norm(x - y) / np.linalg.norm(y))
    print("a rel", rel(a[::s], fix["a"]), "dadt rel", rel(d[::s], fix["dadt"]))
    assert rel(a[::s], fix["a"]) <= 5e-6 and rel(d[::s], fix["dadt"]) <= 5e-3


def test_smooth_properties():
    x = np.random.default_rng(0).normal(size=500)
    for w in ("flat", "hanning", "hamming", "bartlett", "blackman"):
        y = pp.smooth(x, 61, w)
        assert y.size == x.size + 60
        assert np.allclose(pp.smooth(np.full(300, 2.5), 61, w), 2.5)            # normalised window
    assert pp.smooth(x, 1) is not None and np.array_equal(pp.smooth(x, 1), x)   # window_len < 3: returned unchanged
    with pytest.raises(ValueError):
        pp.smooth(x[:10], 61)
    with pytest.raises(ValueError):
        pp.smooth(x.reshape(2, -1))
    # a linear ramp is a fixed point of symmetric smoothing away from the reflected ends
    ramp = np.arange(400, dtype=float)
    assert np.allclose(pp.smooth(ramp, 61)[30:-30][61:-61], ramp[61:-61])


def test_checkpoint_round_trip_and_reference_files(tmp_path):
    w = K.load_weights("s1")
    sd = pp.flat_to_state_dict(w, 5, 200)
    assert list(sd)[:4] == ["net.0.weight", "net.0.bias", "net.2.weight", "net.2.bias"] and sd["net.12.weight"].shape == (1, 200)
    flat, L, N = pp.state_dict_to_flat(sd)
    assert (L, N) == (5, 200) and np.array_equal(flat, w)
    path = str(tmp_path / "checkpoint-2.pt")
    pp.save_checkpoint(path, 401, w, 5, 200, optimizer_state={"state": {}, "param_groups": []}, loss=[0.1058, 0.0271])
    ck = pp.load_checkpoint(path)
    assert ck["epoch"] == 401 and ck["loss"] == [0.1058, 0.0271] and np.array_equal(ck["flat"], w)
    # what the reference's load_ckp does with it (train-r1.py:68-72)
    import ref_style_modules as M
    func = M.NNf(K.P_HH)
    func.load_state_dict(torch.load(path, weights_only=True)["state_dict"])
    assert np.array_equal(func.net[0].weight.detach().numpy().ravel(), w[:400])
    ref = "/root/reference"
    if os.path.exists(os.path.join(ref, "s1", "model-state-dict.pt")):  # build container only: the reference's own files
        assert np.array_equal(pp.load_checkpoint(os.path.join(ref, "s1", "model-state-dict.pt"))["flat"], w)
        best = os.path.join(ref, "r1", "best-model-checkpoint.pt")
        if os.path.exists(best):
            ck = pp.load_checkpoint(best)
            assert ck["epoch"] == 401 and (ck["mlp_layers"], ck["mlp_width"]) == (5, 200) and ck["flat"].size == 201801
            assert abs(float(ck["loss"][0]) - 0.1058) < 1e-3 and "param_groups" in ck["optimizer"]
