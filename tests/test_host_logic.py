"""CPU tests of the host side: RHS recognition, the drop-in odeint's argument handling, the generic torch stepper,
the C-ABI library (loads, exports, packs) -- no compute calls that need a GPU."""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest
import torch

import kat_cases as K
import ref_style_modules as M


def test_capi_exports_every_declared_symbol(ion):
    """libionode.so loads without a GPU and exports every function include/ionode.h declares."""
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "ionode.h")).read()
    declared = set(re.findall(r"\b(ionode_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(ion.capi.EXPORTS), declared ^ set(ion.capi.EXPORTS)
    lib = ctypes.CDLL(ion.capi.LIB_PATH)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert ion.capi.lib().ionode_abi_version() == ion.capi.ABI_VERSION


def test_descriptor_layout_matches_header(ion):
    """ctypes mirror of ionode_desc: field order/names follow the header (guards ABI drift)."""
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "ionode.h")).read()
    body = hdr[hdr.index("typedef struct ionode_desc {"):hdr.index("} ionode_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:int32_t|int64_t|double)\s+\*?([^;]+);", body):
        names += [n.strip().lstrip("*") for n in decl.split(",")]
    assert names == [f[0] for f in ion.capi.IonodeDesc._fields_]


def test_dispatch_geometry_and_errors(ion):
    capi = ion.capi
    d = capi.make_desc(model=capi.MODEL_NNF, n_state=2, n_out=10, n_traj=4096, n_prot=1, prot_n=100, mlp_layers=5,
                       mlp_width=200, n_params=8, prot_dt=0.1, rtol=1e-7, atol=1e-9)
    g = capi.launch_geometry(d)
    assert g["grid"] == 256 and g["block"] == 256 and g["tile_waves"] == 4 and g["lds_bytes"] < 160 * 1024
    assert "ionode_dopri5_kernel<2, double, 4" in capi.kernel_name(d)
    d.model = capi.MODEL_HH2
    layout = 64 * 14 * 8 + 64 * 64 + 512 + 1024   # 16 per wavefront; interpolant rows, objective sums, cursors, work list
    assert layout <= 12800                          # gfx950 LDS granule 1280 B: 12 wavefronts per compute unit
    # lane-wise kernels: a tile is one wavefront (16 trajectories here), a workgroup carries FOUR tiles and the workgroup count is a
    # multiple of 8 (tile t on XCD t % 8): 4096 trajectories = 256 tiles = 64 workgroups of 256 threads.  Even placement: 64 workgroups
    # on 256 compute units = at most one per CU, so the plan reserves the whole 160 KB of LDS per workgroup
    assert capi.launch_geometry(d) == {"grid": 64, "block": 256, "lds_bytes": 160 * 1024, "tile_waves": 1}
    d.n_traj = 16 * 4 * 256 * 2    # two workgroups per compute unit
    assert capi.launch_geometry(d) == {"grid": 512, "block": 256, "lds_bytes": 160 * 1024 // 2 // 1280 * 1280, "tile_waves": 1}
    d.n_traj, d.tile_waves = 16 * 4 * 256 * 40, 16   # far beyond the residency (3 per CU at 4 x 12 800 bytes): the layout itself
    assert capi.launch_geometry(d) == {"grid": 256 * 40, "block": 256, "lds_bytes": 4 * layout, "tile_waves": 1}
    d.n_traj, d.tile_waves = 4096, 64
    assert capi.launch_geometry(d)["grid"] == 16     # 64 tiles of 64 trajectories -> 16 workgroups
    d.tile_waves = 0
    d.n_state = 6  # inconsistent with HH2
    with pytest.raises(capi.IonodeError):
        capi.launch_geometry(d)
    d = capi.make_desc(model=capi.MODEL_NNF, n_state=2, n_out=10, n_traj=1, n_prot=1, prot_n=100, mlp_layers=5,
                       mlp_width=640, n_params=8, prot_dt=0.1, rtol=1e-7, atol=1e-9)
    with pytest.raises(capi.IonodeError, match="width"):      # beyond 512 (N = 64 is served since round 5: the run-time-width tile)
        capi.launch_geometry(d)


@pytest.mark.parametrize("L,N", [(5, 200), (1, 100), (5, 10), (1, 500)])
def test_weight_pack_is_a_permutation_with_zero_padding(ion, L, N):
    """ionode_mlp_pack (host -> host): every weight/bias appears exactly once, the rest of the image is zero."""
    rng = np.random.default_rng(L * 1000 + N)
    n = 2 * N + N + L * (N * N + N) + N + 1
    w = rng.permutation(n).astype(np.float32) + 1.0  # distinct, non-zero, exactly representable
    img = ion.capi.mlp_pack(w, L, N)
    if N <= 16:
        # N <= 16: the image ends with the SCALAR section of the per-lane net (MlpLane), in row PAIRS (the two halves of a
        # v_pk_fma_f32): layer 0 {b0, b0'} {w00, w00'} {w01, w01'} {0, 0} per pair, then per hidden layer and pair the weights
        # {W[2m][k], W[2m+1][k]} in the canonical k order (r-major, q-minor over k = 4 q + r), the bias pair, padding
        NPAIR, PB = (N + 1) // 2, (2 * (N + 1) + 3) & ~3
        n_sc = NPAIR * 8 + L * NPAIR * PB
        s0 = img[-n_sc:][:NPAIR * 8].reshape(NPAIR, 4, 2)
        sc = img[-n_sc:][NPAIR * 8:].reshape(L, NPAIR, PB // 2, 2)
        img = img[:-n_sc]
        korder = [4 * q + r for r in range(4) for q in range(4) if 4 * q + r < N]
        W0 = w[:2 * N].reshape(N, 2); b0 = w[2 * N:3 * N]
        rows = np.arange(N).reshape(NPAIR, 2)   # (N even here)
        assert np.array_equal(s0[:, 0], b0[rows]) and np.array_equal(s0[:, 1], W0[rows, 0]) and np.array_equal(s0[:, 2], W0[rows, 1]) and not s0[:, 3].any()
        off = 2 * N + N
        for l in range(L):
            W = w[off:off + N * N].reshape(N, N); b = w[off + N * N:off + N * N + N]; off += N * N + N
            for m in range(NPAIR):
                for e in range(2):
                    assert np.array_equal(sc[l, m, :N, e], W[2 * m + e, korder]) and sc[l, m, N, e] == b[2 * m + e]
            assert not sc[l, :, N + 1:].any()
    if N == 200:
        # N = 200 (round 5): the image ends with the section of the ONE-trajectory tile (MlpRow1): per layer three full-row wavefronts x 13
        # steps x 4 r x 64 lanes x float4 over q, the remainder wavefront's 4 steps x 4 r x 64 lanes (every hidden weight exactly once more;
        # -0.0 where a partial chain has no fourth k-tile), + 4 x 64 accumulator starts
        lay1 = (3 * 13 * 4 + 4 * 4) * 256 + 256
        r1 = img[-L * lay1:].reshape(L, lay1)
        img = img[:-L * lay1]
        off = 2 * N + N
        for l in range(L):
            full = r1[l, :3 * 13 * 4 * 256].reshape(3, 13, 4, 64, 4)
            rem = r1[l, 3 * 13 * 4 * 256:(3 * 13 * 4 + 16) * 256].reshape(4, 4, 64, 4)
            Wl = w[off:off + N * N].reshape(N, N); bl = w[off + N * N:off + N * N + N]; off += N * N + N
            nzf = np.concatenate([full[full != 0], rem[rem != 0]])
            assert nzf.size == N * N and np.array_equal(np.sort(nzf), np.sort(Wl.reshape(-1)))
            for (wv, st, r, lane, q) in ((0, 0, 0, 0, 0), (2, 5, 2, 47, 3), (1, 12, 3, 17, 1), (0, 7, 1, 63, 2)):
                row, k = 64 * wv + lane, 16 * ((st + (lane >> 4)) % 13) + 4 * q + r
                assert full[wv, st, r, lane, q] == (Wl[row, k] if (row < N and k < N) else 0.0)
            for (j, r, lane, q) in ((0, 0, 0, 0), (2, 3, 21, 1), (3, 1, 5, 0), (3, 2, 40, 1), (1, 0, 63, 3)):
                row, kt = 192 + (lane & 15), (lane >> 4) + 4 * j
                k = 16 * kt + 4 * q + r
                assert rem[j, r, lane, q] == (Wl[row, k] if (kt < 13 and row < N and k < N) else 0.0)
            bias = r1[l, (3 * 13 * 4 + 16) * 256:].reshape(4, 64)
            assert np.array_equal(np.sort(bias[bias != 0]), np.sort(bl))    # every row's bias exactly once (remainder rows: chain 0 only)
            assert int((np.signbit(rem) & (rem == 0)).sum()) == 3 * 16 * 4 * 4   # step 3 of the chains 1..3: lanes x r x q
        # ... in front of it the section of the 4-trajectory tile (MlpTile4), round-5 lane layout: per layer three full-row wavefronts x 13 steps
        # x 4 q x 64 lanes x float4 over r, the remainder wavefront's 4 steps x 4 q x 64 lanes (every hidden weight exactly once more; -0.0 where
        # a partial chain has no fourth k-tile) + 4 x 64 accumulator-start float4s
        lay = (3 * 13 * 4 + 16) * 256 + 4 * 256
        t4 = img[-L * lay:].reshape(L, lay)
        img = img[:-L * lay]
        off = 2 * N + N
        for l in range(L):
            full = t4[l, :3 * 13 * 4 * 256].reshape(3, 13, 4, 64, 4)
            rem = t4[l, 3 * 13 * 4 * 256:(3 * 13 * 4 + 16) * 256].reshape(4, 4, 64, 4)
            Wl = w[off:off + N * N].reshape(N, N); bl = w[off + N * N:off + N * N + N]; off += N * N + N
            nzf = np.concatenate([full[full != 0], rem[rem != 0]])
            assert nzf.size == N * N and np.array_equal(np.sort(nzf), np.sort(Wl.reshape(-1)))
            for (wv, st, q, lane, r) in ((0, 0, 0, 0, 0), (2, 5, 2, 47, 3), (1, 12, 3, 17, 1), (0, 7, 1, 63, 2)):
                i, bb = lane & 3, lane >> 2
                row, k = 16 * (4 * wv + (bb >> 2)) + 4 * (bb & 3) + i, 16 * ((st + (bb >> 2)) % 13) + 4 * q + r
                assert full[wv, st, q, lane, r] == (Wl[row, k] if (row < N and k < N) else 0.0)
            for (jj, q, lane, r) in ((0, 0, 0, 0), (2, 3, 21, 1), (3, 1, 5, 0), (3, 2, 40, 1), (1, 0, 63, 3)):
                i, bb = lane & 3, lane >> 2
                row, kt = 192 + 4 * (bb & 3) + i, (bb >> 2) + 4 * jj
                k = 16 * kt + 4 * q + r
                assert rem[jj, q, lane, r] == (Wl[row, k] if (kt < 13 and row < N and k < N) else 0.0)
            bias = t4[l, (3 * 13 * 4 + 16) * 256:]
            assert np.array_equal(np.sort(bias[bias != 0]), np.sort(np.repeat(bl, 4)))   # every lane of a block (4 trajectories) carries its rows' biases
            assert int((np.signbit(rem) & (rem == 0)).sum()) == 3 * 16 * 4 * 4          # step 3 of the chains 1..3: lanes x q x r
    nz = img[img != 0]
    assert nz.size == n and np.array_equal(np.sort(nz), np.sort(w))
    with pytest.raises(ion.capi.IonodeError):
        ion.capi.mlp_pack(w[:-1], L, N)


def _hh_module():
    m = M.HodgkinHuxley()
    pt, pv, te = K.activation(20)
    m.set_fixed_form_voltage_protocol(pt, pv)
    return m, pt, pv, te


def test_recognise_reference_modules(ion):
    m, pt, pv, te = _hh_module()
    y0 = torch.tensor([[0.0, 1.0]])
    s = ion.recognise(m, y0)
    assert s.model == ion.capi.MODEL_HH2 and s.prot_t is None and s.prot_dt == 1.0 and np.array_equal(s.params, K.P_HH)
    m6 = M.Markov6()
    m6.set_fixed_form_voltage_protocol(*K.ap2hz()[:2])
    s6 = ion.recognise(m6, torch.tensor([[0.0, 1.0, 0, 0, 0, 0]]))
    assert s6.model == ion.capi.MODEL_MARKOV6 and s6.n_state == 6 and s6.prot_t is None and abs(s6.prot_dt - 0.1) < 1e-12
    for cls, model in ((M.NNf, ion.capi.MODEL_NNF), (M.NNd, ion.capi.MODEL_NND)):
        f = cls()
        M.load_flat_weights(f.net, K.load_weights("s1"))
        f.set_fixed_form_voltage_protocol(pt, pv)
        s = ion.recognise(f, y0)
        assert s.model == model and (s.mlp_layers, s.mlp_width) == (5, 200)
        assert np.array_equal(s.weights, K.load_weights("s1"))  # bit-exact hand-over of the state dict
    # the protocol is mutable state of func: a second set_fixed_form_voltage_protocol is seen at the next call
    m.set_fixed_form_voltage_protocol(*K.deactivation(-60)[:2])
    assert ion.recognise(m, y0).prot_v.size == 10001


def test_recognition_rejects_lookalikes(ion):
    y0 = torch.tensor([[0.0, 1.0]])
    m, pt, pv, te = _hh_module()

    class Twisted(M.HodgkinHuxley):  # same attributes, different maths
        def forward(self, t, y):
            return 2.0 * super().forward(t, y)

    tw = Twisted()
    tw.set_fixed_form_voltage_protocol(pt, pv)
    with pytest.raises(ion.UnrecognisedRhs):
        ion.recognise(tw, y0)
    f = M.NNf()
    f.net[1] = torch.nn.ReLU()
    f.set_fixed_form_voltage_protocol(pt, pv)
    with pytest.raises(ion.UnrecognisedRhs):
        ion.recognise(f, y0)
    with pytest.raises(ion.UnrecognisedRhs):
        ion.recognise(M.HodgkinHuxley(), y0)  # no protocol set
    with pytest.raises(ion.UnrecognisedRhs):
        ion.odeint(lambda t, y: -y, y0, torch.linspace(0, 1, 5))


def test_odeint_argument_handling(ion):
    m, pt, pv, te = _hh_module()
    y0 = torch.tensor([[0.0, 1.0]])
    t = torch.linspace(0.0, 100.0, 11)
    with pytest.raises(NotImplementedError):
        ion.odeint(m, y0, t, method="adams")
    with pytest.raises(TypeError):
        ion.odeint(m, y0.numpy(), t)
    if not torch.cuda.is_available():
        # recognised module, no HIP device: fails loudly, never falls back to the CPU
        with pytest.raises(ion.IonodeError, match="no CPU fallback"):
            ion.odeint(m, y0, t)
        with pytest.raises(ion.IonodeError):
            ion.odeint(m, y0, t, method="dopri5", options={"grid_points": None, "eps": 1e-6})


def test_generic_stepper_matches_oracle(ion, oracle):
    """The opt-in generic path (func.forward from Python) runs the same algorithm: HH in fp64 state agrees with the
    oracle to ~1e-9 (torch.exp / ** vs the deterministic exp / fifth root: last-ulp differences only)."""

    class Plain(torch.nn.Module):  # not a reference family: forces the generic path
        def __init__(self):
            super().__init__()
            self.inner = _hh_module()[0]

        def forward(self, t, y):
            return self.inner(t, y)

    f = Plain()
    te = torch.linspace(0.0, 3000.0, 301, dtype=torch.float64)
    y0 = torch.tensor([[0.0, 1.0]], dtype=torch.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = ion.odeint(f, y0, te, options={"allow_generic": True})
    assert got.shape == (301, 1, 2) and got.dtype == torch.float64
    pt, pv, _ = K.activation(20)
    o = oracle.solve(K.MODEL_HH2, K.P_HH, pv, [0.0, 1.0], te.numpy(), prot_t=pt)
    err = np.linalg.norm(got[:, 0, :].numpy() - o["y"][0]) / np.linalg.norm(o["y"][0])
    assert err < 1e-7, err


def test_shard_bounds(ion):
    import importlib
    dist = importlib.import_module("neural-ode-ion-channels_amd.distributed")
    for n in (0, 1, 7, 4096, 65537):
        for w in (1, 2, 3, 8):
            cuts = [dist.shard_bounds(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    # equal-cost shards: contiguous, covering, and no shard heavier than the ideal share plus its own heaviest item
    rng = np.random.default_rng(0)
    for n, w in ((1, 2), (5, 2), (1000, 8), (4097, 3)):
        cost = rng.uniform(1.0, 3.0, n)
        cuts = dist.shard_bounds_by_cost(cost, w)
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
        for lo, hi in cuts:
            assert cost[lo:hi].sum() <= cost.sum() / w + cost.max() + 1e-9
    assert dist.shard_bounds_by_cost([1.0, 1.0, 1.0, 5.0, 1.0], 2) == [(0, 4), (4, 5)]
    assert dist.shard_bounds_by_cost([0.0, 0.0, 0.0], 2) == [(0, 2), (2, 3)]           # no information: equal counts


def test_lpt_order_is_a_stable_descending_permutation(ion):
    cost = np.array([5, 9, 5, 1, 9, 7], dtype=np.int64)
    o = ion.schedule.lpt_order(cost).numpy()
    assert sorted(o.tolist()) == list(range(6))
    assert o.tolist() == [1, 4, 5, 0, 2, 3]            # ties keep the caller's order
    assert ion.schedule.lpt_order(torch.tensor([0.5, 2.5])).tolist() == [1, 0]


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package (Python or C/HIP sources) or the torchdiffeq
    shim may import, include, link or execute it; bench.py may only in its cpu_baseline legs."""
    root = os.path.join(os.path.dirname(__file__), "..")
    offenders = []
    for base in ("neural-ode-ion-channels_amd", "torchdiffeq", "include"):
        for dp, _, files in os.walk(os.path.join(root, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|oracle/|liboracle|dopri5_oracle", txt):
                        offenders.append(os.path.join(dp, f))
    assert not offenders, offenders
    bench = open(os.path.join(root, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle import", bench)]
    assert len(uses) == 1 and bench[:uses[0]].rfind("def cpu_baseline") > bench[:uses[0]].rfind("def main")


def test_tiny_net_auto_dispatch_respects_the_image_granularity(ion):
    """a population of N <= 16 nets padded to 16 / 32 / 48 trajectories per candidate must not be sent to the
    64-per-wavefront kernel just because the batch is large (it would be rejected with IONODE_ERR_ARG at launch)."""
    capi = ion.capi
    kw = dict(model=capi.MODEL_NNF, n_state=2, n_out=10, n_prot=1, prot_n=100, mlp_layers=5, mlp_width=10, n_params=8,
              prot_dt=0.1, rtol=1e-7, atol=1e-9)
    big = 16 * 6000  # above IONODE_TINY64_FROM
    assert "1, 64, 1, 1" in capi.kernel_name(capi.make_desc(n_traj=big, **kw))                                   # one image: 64 per wavefront
    assert "1, 64, 1, 1" in capi.kernel_name(capi.make_desc(n_traj=big, traj_per_image=128, mlp_image_stride=10**6, **kw))
    for s in (16, 32, 48):
        d = capi.make_desc(n_traj=big, traj_per_image=s, mlp_image_stride=10**6, **kw)
        assert "1, 1, 1, 1" in capi.kernel_name(d) and capi.launch_geometry(d)["grid"] == big // 16              # 16 per wavefront


def test_default_checkpoint_loads_into_a_reference_style_load_ckp(ion, tmp_path):
    """save_checkpoint() without an optimizer state must still satisfy train-r1.py:68-72's
    `optimizer.load_state_dict(checkpoint['optimizer'])`."""
    import importlib
    pre = importlib.import_module("neural-ode-ion-channels_amd.preprocess")
    L, N = 1, 10
    class Func(torch.nn.Module):  # the reference keeps the MLP in ODEFunc.net (train-r1.py:148-166): keys net.0.weight ...
        def __init__(self):
            super().__init__()
            self.net = torch.nn.Sequential(torch.nn.Linear(2, N), torch.nn.LeakyReLU(), torch.nn.Linear(N, N), torch.nn.LeakyReLU(),
                                           torch.nn.Linear(N, 1))
    net = Func()
    flat, L2, N2 = pre.state_dict_to_flat(net.state_dict())
    assert (L2, N2) == (L, N)
    path = str(tmp_path / "ck.pt")
    pre.save_checkpoint(path, 7, flat, L2, N2, loss=[0.5, 0.25], lr=2e-3)

    def load_ckp(checkpoint_fpath, model, optimizer):  # the reference's function, restated (train-r1.py:68-72)
        checkpoint = torch.load(checkpoint_fpath, weights_only=True)
        model.load_state_dict(checkpoint["state_dict"])
        optimizer.load_state_dict(checkpoint["optimizer"])
        return model, optimizer, checkpoint["epoch"]

    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    _, opt, epoch = load_ckp(path, net, opt)
    assert epoch == 7 and opt.param_groups[0]["lr"] == 2e-3
    loss = net.net(torch.zeros(3, 2)).sum()
    loss.backward()
    opt.step()  # a usable optimizer


def test_population_objective_rejects_a_cost_vector_of_the_wrong_length(ion):
    with pytest.raises(ion.capi.IonodeError, match="cost has 3 entries for 5 candidates"):
        __import__("importlib").import_module("neural-ode-ion-channels_amd.objective").population_sum_of_squares(np.ones((5, 4)), np.zeros((2, 100)), np.zeros((2, 10)), np.arange(10.0),
                                                base_params=K.P_HH, cost=[1.0, 2.0, 3.0])


def test_stable_step_cap_follows_the_fastest_gate(ion):
    """grad.stable_step_cap: 3 / lambda_max at the protocol's extreme voltages (the r gate at -120 mV is the stiff one)."""
    capi, grad = ion.capi, ion.grad
    pv = torch.tensor([[-120.0, -80.0, 40.0]], dtype=torch.float64)
    p = torch.from_numpy(np.stack([K.P_HH, 2.0 * K.P_HH]))
    k = lambda pp, v: (pp[4] * np.exp(pp[5] * v) + pp[6] * np.exp(-pp[7] * v), pp[0] * np.exp(pp[1] * v) + pp[2] * np.exp(-pp[3] * v))
    lam_r = max(k(2.0 * K.P_HH, v)[0] for v in (-120.0, 40.0, -80.0))
    lam_a = max(k(2.0 * K.P_HH, v)[1] for v in (-120.0, 40.0, -80.0))
    assert np.isclose(grad.stable_step_cap(capi.MODEL_NNF, p, pv), 3.0 / lam_r, rtol=1e-12)
    assert np.isclose(grad.stable_step_cap(capi.MODEL_HH2, p, pv), 3.0 / max(lam_r, lam_a), rtol=1e-12)
    cap6 = grad.stable_step_cap(capi.MODEL_MARKOV6, torch.from_numpy(K.P_M6[None, :]), pv)
    assert 0.0 < cap6 < 3.0 / 0.1


def test_protocol_major_launch_order_is_dealt_out_over_the_xcds(ion):
    """capi._protocol_major (launch_order="auto" of the one-trajectory-per-lane kernels): a permutation; every 64-trajectory wavefront
    holds one protocol; and the wavefronts that share an XCD (workgroup index modulo 8) hold a contiguous eighth of the protocols,
    so that an XCD's L2 sees 8 of 64 protocols, not all of them.  Already-sorted input: no order at all."""
    B, P = 64 * 8 * 24, 64
    pot = (torch.arange(B, dtype=torch.int32) % P).contiguous()
    order = ion.capi._protocol_major(pot).long()
    assert sorted(order.tolist()) == list(range(B))
    prot = pot[order].view(-1, 64)
    assert bool((prot == prot[:, :1]).all())                      # one protocol per wavefront
    per_xcd = [set(prot[i::8, 0].tolist()) for i in range(8)]
    assert all(len(s) == P // 8 for s in per_xcd) and len(set().union(*per_xcd)) == P
    assert all(max(per_xcd[i]) < min(per_xcd[i + 1]) for i in range(7))
    assert ion.capi._protocol_major(torch.sort(pot).values.contiguous()) is None
    ragged = (torch.arange(1000, dtype=torch.int32) % 7).contiguous()   # not a multiple of 512: plain protocol-major
    o2 = ion.capi._protocol_major(ragged).long()
    assert sorted(o2.tolist()) == list(range(1000)) and bool((ragged[o2][1:] >= ragged[o2][:-1]).all())


def test_explicit_64_per_wavefront_rejects_images_that_do_not_fill_a_tile(ion):
    """the tile size of a several-weight-sets launch comes from the kernel VARIANT (a lane-wise workgroup is 4 x 64 lanes,
    so its block size says nothing): tile_waves = 64 with 16 / 32 / 48 trajectories per image must be an argument error, never a
    launch in which 48 of 64 trajectories integrate with the wrong net."""
    capi = ion.capi
    kw = dict(model=capi.MODEL_NNF, n_state=2, n_out=10, n_prot=1, prot_n=100, mlp_layers=5, mlp_width=10, n_params=8,
              prot_dt=0.1, rtol=1e-7, atol=1e-9, n_traj=64 * 600, mlp_image_stride=10**6)
    for s in (16, 32, 48):
        with pytest.raises(capi.IonodeError, match="traj_per_image must be a multiple of the tile size"):
            capi.launch_geometry(capi.make_desc(tile_waves=64, traj_per_image=s, **kw))
        assert capi.kernel_name(capi.make_desc(tile_waves=64, traj_per_image=s, **kw)) == ""
    assert capi.launch_geometry(capi.make_desc(tile_waves=64, traj_per_image=64, **kw))["block"] == 256
    kw200 = dict(kw, mlp_width=200, n_traj=64)
    # 32-trajectory tiles need images of whole 32s: a request for them with 16 per image is served by the 16-trajectory tile
    assert capi.launch_geometry(capi.make_desc(tile_waves=8, traj_per_image=16, **kw200))["grid"] == 4
    assert capi.launch_geometry(capi.make_desc(tile_waves=8, traj_per_image=32, **kw200))["grid"] == 2
    assert capi.launch_geometry(capi.make_desc(tile_waves=2, traj_per_image=4, **kw200))["grid"] == 16   # 4-trajectory tiles
    assert capi.launch_geometry(capi.make_desc(tile_waves=16, traj_per_image=1, **kw200))["grid"] == 64  # one-trajectory tiles: any image granularity
    assert ", 4, 4, 13, 13, 32>" in capi.kernel_name(capi.make_desc(**dict(kw200, mlp_image_stride=0)))  # 64 trajectories: chosen by itself
    deep = dict(kw200, mlp_image_stride=0, mlp_layers=10)                                                  # s02: no room for LDS-resident steps
    assert ", 4, 4, 13, 13, 96>" in capi.kernel_name(capi.make_desc(**deep)) and capi.launch_geometry(capi.make_desc(**deep))["lds_bytes"] < 20 * 1024
    assert capi.launch_geometry(capi.make_desc(**dict(kw200, mlp_image_stride=0)))["lds_bytes"] > 100 * 1024


def test_lane_wise_crossovers_come_from_the_library(ion):
    """capi.py reads the dispatcher's 16 -> 64 trajectories-per-wavefront crossovers through the C ABI."""
    capi = ion.capi
    L = capi.lib()
    kw = dict(n_state=2, n_out=10, n_prot=1, prot_n=100, n_params=8, prot_dt=0.1, rtol=1e-7, atol=1e-9)
    for model, extra in ((capi.MODEL_HH2, {}), (capi.MODEL_MARKOV6, dict(n_state=6, n_params=12)), (capi.MODEL_NNF, dict(mlp_layers=5, mlp_width=10))):
        n = L.ionode_lane_wise_from(model, extra.get("mlp_width", 0))
        assert n > 0
        k = dict(kw, **extra)
        below = capi.launch_geometry(capi.make_desc(model=model, n_traj=n - 1, **k))
        at = capi.launch_geometry(capi.make_desc(model=model, n_traj=n, **k))
        per_wave = lambda g, nt: nt / (g["grid"] * (4 if g["block"] == 256 else 1))
        assert per_wave(below, n - 1) <= 16 and per_wave(at, n) > 16
    assert L.ionode_lane_wise_from(capi.MODEL_NNF, 200) == 0


def test_reduce_slab_count_makes_one_round_of_workgroups(ion):
    """Round 5: the slab count of ionode_grad_reduce() comes from the library (host code kept a formula of its own that went stale when the
    kernel's jobs per slab changed).  Without a device the plan is that of 256 compute units: N = 200 runs three workgroups per unit and two
    column blocks per layer (768 // 11), N = 500 four column blocks at one per unit (256 // 21), N = 100 one block (256 // 6); never fewer
    than four records per slab, never fewer than one slab."""
    L = ion.capi.lib()
    assert L.ionode_grad_reduce_slabs(5, 200, 8276) == 69
    assert L.ionode_grad_reduce_slabs(10, 200, 100000) == 768 // 21
    assert L.ionode_grad_reduce_slabs(5, 500, 8276) == 12
    assert L.ionode_grad_reduce_slabs(5, 100, 8276) == 42
    assert L.ionode_grad_reduce_slabs(5, 200, 40) == 10 and L.ionode_grad_reduce_slabs(5, 200, 3) == 1 and L.ionode_grad_reduce_slabs(0, 200, 8) == 1


def test_every_width_up_to_512_has_a_kernel_and_an_image(ion):
    """Round 5: widths without a tuned tile go to the run-time-width tile (NT slot 0) instead of
    IONODE_ERR_UNSUPPORTED; the packed image of such a width is the generic layout, and beyond N = 512 the call still refuses."""
    capi = ion.capi
    kw = dict(model=capi.MODEL_NNF, n_state=2, n_out=10, n_traj=40, n_prot=1, prot_n=100, mlp_layers=3, n_params=8, prot_dt=0.1, rtol=1e-7, atol=1e-9)
    tuned = {10: "1, 1, 1, 1", 16: "1, 1, 1, 1", 100: "4, 4, 7, 7", 200: "4, 4, 13, 13", 500: "4, 8, 32, 4"}
    for N in (1, 10, 16, 17, 32, 50, 64, 96, 100, 112, 113, 150, 200, 208, 209, 300, 496, 500, 512):
        name = capi.kernel_name(capi.make_desc(mlp_width=N, **kw))
        NT = (N + 15) // 16
        if NT in (1, 7, 13, 32):
            assert name, N
            if N in tuned:
                assert tuned[N] in name
        else:
            assert ", 4, 1, 0, 1, " in name, (N, name)
            g = capi.launch_geometry(capi.make_desc(mlp_width=N, **kw))
            assert g["grid"] == 3 and g["block"] == 256 and g["lds_bytes"] <= 160 * 1024
            n_w = 2 * N + N + 3 * (N * N + N) + N + 1
            img = capi.mlp_pack(np.arange(1, n_w + 1, dtype=np.float32), 3, N)
            NP = 16 * NT
            assert img.size == 4 * NP + 3 * (NT * NT * 256 + NP) + NP + 4
            # fragment (rt, kt) lane 16 q + m, component r  ==  W[16 rt + m][16 kt + 4 q + r] of hidden layer 0
            W1 = np.arange(1, n_w + 1, dtype=np.float32)[3 * N:3 * N + N * N].reshape(N, N)
            frag = img[4 * NP:4 * NP + NT * NT * 256].reshape(NT, NT, 64, 4)
            for (rt, kt, lane, r) in ((0, 0, 0, 0), (NT - 1, 1, 37, 2), (1, NT - 1, 63, 3)):
                row, k = 16 * rt + (lane & 15), 16 * kt + 4 * (lane >> 4) + r
                assert frag[rt, kt, lane, r] == (W1[row, k] if row < N and k < N else 0.0)
    assert capi.kernel_name(capi.make_desc(mlp_width=513, **kw)) == ""
    with pytest.raises(capi.IonodeError):
        capi.mlp_pack(np.zeros(2 * 600 + 600 + 600 + 1, dtype=np.float32), 0, 600)
    assert capi.kernel_name(capi.make_desc(mlp_width=64, tile_waves=1, **kw)) == ""   # the generic tile is a four-wavefront tile
