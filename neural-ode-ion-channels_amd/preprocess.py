"""Derivative-estimation preprocessing and checkpoint compatibility -- CPU-side host code around the hot
path: numpy / SciPy, as in the reference.  It turns measured current traces into the (V, a) -> da/dt rows the MLP
regression step (regression.py) trains on, and reads / writes the reference's checkpoint files.

Reference: smoothing.py:73-129 (`smooth`), train-s1.py:603-808 (estimating a and da/dt from i(t)), train-r1.py:422-679 (the
real-data route: tri- / bi-exponential fits of a(t) per voltage step with analytic derivatives), train-r1.py:61-72
(`save_ckp` / `load_ckp`), train-s1.py:891 ff. (the cached s1/{v,a,dadt}.pt these functions reproduce).
"""
import numpy as np

_WINDOWS = {"flat": lambda n: np.ones(n, "d"), "hanning": np.hanning, "hamming": np.hamming, "bartlett": np.bartlett,
            "blackman": np.blackman}


def smooth(x, window_len=11, window="hanning"):
    """Window smoothing by convolution (smoothing.py:73-129): the signal is extended by window_len - 1 reflected samples
    at both ends and convolved ('valid') with the normalised window, so the result has len(x) + window_len - 1 samples
    (callers cut window_len // 2 from each end, e.g. `smooth(i, 61)[30:-30]`, train-s1.py:682)."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("smooth only accepts 1 dimension arrays.")
    if x.size < window_len:
        raise ValueError("Input vector needs to be bigger than window size.")
    if window_len < 3:
        return x
    if window not in _WINDOWS:
        raise ValueError("Window is on of 'flat', 'hanning', 'hamming', 'bartlett', 'blackman'")
    ext = np.concatenate([x[window_len - 1:0:-1], x, x[-2:-window_len - 1:-1]])
    w = _WINDOWS[window](window_len)
    return np.convolve(w / w.sum(), ext, mode="valid")


def segment_ends(prot_t, prot_v):
    """End times of the constant-voltage segments of a step protocol (+ one past the end), train-s1.py:672-673."""
    prot_t, prot_v = np.asarray(prot_t), np.asarray(prot_v)
    ends = prot_t[np.append([False], prot_v[:-1] != prot_v[1:])]
    return np.append(ends, prot_t[-1] + 1)


def fit_current(t, i, prot_t, prot_v, window_len=61):
    """Smoothed current and its time derivative, one cubic interpolating spline per voltage segment (train-s1.py:668-690):
    Hanning smoothing of the segment, UnivariateSpline(k=3) with smoothing factor 0, evaluated at the data times."""
    from scipy.interpolate import UnivariateSpline
    t, i = np.asarray(t, dtype=np.float64), np.asarray(i, dtype=np.float64).reshape(-1)
    i_fit, didt = [], []
    t_lo = 0
    half = window_len // 2
    for t_hi in segment_ends(prot_t, prot_v):
        idx = np.where((t >= t_lo) & (t < t_hi))[0]
        spl = UnivariateSpline(t[idx], smooth(i[idx], window_len)[half:-half], k=3)
        spl.set_smoothing_factor(0)
        i_fit = np.append(i_fit, spl(t[idx]))
        didt = np.append(didt, spl.derivative()(t[idx]))
        t_lo = t_hi
    return i_fit, didt


# ---- real-data route (train-r1.py:422-679): a(t) of a voltage step is a sum of decaying exponentials; fit it, differentiate analytically ----

TRI_EXP_X0 = (1.0, 1.0 / 100.0, 0.5, 1.0 / 200.0, 0.25, 1.0 / 400.0, 0.1)      # train-r1.py:426 (activation protocol)
TRI_EXP_X0_SLOW = (0.7, 1.0 / 50.0, 0.2, 1.0 / 100.0, 0.1, 1.0 / 200.0, 0.01)   # train-r1.py:425 (sine-wave / AP protocols)
BI_EXP_X0 = (0.7, 1.0 / 50.0, 0.2, 1.0 / 100.0, 0.01)                           # train-r1.py:440


def multi_exp(t, x, order=0):
    """sum_j A_j exp(-b_j t) + offset and its first / second time derivative: x = (A_1, b_1, ..., A_n, b_n, offset) with n = 3
    (tri_exp / dtri_exp / d2tri_exp, train-r1.py:427-438) or n = 2 (bi_exp / dbi_exp / d2bi_exp, train-r1.py:441-451)."""
    t = np.asarray(t, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    if x.size not in (5, 7):
        raise ValueError("multi_exp: 5 (bi-exponential) or 7 (tri-exponential) parameters")
    amp, rate = x[0:-1:2], x[1:-1:2]
    out = np.zeros(t.shape)
    for a, b in zip(amp, rate):   # in the reference's order a*exp(-b t) + c*exp(-d t) + e*exp(-f t) [+ g]
        out = out + ((-b) ** order * a) * np.exp(-b * t)
    return out + x[-1] if order == 0 else out


def fit_multi_exp(t, a, x0, restarts=0, seed=0):
    """Nelder-Mead minimisation of the RMSE of multi_exp(t, x) against a (scipy.optimize.fmin, as train-r1.py:487-489).  The
    reference hands its hardest segments (the -90 mV steps) to PINTS' CMA-ES (train-r1.py:553-555); PINTS is an absent third-party
    dependency, so `restarts` > 0 re-runs the simplex from log-normal perturbations of x0 (seeded) and keeps the best."""
    from scipy import optimize
    t, a = np.asarray(t, dtype=np.float64), np.asarray(a, dtype=np.float64)
    f = lambda x: np.sqrt(np.mean((multi_exp(t, x) - a) ** 2))
    best = optimize.fmin(f, np.asarray(x0, dtype=np.float64), disp=False)
    rng = np.random.default_rng(seed)
    for _ in range(int(restarts)):
        x = optimize.fmin(f, np.asarray(x0) * np.exp(rng.normal(0.0, 0.5, len(x0))), disp=False)
        if f(x) < f(best):
            best = x
    return best


def fit_activation(prot_t, a, change_mask, cap_mask, std_cutoff=0.01, x0=TRI_EXP_X0, bi_exp_at=(), spline_at=(), restart_segments=(),
                   restarts=4, spline_window=51, spline_k=4, spline_s=0.2):
    """a(t), da/dt, d2a/dt2 on the whole time grid of one real-data protocol, segment by segment (train-r1.py:453-679).

    prot_t [n] sample times; a [n] the activation estimate i / (g r (V - E)); change_mask [n] False exactly AT the voltage change
    points (the reference's `change_pt*`: t_split = prot_t[~change_mask]); cap_mask [n] True where the sample is usable (capacitive
    spikes removed, `cap_mask*`).  Per constant-voltage segment:
      * std(a) > std_cutoff: the gate is moving -- tri-exponential fit of the raw samples (bi-exponential where the segment contains
        one of `bi_exp_at`, train-r1.py:631-636), evaluated with its analytic derivatives on the segment's full grid;
        segments whose index is in `restart_segments` take the restarted simplex (the reference's CMA-ES cases);
      * otherwise, or where the segment contains one of `spline_at` (the sine-wave window at 3500 ms: smooth(a, 21), k = 5,
        train-r1.py:565-573): Hanning smoothing + UnivariateSpline(k = spline_k, s = spline_s), derivatives from the spline.
    Returns (a_fit, dadt, d2adt2, kinds): arrays [n] (zero outside every fitted segment, as the reference leaves them) and the
    list of (t_first, t_last, 'tri-exp' | 'bi-exp' | 'spline') per segment."""
    from scipy.interpolate import UnivariateSpline
    prot_t, a = np.asarray(prot_t, dtype=np.float64), np.asarray(a, dtype=np.float64).reshape(-1)
    change_mask, cap_mask = np.asarray(change_mask, dtype=bool), np.asarray(cap_mask, dtype=bool)
    within = lambda r, x: (np.min(r) < x) and (np.max(r) > x)
    t_split = np.append(prot_t[~change_mask], prot_t[-1] + 1)
    tt, aa = prot_t[cap_mask], a[cap_mask]
    ao, d1, d2 = np.zeros(prot_t.shape), np.zeros(prot_t.shape), np.zeros(prot_t.shape)
    kinds = []
    t_lo = 0
    for seg, t_hi in enumerate(t_split):
        idx = np.where((tt >= t_lo) & (tt < t_hi))[0]
        t_lo = t_hi
        if idx.size == 0:
            continue
        tfit = tt[idx]
        full = np.where((prot_t >= tfit[0]) & (prot_t <= tfit[-1]))[0]
        sine = any(within(tfit, x) for x in spline_at)
        if np.std(aa[idx]) > std_cutoff and not sine:
            bi = any(within(tfit, x) for x in bi_exp_at)
            t0 = tfit - tfit[0]
            x = fit_multi_exp(t0, aa[idx], BI_EXP_X0 if bi else x0, restarts=restarts if seg in restart_segments else 0)
            tf = prot_t[full] - tfit[0]
            ao[full], d1[full], d2[full] = multi_exp(tf, x), multi_exp(tf, x, 1), multi_exp(tf, x, 2)
            kinds.append((tfit[0], tfit[-1], "bi-exp" if bi else "tri-exp"))
        else:
            w, k = (21, 5) if sine else (spline_window, spline_k)
            spl = UnivariateSpline(tfit, smooth(aa[idx], w)[w // 2:-(w // 2)], k=k)
            spl.set_smoothing_factor(spline_s)
            ao[full], d1[full], d2[full] = spl(prot_t[full]), spl(prot_t[full], 1), spl(prot_t[full], 2)
            kinds.append((tfit[0], tfit[-1], "spline"))
    return ao, d1, d2, kinds


def state_space_samples(i_fit, didt, r, drdt, v, g=1.0, e=-86.0, dvdt=0.0):
    """a = i / (g r (V - E)) and da/dt = (1/r) ((di/dt / g - a r dV/dt) / (V - E) - a dr/dt)   (train-s1.py:733-746)."""
    a = i_fit / (g * r * (v - e))
    dadt = r ** (-1) * ((didt / g - a * r * dvdt) / (v - e) - a * drdt)
    return a, dadt


def step_mask(n, step_indices, before=5, after=50):
    """Drop `before` samples ahead of and `after` samples behind every voltage step (train-s1.py:52-63)."""
    m = np.ones(n, dtype=bool)
    for s in step_indices:
        m[s - before:s + after] = False
    return m


def training_rows(v_list, a_list, dadt_list, masks, skip=5, sparse=11):
    """Mask, drop the first `skip` samples, keep every `sparse`-th, concatenate over protocols (train-s1.py:783-803)."""
    pick = lambda x, m: np.asarray(x)[m][skip::sparse]
    cat = lambda xs: np.concatenate([pick(x, m) for x, m in zip(xs, masks)])
    return cat(v_list), cat(a_list), cat(dadt_list)


# ---- checkpoints (train-r1.py:61-72, :947-964): {epoch, state_dict, optimizer, loss}; state_dict keys net.{0,2,..}.weight/bias ----

def flat_to_state_dict(flat, mlp_layers, mlp_width, prefix="net."):
    """Flat fp32 state dict (the C ABI's order) -> {'net.0.weight': tensor, 'net.0.bias': ..., 'net.2.weight': ...}."""
    import torch
    flat = np.asarray(flat, dtype=np.float32).reshape(-1)
    N, L = mlp_width, mlp_layers
    sd, off = {}, 0
    for n, (o, i) in enumerate([(N, 2)] + [(N, N)] * L + [(1, N)]):
        sd[f"{prefix}{2 * n}.weight"] = torch.from_numpy(flat[off:off + o * i].reshape(o, i).copy()); off += o * i
        sd[f"{prefix}{2 * n}.bias"] = torch.from_numpy(flat[off:off + o].copy()); off += o
    if off != flat.size:
        raise ValueError("flat state dict does not match (L, N)")
    return sd


def state_dict_to_flat(state_dict, prefix="net."):
    """Reference state dict -> (flat fp32, L, N); extra keys (p1..p8 buffers etc.) are ignored."""
    keys = sorted((int(k[len(prefix):].split(".")[0]) for k in state_dict if k.startswith(prefix) and k.endswith(".weight")))
    parts = []
    for n in keys:
        parts += [state_dict[f"{prefix}{n}.weight"].detach().cpu().numpy().astype(np.float32).reshape(-1),
                  state_dict[f"{prefix}{n}.bias"].detach().cpu().numpy().astype(np.float32).reshape(-1)]
    first = state_dict[f"{prefix}{keys[0]}.weight"]
    return np.concatenate(parts), len(keys) - 2, int(first.shape[0])


def save_checkpoint(path, epoch, flat, mlp_layers, mlp_width, optimizer_state=None, loss=None, lr=1e-3):
    """Write a checkpoint the reference's `load_ckp` reads: torch.save({'epoch', 'state_dict', 'optimizer', 'loss'}).
    `load_ckp` calls optimizer.load_state_dict(checkpoint['optimizer']) unconditionally (train-r1.py:68-72), so without an
    optimizer_state (e.g. adam_state_from_regression(reg)) the checkpoint carries the state dict of a FRESH Adam over the net's
    2 (L + 2) tensors (no moments yet, learning rate `lr`) -- loadable by a real torch.optim.Adam, never an empty dict."""
    import torch
    sd = flat_to_state_dict(flat, mlp_layers, mlp_width)
    if optimizer_state is None:
        optimizer_state = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1)) for _ in sd], lr=float(lr)).state_dict()
    torch.save({"epoch": int(epoch), "state_dict": sd, "optimizer": optimizer_state, "loss": loss}, path)


def load_checkpoint(path):
    """Read a reference checkpoint (or a bare state dict such as s1/model-state-dict.pt): dict(flat, L, N, epoch, loss, optimizer)."""
    import torch
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    flat, L, N = state_dict_to_flat(sd)
    meta = ck if isinstance(ck, dict) and "state_dict" in ck else {}
    return {"flat": flat, "mlp_layers": L, "mlp_width": N, "epoch": meta.get("epoch"), "loss": meta.get("loss"),
            "optimizer": meta.get("optimizer")}


def adam_state_to_flat(optimizer_state):
    """torch.optim.Adam.state_dict() (as stored in a reference checkpoint) -> (exp_avg flat, exp_avg_sq flat, step, lr):
    parameters are in nn.Sequential order, i.e. the flat state-dict order."""
    st = optimizer_state["state"]
    idx = sorted(st)
    m = np.concatenate([st[i]["exp_avg"].detach().cpu().numpy().astype(np.float32).reshape(-1) for i in idx])
    v = np.concatenate([st[i]["exp_avg_sq"].detach().cpu().numpy().astype(np.float32).reshape(-1) for i in idx])
    step = int(float(st[idx[0]]["step"]))
    return m, v, step, float(optimizer_state["param_groups"][0]["lr"])


def adam_state_from_regression(reg):
    """torch.optim.Adam.state_dict() equivalent of an MlpRegression trainer (per-tensor exp_avg / exp_avg_sq / step)."""
    import torch
    N, L = reg.N, reg.L
    m, v = reg.m.detach().cpu(), reg.v.detach().cpu()
    state, off, idx = {}, 0, 0
    for (o, i) in [(N, 2)] + [(N, N)] * L + [(1, N)]:
        for shape in ((o, i), (o,)):
            n = int(np.prod(shape))
            state[idx] = {"step": torch.tensor(float(reg.t)), "exp_avg": m[off:off + n].reshape(shape).clone(),
                          "exp_avg_sq": v[off:off + n].reshape(shape).clone()}
            off += n; idx += 1
    group = {"lr": reg.lr(), "betas": tuple(reg.betas), "eps": reg.eps, "weight_decay": 0, "amsgrad": False,
             "params": list(range(idx))}
    return {"state": state, "param_groups": [group]}
