"""Derivative-estimation preprocessing and checkpoint compatibility -- CPU-side host code around the hot
path: numpy / SciPy, as in the reference.  It turns measured current traces into the (V, a) -> da/dt rows the MLP
regression step (regression.py) trains on, and reads / writes the reference's checkpoint files.

Reference: smoothing.py:73-129 (`smooth`), train-s1.py:603-808 (estimating a and da/dt from i(t)), train-r1.py:61-72
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
