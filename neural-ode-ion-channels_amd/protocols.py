"""Synthetic voltage-clamp protocols and the reference's step protocols.

The reference's sinewave / staircase CSVs are absent from its tree (.MISSING_LARGE_BLOBS), so the
benchmark protocols are synthesised: same sampling (0.1 ms), same duration, same kind of wave-form.
Per-trajectory diversity: amplitudes and frequencies scaled by U(0.8, 1.2) drawn from
numpy.random.default_rng(1234 + i), so trajectory i's protocol depends on i only (rank-invariant).

`xp` is numpy (tests, CPU baseline) or torch (bench: generated directly in HBM).
"""
import numpy as np

SINE_SEED = 1234


def sinewave_scales(first, count):
    """[count, 6] U(0.8, 1.2) scale factors (3 amplitudes, 3 frequencies) of trajectories first..first+count-1."""
    return np.stack([np.random.default_rng(SINE_SEED + i).uniform(0.8, 1.2, 6) for i in range(first, first + count)])


def _segments(t, xp):
    """Step skeleton of the sine-wave protocol (Beattie et al. 2018 shape, train-r1.py:108 for the sine window)."""
    v = xp.full_like(t, -80.0)
    v = xp.where((t >= 250.0) & (t < 300.0), xp.full_like(t, -120.0), v)
    v = xp.where((t >= 500.0) & (t < 1500.0), xp.full_like(t, 40.0), v)
    v = xp.where((t >= 1500.0) & (t < 2000.0), xp.full_like(t, -120.0), v)
    v = xp.where((t >= 6500.0) & (t < 7000.0), xp.full_like(t, -120.0), v)
    return v


def sinewave(scales, n_samples=100001, dt=0.1, xp=np, device=None, chunk=128):
    """[B, n_samples] fp64 mV.  n_samples = 80001 is the 8 s protocol, 100001 the '100k' variant (10 s, longer
    final -80 mV hold).  scales: [B, 6] from sinewave_scales()."""
    scales = np.asarray(scales, dtype=np.float64)
    B = scales.shape[0]
    if xp is np:
        t = np.arange(n_samples, dtype=np.float64) * dt
        base = _segments(t, np)
        tau = t - 2500.0
        win = (t >= 3000.0) & (t < 6500.0)
        out = np.empty((B, n_samples), dtype=np.float64)
        for b in range(B):
            a1, a2, a3, f1, f2, f3 = scales[b]
            s = -30.0 + 54.0 * a1 * np.sin(0.007 * f1 * tau) + 26.0 * a2 * np.sin(0.037 * f2 * tau) \
                + 10.0 * a3 * np.sin(0.19 * f3 * tau)
            out[b] = np.where(win, s, base)
        return out
    import torch
    t = torch.arange(n_samples, dtype=torch.float64, device=device) * dt
    base = _segments(t, torch)
    tau = t - 2500.0
    win = (t >= 3000.0) & (t < 6500.0)
    out = torch.empty((B, n_samples), dtype=torch.float64, device=device)
    sc = torch.from_numpy(scales).to(device)
    for b0 in range(0, B, chunk):
        c = sc[b0:b0 + chunk]
        s = -30.0 + 54.0 * c[:, 0:1] * torch.sin(0.007 * c[:, 3:4] * tau) \
            + 26.0 * c[:, 1:2] * torch.sin(0.037 * c[:, 4:5] * tau) \
            + 10.0 * c[:, 2:3] * torch.sin(0.19 * c[:, 5:6] * tau)
        out[b0:b0 + chunk] = torch.where(win, s, base)
    return out


def staircase(n_samples=150001, dt=0.1):
    """[n_samples] fp64: 15 s synthetic staircase (train-s1.py:268 duration): 500 ms plateaus stepping
    -40,-60,-20,-40,0,-20,... up to +40 and back down, with +/-120 mV tails.  The published shape is not in the
    reference tree."""
    t = np.arange(n_samples, dtype=np.float64) * dt
    v = np.full(n_samples, -80.0)
    levels = [-40, -60, -20, -40, 0, -20, 20, 0, 40, 20, 40, 0, 20, -20, 0, -40, -20, -60, -40]
    t0 = 1000.0
    v[(t >= 250) & (t < 300)] = -120.0
    for k, lv in enumerate(levels):
        v[(t >= t0 + 500.0 * k) & (t < t0 + 500.0 * (k + 1))] = lv
    end = t0 + 500.0 * len(levels)
    v[(t >= end) & (t < end + 500.0)] = -120.0
    v[(t >= end + 1000.0) & (t < end + 1400.0)] = -70.0
    return v


def activation_pr3(v_step, dt=0.1):
    """Pr3 steady-activation sweep on the data grid, train-s1.py:69-80 (8 s, 0.1 ms, 80001 samples)."""
    n = int(round(8000.0 / dt)) + 1
    s = int(round(0.1 / dt))
    v = np.zeros(n)
    v[:10000 * s] = -80
    v[10000 * s:60000 * s] = v_step
    v[60000 * s:70000 * s] = -40
    v[70000 * s:75000 * s] = -120
    v[75000 * s:] = -80
    return v


def deactivation_pr5(v_step, dt=0.1):
    """Pr5 deactivation sweep, train-s1.py:84-95 (10 s, 0.1 ms, 100001 samples)."""
    n = int(round(10000.0 / dt)) + 1
    s = int(round(0.1 / dt))
    v = np.zeros(n)
    v[:10000 * s] = -80
    v[10000 * s:30000 * s] = 50
    v[30000 * s:90000 * s] = v_step
    v[90000 * s:95000 * s] = -120
    v[95000 * s:] = -80
    return v


def pr4_synthetic(k, n_samples=29006, dt=0.1):
    """Pr4 sweep k of 16.  The reference only records Pr4's SHAPE (16 sweeps x 29 006 samples of 0.1 ms,
    train-r1.py:353 + figure-3/y1-pr4.pt); its wave-form lives in the missing data CSVs, so this is a synthetic
    three-step sweep of that shape: -80 mV hold, 1 s conditioning step to -50..+40 mV (6 mV apart), 0.5 s at -120 mV,
    back to -80 mV."""
    if not 0 <= k < 16:
        raise ValueError("Pr4 has 16 sweeps")
    t = np.arange(n_samples, dtype=np.float64) * dt
    v = np.full(n_samples, -80.0)
    v[(t >= 500.0) & (t < 1500.0)] = -50.0 + 6.0 * k
    v[(t >= 1500.0) & (t < 2000.0)] = -120.0
    return v


PR3_STEPS = (-60, -40, -20, 0, 20, 40, 60)                   # train-s1.py:77
PR5_STEPS = (-120, -110, -100, -90, -80, -70, -60, -50, -40)  # train-s1.py:92
