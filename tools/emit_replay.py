"""Dev tool (CPU): how many dense-output passes does a wavefront of the lane-wise kernels need per step attempt?

python tools/emit_replay.py [hh|m6]

Replays the step logs of 64 trajectories of the closed-form bench workload (one wavefront in protocol-major order; the CPU oracle
supplies the logs) and counts the 64-sample passes per attempt of three emission schedules:
  * round 3: groups of 8 lanes take one emitting trajectory at a time; a pass lasts until the longest of its 8 trajectories is done
  * round 4: a work list of 8-sample chunks, a pass takes the next 8 chunks whatever trajectories they belong to
  * one trajectory per pass with all 64 lanes (the general fallback)
against the ideal (samples / 64).  This replay is what showed the round-3 schedule at 78 iterations per attempt where 36 would do
(DESIGN.md 5.1); it needs no GPU.
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
ion = importlib.import_module("neural-ode-ion-channels_amd")
import kat_cases as K  # noqa: E402
from oracle import oracle  # noqa: E402  (dev tool: the oracle is the step-log source)

model = 1 if (len(sys.argv) > 1 and sys.argv[1] == "m6") else 0
Nt = 20001
pv = ion.protocols.sinewave(ion.protocols.sinewave_scales(0, 64), n_samples=Nt, xp=torch, device="cpu").numpy()
rng = np.random.default_rng(0)
p0 = K.P_HH if model == 0 else K.P_M6
y0 = [0.0, 1.0] if model == 0 else [0.0, 1.0, 0, 0, 0, 0]
params = p0[None, :] * rng.uniform(0.8, 1.25, (393216, p0.size))
te = np.arange(Nt) * 0.1
logs = [oracle.solve(model, params[b:b + 1], pv[:1], y0, te, prot_t0=0.0, prot_dt=0.1, step_log_cap=4000)["step_log"] for b in range(0, 64 * 64, 64)]
A = max(len(l) for l in logs)
nout = np.zeros((A, 64), int)
for j, l in enumerate(logs):
    oi = 1
    for a, (t0, dt, _ratio, acc) in enumerate(l):
        if acc:
            g = oi
            while g < Nt and te[g] <= t0 + dt:
                g += 1
            nout[a, j], oi = g - oi, g
rng2 = np.random.default_rng(1)
it_r3 = it_list = it_one = 0
for a in range(A):
    em = nout[a][nout[a] > 0]
    if len(em) == 0:
        continue
    ch = (em + rng2.integers(0, 8, len(em)) + 7) // 8          # chunks of a step (random misalignment of its first sample)
    it_r3 += sum(ch[g:g + 8].max() for g in range(0, len(em), 8))
    it_list += (ch.sum() + 7) // 8
    it_one += ((em + 63) // 64).sum()
tot = nout.sum()
print(f"attempts of the wavefront {A}; emitting lanes per attempt {np.mean((nout > 0).sum(1)):.1f}; samples per step p10/p50/p90 "
      f"{np.percentile(nout[nout > 0], [10, 50, 90])}")
print(f"passes per attempt: ideal {tot / 64 / A:.1f} | round-3 groups {it_r3 / A:.1f} | work list {it_list / A:.1f} | one trajectory per pass {it_one / A:.1f}")
