"""Generate the committed parity fixtures from the reference's DATA artefacts.

Run once in the build container (the reference tree is not available on the GPU box):

    python tests/golden/make_fixtures.py [/root/reference]

Only *data* is read (state dicts, CSV protocol, recorded logs, a cached tensor);
no reference source file is imported, executed or copied.  Outputs (all little-endian):

  weights_<model>.f32   raw fp32, concatenation of net.{0,2,...,12}.{weight,bias} in
                        state-dict order, bit-exact copies (never re-quantised);
                        <model> in s1, s2, d1, d2           
  ap2hz.f64             35000 x 2 fp64: time [ms] (csv seconds * 1e3, as the reference
                        scales it, train-s1.py:44-45) and voltage [mV]
  kat_losses.json       the 26 recorded '--pred' losses of s1/s2/d1/d2 'log2'
  fig0s_hh_r.f32        the reference's r(t) state trace of figure-0-s (fp32, every 10th sample): i_n / (a_n (v + 86)),
                        figure-0-s.py:147-153,196-200
  fig0s_hh_current.f64  figure-0-s/i_n.pt minus its seeded noise (figure-0-s.py:31,141-144),
                        decimated x10 -> 8001 fp64 samples of the HH current trace
"""
import json
import os
import re
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def weights():
    meta = {}
    for m in ("s1", "s2", "d1", "d2"):
        sd = torch.load(os.path.join(REF, m, "model-state-dict.pt"), weights_only=True)
        keys = list(sd.keys())
        blobs, shapes = [], []
        for k in keys:
            t = sd[k]
            assert t.dtype == torch.float32
            blobs.append(t.contiguous().numpy().reshape(-1))
            shapes.append([k, list(t.shape)])
        flat = np.concatenate(blobs).astype("<f4")
        flat.tofile(os.path.join(OUT, f"weights_{m}.f32"))
        meta[m] = shapes
    with open(os.path.join(OUT, "weights_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


def ap2hz():
    p = np.loadtxt(os.path.join(REF, "test-protocols", "ap2hz.csv"), skiprows=1, delimiter=",")
    p[:, 0] *= 1e3  # s -> ms, same operation as the reference applies
    p.astype("<f8").tofile(os.path.join(OUT, "ap2hz.f64"))


def kats():
    out = {}
    for m in ("s1", "s2", "d1", "d2"):
        sec, d = None, {}
        for line in open(os.path.join(REF, m, "log2")):
            line = line.rstrip()
            mm = re.match(r"^(AP 2Hz|APs|Sinewave|Staircase) prediction \| Total Loss ([0-9.]+)", line)
            if mm:
                d[mm.group(1)] = float(mm.group(2))
                continue
            if line.startswith("Activation prediction"):
                sec = "act"
            elif line.startswith("Deactivation prediction"):
                sec = "deact"
            elif line.startswith("Activation time constant"):
                sec = "atau"
            mm = re.match(r"^\s+(-?[0-9.]+)m[Vs] \| Total Loss ([0-9.]+)", line)
            if mm and sec:
                d.setdefault(sec, {})[mm.group(1)] = float(mm.group(2))
        out[m] = d
    with open(os.path.join(OUT, "kat_losses.json"), "w") as f:
        json.dump(out, f, indent=1)


def fig0s():
    i_n = torch.load(os.path.join(REF, "figure-0-s", "i_n.pt"), weights_only=True)[0].numpy().reshape(-1)
    np.random.seed(0)
    noise = np.random.normal(0, 0.1, 80001)
    clean = i_n - noise
    clean[::10].astype("<f8").tofile(os.path.join(OUT, "fig0s_hh_current.f64"))
    # the reference's own r(t) = odeint(...)[:, 0, 1] of the same solve family (figure-0-s.py:147-153), recovered from what the
    # script cached: a_noisy = i_noisy / (g * r * (v - e)) (figure-0-s.py:196-200, g = 1, e = -86) => r = i_n / (a_n * (v + 86)).
    # The quotient lands on the fp32 grid to 2e-16: these ARE the fp32 state values torchdiffeq returned -- a state-level golden.
    a_n = torch.load(os.path.join(REF, "figure-0-s", "a_n.pt"), weights_only=True)[0].numpy().reshape(-1).astype(np.float64)
    v = torch.load(os.path.join(REF, "figure-0-s", "v.pt"), weights_only=True)[0].numpy().reshape(-1).astype(np.float64)
    r = i_n.astype(np.float64) / (a_n * (v + 86.0))
    assert np.isfinite(r).all() and np.abs(r - r.astype(np.float32)).max() <= 1e-15 * np.abs(r).max() * 4
    r[::10].astype("<f4").tofile(os.path.join(OUT, "fig0s_hh_r.f32"))


if __name__ == "__main__":
    weights(); ap2hz(); kats(); fig0s()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
