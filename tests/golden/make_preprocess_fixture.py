#!/usr/bin/env python3
"""Fixture for the derivative-estimation preprocessing: every 50th row of the reference's cached regression
samples <ref>/s1/{v,a,dadt}.pt (written by train-s1.py:806-808 from seeded synthetic data) -- DATA, 132 410 rows -> 2 649.

    python tests/golden/make_preprocess_fixture.py [/root/reference]   ->  tests/golden/preproc_s1_rows.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main(ref="/root/reference"):
    v, a, d = (torch.load(os.path.join(ref, "s1", f + ".pt")).reshape(-1).numpy() for f in ("v", "a", "dadt"))
    np.savez_compressed(os.path.join(HERE, "preproc_s1_rows.npz"), n_rows=v.size, stride=50, v=v[::50], a=a[::50], dadt=d[::50])
    print(v.size, "rows ->", v[::50].size)
    # figure-0-s (round 5): the same pipeline on ONE sweep, every intermediate cached by the reference (figure-0-s.py:160-214):
    # i.pt (spline of the smoothed noisy current), didt.pt, a.pt, dadt.pt -- every 50th of 80 001 samples each
    f0 = {n: torch.load(os.path.join(ref, "figure-0-s", n + ".pt"), weights_only=True)[0].numpy().reshape(-1).astype(np.float64)
          for n in ("i", "didt", "a", "dadt")}
    np.savez_compressed(os.path.join(HERE, "preproc_fig0s.npz"), n=f0["i"].size, stride=50, **{k: x[::50] for k, x in f0.items()})
    print("figure-0-s", f0["i"].size, "samples ->", f0["i"][::50].size)


if __name__ == "__main__":
    main(*sys.argv[1:])
