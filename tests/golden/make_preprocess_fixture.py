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


if __name__ == "__main__":
    main(*sys.argv[1:])
