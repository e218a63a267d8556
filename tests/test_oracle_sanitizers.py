"""CPU: the oracle's C sources under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not
available on the pool, so the restatement the kernels are compared with is the code that gets the memory / UB check).

The sanitizer build (oracle/Makefile target `asan`) is loaded into a child python with libasan preloaded and replays three short
known-answer cases (one per RHS family: NN-f with the shipped s1 weights, HH 2-state, 6-state) in both state dtypes; the child
compares against the regular build, so a sanitizer report (non-zero exit) or a differing bit fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import kat_cases as K
from oracle import oracle
assert oracle._LIB.endswith("liboracle_asan.so"), oracle._LIB
w = K.load_weights("s1")
t_ms, v_mv, te = K.activation(40)
te = te[:1201]
ref = np.load(sys.argv[1])
k = 0
for f32 in (False, True):
    for model, p, y0, kw in ((K.MODEL_NNF, K.P_HH, K.NN_Y0, dict(weights=w, mlp_layers=K.MLP_L, mlp_width=K.MLP_N)),
                             (K.MODEL_HH2, K.P_HH, [0.0, 1.0], {}),
                             (K.MODEL_MARKOV6, K.P_M6, [0.0, 1.0, 0.0, 0.0, 0.0, 0.0], {})):
        o = oracle.solve(model, p, v_mv, y0, te, prot_t0=float(t_ms[0]), prot_dt=float(t_ms[1] - t_ms[0]), state_f32=f32,
                         step_log_cap=4096, **kw)
        assert o["status"][0] == 0
        assert np.array_equal(o["y"], ref[f"y{k}"]) and np.array_equal(o["stats"], ref[f"s{k}"]), (model, f32)
        k += 1
# ragged inputs: an output time beyond the protocol, a step budget that trips, several trajectories over two protocols
o = oracle.solve(K.MODEL_HH2, np.tile(K.P_HH, (5, 1)), np.stack([v_mv, v_mv[::-1]]), [0.0, 1.0], np.array([0.0, 10.0, 9000.0]),
                 prot_t0=float(t_ms[0]), prot_dt=float(t_ms[1] - t_ms[0]), max_total_steps=50,
                 prot_of_traj=np.array([0, 1, 0, 1, 1], dtype=np.int32))
assert (o["status"] == 3).all()
print("sanitizer replay ok")
'''


def _regular(tmp_path):
    """the same cases through the regular build, saved for the child"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import kat_cases as K
    from oracle import oracle
    oracle.build()
    w = K.load_weights("s1")
    t_ms, v_mv, te = K.activation(40)
    te = te[:1201]
    out, k = {}, 0
    for f32 in (False, True):
        for model, p, y0, kw in ((K.MODEL_NNF, K.P_HH, K.NN_Y0, dict(weights=w, mlp_layers=K.MLP_L, mlp_width=K.MLP_N)),
                                 (K.MODEL_HH2, K.P_HH, [0.0, 1.0], {}),
                                 (K.MODEL_MARKOV6, K.P_M6, [0.0, 1.0, 0.0, 0.0, 0.0, 0.0], {})):
            o = oracle.solve(model, p, v_mv, y0, te, prot_t0=float(t_ms[0]), prot_dt=float(t_ms[1] - t_ms[0]), state_f32=f32, **kw)
            out[f"y{k}"], out[f"s{k}"] = o["y"], o["stats"]
            k += 1
    path = os.path.join(tmp_path, "regular.npz")
    np.savez(path, **out)
    return path


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    ref = _regular(str(tmp_path))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", IONODE_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"),
               OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD, ref], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitizer replay ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
