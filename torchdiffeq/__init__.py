"""Drop-in shim: `from torchdiffeq import odeint` (train-s1.py:29-32) resolves to the MI355X-native integrator.

Put the repository root on PYTHONPATH ahead of any installed torchdiffeq; the reference scripts then run unchanged.
Nothing of the upstream torchdiffeq package is contained here -- this file only re-exports two names.
"""
import importlib as _importlib

_impl = _importlib.import_module("neural-ode-ion-channels_amd.odeint")
odeint = _impl.odeint
odeint_adjoint = _impl.odeint_adjoint
__version__ = "0.2.1+ionode"
__all__ = ["odeint", "odeint_adjoint"]
