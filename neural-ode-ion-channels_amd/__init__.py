"""MI355X-native batched neural-ODE integrator for ion-channel gating models.

The directory name is not a Python identifier; import it with
    ion = importlib.import_module("neural-ode-ion-channels_amd")
or use the drop-in `torchdiffeq` shim at the repository root (`from torchdiffeq import odeint`).
"""
from . import capi  # noqa: F401
from .capi import IonodeError, build  # noqa: F401

__all__ = ["capi", "IonodeError", "build"]
