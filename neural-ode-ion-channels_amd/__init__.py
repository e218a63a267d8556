"""MI355X-native batched neural-ODE integrator for ion-channel gating models.

The directory name is not a Python identifier; import it with
    ion = importlib.import_module("neural-ode-ion-channels_amd")
or use the drop-in `torchdiffeq` shim at the repository root (`from torchdiffeq import odeint`).
"""
from . import batched, capi, grad, protocols, rhs, schedule  # noqa: F401
from .batched import Solution, solve  # noqa: F401
from .capi import IonodeError, build  # noqa: F401
from .odeint import odeint, odeint_adjoint  # noqa: F401
from .rhs import UnrecognisedRhs, recognise  # noqa: F401

__all__ = ["capi", "batched", "grad", "protocols", "rhs", "schedule", "IonodeError", "UnrecognisedRhs", "build", "solve", "Solution",
           "odeint", "odeint_adjoint", "recognise"]
