"""athenak_amd -- MI355X-native MeshBlock finite-volume update (AthenaK hot path).

Layout: csrc/ (HIP kernels + C ABI, built into lib/libakmi.so), capi.py (ctypes binding),
and the host-side mirror of the reference's operator surface for this path:
parameter_input, mesh (Mesh/MeshBlock/MeshBlockPack), tasklist, driver, hydro, mhd, bvals,
pgen.  See DESIGN.md and INTEGRATION.md.
"""
from .parameter_input import ParameterInput  # noqa: F401
from .main import Simulation, run_deck  # noqa: F401

__all__ = ["ParameterInput", "Simulation", "run_deck"]
