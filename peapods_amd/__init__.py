"""peapods_amd: MI355X-native engine for the spin-sim hot path of PeaBrane/peapods.

`Ising` mirrors `peapods.Ising`; `IsingSimulation` mirrors the reference's native
class.  Both need libpeapods_amd.so (python -m peapods_amd.build) and a HIP device.
"""

from ._core import IsingSimulation
from .spin_models import Ising

__all__ = ["Ising", "IsingSimulation"]
__version__ = "0.1.0"
