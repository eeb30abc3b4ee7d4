"""starks_amd -- the STARK proving hot path of computablelabs/starks (NTT / LDE / Merkle / FRI commit
over the MiMC prime) on AMD MI355X, behind the reference's own Python call sites.  See DESIGN.md."""
from ._lib import MIMC_P, StarkHipError  # noqa: F401
from .modp import IntegersModP  # noqa: F401

__all__ = ["MIMC_P", "StarkHipError", "IntegersModP"]
