"""autoforce_amd — MI355X-native (gfx950) SGPR force-field evaluator behind AutoForce's
ActiveCalculator surface.  All numerics live in libsgpr_hip.so (hand-written HIP); see
DESIGN.md.  Importing this package never imports the test oracle."""
from ._lib import SgprError, device_count, load  # noqa: F401
from .model import Local, SGPRModel, default_radii  # noqa: F401

__all__ = ["SGPRModel", "Local", "SgprError", "device_count", "load", "default_radii"]
