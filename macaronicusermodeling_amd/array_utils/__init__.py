"""Drop-in for the reference's `array_utils` package: `from array_utils import c_array_utils as au`
(LBP.py:6) becomes `from macaronicusermodeling_amd.array_utils import c_array_utils as au`."""
from . import c_array_utils  # noqa: F401
