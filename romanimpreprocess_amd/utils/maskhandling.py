"""Grown pixel masks on the GPU -- same call surface as the reference's ``utils/maskhandling.py`` (``CombinedMask``,
``PixelMask1``).  ``build`` runs ``rip_stage_build_mask`` (bit-exact with the reference's per-bit convolutions)."""

import numpy as np

from .. import _native
from ..dqflags import pixel


class CombinedMask:
    """``maskdict``: flag name (or bit number) -> growth (1 copy, 5 plus-shaped, 9 3x3, 25 5x5), as
    ``maskhandling.py:38-58``."""

    def __init__(self, maskdict):
        self.array = np.zeros(32, dtype=np.uint8)
        for key, grow in maskdict.items():
            if isinstance(key, str):
                bit = int(getattr(pixel, key.upper())).bit_length() - 1
            else:
                bit = int(key)
            self.array[bit] = int(grow)

    def build(self, dq, ctx=None):
        """Boolean mask (True = masked) from a 2-D uint32 dq array (``maskhandling.py:82-117``)."""
        ctx = ctx or _native.default_context()
        dq = np.ascontiguousarray(dq, dtype=np.uint32)
        ny, nx = dq.shape
        out = np.empty((ny, nx), np.uint8)
        ctx.check(ctx.lib.rip_stage_build_mask(ctx.h, dq.ctypes.data, ny, nx, self.array.ctypes.data, out.ctypes.data))
        return out.astype(bool)


# the reference's standard choice (maskhandling.py:152-180)
PixelMask1 = CombinedMask({
    "DO_NOT_USE": 1, "JUMP_DET": 5, "DROPOUT": 25, "GW_AFFECTED_DATA": 1, "PERSISTENCE": 1, "AD_FLOOR": 5,
    "UNRELIABLE_ERROR": 1, "NON_SCIENCE": 1, "DEAD": 9, "HOT": 9, "WARM": 1, "LOW_QE": 9, "TELEGRAPH": 1,
    "NO_FLAT_FIELD": 9, "NO_GAIN_VALUE": 9, "NO_LIN_CORR": 9, "NO_SAT_CHECK": 9, "UNRELIABLE_BIAS": 1,
    "UNRELIABLE_DARK": 9, "UNRELIABLE_SLOPE": 9, "UNRELIABLE_FLAT": 9, "UNRELIABLE_RESET": 9, "OTHER_BAD_PIXEL": 9,
})
