"""Data-quality bit table (the one place where bit values are written down).

The reference takes these from ``roman_datamodels.dqflags`` (imported at
``gen_cal_image.py:33``, ``fitting.py:17``, ``ipc_linearity.py:32``,
``flatutils.py:15``); that package is not available offline, so the table is
restated here.  Members are ``np.uint32`` scalars, so
``array & pixel.X`` keeps unsigned dtypes exactly as in the reference.
"""

import enum

import numpy as np


class pixel(np.uint32, enum.Enum):
    """2-D pixel DQ flags (uint32)."""

    GOOD = 0
    DO_NOT_USE = 2**0
    SATURATED = 2**1
    JUMP_DET = 2**2
    DROPOUT = 2**3
    GW_AFFECTED_DATA = 2**4
    PERSISTENCE = 2**5
    AD_FLOOR = 2**6
    OUTLIER = 2**7
    UNRELIABLE_ERROR = 2**8
    NON_SCIENCE = 2**9
    DEAD = 2**10
    HOT = 2**11
    WARM = 2**12
    LOW_QE = 2**13
    TELEGRAPH = 2**15
    NONLINEAR = 2**16
    BAD_REF_PIXEL = 2**17
    NO_FLAT_FIELD = 2**18
    NO_GAIN_VALUE = 2**19
    NO_LIN_CORR = 2**20
    NO_SAT_CHECK = 2**21
    UNRELIABLE_BIAS = 2**22
    UNRELIABLE_DARK = 2**23
    UNRELIABLE_SLOPE = 2**24
    UNRELIABLE_FLAT = 2**25
    UNRELIABLE_RESET = 2**28
    OTHER_BAD_PIXEL = 2**30
    REFERENCE_PIXEL = 2**31


class group(np.uint32, enum.Enum):
    """Per-group (resultant) DQ flags; stored as uint8 in ``groupdq``."""

    GOOD = 0
    DO_NOT_USE = 2**0
    SATURATED = 2**1
    JUMP_DET = 2**2
    DROPOUT = 2**3
    AD_FLOOR = 2**6
