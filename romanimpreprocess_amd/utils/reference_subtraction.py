"""Reference-pixel subtraction on the GPU -- same call surface as the reference's
``utils/reference_subtraction.py`` (``ref_subtraction_channel`` :16, ``ref_subtraction_row`` :77) for
the configuration the L1->L2 driver uses: 33 channels (``use_ref_channel=True``) and a given ``slope``.

The image is (ny, nx+128) float32: science frame with the reference output appended on the right.
It is updated in place and returned.
"""

import numpy as np

from .. import _native


def _run(image, slope, do_row, do_channel, lines, ctx):
    ctx = ctx or _native.default_context()
    if image.dtype != np.float32 or not image.flags.c_contiguous or image.ndim != 2:
        raise TypeError("image must be a C-contiguous 2-D float32 array (updated in place)")
    ny, w = image.shape
    nx = w - 128
    if nx <= 0 or nx % 128:
        raise ValueError("image must be (ny, 128*nchannel + 128): science channels plus the reference output")
    ln = None if lines is None else np.ascontiguousarray(lines, dtype=np.float64)
    ctx.check(ctx.lib.rip_stage_refpix_image(ctx.h, image.ctypes.data, ny, nx, float(slope), int(do_row), int(do_channel),
                                             None if ln is None else ln.ctypes.data, None, None, None))
    return image


def ref_subtraction_row(image, use_ref_channel=False, slope=None, ctx=None):
    """image[r,:] -= slope * (median(image[r, -128:]) - median of those medians), per row."""
    if not use_ref_channel or slope is None:
        raise NotImplementedError(
            "the GPU path implements the configuration of calibrateimage (use_ref_channel=True with a given slope); "
            "the polyfit / border-pixel variant is not on the L1->L2 path")
    return _run(image, slope, 1, 0, None, ctx)


def ref_subtraction_channel(image, channel_start=0, channel_end=128, use_ref_channel=False, lines=None, ctx=None):
    """Per 128-column channel: subtract the line through the medians of the bottom and top 4 rows."""
    if not use_ref_channel or channel_start != 0 or channel_end != 128:
        raise NotImplementedError("the GPU path implements the 33-channel configuration of calibrateimage")
    return _run(image, 0.0, 0, 1, lines, ctx)
