"""Reference-pixel subtraction on the GPU -- same call surface as the reference's
``utils/reference_subtraction.py`` (``ref_subtraction_channel`` :16, ``ref_subtraction_row`` :77), every argument included.

The image is a 2-D C-contiguous float32 array, science columns first and (where ``use_ref_channel`` is used) the 128 columns
of the reference output appended on the right; it is updated in place and returned, as the reference does.  The reference
hard-codes a 4096-pixel side (border pixels 0:4 and 4092:4096, rows 4092:4096, 32 channels) -- the defaults here; the
extra keywords ``nside`` / ``n_channels`` (and the image's own row count) exist so that small frames can be tested.

Two things are decided on the host exactly as numpy decides them in the reference:
  * ``slope=None``: the science-row and reference-row medians come from the device (exact selections), the fit is
    ``np.polyfit`` itself on those float32 medians (:114; it returns a float32 slope);
  * the dtype of the row update follows numpy's promotion (:123): a numpy float64 ``slope`` makes it float64
    (``RIP_ROW_SLOPE_F64``), a Python float or a float32 one (the polyfit result) keeps it float32 (``RIP_ROW_SLOPE_F32``).
"""

import numpy as np

from .. import _native

ROW_MEDIANS_ONLY, ROW_SLOPE_F64, ROW_SLOPE_F32 = 0, 1, 2


def _check(image):
    if not isinstance(image, np.ndarray) or image.dtype != np.float32 or not image.flags.c_contiguous or image.ndim != 2:
        raise TypeError("image must be a C-contiguous 2-D float32 array (updated in place)")


def _nside(image, nside, use_ref_channel):
    w = image.shape[1]
    if nside is None:
        nside = 4096   # reference_subtraction.py:107-111
    if nside < 16 or nside > w or (use_ref_channel and nside + 128 > w):
        raise ValueError(f"an image {w} columns wide does not hold {nside} science columns"
                         + (" plus the reference output" if use_ref_channel else ""))
    return int(nside)


def _run(image, slope, do_row, do_channel, lines, ctx):
    """the configuration of calibrateimage (33 channels, reference-output row medians, float64 slope): one call"""
    ctx = ctx or _native.default_context()
    _check(image)
    ny, w = image.shape
    nx = w - 128
    if nx <= 0 or nx % 128:
        raise ValueError("image must be (ny, 128*nchannel + 128): science channels plus the reference output")
    ln = None if lines is None else np.ascontiguousarray(lines, dtype=np.float64)
    ctx.check(ctx.lib.rip_stage_refpix_image(ctx.h, image.ctypes.data, ny, nx, float(slope), int(do_row), int(do_channel),
                                             None if ln is None else ln.ctypes.data, None, None, None))
    return image


def _is_f64_scalar(x):
    return isinstance(x, (np.floating, np.ndarray)) and np.asarray(x).dtype == np.float64


def row_medians(image, use_ref_channel=False, nside=None, science=True, ctx=None):
    """(ref_medians, sci_medians or None, ctr) of reference_subtraction.py:104-115, float32, from the device."""
    ctx = ctx or _native.default_context()
    _check(image)
    ny, w = image.shape
    ns = _nside(image, nside, use_ref_channel)
    ref = np.empty(ny, np.float32)
    sci = np.empty(ny, np.float32) if science else None
    ctr = np.empty(1, np.float32)
    ctx.check(ctx.lib.rip_stage_refpix_row(ctx.h, image.ctypes.data, ny, w, ns, int(bool(use_ref_channel)), ROW_MEDIANS_ONLY, 0.0,
                                           ref.ctypes.data, None if sci is None else sci.ctypes.data, ctr.ctypes.data))
    return ref, sci, ctr[0]


def ref_subtraction_row(image, use_ref_channel=False, slope=None, ctx=None, nside=None):
    """image[r, :] -= slope * (ref_med[r] - median(ref_med)) per row, ``ref_med[r]`` the median of the reference output of
    the row (``use_ref_channel``) or of its 4 + 4 border pixels; ``slope=None``: fitted (science-row medians against
    reference medians, ``np.polyfit``)."""
    ctx = ctx or _native.default_context()
    _check(image)
    ny, w = image.shape
    ns = _nside(image, nside, use_ref_channel)
    if slope is None:
        ref, sci, _ctr = row_medians(image, use_ref_channel, ns, True, ctx)
        slope, _ = np.polyfit(ref, sci, 1)          # reference_subtraction.py:114, on float32 medians
    if use_ref_channel and _is_f64_scalar(slope) and ns + 128 == w and ns % 128 == 0:
        return _run(image, slope, 1, 0, None, ctx)   # the driver's configuration: the single-purpose kernels
    mode = ROW_SLOPE_F64 if _is_f64_scalar(slope) else ROW_SLOPE_F32
    ctx.check(ctx.lib.rip_stage_refpix_row(ctx.h, image.ctypes.data, ny, w, ns, int(bool(use_ref_channel)), mode, float(slope),
                                           None, None, None))
    return image


def ref_subtraction_channel(image, channel_start=0, channel_end=128, use_ref_channel=False, lines=None, ctx=None, n_channels=None):
    """Per window of columns ``[channel_start + 128 k, channel_end + 128 k)``, k < 32 (33 with ``use_ref_channel``): subtract
    the line through the medians of the bottom and top 4 rows.  ``lines``: optional (nchannel, 2) float64 (m, c) from the
    caller's LAPACK instead of the device's two-point formula (channel line fit)."""
    ctx = ctx or _native.default_context()
    _check(image)
    ny, w = image.shape
    nchan = (32 if n_channels is None else int(n_channels)) + (1 if use_ref_channel else 0)   # :42-44
    if channel_end + (nchan - 1) * 128 > w:
        raise ValueError(f"{nchan} windows of columns [{channel_start}, {channel_end}) + 128 k do not fit an image {w} wide")
    if use_ref_channel and channel_start == 0 and channel_end == 128 and nchan * 128 == w:
        return _run(image, 0.0, 0, 1, lines, ctx)
    ln = None if lines is None else np.ascontiguousarray(lines, dtype=np.float64)
    if ln is not None and ln.shape != (nchan, 2):
        raise ValueError(f"lines must be ({nchan}, 2)")
    ctx.check(ctx.lib.rip_stage_refpix_channel(ctx.h, image.ctypes.data, ny, w, int(channel_start), int(channel_end), nchan,
                                               None if ln is None else ln.ctypes.data, None))
    return image


def refpix_tables(data, dark, amp33, amp33_med, slope, form=-1, ctx=None):
    """The tables the chain's reference-pixel step applies to a ramp (gen_cal_image.py:531-556): ``rowcorr`` (ngrp, ny) float64 =
    ``slope * float64(float32(row median of the reference output - ctr))`` and ``lines`` (ngrp, nx // 128, 2) float64 = (m, c) of
    the science channels (two-point formula, DESIGN.md "channel line fit").  ``form``: 1 the single-launch kernel, 0 the
    multi-launch kernels, -1 the library's default; identical bits.  Returns (rowcorr, lines, status)."""
    import ctypes

    ctx = ctx or _native.default_context()
    data = np.ascontiguousarray(data)
    if data.dtype not in (np.uint16, np.float32):
        raise TypeError("data must be uint16 or float32")
    G, ny, nx = data.shape
    dark = np.ascontiguousarray(dark[:G], dtype=np.float32)
    amp33 = np.ascontiguousarray(amp33, dtype=np.uint16)
    med = np.ascontiguousarray(amp33_med, dtype=np.float32)
    if dark.shape != (G, ny, nx) or amp33.shape != (G, ny, 128) or med.shape != (ny, 128):
        raise ValueError("shapes: data (G, ny, nx), dark (>= G, ny, nx), amp33 (G, ny, 128), amp33_med (ny, 128)")
    rowcorr = np.empty((G, ny), np.float64)
    lines = np.empty((G, nx // 128, 2), np.float64)
    status = ctypes.c_int(0)
    ctx.check(ctx.lib.rip_stage_refpix_tables(ctx.h, data.ctypes.data, _native.dtype_code(data), dark.ctypes.data, amp33.ctypes.data,
                                              med.ctypes.data, float(slope), G, ny, nx, int(form), rowcorr.ctypes.data,
                                              lines.ctypes.data, ctypes.addressof(status)))
    return rowcorr, lines, status.value
