"""Oracle: reference-pixel correction.  Test infrastructure only.

Follows ``gen_cal_image.py:531-556`` (per-group loop), and
``utils/reference_subtraction.py:77-125`` (row step) / ``:16-74`` (channel step).

dtype recipe (numpy 2 / NEP 50), per group j:
  img[:, :N]   = f32(data[j]) - f32(dark[j])                      (f32)
  img[:, N:]   = u16(amp33[j]) - f32(med)  -> f32 ; minus its own np.median (f32)
  row step     : ref_med[r] = median_f32(img[r, N:N+128])   (even count: f32(lo+hi)/2)
                 ctr        = median_f32(ref_med)
                 img[r, :]  = f32( f64(img[r, :]) - f64(slope) * f64( f32(ref_med[r]-ctr) ) )
  channel step : for each 128-column channel (32 science + the reference output):
                 b = median_f32(img[0:4, ch]) ; t = median_f32(img[R-4:R, ch])      (R = number of rows)
                 (m, c) = lstsq line through (1.5, b), (R-2.5, t)   (f64, LAPACK gelsd)
                 img[r, ch] = f32( f64(img[r, ch]) - (m*r + c) )
  data[j]      = img[:, :N] + dark[j]                                (f32)
The frame size (R rows x N science columns) is a parameter here; the reference hard-codes
R = N = 4096 (4092, 4093.5).
"""

import numpy as np

CHANNEL_WIDTH = 128


def optimal_refout_slope(m_pink, ru_pink, c_pink, amp33_std):
    """Scalar weight of the reference-output row correction (``gen_cal_image.py:542-553``).

    m_pink, ru_pink, c_pink are Python floats (YAML scalars of the ``read``
    file); ``np.median(std)`` is an f32 scalar, ``/128`` keeps f32,
    ``/np.log(4096)`` (an f64 numpy scalar) promotes to f64.
    """
    cvar = c_pink**2
    tail = np.median(amp33_std) ** 2 / 128 / np.log(4096)
    return m_pink * cvar / (m_pink**2 * cvar + ru_pink**2 + tail)


def _polyfit_slope(ref_med, sci_med):
    m, _ = np.polyfit(ref_med, sci_med, 1)
    return m


def row_step(image, nside, use_ref_channel, slope):
    """Row-wise correction, in place on ``image`` (rows x (nside [+128])), f32.

    ``reference_subtraction.py:104-123``.  Returns (image, ref_med, ctr).
    """
    if use_ref_channel:
        ref_med = np.median(image[:, nside : nside + CHANNEL_WIDTH], axis=1)
    else:
        edge = np.concatenate((image[:, 0:4], image[:, nside - 4 : nside]), axis=1)
        ref_med = np.median(edge, axis=1)
    ref_med = np.asarray(ref_med, dtype=image.dtype)
    if slope is None:
        sci_med = np.median(image[:, 4 : nside - 4], axis=1)
        slope = _polyfit_slope(ref_med, sci_med)
    ctr = np.median(ref_med)
    if isinstance(slope, np.floating) and slope.dtype == np.float64 or isinstance(slope, np.ndarray) and slope.dtype == np.float64:
        # a numpy f64 scalar (gen_cal_image.py:542-553, np.polyfit): (ref_med[r]-ctr) is rounded in the image dtype first,
        # then multiplied and subtracted in f64, the row cast back on assignment
        corr = np.float64(slope) * (ref_med - ctr).astype(np.float64)
        image[:, :] = (image.astype(np.float64) - corr[:, None]).astype(image.dtype)
    else:
        # a Python float (weak under NEP 50) or a numpy f32 scalar: `m_med * (ref_medians[i] - ctr)` and the subtraction stay
        # in the image dtype (reference_subtraction.py:123)
        corr = image.dtype.type(slope) * (ref_med - ctr)
        image[:, :] = image - corr[:, None]
    return image, ref_med, ctr


def channel_line(bottom_med, top_med, nrows):
    """Line through (1.5, bottom) and (nrows-2.5, top) exactly as ``reference_subtraction.py:57-60``."""
    A = np.vstack([(1.5, nrows - 2.5), np.ones(2)]).T
    m_cor, c_cor = np.linalg.lstsq(A, (bottom_med, top_med), rcond=None)[0]
    return m_cor, c_cor


def channel_step(image, nside, use_ref_channel, channel_start=0, channel_end=CHANNEL_WIDTH):
    """Channel-wise correction, in place (``reference_subtraction.py:44-74``): windows of columns
    ``[channel_start + 128 k, channel_end + 128 k)``, one after the other.

    Returns (image, table) with table[ch] = (bottom_med, top_med, m, c).
    """
    nch = nside // CHANNEL_WIDTH + (1 if use_ref_channel else 0)
    nrows = image.shape[0]
    rows = np.arange(nrows, dtype=np.float64)
    table = np.zeros((nch, 4), dtype=np.float64)
    for ch in range(nch):
        sl = slice(channel_start + ch * CHANNEL_WIDTH, channel_end + ch * CHANNEL_WIDTH)
        b = np.median(image[0:4, sl])
        t = np.median(image[nrows - 4 : nrows, sl])
        m_cor, c_cor = channel_line(b, t, nrows)
        line = m_cor * rows + c_cor  # f64
        image[:, sl] = (image[:, sl].astype(np.float64) - line[:, None]).astype(image.dtype)
        table[ch] = (b, t, m_cor, c_cor)
    return image, table


def correct_group(data_j, dark_j, amp33_j, amp33_med, slope):
    """One pass of the per-group loop body of ``gen_cal_image.py:533-556``.

    data_j f32 (N,N); dark_j f32 (N,N); amp33_j u16 (N,128) or None; amp33_med f32 (N,128).
    Returns (corrected data_j f32, diagnostics dict).
    """
    nside = data_j.shape[1]
    image = np.zeros((data_j.shape[0], nside + CHANNEL_WIDTH), dtype=np.float32)
    image[:, :nside] = data_j - dark_j
    diag = {}
    if amp33_j is not None:
        image[:, nside:] = amp33_j - amp33_med
        gmed = np.median(image[:, nside:])
        image[:, nside:] -= gmed
        diag["amp33_median"] = gmed
        image, ref_med, ctr = row_step(image, nside, True, slope)
        diag["ref_med"] = ref_med
        diag["ctr"] = ctr
    # (no amp33 in the read file: the reference's row step degenerates to the identity on an
    #  all-zero reference block -- see DESIGN.md)
    image, table = channel_step(image, nside, True)
    diag["channels"] = table
    return image[:, :nside] + dark_j, diag


def correct_cube(data, dark, amp33, amp33_med, slope):
    """All groups; returns a new f32 cube (the reference overwrites ``data`` in place)."""
    out = np.empty_like(data, dtype=np.float32)
    diags = []
    for j in range(data.shape[0]):
        out[j], d = correct_group(data[j], dark[j], None if amp33 is None else amp33[j], amp33_med, slope)
        diags.append(d)
    return out, diags
