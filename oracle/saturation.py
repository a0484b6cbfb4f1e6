"""Oracle: dq-init + saturation flagging.  Test infrastructure only.

numpy restatement of the reference's call ``saturation.flag_saturation(...)`` at ``gen_cal_image.py:172-185`` (romancal /
stcal ``flag_saturated_pixels`` with ``n_pix_grow_sat = 1``) from its arguments and ``docs/L1_to_L2_README.rst:139-141``.
stcal's source is not in the reference tree and not installed here: PARITY UNPINNED -- the device kernel
(``misc.hip: sat_exceed_kernel / sat_flags_kernel``) is tested bit for bit against this restatement.
"""

import numpy as np

SATURATED = 2
NO_SAT_CHECK = np.uint32(2**21)


def read_pattern_dilution(read_pattern):
    """Per group: mean(read indices) / last read index -- the factor by which the average of a group's reads of a linear
    ramp lies below its last read.  stcal's ``flag_saturated_pixels`` compares a group with ``sat_thresh * dilution`` when
    it is given the read pattern ("checks for groups with some reads saturated", docs/L1_to_L2_README.rst:139-141; the
    reference passes the pattern at gen_cal_image.py:172-185).  The pattern is used as the files hold it (the reference's
    simulated files count reads from 0, so a group holding only read 0 gives 0/0 = NaN, which never flags)."""
    with np.errstate(all="ignore"):
        return np.array([np.float64(np.mean(r)) / np.float64(r[-1]) for r in read_pattern], dtype=np.float64)


def flag_saturation(ramp, sat_threshold, backup=1, skip_firstn=1, n_pix_grow_sat=1, sat_dq=None, read_pattern=None):
    """Saturation flags (restatement of the call at gen_cal_image.py:172-185; stcal's source is not available:
    PARITY UNPINNED).  A resultant of group g >= skip_firstn is SATURATED where data >= threshold; the flag is
    grown by ``n_pix_grow_sat`` pixels (3x3 box for 1), is sticky for all later groups, and is also set on the
    ``backup`` preceding groups (but never on the first ``skip_firstn`` ones).  Pixels whose threshold is NaN or
    flagged NO_SAT_CHECK are not checked.  pixeldq receives SATURATED where any group is flagged.
    ``read_pattern``: when given, group g is compared with f64(threshold) * read_pattern_dilution[g] (partial saturation of
    a group's later reads).  stcal's further special case for the third group is not restated."""
    data, gdq, pdq = ramp["data"], ramp["groupdq"], ramp["pixeldq"]
    G = data.shape[0]
    thr = np.array(sat_threshold, dtype=np.float32)
    nocheck = ~np.isfinite(thr)
    if sat_dq is not None:
        nocheck |= (np.asarray(sat_dq) & NO_SAT_CHECK) != 0
    sat = np.zeros(data.shape, dtype=bool)
    dil = None if read_pattern is None else read_pattern_dilution(read_pattern)
    for g in range(skip_firstn, G):
        if dil is None:
            s = (data[g] >= thr) & ~nocheck
        else:
            with np.errstate(all="ignore"):
                s = (data[g].astype(np.float64) >= thr.astype(np.float64) * dil[g]) & ~nocheck
        for _ in range(n_pix_grow_sat):
            grown = s.copy()
            grown[1:, :] |= s[:-1, :]
            grown[:-1, :] |= s[1:, :]
            s = grown.copy()
            s[:, 1:] |= grown[:, :-1]
            s[:, :-1] |= grown[:, 1:]
        sat[g] = s
    for g in range(skip_firstn + 1, G):
        sat[g] |= sat[g - 1]
    for _ in range(int(backup)):
        for g in range(skip_firstn, G - 1):
            sat[g] |= sat[g + 1]
    gdq |= np.where(sat, np.uint8(SATURATED), np.uint8(0))
    pdq |= np.where(sat.any(axis=0), np.uint32(SATURATED), np.uint32(0))
