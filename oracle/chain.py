"""Oracle: the whole per-pixel chain in the order of ``calibrateimage``
(``gen_cal_image.py:507-629``), on arrays instead of ASDF files.  Test infrastructure only.

Inputs are plain dicts of numpy arrays mirroring the ``roman`` branch of each CALDIR file
(SURVEY.md Appendix B):
  cal["dark"]              : data f32 (>=G,N,N), dark_slope f32 (N,N), dq u32 (N,N)
  cal["read"]              : data f32 (N,N), amp33 {med, std, M_PINK, RU_PINK} (optional), anc {C_PINK}
  cal["gain"]              : data f32|f64 (N,N)
  cal["linearitylegendre"] : data f32 (P+1,N,N), Smin, Smax, Sref f32 (N,N), dq u32 (N,N)
  cal["ipc4d"]             : data f32|f64 (3,3,Na,Na)         (optional)
  cal["flat"]              : data f32 (N,N)                    (optional: no flat division)
  cal["biascorr"]          : data f32 (>=G,Na,Na)              (optional)
  ramp                     : data u16|f32 (G,N,N), amp33 u16 (G,N,128) or None, groupdq u8 (G,N,N),
                             pixeldq u32 (N,N), read_pattern, frame_time
``groupdq``/``pixeldq`` are the arrays AFTER dq-init and saturation flagging (romancal/stcal
steps whose source is not in the reference tree).
"""

import numpy as np

from . import finish as fin
from . import ipc, linearity, rampfit, refpix

DEFAULT_RAMP_OPT_PARS = {"slope": 0.4, "gain": 1.8, "sigma_read": 6.5}


def calibrate_arrays(ramp, cal, exclude_first=True, ramp_opt_pars=None, jump_pars=None,
                     area_factor=None, nborder=4, stages=None, timings=None):
    """Returns dict(slope, err_read, err_poisson, pixeldq, groupdq, data (the corrected cube), K, meta).
    ``timings``: optional dict that receives the seconds spent per stage (refpix, bias_lin, ipc, rampfit, finish)."""
    import time as _time

    _t = [_time.perf_counter()]

    def _lap(name):
        if timings is not None:
            now = _time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - _t[0]
            _t[0] = now

    nb = nborder
    data = np.array(ramp["data"], dtype=np.float32)  # dq-init: u16 -> f32
    rdq = np.array(ramp["groupdq"], dtype=np.uint8)
    pdq = np.array(ramp["pixeldq"], dtype=np.uint32)
    G, ny, nx = data.shape
    meta = rampfit.ma_table_meta(ramp["read_pattern"], ramp["frame_time"])
    meta["nborder"] = nb
    if exclude_first:
        rdq[0] |= np.uint8(1)

    # reference pixels (gen_cal_image.py:531-556)
    rd = cal["read"]
    a33 = rd.get("amp33")
    refdiag = None
    if stages is not None and not stages.get("refpix", True):
        pass
    elif a33 is not None:
        slope_ref = refpix.optimal_refout_slope(a33["M_PINK"], a33["RU_PINK"], rd["anc"]["C_PINK"], a33["std"])
        data, refdiag = refpix.correct_cube(data, cal["dark"]["data"], ramp["amp33"], a33["med"], slope_ref)
    else:
        data, refdiag = refpix.correct_cube(data, cal["dark"]["data"], None, None, None)

    _lap("refpix")
    # bias (:559-565)
    if "biascorr" in cal:
        b = cal["biascorr"]["data"]
        de = b.shape[0] - G
        data[:, nb:-nb, nb:-nb] -= b[de:]

    # linearity (:580-588)
    L = cal["linearitylegendre"]
    data, dq_lin = linearity.multilin(
        data, L["data"], L["Smin"], L["Smax"], L["Sref"], L["dq"],
        do_not_flag_first=(list(ramp["read_pattern"][0]) == [0]),
        attempt_corr=linearity.attempt_corr_from_groupdq(rdq),
    )
    pdq |= dq_lin

    _lap("bias_lin")
    # IPC (:594-597)
    kern = cal["ipc4d"]["data"] if "ipc4d" in cal else None
    gain = cal["gain"]["data"]
    if kern is not None:
        ipc.correct_cube(data, kern, gain)

    _lap("ipc")
    # ramp fit (:434-463)
    uopt = ramp_opt_pars or DEFAULT_RAMP_OPT_PARS
    u_ = float(uopt["slope"]) / float(uopt["gain"]) / float(uopt["sigma_read"]) ** 2
    meta["K"] = rampfit.construct_weights(u_, meta, exclude_first)
    slope, er, ep = rampfit.ramp_fit(data, rdq, pdq, gain, rd["data"], meta, exclude_first, jump_pars)

    _lap("rampfit")
    # dark rate, error algebra, flat (:188-233, :607-629)
    dk = cal["dark"]
    dark_rate = fin.dark_rate_deconvolved(dk["dark_slope"], kern, gain)
    flat_dn = None
    # order in the reference: image-model packaging, dark, then get_flat flags pdq, then divide
    slope, er, ep = fin.finish(slope, er, ep, pdq, nb, dark_rate, dk.get("dq"), None, None)
    if "flat" in cal:
        flat_dn = fin.get_flat(cal["flat"]["data"], gain, kern, nb, pdq, ipc_deconvolve=kern is not None)
        af = 1.0 if area_factor is None else area_factor
        flat = (flat_dn / af).astype(np.float32)
        slope /= flat
        er /= flat
        ep /= flat
    _lap("finish")
    return {
        "slope": slope, "err_read": er, "err_poisson": ep, "pixeldq": pdq, "groupdq": rdq,
        "data": data, "K": meta["K"], "meta": meta, "flat": flat_dn, "refpix_diag": refdiag,
    }
