"""L1 -> L2 driver on the GPU: same entry point and configuration surface as the reference's
``L1_to_L2/gen_cal_image.py`` (``calibrateimage(config, verbose=True)`` :480, CLI ``python -m
romanimpreprocess_amd.L1_to_L2.gen_cal_image config.yaml`` :742-746).

What runs where
  * the per-pixel chain of ``calibrateimage`` lines 531-629 (reference pixels, bias, linearity, IPC, ramp fit
    with jump detection, dark rate, error split, flat) -> one ``rip_calibrate`` call (HIP kernels);
  * the host keeps what the reference does on the host in microseconds: metadata (:123-145), ramp weights
    (:434-444), the process log, file I/O (ASDF through ``asdf`` when installed, else the in-repo reader/writer,
    or ``.npz`` mirrors -- see ``calio``);
  * steps whose source is NOT in the reference tree are restated from their call sites and documented
    behaviour (parity unpinned): dq-init (:118, mask -> pixeldq, zero groupdq), saturation flagging
    (:148-185, on the device: ``rip_ramp_desc::flag_saturation``; its numpy restatement is ``oracle/saturation.py``), and
    the L2 tree layout of ``romanisim.image.make_asdf`` (:653-662);
  * the post-path reductions (:632-651, 697-712: median gain, sky mode, SKYORDER model, SLICEOUT) -> ``utils/sky.py``,
    ``utils/maskhandling.py`` (HIP kernels);
  * not done here (outside the hot path): WCS->gwcs and the pixel-area map (pass ``AREAFACTOR``: file with
    an (N,N) f64 array, else 1), dark decay, WFI18 transient, romancal likelihood ramp fit, FITS output.  Asking for one
    of those raises NotImplementedError.
"""

import sys

import numpy as np

from .. import calio, pars, pipeline, plan as planmod
from ..dqflags import group
from ..utils import maskhandling, processlog, sky

_cal_cache = {}  # (ctx id, tuple of CALDIR paths) -> slot


def initializationstep(config, caldir, mylog):
    """Read the L1 file and the mask: data (u16 cube), amp33, groupdq (zeros), pixeldq (mask dq), meta."""
    with calio.open_tree(config["IN"]) as f:
        r = f["roman"]
        data = np.ascontiguousarray(r["data"])
        amp33 = np.ascontiguousarray(r["amp33"]) if "amp33" in r else None
        exposure = r["meta"]["exposure"]
        frame_time = float(exposure["frame_time"])
        read_pattern = [list(map(int, g)) for g in exposure["read_pattern"]]
        l1meta = calio._materialise(r["meta"])
    if data.dtype != np.uint16:
        data = data.astype(np.float32)
    if "mask" in caldir:
        with calio.open_tree(caldir["mask"]) as f:
            pixeldq = np.array(f["roman"]["dq"], dtype=np.uint32)
    else:
        pixeldq = np.zeros(data.shape[1:], dtype=np.uint32)
    groupdq = np.zeros(data.shape, dtype=np.uint8)
    meta = planmod.exposure_meta(read_pattern, frame_time)
    if config.get("EXCLUDE_FIRST", True):
        groupdq[0] |= np.uint8(group.DO_NOT_USE)
    return {"data": data, "amp33": amp33, "groupdq": groupdq, "pixeldq": pixeldq, "meta": l1meta}, meta


def load_caldir_arrays(caldir):
    """``roman`` branches of the CALDIR files the per-pixel chain needs (KeyError on a missing required key)."""
    cal = {}
    for key in ("dark", "read", "gain", "linearitylegendre", "flat"):
        cal[key] = calio.roman_branch(caldir[key])
    for key in ("ipc4d", "biascorr", "saturation"):   # optional (no ipc4d: "skipping IPC correction", ipc_linearity.py:170-176)
        if key in caldir:
            cal[key] = calio.roman_branch(caldir[key])
    return cal


def _file_stamp(path):
    """(mtime, size) of a calibration file: a file rewritten in place must not be served stale from HBM"""
    import os

    try:
        st = os.stat(path)
        return (st.st_mtime_ns, st.st_size)
    except OSError:
        return None


def _caldir_slot(cb, caldir):
    """The CALDIR slot holding this set of files on the calibrator's context, uploading it first if need be.  The cache key
    holds the paths AND each file's (mtime, size); a slot is taken from the top (31 downwards) only if nobody owns it or its
    owner is one of this cache's own, older entries -- a slot a caller loaded explicitly (``Calibrator.load_caldir``) is never
    replaced silently."""
    files = tuple(sorted((k, str(v), _file_stamp(str(v))) for k, v in caldir.items() if isinstance(v, str)))
    key = (id(cb.ctx), files)
    slot = _cal_cache.get(key)
    if slot is not None and cb.slot_owner(slot) == key:   # still ours
        return slot
    for cand in range(31, 7, -1):   # from the top: low slot numbers are left to explicit load_caldir calls
        if cand not in cb.shapes:    # never loaded on this context
            slot = cand
            break
    else:
        # every slot is taken: evict the oldest entry of this cache (never an explicitly loaded slot)
        slot = None
        for k_old, s_old in list(_cal_cache.items()):
            if k_old[0] == id(cb.ctx) and cb.slot_owner(s_old) == k_old:
                slot = s_old
                del _cal_cache[k_old]
                break
        if slot is None:
            raise RuntimeError("no free CALDIR slot: slots 8..31 are all held by explicit load_caldir calls")
    cb.load_caldir(slot, load_caldir_arrays(caldir), owner=key)
    _cal_cache[key] = slot
    return slot


def calibrateimage(config, verbose=True, calibrator=None):
    """Run the calibrations specified by ``config`` (dict, normally from YAML) and write the L2 file.  Beyond the reference's
    surface: ``config["IN"]`` may be an L1 tree (dict) instead of a path, and with ``config["OUT"] = None`` the L2 tree is
    returned instead of written (the noise-layer driver keeps its intermediate exposures in memory that way)."""
    for unsupported in ("romancal_ramp_fit", "correct_wfi18_transient"):
        if config.get(unsupported):
            raise NotImplementedError(f"{unsupported} is outside the GPU L1->L2 path of this package")
    if config.get("FITSOUT"):
        raise NotImplementedError("FITSOUT needs astropy, which this package does not depend on")
    mylog = processlog.ProcessLog()
    caldir = config["CALDIR"]
    if "dark_decay" in caldir:
        raise NotImplementedError("dark_decay is a romancal step outside the GPU L1->L2 path of this package")
    backup = config.get("SATURATION_BACKUP", 1)

    ramp, meta = initializationstep(config, caldir, mylog)
    meta["nborder"] = pars.nborder
    nb = pars.nborder
    mylog.append("Initialized data\n")

    cb = calibrator or pipeline.Calibrator()
    slot = _caldir_slot(cb, caldir)
    # saturation flagging (gen_cal_image.py:148-185): on the device, inside the calibrate call
    if "saturation" not in caldir:
        raise KeyError("saturation")  # the reference opens caldir["saturation"] unconditionally (:174)
    sat_on_device = True
    mylog.append("Saturation check on the device\n")
    exclude_first = config.get("EXCLUDE_FIRST", True)
    area = None
    if "AREAFACTOR" in config:
        with calio.open_tree(config["AREAFACTOR"]) as f:
            area = np.asarray(f["roman"]["data"], dtype=np.float64)
    ramp["read_pattern"], ramp["frame_time"] = meta["read_pattern"], meta["frame_time"]
    if sat_on_device:
        ramp["groupdq"] = None  # zeros + DO_NOT_USE on the first group: made on the device
    res = cb.calibrate(slot, ramp, exclude_first=exclude_first, ramp_opt_pars=config.get("RAMP_OPT_PARS"),
                       jump_pars=config.get("JUMP_DETECT_PARS"), area_factor=area, flag_saturation=sat_on_device, saturation_read_pattern=True,
                       saturation_backup=backup)
    K = res["K"]
    uopt = config.get("RAMP_OPT_PARS", planmod.DEFAULT_RAMP_OPT_PARS)
    mylog.append(f"\n\nRamp fit optimized for u = {planmod.ramp_opt_u(uopt):11.5E} s**-1\n")
    mylog.append(f"weights = {K}\n")
    mylog.append("Reference pixels, bias, linearity, IPC, ramp fit, dark current, flat: complete (GPU)\n")

    slope, pdq, rdq = res["slope"], res["pixeldq"], res["groupdq"]
    err_read, err_poisson = res["err_read"], res["err_poisson"]
    with calio.open_tree(caldir["gain"]) as g_:
        gain_plane = np.asarray(g_["roman"]["data"])
    if gain_plane.dtype == np.float32 and gain_plane.ndim == 2:
        medgain = float(sky.block_nanmedians(gain_plane, 1, ctx=cb.ctx)[0, 0])  # = np.median for a NaN-free f32 plane
    else:
        medgain = float(np.median(gain_plane))
    mylog.append(f"median gain = {medgain:8.5f} e/DN\n")

    # sky information (gen_cal_image.py:639-651): grown mask, mode of the 4x4-binned unmasked image, optional low-order
    # sky model subtraction -- all on the GPU (utils/maskhandling.py, utils/sky.py)
    act = (slice(nb, -nb), slice(nb, -nb))
    m = maskhandling.PixelMask1.build(pdq, ctx=cb.ctx)
    medsky, _ = sky.smooth_mode(sky.binkxk(slope, 4, mask=m, ctx=cb.ctx), ctx=cb.ctx)
    medsky = float(medsky)
    data_act = np.ascontiguousarray(slope[act])
    data_withsky = data_act.copy()
    if "SKYORDER" in config:
        skyorder = int(config["SKYORDER"])
        skycoefs, _ = sky.medfit(data_act, order=skyorder, subtract=True, ctx=cb.ctx)
        skycoefs = np.asarray(skycoefs)
    else:
        skycoefs = np.array([]).astype(np.float32)
        skyorder = -1  # not used

    var_r, var_p = err_read[act] ** 2, err_poisson[act] ** 2
    im2 = {
        "meta": ramp["meta"],
        "data": data_act,
        "dq": pdq[act].copy(),
        "var_poisson": var_p,
        "var_rnoise": var_r,
        "var_flat": np.zeros_like(var_r),
        "err": np.sqrt(var_r + var_p),
        "data_withsky": data_withsky,
    }
    if ramp["amp33"] is not None:
        im2["amp33"] = ramp["amp33"]
    processinfo = {
        "medsky": medsky, "medgain": medgain, "skyorder": skyorder, "skycoefs": skycoefs,
        "ramp_opt_pars": dict(uopt), "weights": K, "log": mylog.output,
        "config": {k: v for k, v in config.items() if not (k == "IN" and isinstance(v, dict))},
        "exclude_first": bool(exclude_first),
        # the full meta, read_pattern included (a list of lists): the reference's noise driver reads it from the L2 file
        # (gen_noise_image.py:208-212), so files written here can be fed to it
        "meta": {k: ([list(map(int, g)) for g in v] if k == "read_pattern" else v) for k, v in meta.items()},
    }
    if config.get("SLICEOUT"):
        endslice = sky.endslice(rdq, nb, ctx=cb.ctx)  # raises ValueError("too many groups") for >= 128 groups
        processinfo["endslice"] = endslice
    if verbose:
        print(mylog.output)
    if config.get("OUT") is None:   # in-memory use (the noise-layer driver): the L2 tree instead of a file
        return {"roman": im2, "processinfo": processinfo}
    calio.write_asdf(config["OUT"], {"roman": im2, "processinfo": processinfo})
    return


if __name__ == "__main__":
    import yaml

    with open(sys.argv[1]) as f:
        calibrateimage(yaml.safe_load(f))
