"""Array-level driver of the L1->L2 chain on one GPU.

``Calibrator`` owns a context, the device-resident CALDIR sets (one slot per SCA) and the ramp-fit
plans; ``calibrate`` runs the chain of ``calibrateimage`` (``gen_cal_image.py:531-629``) on numpy
arrays (host) or on device pointers (e.g. torch tensors' ``data_ptr()``), through ``rip_calibrate``.
"""


import numpy as np

from . import _native, pars, plan as planmod
from ._native import (STAGE_ALL, STAGE_BIAS, STAGE_DARK, STAGE_FLAT, STAGE_IPC, STAGE_LIN, STAGE_RAMPFIT,  # noqa: F401
                      STAGE_REFPIX)


def read_pattern_dilution(read_pattern):
    """mean(read indices) / last read index per group (f64; 0/0 = NaN for a group holding only read 0: never flags): the
    factor stcal applies to the saturation threshold of groups that average several reads (rip_ramp_desc::sat_dilution)."""
    with np.errstate(all="ignore"):
        return np.ascontiguousarray([np.float64(np.mean(r)) / np.float64(r[-1]) for r in read_pattern], dtype=np.float64)


class _Block:
    """Memory of a result array that goes back to its pool when the last array (or view of one) that refers to it is gone."""

    def __init__(self, pool, mem):
        self._pool, self._mem = pool, mem
        self.__array_interface__ = {"shape": mem.shape, "typestr": "|u1", "data": (mem.ctypes.data, False), "version": 3}

    def __del__(self):
        try:
            self._pool._give_back(self._mem)
        except Exception:   # interpreter shutdown
            pass


class ResultPool:
    """Result arrays made of memory that has been touched before.  A fresh ``np.empty`` of the four result planes and the group
    flags of a 4096 x 4096 x 8 ramp (0.4 GB) costs 100 000 page faults when the copy from the device first writes it: 33 ms on top
    of the 16 ms the calibration takes (bench.py, ``host_path``).  Arrays from this pool are ordinary numpy arrays that own their
    memory through a base object; when the caller drops the last reference the memory returns here and serves the next call.  At
    most ``keep`` spare blocks per size are held."""

    def __init__(self, keep=12):
        self.keep, self.free = keep, {}

    def _give_back(self, mem):
        spare = self.free.setdefault(mem.nbytes, [])
        if len(spare) < self.keep:
            spare.append(mem)

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if nbytes < (1 << 20):
            return np.empty(shape, dtype)
        spare = self.free.get(nbytes)
        mem = spare.pop() if spare else np.empty(nbytes, np.uint8)
        return np.asarray(_Block(self, mem)).view(dtype).reshape(shape)


class Calibrator:
    def __init__(self, device=None, ctx=None):
        self.ctx = ctx if ctx is not None else _native.default_context(device)
        # CALDIR slots and plans belong to the context: every Calibrator on it sees them
        self.shapes = self.ctx.__dict__.setdefault("_caldir_shapes", {})
        self._plans = self.ctx.__dict__.setdefault("_plan_cache", {})
        self._results = self.ctx.__dict__.setdefault("_result_pool", ResultPool())

    # ---- CALDIR ---------------------------------------------------------------------------
    def load_caldir(self, slot, cal, nborder=pars.nborder, owner=None):
        """Upload one SCA's calibration arrays (dict of dicts, ``roman`` branch layout) into ``slot``.  ``owner``: tag kept
        with the slot (``slot_owner``) so that a cache of file-based CALDIR sets can tell when its slot was reused."""
        rslope = planmod.refout_slope(cal["read"])
        self.shapes[slot] = self.ctx.upload_caldir(slot, cal, nborder=nborder, refout_slope=rslope)
        self.ctx.__dict__.setdefault("_caldir_owner", {})[slot] = owner
        return self.shapes[slot]

    def slot_owner(self, slot):
        return self.ctx.__dict__.get("_caldir_owner", {}).get(slot)

    # ---- plans ----------------------------------------------------------------------------
    def plan_for(self, read_pattern, frame_time, exclude_first=True, ramp_opt_pars=None, jump_pars=None):
        """(plan id, meta) for an MA table; cached per configuration."""
        key = (repr(read_pattern), float(frame_time), bool(exclude_first), repr(ramp_opt_pars), repr(jump_pars))
        if key in self._plans:
            return self._plans[key]
        meta = planmod.exposure_meta(read_pattern, frame_time)
        meta["nborder"] = pars.nborder
        meta["K"] = planmod.construct_weights(planmod.ramp_opt_u(ramp_opt_pars), meta, exclude_first)
        if jump_pars:
            meta["jump_detect_pars"] = jump_pars
        desc = planmod.plan_desc(meta, meta["K"], exclude_first, list(read_pattern[0]) == [0], jump_pars)
        pid = self.ctx.create_plan(desc)
        self._plans[key] = (pid, meta)
        return pid, meta

    # ---- host arrays in, host arrays out --------------------------------------------------
    def calibrate(self, slot, ramp, exclude_first=True, ramp_opt_pars=None, jump_pars=None, area_factor=None,
                  stages=STAGE_ALL, want_groupdq=True, want_cube=False, channel_lines=None, flag_saturation=False,
                  saturation_backup=1, saturation_skip_firstn=1, out=None, saturation_read_pattern=False):
        """Run the chain on one ramp given as numpy arrays.

        ``ramp``: dict(data u16|f32 (G,ny,nx), amp33 u16 (G,ny,128)|None, groupdq u8, pixeldq u32,
        read_pattern, frame_time).  Returns dict(slope, err_read, err_poisson, pixeldq[, groupdq][, cube], K, meta).

        ``flag_saturation``: dq-init + saturation flagging on the device before the chain (the CALDIR slot must hold
        ``saturation``); ``ramp["groupdq"]`` may then be None and ``ramp["pixeldq"]`` is the mask dq.
        ``saturation_read_pattern``: compare groups of several reads with threshold * mean(reads) / last read (the read-pattern
        rule of stcal the reference's call enables, gen_cal_image.py:172-185).

        ``out``: optional dict of preallocated C-contiguous result arrays (any of slope, err_read, err_poisson f32 (ny,nx),
        pixeldq u32 (ny,nx), groupdq u8 (G,ny,nx), cube f32 (G,ny,nx)) that are filled instead of new ones.  With page-locked
        arrays (``Calibrator.pinned_empty``) on both sides the copies run at PCIe rate (tools/gpu_checks/host_path_timing.py).
        """
        ny, nx = self.shapes[slot]
        pid, meta = self.plan_for(ramp["read_pattern"], ramp["frame_time"], exclude_first, ramp_opt_pars, jump_pars)
        data = np.ascontiguousarray(ramp["data"])
        if data.dtype not in (np.uint16, np.float32):
            data = data.astype(np.float32)
        G = data.shape[0]
        if data.shape != (G, ny, nx):
            raise ValueError(f"ramp shape {data.shape} does not match the CALDIR frame {(ny, nx)}")
        gdq = None
        if ramp.get("groupdq") is not None:
            gdq = np.ascontiguousarray(ramp["groupdq"], dtype=np.uint8)   # (DO_NOT_USE on an excluded first group: set on the
            #                                                              library's device copy, rd.or_first_group below)
        elif not flag_saturation:
            raise ValueError("ramp['groupdq'] is required unless flag_saturation is set")
        pdq = np.ascontiguousarray(ramp["pixeldq"], dtype=np.uint32)
        amp33 = None if ramp.get("amp33") is None else np.ascontiguousarray(ramp["amp33"], dtype=np.uint16)
        area = None if area_factor is None else np.ascontiguousarray(area_factor, dtype=np.float64)
        lines = None if channel_lines is None else np.ascontiguousarray(channel_lines, dtype=np.float64)

        rd = _native.RampDesc()
        rd.location, rd.ngrp = _native.RIP_HOST, G
        rd.data, rd.data_dtype = data.ctypes.data, _native.dtype_code(data)
        rd.amp33 = None if amp33 is None else amp33.ctypes.data
        rd.groupdq, rd.pixeldq = (None if gdq is None else gdq.ctypes.data), pdq.ctypes.data
        rd.flag_saturation = 1 if flag_saturation else 0
        rd.or_first_group = 1 if (exclude_first and gdq is not None) else 0   # gen_cal_image.py:142-143, the caller's array untouched
        rd.sat_backup, rd.sat_skip_firstn = int(saturation_backup), int(saturation_skip_firstn)
        dil = read_pattern_dilution(ramp["read_pattern"]) if (flag_saturation and saturation_read_pattern) else None
        rd.sat_dilution = None if dil is None else dil.ctypes.data
        rd.area_factor = None if area is None else area.ctypes.data
        rd.channel_lines = None if lines is None else lines.ctypes.data

        given = out

        def result(name, shape, dtype):
            a = None if given is None else given.get(name)
            if a is None:
                return self._results.empty(shape, dtype)
            if a.shape != shape or a.dtype != dtype or not a.flags.c_contiguous:
                raise ValueError(f"out[{name!r}] must be a C-contiguous {np.dtype(dtype).name} array of shape {shape}")
            return a

        res = {
            "slope": result("slope", (ny, nx), np.float32), "err_read": result("err_read", (ny, nx), np.float32),
            "err_poisson": result("err_poisson", (ny, nx), np.float32), "pixeldq": result("pixeldq", (ny, nx), np.uint32),
        }
        out = _native.Outputs()
        out.location = _native.RIP_HOST
        out.slope, out.err_read = res["slope"].ctypes.data, res["err_read"].ctypes.data
        out.err_poisson, out.pixeldq = res["err_poisson"].ctypes.data, res["pixeldq"].ctypes.data
        if want_groupdq:
            res["groupdq"] = result("groupdq", (G, ny, nx), np.uint8)
            out.groupdq = res["groupdq"].ctypes.data
        if want_cube:
            res["cube"] = result("cube", (G, ny, nx), np.float32)
            out.cube = res["cube"].ctypes.data
        self.ctx.calibrate_raw(slot, pid, stages, rd, out)
        if not (stages & STAGE_RAMPFIT):
            for k in ("slope", "err_read", "err_poisson"):
                res.pop(k)
            res.pop("groupdq", None)
        res["K"], res["meta"] = meta["K"], meta
        return res

    # ---- a batch of ramps in host memory, pipelined over PCIe --------------------------------
    def calibrate_many(self, slot, ramps, exclude_first=True, ramp_opt_pars=None, jump_pars=None, want_groupdq=False,
                       flag_saturation=False, saturation_backup=1, saturation_skip_firstn=1, out=None, saturation_read_pattern=False):
        """Run the whole chain on a list of ramps (dicts as for ``calibrate``; same read pattern, frame time and data
        dtype) with ``rip_calibrate_batch``: upload, chain and download of consecutive ramps overlap.  ``out``: optional
        list of dicts of preallocated result arrays (see ``calibrate``).  Returns the list of result dicts.  Page-locked
        arrays (``pinned_empty``) on both sides give the full PCIe rate.  ``saturation_read_pattern``: as in ``calibrate``
        (the read-pattern rule of the saturation step; ``calibrateimage`` and the harness switch it on)."""
        ramps = list(ramps)
        if not ramps:
            return []
        ny, nx = self.shapes[slot]
        pid, meta = self.plan_for(ramps[0]["read_pattern"], ramps[0]["frame_time"], exclude_first, ramp_opt_pars, jump_pars)
        n = len(ramps)
        # the read-pattern rule of the saturation step, as calibrateimage applies it (gen_cal_image.py:172-185)
        dil = read_pattern_dilution(ramps[0]["read_pattern"]) if (flag_saturation and saturation_read_pattern) else None
        descs, outs, results, keep = (_native.RampDesc * n)(), (_native.Outputs * n)(), [], []
        for i, ramp in enumerate(ramps):
            if list(map(list, ramp["read_pattern"])) != list(map(list, ramps[0]["read_pattern"])):
                raise ValueError(f"ramp {i}: read pattern differs from ramp 0")
            data = np.ascontiguousarray(ramp["data"])
            if data.dtype not in (np.uint16, np.float32):
                data = data.astype(np.float32)
            G = data.shape[0]
            if data.shape != (G, ny, nx):
                raise ValueError(f"ramp {i}: shape {data.shape} does not match the CALDIR frame {(ny, nx)}")
            gdq = None
            if ramp.get("groupdq") is not None:
                gdq = np.ascontiguousarray(ramp["groupdq"], dtype=np.uint8)   # (DO_NOT_USE on an excluded first group: rd.or_first_group)
            elif not flag_saturation:
                raise ValueError(f"ramp {i}: groupdq is required unless flag_saturation is set")
            pdq = np.ascontiguousarray(ramp["pixeldq"], dtype=np.uint32)
            amp33 = None if ramp.get("amp33") is None else np.ascontiguousarray(ramp["amp33"], dtype=np.uint16)
            keep.append((data, gdq, pdq, amp33))
            rd = descs[i]
            rd.location, rd.ngrp = _native.RIP_HOST, G
            rd.data, rd.data_dtype = data.ctypes.data, _native.dtype_code(data)
            rd.amp33 = None if amp33 is None else amp33.ctypes.data
            rd.groupdq, rd.pixeldq = (None if gdq is None else gdq.ctypes.data), pdq.ctypes.data
            rd.flag_saturation = 1 if flag_saturation else 0
            rd.or_first_group = 1 if (exclude_first and gdq is not None) else 0   # gen_cal_image.py:142-143 on the device copy
            rd.sat_backup, rd.sat_skip_firstn = int(saturation_backup), int(saturation_skip_firstn)
            if dil is not None:
                rd.sat_dilution = dil.ctypes.data
            given = None if out is None else out[i]

            def result(name, shape, dtype, given=given):
                a = None if given is None else given.get(name)
                if a is None:
                    return self._results.empty(shape, dtype)
                if a.shape != shape or a.dtype != dtype or not a.flags.c_contiguous:
                    raise ValueError(f"out[{i}][{name!r}] must be a C-contiguous {np.dtype(dtype).name} array of shape {shape}")
                return a

            res = {"slope": result("slope", (ny, nx), np.float32), "err_read": result("err_read", (ny, nx), np.float32),
                   "err_poisson": result("err_poisson", (ny, nx), np.float32), "pixeldq": result("pixeldq", (ny, nx), np.uint32)}
            od = outs[i]
            od.location = _native.RIP_HOST
            od.slope, od.err_read = res["slope"].ctypes.data, res["err_read"].ctypes.data
            od.err_poisson, od.pixeldq = res["err_poisson"].ctypes.data, res["pixeldq"].ctypes.data
            if want_groupdq:
                res["groupdq"] = result("groupdq", (G, ny, nx), np.uint8)
                od.groupdq = res["groupdq"].ctypes.data
            res["K"], res["meta"] = meta["K"], meta
            results.append(res)
        self.ctx.check(self.ctx.lib.rip_calibrate_batch(self.ctx.h, int(slot), int(pid), int(STAGE_ALL), n, descs, outs))
        del keep
        return results

    # ---- device pointers in, device pointers out (asynchronous) ---------------------------
    def calibrate_device(self, slot, plan_id, ngrp, data_ptr, data_is_u16, amp33_ptr, groupdq_ptr, pixeldq_ptr,
                         slope_ptr, err_read_ptr, err_poisson_ptr, pixeldq_out_ptr, groupdq_out_ptr=None,
                         area_ptr=None, stages=STAGE_ALL, flag_saturation=False, saturation_backup=1,
                         saturation_skip_firstn=1, read_pattern=None, inputs_complete=False, ready_event=None):
        """``groupdq_ptr`` may be None with ``flag_saturation`` (dq-init + saturation flagging on the device).
        ``read_pattern``: with ``flag_saturation``, the exposure's read pattern for the read-pattern rule of the saturation
        step (groups averaging several reads are compared with a diluted threshold, as ``calibrateimage`` does: the reference
        hands the pattern to stcal at gen_cal_image.py:172-185); None = every group against the full threshold.
        Ordering of the inputs (``rip_ramp_desc::inputs_ready`` / ``ready_event``): by default the call is ordered behind
        everything queued on the context's stream; ``inputs_complete=True`` = the caller vouches that the device arrays are
        complete now (keeps the overlap of the reference-pixel pre-pass with the previous call's kernel); ``ready_event`` =
        a raw ``hipEvent_t`` (e.g. ``torch.cuda.Event.cuda_event``) recorded behind the work that writes them."""
        rd = _native.RampDesc()
        rd.location, rd.ngrp = _native.RIP_DEVICE, int(ngrp)
        rd.data, rd.data_dtype = data_ptr, (_native.RIP_U16 if data_is_u16 else _native.RIP_F32)
        rd.amp33, rd.groupdq, rd.pixeldq, rd.area_factor = amp33_ptr, groupdq_ptr, pixeldq_ptr, area_ptr
        rd.flag_saturation = 1 if flag_saturation else 0
        rd.sat_backup, rd.sat_skip_firstn = int(saturation_backup), int(saturation_skip_firstn)
        dil = read_pattern_dilution(read_pattern) if (flag_saturation and read_pattern is not None) else None
        if dil is not None:
            rd.sat_dilution = dil.ctypes.data   # host pointer, read during the call
        rd.inputs_ready = _native.RIP_INPUTS_COMPLETE if inputs_complete else _native.RIP_INPUTS_STREAM_ORDERED
        rd.ready_event = ready_event
        out = _native.Outputs()
        out.location = _native.RIP_DEVICE
        out.slope, out.err_read, out.err_poisson = slope_ptr, err_read_ptr, err_poisson_ptr
        out.pixeldq, out.groupdq = pixeldq_out_ptr, groupdq_out_ptr
        self.ctx.calibrate_raw(slot, plan_id, stages, rd, out)

    def pinned_empty(self, shape, dtype):
        """A page-locked numpy array: host buffers the library copies from / to at PCIe rate."""
        return self.ctx.pinned_empty(shape, dtype)

    def synchronize(self):
        self.ctx.synchronize()


def lapack_channel_lines(bottom_top, nrows):
    """(m, c) of the line through (1.5, bottom), (nrows-2.5, top) from LAPACK's least squares, exactly as
    ``reference_subtraction.py:57-60`` obtains it; ``bottom_top`` (..., 2) f32 -> (..., 2) f64."""
    bt = np.asarray(bottom_top)
    A = np.vstack([(1.5, nrows - 2.5), np.ones(2)]).T
    out = np.zeros(bt.shape, dtype=np.float64)
    flat_in, flat_out = bt.reshape(-1, 2), out.reshape(-1, 2)
    for i in range(flat_in.shape[0]):
        flat_out[i] = np.linalg.lstsq(A, (flat_in[i, 0], flat_in[i, 1]), rcond=None)[0]
    return out
