"""ctypes binding of ``libromanhip.so`` (C-ABI: ``include/romanhip.h``).

This is the only door between the Python host code and the HIP kernels.  There is no CPU
fallback: if the library is missing or no MI355X is visible, the calls raise.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROMANHIP_LIB") or os.path.join(_HERE, "libromanhip.so")   # the env var: A/B timing of library variants

RIP_MAX_GROUPS = 64
RIP_TIMING_BUILD_FLAG = 1000000   # include/romanhip.h
RIP_F32, RIP_F64, RIP_U16 = 0, 1, 2
RIP_HOST, RIP_DEVICE = 0, 1
RIP_INPUTS_STREAM_ORDERED, RIP_INPUTS_COMPLETE = 0, 1

STAGE_REFPIX, STAGE_BIAS, STAGE_LIN, STAGE_IPC, STAGE_RAMPFIT, STAGE_DARK, STAGE_FLAT = (1 << i for i in range(7))
STAGE_ALL = 0x7F

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)
c_u16_p = C.POINTER(C.c_uint16)
c_u32_p = C.POINTER(C.c_uint32)


class CaldirDesc(C.Structure):
    _fields_ = [
        ("ny", C.c_int32), ("nx", C.c_int32), ("nborder", C.c_int32),
        ("ngrp_dark", C.c_int32), ("dark_data", c_float_p), ("dark_slope", c_float_p), ("dark_dq", c_u32_p),
        ("read_noise", c_float_p), ("amp33_med", c_float_p), ("refout_slope", C.c_double),
        ("gain", C.c_void_p), ("gain_dtype", C.c_int32),
        ("lin_nplanes", C.c_int32), ("lin_coefs", c_float_p),
        ("lin_smin", c_float_p), ("lin_smax", c_float_p), ("lin_sref", c_float_p), ("lin_dq", c_u32_p),
        ("ipc4d", C.c_void_p), ("ipc_dtype", C.c_int32),
        ("flat", c_float_p),
        ("ngrp_bias", C.c_int32), ("biascorr", c_float_p),
        ("saturation", c_float_p), ("saturation_dq", c_u32_p),
    ]


class PlanDesc(C.Structure):
    _fields_ = [
        ("ngrp", C.c_int32), ("exclude_first", C.c_int32), ("do_not_flag_first", C.c_int32),
        ("tbar", C.c_float * RIP_MAX_GROUPS), ("tau", C.c_float * RIP_MAX_GROUPS),
        ("nreads", C.c_int16 * RIP_MAX_GROUPS), ("K", C.c_float * RIP_MAX_GROUPS),
        ("nvariants", C.c_int32), ("variant_g", C.c_int32 * RIP_MAX_GROUPS),
        ("variant_coef", C.c_float * RIP_MAX_GROUPS), ("variant_rfac", C.c_float * RIP_MAX_GROUPS),
        ("sthresh_a", C.c_double), ("sthresh_b", C.c_double), ("ithresh_a", C.c_double), ("ithresh_b", C.c_double),
    ]


class RampDesc(C.Structure):
    _fields_ = [
        ("location", C.c_int32), ("ngrp", C.c_int32), ("data", C.c_void_p), ("data_dtype", C.c_int32),
        ("amp33", C.c_void_p), ("groupdq", C.c_void_p), ("pixeldq", C.c_void_p), ("area_factor", C.c_void_p),
        ("channel_lines", C.c_void_p),
        ("flag_saturation", C.c_int32), ("sat_backup", C.c_int32), ("sat_skip_firstn", C.c_int32),
        ("sat_dilution", C.c_void_p),
        ("inputs_ready", C.c_int32), ("ready_event", C.c_void_p), ("or_first_group", C.c_int32),
    ]


class Outputs(C.Structure):
    _fields_ = [
        ("location", C.c_int32), ("slope", C.c_void_p), ("err_read", C.c_void_p), ("err_poisson", C.c_void_p),
        ("pixeldq", C.c_void_p), ("groupdq", C.c_void_p), ("cube", C.c_void_p),
    ]


class SynthCal(C.Structure):   # rip_synth_cal: DEVICE pointers
    _fields_ = [
        ("ny", C.c_int32), ("nx", C.c_int32), ("nb", C.c_int32), ("channelwidth", C.c_int32), ("nplanes", C.c_int32),
        ("gain_dtype", C.c_int32), ("ipc_dtype", C.c_int32), ("amp33_valid", C.c_int32),
        ("gain", C.c_void_p), ("read_noise", C.c_void_p), ("resetnoise", C.c_void_p), ("dark_slope", C.c_void_p),
        ("dark", C.c_void_p), ("lin_coefs", C.c_void_p), ("smin", C.c_void_p), ("smax", C.c_void_p), ("ipc4d", C.c_void_p),
        ("biascorr", C.c_void_p), ("tbias", C.c_double), ("amp33_med", C.c_void_p), ("amp33_std", C.c_void_p),
        ("m_pink", C.c_double), ("ru_pink", C.c_double), ("u_pink", C.c_double), ("c_pink", C.c_double),
    ]


# every symbol include/romanhip.h declares: name -> (restype, argtypes)
_VP = C.c_void_p
_I = C.c_int
SYMBOLS = {
    "rip_version": (_I, []),
    "rip_ctx_create": (_I, [_I, C.POINTER(_VP)]),
    "rip_ctx_destroy": (None, [_VP]),
    "rip_last_error": (C.c_char_p, [_VP]),
    "rip_synchronize": (_I, [_VP]),
    "rip_stream": (_VP, [_VP]),
    "rip_host_alloc": (_VP, [_VP, C.c_size_t]),
    "rip_host_free": (None, [_VP, _VP]),
    "rip_caldir_upload": (_I, [_VP, _I, C.POINTER(CaldirDesc)]),
    "rip_caldir_drop": (_I, [_VP, _I]),
    "rip_plan_create": (_I, [_VP, C.POINTER(PlanDesc), C.POINTER(_I)]),
    "rip_plan_destroy": (_I, [_VP, _I]),
    "rip_calibrate": (_I, [_VP, _I, _I, C.c_uint, C.POINTER(RampDesc), C.POINTER(Outputs)]),
    "rip_calibrate_batch": (_I, [_VP, _I, _I, C.c_uint, _I, C.POINTER(RampDesc), C.POINTER(Outputs)]),
    "rip_calibrate_batch_completed": (_I, [_VP]),
    "rip_stage_pearson": (_I, [_VP, C.c_size_t, _VP, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_uint32, _VP, _VP, _VP]),
    "rip_stage_refpix_image": (_I, [_VP, _VP, _I, _I, C.c_double, _I, _I, _VP, _VP, _VP, _VP]),
    "rip_stage_refpix_row": (_I, [_VP, _VP, _I, _I, _I, _I, _I, C.c_double, _VP, _VP, _VP]),
    "rip_stage_refpix_channel": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _VP, _VP]),
    "rip_stage_refpix_tables": (_I, [_VP, _VP, _I, _VP, _VP, _VP, C.c_double, _I, _I, _I, _I, _VP, _VP, _VP]),
    "rip_synth_frames_ahead": (_I, [_VP, _I, _I, _I, C.c_uint64]),
    "rip_stage_jump_detect": (_I, [_VP, _I, _VP, _VP, _I, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "rip_stage_multilin": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "rip_stage_ipc_image": (_I, [_VP, _I, _I, _VP, _I, _I, _I, _VP, _I, _VP, _I, _VP]),
    "rip_stage_correct_cube": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _I, _VP, _I]),
    "rip_stage_ramp_fit": (_I, [_VP, _I, _VP, _VP, _VP, _I, _I, _I, _VP, _I, _VP, _VP, _VP, _VP]),
    "rip_stage_get_flat": (_I, [_VP, _VP, _I, _I, _I, _VP, _I, _VP, _I, _I, _VP, _VP]),
    "rip_stage_build_mask": (_I, [_VP, _VP, _I, _I, _VP, _VP]),
    "rip_stage_endslice": (_I, [_VP, _VP, _I, _I, _I, _I, _VP]),
    "rip_stage_bin_mean": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "rip_stage_select_ranks": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "rip_stage_gauss_hist": (_I, [_VP, _VP, C.c_int64, _VP, _I, C.c_double, _VP]),
    "rip_stage_legendre2d": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _I, _VP]),
    "rip_stage_invlinearity": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _VP]),
    "rip_stage_noise_inject": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, C.c_uint64, C.c_uint32, _VP]),
    "rip_stage_noise_1f": (_I, [_VP, _I, _I, _I, _VP, C.c_uint64, C.c_uint32, _VP]),
    "rip_stage_poisson_resample": (_I, [_VP, _VP, _VP, _I, C.c_size_t, C.c_double, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I,
                                        C.c_uint64, C.c_uint32, _VP]),
    "rip_stats_l1_diff": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "rip_stats_l2_pack": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "rip_stats_reduce": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP]),
    "rip_synth_apportion": (_I, [_VP, _VP, _I, _I, _I, _I, _VP, C.c_uint64, _VP]),
    "rip_synth_resultants": (_I, [_VP, C.POINTER(SynthCal), _I, _VP, _VP, _VP, _VP, C.c_uint64, _VP, _VP, _VP]),
    "rip_synth_fill": (_I, [_VP, C.POINTER(SynthCal), _I, _VP, _I, _VP, _VP, _VP, C.c_uint64, _VP, _VP]),
    "rip_synth_noise_1f": (_I, [_VP, _I, _I, _I, C.c_uint64, C.c_uint32, _VP]),
    "rip_synth_extract_ref": (_I, [_VP, _VP, _I, C.c_size_t, _I, _VP]),
    "rip_set_option_f64": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "rip_set_option": (_I, [_VP, C.c_char_p, _I]),
    "rip_last_chain_form": (_I, [_VP]),
    "rip_profile_enable": (_I, [_VP, _I]),
    "rip_profile_read": (_I, [_VP, C.POINTER(C.c_double), C.POINTER(_I)]),
}

_lib = None


def load_library():
    """dlopen the in-tree library and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C romanimpreprocess_amd/csrc`).  romanimpreprocess_amd has no CPU fallback."
        )
    # PyTorch (used by the device-resident mirrors for device memory) ships its own HIP runtime under the same soname: whichever
    # of the two is loaded first serves both, and only PyTorch's own copy leaves PyTorch with its devices.  So it goes first,
    # whenever it is installed; without it the library runs on the system's runtime alone.
    try:
        import torch  # noqa: F401, PLC0415
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    ver = lib.rip_version()
    if ver >= RIP_TIMING_BUILD_FLAG and os.environ.get("ROMANHIP_ALLOW_TIMING_BUILD") != "1":
        raise RuntimeError(f"{LIB_PATH} is a timing build (compiled with -DRIP_TIMING_BUILD: its kernels may skip phases, results "
                           "invalid by construction); set ROMANHIP_ALLOW_TIMING_BUILD=1 to load it for timing experiments")
    _lib = lib
    return lib


def dtype_code(arr):
    if arr.dtype == np.float32:
        return RIP_F32
    if arr.dtype == np.float64:
        return RIP_F64
    if arr.dtype == np.uint16:
        return RIP_U16
    raise TypeError(f"unsupported dtype {arr.dtype}")


def ptr(arr):
    """address of a C-contiguous numpy array (the caller keeps it alive), or None."""
    if arr is None:
        return None
    if not arr.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return arr.ctypes.data


def _c(arr, dtype=None):
    """C-contiguous view/copy with the given dtype (None keeps the dtype)."""
    if arr is None:
        return None
    return np.ascontiguousarray(arr, dtype=dtype)


class Context:
    """One rip_ctx (one GPU).  Raises RuntimeError when the library or the GPU is unavailable."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.rip_ctx_create(int(device), C.byref(h))
        if rc != 0:
            msg = self.lib.rip_last_error(None)
            raise RuntimeError(f"rip_ctx_create(device={device}) failed: {msg.decode() if msg else rc}")
        self.h = h
        self.device = int(device)
        self._keep = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.rip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, exc=RuntimeError):
        if rc != 0:
            msg = self.lib.rip_last_error(self.h)
            text = msg.decode() if msg else f"status {rc}"
            raise (ValueError if rc == -1 else exc)(f"libromanhip: {text}")

    def synchronize(self):
        self.check(self.lib.rip_synchronize(self.h))

    def pinned_empty(self, shape, dtype):
        """A numpy array in page-locked host memory (``rip_host_alloc``), freed with the array."""
        import weakref

        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.rip_host_alloc(self.h, nbytes)
        if not p:
            self.check(-2, MemoryError)
        buf = (C.c_ubyte * max(nbytes, 1)).from_address(p)
        arr = np.frombuffer(buf, dtype=np.uint8, count=nbytes).view(dtype).reshape(shape)
        weakref.finalize(buf, self.lib.rip_host_free, self.h, p).atexit = False   # at exit the process returns it
        return arr

    def set_option(self, name, value):
        self.check(self.lib.rip_set_option(self.h, name.encode(), int(value)))

    def set_option_f64(self, name, value):
        """floating-point options of this context ("guard_band")"""
        self.check(self.lib.rip_set_option_f64(self.h, name.encode(), float(value)))

    def last_chain_form(self):
        """0 = stage kernels, 2 = the fused kernel (last calibrate call; 1 and 3 were the general and wave-private fused kernels of rounds 1-2)."""
        return int(self.lib.rip_last_chain_form(self.h))

    def profile(self, on=True):
        self.check(self.lib.rip_profile_enable(self.h, int(bool(on))))

    def profile_read(self):
        """(ms per stage [refpix pre-pass, cube stage, ipc, ramp fit], number of calls) since the last read."""
        ms = (C.c_double * 4)()
        n = C.c_int(0)
        self.check(self.lib.rip_profile_read(self.h, ms, C.byref(n)))
        return list(ms), n.value

    @property
    def stream(self):
        return self.lib.rip_stream(self.h)

    # ---- CALDIR ------------------------------------------------------------------------
    def upload_caldir(self, slot, cal, nborder=4, refout_slope=None):
        """``cal``: dict of dicts of numpy arrays, layout of the ``roman`` branch of each CALDIR file."""
        d = CaldirDesc()
        keep = []

        def P(a, dt, ptype):
            if a is None:
                return None
            a = _c(a, dt)
            keep.append(a)
            return a.ctypes.data_as(ptype)

        gain = _c(cal["gain"]["data"])
        if gain.dtype not in (np.float32, np.float64):
            gain = gain.astype(np.float64)
        keep.append(gain)
        ny, nx = gain.shape
        d.ny, d.nx, d.nborder = ny, nx, nborder
        dark = cal.get("dark")
        if dark is not None:
            if "data" in dark and dark["data"] is not None:
                d.ngrp_dark = dark["data"].shape[0]
                d.dark_data = P(dark["data"], np.float32, c_float_p)
            d.dark_slope = P(dark.get("dark_slope"), np.float32, c_float_p)
            d.dark_dq = P(dark.get("dq"), np.uint32, c_u32_p)
        rd = cal["read"]
        d.read_noise = P(rd["data"], np.float32, c_float_p)
        a33 = rd.get("amp33")
        if a33 is not None:
            d.amp33_med = P(a33["med"], np.float32, c_float_p)
            d.refout_slope = float(refout_slope)
        d.gain = gain.ctypes.data
        d.gain_dtype = dtype_code(gain)
        lin = cal.get("linearitylegendre")
        if lin is not None:
            d.lin_nplanes = lin["data"].shape[0]
            d.lin_coefs = P(lin["data"], np.float32, c_float_p)
            d.lin_smin = P(lin["Smin"], np.float32, c_float_p)
            d.lin_smax = P(lin["Smax"], np.float32, c_float_p)
            d.lin_sref = P(lin["Sref"], np.float32, c_float_p)
            d.lin_dq = P(lin["dq"], np.uint32, c_u32_p)
        if cal.get("ipc4d") is not None:
            k = _c(cal["ipc4d"]["data"])
            if k.dtype not in (np.float32, np.float64):
                k = k.astype(np.float64)
            if k.shape != (3, 3, ny - 2 * nborder, nx - 2 * nborder):
                raise ValueError(f"ipc4d shape {k.shape} does not match frame {ny}x{nx} with border {nborder}")
            keep.append(k)
            d.ipc4d = k.ctypes.data
            d.ipc_dtype = dtype_code(k)
        if cal.get("flat") is not None:
            d.flat = P(cal["flat"]["data"], np.float32, c_float_p)
        if cal.get("biascorr") is not None:
            b = cal["biascorr"]["data"]
            d.ngrp_bias = b.shape[0]
            d.biascorr = P(b, np.float32, c_float_p)
        if cal.get("saturation") is not None:  # only needed for flag_saturation (device dq-init + saturation flagging)
            d.saturation = P(cal["saturation"]["data"], np.float32, c_float_p)
            if cal["saturation"].get("dq") is not None:
                d.saturation_dq = P(cal["saturation"]["dq"], np.uint32, c_u32_p)
        self.check(self.lib.rip_caldir_upload(self.h, int(slot), C.byref(d)))
        return (ny, nx)

    def drop_caldir(self, slot):
        self.check(self.lib.rip_caldir_drop(self.h, int(slot)))

    # ---- plans -------------------------------------------------------------------------
    def create_plan(self, desc):
        pid = C.c_int(-1)
        self.check(self.lib.rip_plan_create(self.h, C.byref(desc), C.byref(pid)))
        return pid.value

    def destroy_plan(self, pid):
        self.check(self.lib.rip_plan_destroy(self.h, int(pid)))

    # ---- the chain ---------------------------------------------------------------------
    def calibrate_raw(self, slot, plan, stages, ramp_desc, outputs):
        self.check(self.lib.rip_calibrate(self.h, int(slot), int(plan), int(stages), C.byref(ramp_desc), C.byref(outputs)))


_default = {}


def default_context(device=None):
    """Process-wide context for ``device`` (default: LOCAL_RANK, else 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]
