"""Level-1 synthesis on the GPU -- the per-pixel functions of the reference's ``from_sim/sim_to_isim.py`` with their names and
argument meaning: ``make_l1_fullcal`` (:163-262), ``noise_1f_frame`` (:265-303), ``fill_in_refdata_and_1f`` (:306-403) and the
``EXTRACT_REF`` block of ``Image2D.simulate`` (:711-730) as ``extract_ref``.  Scene rendering, WCS and the file writers around
them (galsim / romanisim / astropy objects) are outside the hot path.

Two ways in:
  * the reference's function signatures, numpy arrays in and out (``caldir``: dict of file paths as in the YAML, or of trees);
  * ``L1Synth``: the calibration arrays uploaded once, every product left in HBM as torch tensors -- what the
    many-realisations harness uses (256 exposures of 4096 x 4096 x 8 without a byte crossing PCIe).

Random numbers: the reference threads a ``galsim.BaseDeviate`` through these functions; here ``rng`` is an integer seed or a
``numpy.random.Generator`` (one integer is drawn from it per call) and every deviate comes from a counter-based generator
on the device keyed by (seed, plane, pixel).  The distributions are reproduced, not galsim's or numpy's streams; with
deviates handed in (``L1Synth`` methods) the arithmetic is bit-identical to the reference's functions (tests/golden/l1sim.npz).
Cosmic rays and persistence (``romanisim.cr`` / ``romanisim.persistence``, called by romanisim's apportioning loop) are not
injected: passing ``persistence`` raises.
"""

import ctypes as C

import numpy as np

from .. import _native, calio, pars

READ_TIME = 3.04   # seconds per read: romanisim.parameters.read_time, which the reference's read_pattern_to_tij uses


def read_pattern_to_tij(read_pattern, read_time=READ_TIME):
    """Time stamps of the reads of every resultant (``romanisim.l1.read_pattern_to_tij``)."""
    return [read_time * np.array(reads) for reads in read_pattern]


def _seed_of(rng):
    if rng is None:
        raise ValueError("rng must be given (an integer seed or a numpy Generator)")
    if isinstance(rng, (int, np.integer)):
        return int(rng) & (2**64 - 1)
    if hasattr(rng, "integers"):
        return int(rng.integers(0, 2**63 - 1))
    if hasattr(rng, "raw"):   # a galsim deviate, if someone has galsim
        return int(rng.raw())
    raise TypeError(f"cannot derive a seed from {type(rng).__name__}")


def _branch(entry):
    """``roman`` branch of a CALDIR entry given as a path or as a tree / branch dict."""
    if isinstance(entry, dict):
        return entry["roman"] if "roman" in entry else entry
    return calio.roman_branch(entry)


def caldir_arrays(caldir):
    """The arrays the synthesis needs, from a CALDIR dict of paths (or of trees)."""
    cal = {k: _branch(caldir[k]) for k in ("read", "gain", "dark", "linearitylegendre")}
    for k in ("ipc4d", "biascorr"):
        if k in caldir:
            cal[k] = _branch(caldir[k])
    return cal


class L1Synth:
    """Calibration arrays of one CALDIR set resident on the device + the synthesis entry points on device tensors.
    ``apportion`` / ``resultants`` / ``fill`` / ``extract_ref`` are asynchronous on the context's stream, which is NOT torch's and
    does not synchronise with it: tensors handed in must be complete (``torch.cuda.synchronize()`` after the torch operations
    that made them; numpy arrays are copied synchronously), and ``ctx.synchronize()`` comes before reading results with torch or
    dropping inputs.  ``make`` takes care of both for what it allocates."""

    def __init__(self, cal, read_pattern, read_time, ctx=None, nb=pars.nborder, channelwidth=None):
        import torch

        self.torch = torch
        self.ctx = ctx or _native.default_context()
        self.dev = torch.device("cuda", self.ctx.device)
        self.rp = [list(map(int, g)) for g in read_pattern]
        self.read_time = float(read_time)
        self.ngrp = len(self.rp)
        self.nreads = sum(len(g) for g in self.rp)
        self.t_reads = np.ascontiguousarray(np.concatenate(read_pattern_to_tij(self.rp, self.read_time)), dtype=np.float64)
        self.group_count = np.array([len(g) for g in self.rp], dtype=np.int32)
        gain = np.ascontiguousarray(cal["gain"]["data"])
        if gain.dtype not in (np.float32, np.float64):
            gain = gain.astype(np.float64)
        self.ny, self.nx = gain.shape
        self.nb = nb
        self.nya, self.nxa = self.ny - 2 * nb, self.nx - 2 * nb
        # channels are pars.channelwidth columns wide wherever the frame allows it (the calibration chain's geometry: nx / 128
        # channels); otherwise the reference's 32 channels
        self.cw = channelwidth or (pars.channelwidth if self.nx % pars.channelwidth == 0 else self.nx // 32)
        self.nch = self.nx // self.cw
        self._keep = []

        def up(a, dt=None):
            a = np.ascontiguousarray(a, dtype=dt)
            if not a.flags.writeable:   # arrays served from a file's bytes: torch wants to own writable memory
                a = a.copy()
            t = torch.from_numpy(a).to(self.dev)
            self._keep.append(t)
            return t.data_ptr()

        d = _native.SynthCal()
        d.ny, d.nx, d.nb, d.channelwidth = self.ny, self.nx, nb, self.cw
        lin = cal.get("linearitylegendre")   # only the resultants need it; reference pixels / correlated noise do not
        d.gain, d.gain_dtype = up(gain), _native.dtype_code(gain)
        rd = cal["read"]
        d.read_noise, d.resetnoise = up(rd["data"], np.float32), up(rd["resetnoise"], np.float32)
        dark = cal["dark"]
        d.dark_slope = up(dark["dark_slope"], np.float32)
        dd = np.asarray(dark["data"])
        d.dark = up(dd[dd.shape[0] - self.ngrp:], np.float32)
        self.lin_dq = None
        if lin is not None:
            d.nplanes = lin["data"].shape[0]
            d.lin_coefs, d.smin, d.smax = up(lin["data"], np.float32), up(lin["Smin"], np.float32), up(lin["Smax"], np.float32)
            self.lin_dq = np.array(lin["dq"], dtype=np.uint32)
        if cal.get("ipc4d") is not None:
            k = np.ascontiguousarray(cal["ipc4d"]["data"])
            if k.dtype not in (np.float32, np.float64):
                k = k.astype(np.float64)
            if k.shape != (3, 3, self.nya, self.nxa):
                raise ValueError(f"ipc4d shape {k.shape} does not match frame {self.ny}x{self.nx} with border {nb}")
            d.ipc4d, d.ipc_dtype = up(k), _native.dtype_code(k)
        if cal.get("biascorr") is not None:
            b = np.asarray(cal["biascorr"]["data"])
            d.biascorr = up(b[b.shape[0] - self.ngrp:], np.float32)
            d.tbias = float(cal["biascorr"]["t0"])
        a33 = rd.get("amp33") if hasattr(rd, "get") else (rd["amp33"] if "amp33" in rd else None)
        if a33 is not None and bool(a33["valid"]):
            d.amp33_valid = 1
            d.amp33_med, d.amp33_std = up(a33["med"], np.float32), up(a33["std"], np.float32)
            d.m_pink, d.ru_pink = float(a33["M_PINK"]), float(a33["RU_PINK"])
        if "anc" in rd:
            d.u_pink, d.c_pink = float(rd["anc"]["U_PINK"]), float(rd["anc"]["C_PINK"])
        self.desc = d

    # ---- device-tensor entry points ------------------------------------------------------------------------------------
    def _t(self, a, dt):
        if a is None:
            return None
        if self.torch.is_tensor(a):
            assert a.is_contiguous() and a.device == self.dev
            return a
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.dev)

    @staticmethod
    def _p(t):
        return None if t is None else t.data_ptr()

    def apportion(self, counts, seed, poisson=False):
        """(nreads, nya, nxa) int32 tensor: electrons collected up to every read.  ``counts`` (nya, nxa) f32: integers, or with
        ``poisson`` the mean of a Poisson draw made first."""
        torch = self.torch
        c = self._t(counts, np.float32)
        if tuple(c.shape) != (self.nya, self.nxa) or c.dtype != torch.float32:
            raise ValueError(f"counts must be float32 of shape {(self.nya, self.nxa)}")
        out = torch.empty((self.nreads, self.nya, self.nxa), dtype=torch.int32, device=self.dev)
        self.ctx.check(self.ctx.lib.rip_synth_apportion(self.ctx.h, c.data_ptr(), self.nya, self.nxa, int(bool(poisson)), self.nreads,
                                                        self.t_reads.ctypes.data, int(seed), out.data_ptr()))
        self._hold = c
        return out

    def resultants(self, reads_e, seed, normals_reset=None, normals_read=None, want_resultants=False, want_cube=True,
                   want_start=False):
        """dict of device tensors: "cube" (ngrp, ny, nx) int16 holding the u16 bits, "resultants" (ngrp, nya, nxa) f32, "start_e"."""
        torch = self.torch
        if self.lin_dq is None:
            raise KeyError("linearitylegendre")   # built for the reference-pixel / correlated-noise step only
        nr, nd = self._t(normals_reset, np.float32), self._t(normals_read, np.float32)
        out = {}
        if want_resultants:
            out["resultants"] = torch.empty((self.ngrp, self.nya, self.nxa), dtype=torch.float32, device=self.dev)
        if want_cube:
            out["cube"] = torch.empty((self.ngrp, self.ny, self.nx), dtype=torch.int16, device=self.dev)
        if want_start:
            out["start_e"] = torch.empty((self.nya, self.nxa), dtype=torch.float32, device=self.dev)
        self.ctx.check(self.ctx.lib.rip_synth_resultants(
            self.ctx.h, C.byref(self.desc), self.ngrp, self.group_count.ctypes.data, reads_e.data_ptr(), self._p(nr), self._p(nd),
            int(seed), self._p(out.get("start_e")), self._p(out.get("resultants")), self._p(out.get("cube"))))
        self._hold2 = (nr, nd, reads_e)
        return out

    def fill(self, cube, amp33, seed, banding=True, normals=None, frames=None, white33=None):
        """``fill_in_refdata_and_1f`` in place on ``cube`` (ngrp, ny, nx) and ``amp33`` (ngrp, ny, cw) int16 tensors (u16 bits)."""
        n, f, w = self._t(normals, np.float32), self._t(frames, np.float32), self._t(white33, np.float32)
        self.ctx.check(self.ctx.lib.rip_synth_fill(self.ctx.h, C.byref(self.desc), self.ngrp, self.group_count.ctypes.data,
                                                   int(bool(banding)), self._p(n), self._p(f), self._p(w), int(seed), cube.data_ptr(),
                                                   self._p(amp33)))
        self._hold3 = (n, f, w)

    def make(self, counts, seed, poisson=False, banding=True):
        """One exposure: (cube (ngrp, ny, nx), amp33 (ngrp, ny, cw)) int16 device tensors holding the u16 bits."""
        torch = self.torch
        reads_e = self.apportion(counts, seed, poisson)
        if banding:   # the 1/f frames of this exposure's fill: on the second stream beside the resultants (f64 arithmetic), behind
            # the apportioning (bound by HBM like the transforms)
            self.ctx.check(self.ctx.lib.rip_synth_frames_ahead(self.ctx.h, self.ny, self.cw, self.ngrp * (self.nx // self.cw + 2),
                                                               int(seed)))
        cube = self.resultants(reads_e, seed)["cube"]
        amp33 = torch.zeros((self.ngrp, self.ny, self.cw), dtype=torch.int16, device=self.dev)
        torch.cuda.current_stream(self.dev).synchronize()   # torch's zero fill runs on torch's stream, the kernels below on the context's
        self.fill(cube, amp33, seed, banding)
        self.ctx.synchronize()   # the intermediate tensors (read electrons, deviates) may go once the kernels are done
        return cube, amp33

    def extract_ref(self, data, offset=0):
        """``EXTRACT_REF`` on a device tensor (ngrp, ...) int16: (reference read, data[1:]) -- ``data`` is modified in place."""
        ref = self.torch.empty_like(data[0])
        self.ctx.check(self.ctx.lib.rip_synth_extract_ref(self.ctx.h, data.data_ptr(), int(data.shape[0]), int(data[0].numel()),
                                                          int(offset), ref.data_ptr()))
        return ref, data[1:]


# ---- the reference's function surface (numpy in and out) ---------------------------------------------------------------------
def make_l1_fullcal(counts, read_pattern, caldir, rng=None, persistence=None, tstart=None, read_time=None, ctx=None):
    """Resultants (ngrp, na, na) f32 in DN (rounded) and the (ngrp, na, na) u32 dq cube of the linearity file, as the reference's
    ``make_l1_fullcal`` returns them.  ``counts``: (na, na) array of integer electron counts, or an object with ``.array``.
    ``read_time``: seconds per read (default ``READ_TIME``)."""
    if persistence is not None:
        raise NotImplementedError("persistence is a romanisim model outside this package")
    arr = np.asarray(getattr(counts, "array", counts), dtype=np.float32)
    if not np.all(arr == np.round(arr)):
        raise ValueError("apportion_counts_to_resultants expects the counts to be integers!")
    cal = caldir_arrays(caldir)
    s = L1Synth(cal, read_pattern, READ_TIME if read_time is None else read_time, ctx=ctx)
    seed = _seed_of(rng)
    out = s.resultants(s.apportion(arr, seed), seed, want_resultants=True, want_cube=False)
    s.ctx.synchronize()
    nb = s.nb
    dq = np.zeros((s.ngrp, s.nya, s.nxa), dtype=np.uint32)
    dq |= s.lin_dq[None, nb:s.ny - nb, nb:s.nx - nb]
    return out["resultants"].cpu().numpy(), dq


def noise_1f_frame(rng, ctx=None):
    """One (4096, 128) f32 block of 1/f noise, unit variance per logarithmic frequency range."""
    ctx = ctx or _native.default_context()
    out = np.empty((pars.nside, pars.channelwidth), dtype=np.float32)
    ctx.check(ctx.lib.rip_stage_noise_1f(ctx.h, pars.nside, pars.channelwidth, 1, None, _seed_of(rng), 0, out.ctypes.data))
    return out


def fill_in_refdata_and_1f(im, caldir, rng, tij, fill_in_banding=True, amp33=None, ctx=None):
    """Fill the reference pixels of ``im`` (ngrp, ny, nx) u16, add 1/f noise, build the reference output ``amp33`` (ngrp, ny,
    nx/32) u16 when given -- in place, like the reference."""
    import torch

    cal = caldir_arrays(caldir)
    rp = [list(range(len(t))) for t in tij]   # only the number of reads per resultant enters
    s = L1Synth(cal, rp, 1.0, ctx=ctx, channelwidth=None if amp33 is None else int(np.shape(amp33)[-1]))
    if im.dtype != np.uint16 or im.shape != (s.ngrp, s.ny, s.nx):
        raise ValueError(f"im must be uint16 of shape {(s.ngrp, s.ny, s.nx)}")
    cube = torch.from_numpy(np.ascontiguousarray(im).view(np.int16)).to(s.dev)
    a33 = None if amp33 is None else torch.from_numpy(np.ascontiguousarray(amp33).view(np.int16)).to(s.dev)
    s.fill(cube, a33, _seed_of(rng), banding=fill_in_banding)
    s.ctx.synchronize()
    im[...] = cube.cpu().numpy().view(np.uint16)
    if amp33 is not None:
        amp33[...] = a33.cpu().numpy().view(np.uint16)


def extract_ref(im, config):
    """The ``EXTRACT_REF`` block on an L1 tree ``im`` (dict with "data", optionally "amp33" and "meta"), in place."""
    offset = int(config["EXTRACT_REF"].get("data_encoding_offset", 0))

    def shift(stack):
        first = stack[0].copy()
        delta = stack[0].astype(np.int32) - offset
        rest = np.clip(stack[1:].astype(np.int32) - delta[None], 0, 65535).astype(np.uint16)
        return first, rest

    im["reference_read"], im["data"] = shift(np.asarray(im["data"]))
    if im.get("amp33") is not None:
        im["reference_amp33"], im["amp33"] = shift(np.asarray(im["amp33"]))
    meta = im.get("meta")
    if meta is not None:
        meta.setdefault("instrument", {})["data_encoding_offset"] = offset
        meta["exposure"]["read_pattern"] = meta["exposure"]["read_pattern"][1:]
    return im
