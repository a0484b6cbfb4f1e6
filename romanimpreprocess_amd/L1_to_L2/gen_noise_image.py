"""Noise layers -- same entry points and configuration surface as the reference's ``L1_to_L2/gen_noise_image.py``
(``make_noise_cube`` :60, ``generate_all_noise`` :334, CLI :392-400); SURVEY.md 8f row 3.

A layer re-runs the whole L1 -> L2 chain on a noise-injected copy of the Level-1 cube and keeps the difference of the
two slope images.  What runs where
  * both chain runs: ``calibrateimage`` of this package (HIP kernels);
  * the injection of white read noise into the cube (``:120-134``): ``rip_stage_noise_inject`` (HIP kernel; exact given the
    normal deviates);
  * clipping (``z``) and sky-mode removal (``S``): ``utils/sky.py`` (HIP kernels);
Random numbers.  The reference draws from ``galsim`` generators, which are not available offline and whose streams cannot be
reproduced; here ``rng`` may be ``None`` / an integer seed (deviates drawn ON THE DEVICE from a counter-based generator:
reproducible, no host random numbers) or a ``numpy.random.Generator`` (deviates drawn on the host in the reference's order
and handed to the kernel).  Layers are therefore statistically, not bit-wise, comparable with the reference's.
  * resampled Poisson layers (``P..r``, ``:262-331``): ``rip_stage_poisson_resample`` (HIP kernel: deviates, their
    accumulation into the resultants and the ramp-fit weights of every pixel's end slice in one pass; exact given the
    deviates).
  * the correlated part of a read-noise layer (``sim_to_isim.fill_in_refdata_and_1f`` :306-402: fresh reference pixels,
    reference output, 1/f noise): ``rip_synth_fill`` on the device (``from_sim.sim_to_isim.L1Synth.fill``: pinned by goldens made by
    executing the reference's function); the 1/f frames -- 34 Fourier transforms of 2^20 points per group -- come from hipFFT
    with device deviates, the white deviates from the device or, with a host generator, from it in the reference's order
    (``NOISE: {CORRELATED: false}`` switches the step off).
The pseudo-Poisson layers (``O``): moment ratios on the host (``GalPoisson/find_tilnus.py``), Pearson-family deviates on the
    device (``rip_stage_pearson``, ``GalPoisson/draw_with_tilnus.py``; parameters pinned by goldens, deviates from the device's
    counter-based generator).
"""

import re
import sys
from copy import deepcopy

import numpy as np

from .. import _native, calio, pars
from ..utils import sky
from .GalPoisson.draw_with_tilnus import draw_from_Pearson
from .GalPoisson.draw_with_tilnus import _seed_from as draw_seed
from .GalPoisson.find_tilnus import get_tilde_nus
from .gen_cal_image import calibrateimage


def _get_subscript(arr, ch):
    """The subscript of directive ``ch``: what follows it up to the next capital letter
    (``_get_subscript('RS2Pg4', 'S') -> '2'``, ``('RS2Pg4', 'P') -> 'g4'``; gen_noise_image.py:32-57)."""
    return re.split(r"(?=[A-Z])", arr.split(ch)[-1])[0]


def inject_read_noise(data, read_noise, read_pattern, nb=pars.nborder, normals=None, seed=0, layer=0, ctx=None):
    """White read noise into a u16 cube (gen_noise_image.py:120-134).  ``normals`` (ngrp, ny-2nb, nx-2nb) f32 or None
    (drawn on the device from ``seed``, ``layer``)."""
    ctx = ctx or _native.default_context()
    data = np.ascontiguousarray(data)
    if data.dtype != np.uint16:
        raise TypeError(f"the Level-1 cube is {data.dtype}, expected uint16")
    G, ny, nx = data.shape
    read = np.ascontiguousarray(read_noise, dtype=np.float32)
    nreads = np.array([len(g) for g in read_pattern], dtype=np.int32)
    if normals is not None:
        normals = np.ascontiguousarray(normals, dtype=np.float32)
        if normals.shape != (G, ny - 2 * nb, nx - 2 * nb):
            raise ValueError(f"normals have shape {normals.shape}")
    out = np.empty_like(data)
    ctx.check(ctx.lib.rip_stage_noise_inject(ctx.h, data.ctypes.data, G, ny, nx, nb, read.ctypes.data, nreads.ctypes.data,
                                             None if normals is None else normals.ctypes.data, int(seed) & (2**64 - 1),
                                             int(layer), out.ctypes.data))
    return out


def noise_1f_frames(nframes, rows=pars.nside, width=pars.channelwidth, normals=None, seed=0, stream=0, ctx=None):
    """``nframes`` frames (rows, width) f32 of 1/f noise (``sim_to_isim.noise_1f_frame`` :265-303) on the GPU (hipFFT).
    ``normals`` (nframes, 4*rows*width) f64 standard normal deviates or None (drawn on the device)."""
    ctx = ctx or _native.default_context()
    if normals is not None:
        normals = np.ascontiguousarray(normals, dtype=np.float64)
        if normals.shape != (nframes, 4 * rows * width):
            raise ValueError(f"normals have shape {normals.shape}, expected {(nframes, 4 * rows * width)}")
    out = np.empty((nframes, rows, width), np.float32)
    ctx.check(ctx.lib.rip_stage_noise_1f(ctx.h, rows, width, nframes, None if normals is None else normals.ctypes.data,
                                         int(seed) & (2**64 - 1), int(stream) & 0xFFFFFFFF, out.ctypes.data))
    return out


def ramp_weight_vectors(processinfo, ngrp):
    """The weights the ramp fit applied to a pixel as a function of its end slice (gen_noise_image.py:236-252): the stored
    optimal weights for a full ramp, the two-point weights for one truncated at ``iend``.  Returns (w (ngrp,ngrp) f32,
    has (ngrp,) u8, endslice i8 with non-positive entries mapped to ngrp - 1)."""
    meta = processinfo["meta"]
    w = np.zeros((ngrp, ngrp), dtype=np.float32)
    has = np.zeros(ngrp, dtype=np.uint8)
    w[-1] = np.asarray(processinfo["weights"], dtype=np.float32)
    has[-1] = 1
    start = 1 if processinfo["exclude_first"] else 0
    for iend in range(start + 2, ngrp):
        Kt = np.zeros(ngrp, dtype=np.float32)
        Kt[iend - 1] = 1.0 / (meta["tbar"][iend - 1] - meta["tbar"][start])
        Kt[start] = -Kt[iend - 1]
        w[iend - 1] = Kt
        has[iend - 1] = 1
    es = np.asarray(processinfo["endslice"])
    endslice = np.where(es > 0, es, ngrp - 1).astype(np.int8)
    return w, has, endslice


def poisson_resample(diff, skylevel, gain, frame_time, read_pattern, weights, has_weights, endslice, samples=None, seed=0,
                     layer=0, ctx=None):
    """Adds a resampled-Poisson realisation to ``diff`` (f32, in place; gen_noise_image.py:262-331).  ``gain`` already
    clipped; ``samples`` (nsamp, ny, nx) f64 Poisson deviates of mean clip(skylevel*gain*frame_time, 0) or None (device)."""
    ctx = ctx or _native.default_context()
    from ..devarray import is_dev   # every array may be a DevArray (resident in HBM): the entry point takes either kind of pointer

    if not ((isinstance(diff, np.ndarray) or is_dev(diff)) and diff.dtype == np.float32 and diff.flags.c_contiguous):
        raise TypeError("diff must be a C-contiguous float32 array (updated in place)")
    ngrp = len(read_pattern)
    first = np.array([g[0] for g in read_pattern], dtype=np.int32)
    count = np.array([len(g) for g in read_pattern], dtype=np.int32)
    for g in read_pattern:
        if list(g) != list(range(g[0], g[0] + len(g))):
            raise ValueError("groups must hold consecutive reads")
    nsamp = int(read_pattern[-1][-1]) + 1
    sky_ = skylevel if is_dev(skylevel) else np.ascontiguousarray(skylevel, dtype=np.float32)
    g_ = gain if is_dev(gain) else np.ascontiguousarray(gain)
    if g_.dtype not in (np.float32, np.float64):
        g_ = g_.astype(np.float64)
    if sky_.dtype != np.float32 or sky_.size != diff.size or g_.size != diff.size:
        raise ValueError("skylevel (float32) and gain must have the shape of diff")
    w = np.ascontiguousarray(weights, dtype=np.float32)
    hw = np.ascontiguousarray(has_weights, dtype=np.uint8)
    es = endslice if is_dev(endslice) else np.ascontiguousarray(endslice, dtype=np.int8)
    if es.dtype != np.int8 or es.size != diff.size:
        raise ValueError("endslice must be int8 of the shape of diff")
    if samples is not None:
        samples = np.ascontiguousarray(samples, dtype=np.float64)
        if samples.shape != (nsamp,) + diff.shape:
            raise ValueError(f"samples have shape {samples.shape}, expected {(nsamp,) + diff.shape}")
    ctx.check(ctx.lib.rip_stage_poisson_resample(
        ctx.h, sky_.ctypes.data, g_.ctypes.data, _native.dtype_code(g_), diff.size, float(frame_time), ngrp, first.ctypes.data,
        count.ctypes.data, w.ctypes.data, hw.ctypes.data, es.ctypes.data, None if samples is None else samples.ctypes.data,
        nsamp, int(seed) & (2**64 - 1), int(layer), diff.ctypes.data))
    return diff


class _Files:
    """The files a layer list keeps coming back to (read, dark, gain, the L2 output), each read once per call."""

    def __init__(self):
        self.trees = {}

    def tree(self, src):
        if isinstance(src, dict):
            return src if "roman" in src else {"roman": src}
        key = str(src)
        if key not in self.trees:
            with calio.open_tree(src) as f:
                self.trees[key] = calio._materialise(f if isinstance(f, dict) else dict(f))
        return self.trees[key]

    def roman(self, src):
        return self.tree(src)["roman"]


def _device_path_applies(config, rng):
    """The HBM-resident layer loop runs when the exposures stay in memory, the deviates are the device's and nothing asks for a
    host-side ingredient (pixel-area map)."""
    if not bool(config["NOISE"].get("IN_MEMORY", True)) or not bool(config["NOISE"].get("DEVICE_RESIDENT", True)):
        return False
    if isinstance(rng, np.random.Generator) or "AREAFACTOR" in config or not config["NOISE"].get("CORRELATED", True):
        return False
    return "saturation" in config["CALDIR"]


def _make_noise_cube_device(config, seed, files, base_tree):
    """``make_noise_cube`` with every full-frame array resident in HBM across the layers ("natural batch for
    the GPU"): the Level-1 cube and its dark-based counterpart, the L2 planes the layers refer to, the masks and weights are
    uploaded ONCE; a layer is then injection -> fresh reference pixels + correlated noise -> the fused chain -> sky model ->
    difference -> clip / resampled Poisson / pseudo-Poisson -> sky model, all through the same kernels as the host-array path
    (whose results it reproduces bit for bit: tests/test_gpu_noise.py), with only the finished layer (67 MB) going back to the
    host.  The few scalar steps (percentile interpolation, the 6 x 6 normal equations of the sky model, the moment ratios of the
    pseudo-Poisson layers) stay on the host as in the mirrors."""
    import torch

    from .. import pipeline
    from ..devarray import DevArray
    from ..from_sim.sim_to_isim import L1Synth
    from .gen_cal_image import _caldir_slot

    layers = config["NOISE"]["LAYER"]
    nb = pars.nborder
    caldir = config["CALDIR"]
    cb = pipeline.Calibrator()
    ctx = cb.ctx
    slot = _caldir_slot(cb, caldir)
    dev = torch.device("cuda", ctx.device)
    roman = base_tree["roman"]
    read_pattern = [list(map(int, g)) for g in roman["meta"]["exposure"]["read_pattern"]]
    frame_time = float(roman["meta"]["exposure"]["frame_time"])
    exclude_first = config.get("EXCLUDE_FIRST", True)
    pid, _meta = cb.plan_for(read_pattern, frame_time, exclude_first, config.get("RAMP_OPT_PARS"), config.get("JUMP_DETECT_PARS"))
    backup = config.get("SATURATION_BACKUP", 1)
    skyorder = int(config["SKYORDER"]) if "SKYORDER" in config else None

    def up(a, view=None):
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(view) if view is not None else a).to(dev)

    base_cube = np.ascontiguousarray(roman["data"])
    if base_cube.dtype != np.uint16:
        return None   # the chain's u16 path is what this loop drives; anything else takes the host-array path
    G, ny, nx = base_cube.shape
    t_base = up(base_cube, np.int16)
    t_a33_base = up(roman["amp33"], np.int16) if roman.get("amp33") is not None else None
    if "mask" in caldir:
        t_mask = up(np.array(files.roman(caldir["mask"])["dq"], dtype=np.uint32), np.int32)
    else:
        t_mask = torch.zeros((ny, nx), dtype=torch.int32, device=dev)
    read = np.asarray(files.roman(caldir["read"])["data"], dtype=np.float32)
    t_read = up(read)
    nreads = np.array([len(g) for g in read_pattern], dtype=np.int32)
    l2 = files.tree(config["OUT"])
    t_orig = up(np.asarray(l2["roman"]["data"], dtype=np.float32))
    na = tuple(t_orig.shape)
    outs = [torch.empty((ny, nx), dtype=torch.float32, device=dev) for _ in range(3)] + [torch.empty((ny, nx), dtype=torch.int32, device=dev)]
    synth_dev = None
    t_dark = t_dark_ref = None
    t_gain_act = t_withsky = t_endslice = None
    sky_models = {}
    noiseimage = np.zeros((len(layers),) + na, dtype=np.float32)

    def tsync():
        torch.cuda.current_stream(dev).synchronize()

    def l2_data(t_cube, t_a33):
        """roman.data of calibrateimage for a device-resident exposure: the chain, the active region, minus the sky model"""
        tsync()
        cb.calibrate_device(slot, pid, G, t_cube.data_ptr(), True, None if t_a33 is None else t_a33.data_ptr(), None, t_mask.data_ptr(),
                            outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), flag_saturation=True,
                            saturation_backup=backup, read_pattern=read_pattern)
        cb.synchronize()
        data = outs[0][nb:ny - nb, nb:nx - nb].contiguous()
        if skyorder is not None:
            tsync()
            sky.medfit(DevArray(data), order=skyorder, subtract=True, ctx=ctx, want_model=False)
        return data

    for i_noise, cmd in enumerate(layers):
        diff = torch.zeros(na, dtype=torch.float32, device=dev)
        if "R" in cmd:
            noiseflags = _get_subscript(cmd, "R")
            t_ref = t_orig
            t_cube = t_base.clone()
            t_a33 = None if t_a33_base is None else t_a33_base.clone()
            if "a" not in noiseflags:   # start from the dark instead of the data
                if t_dark is None:
                    dark = np.asarray(files.roman(caldir["dark"])["data"])
                    de = dark.shape[0] - G
                    if de not in [0, 1]:
                        raise ValueError("Dark date cube has the wrong shape.")
                    t_dark = up(dark.astype(np.uint16)[de:], np.int16)
                    t_dark_ref = l2_data(t_dark, t_a33_base)
                t_cube = t_dark.clone()
                t_ref = t_dark_ref
            tsync()
            ctx.check(ctx.lib.rip_stage_noise_inject(ctx.h, t_cube.data_ptr(), G, ny, nx, nb, t_read.data_ptr(), nreads.ctypes.data, None,
                                                     int(seed) & (2**64 - 1), int(i_noise), t_cube.data_ptr()))
            if synth_dev is None:
                cal_fill = {k: files.roman(caldir[k]) for k in ("read", "gain", "dark")}
                synth_dev = L1Synth(cal_fill, read_pattern, 1.0, ctx=ctx, nb=nb)
            synth_dev.fill(t_cube, t_a33, (int(seed) + 7919 * (i_noise + 1)) & (2**64 - 1), banding=True)
            diff = l2_data(t_cube, t_a33) - t_ref
            if "z" in noiseflags:
                zclip = float(_get_subscript(noiseflags.upper(), "Z"))
                tsync()
                p25, med, p75 = sky.nanpercentiles(DevArray(diff), [25.0, 50.0, 75.0], ctx=ctx)
                iqr = p75 - p25
                print("***", noiseflags, zclip, iqr, med)
                diff = torch.clamp(diff, float(med - zclip * iqr / 1.34896), float(med + zclip * iqr / 1.34896))
        if "O" in cmd or "P" in cmd:
            pinfo = l2["processinfo"]
            if t_withsky is None:
                gain = np.clip(np.asarray(files.roman(caldir["gain"])["data"]), 1e-4, 1e4)
                withsky = np.asarray(l2["roman"]["data_withsky"], dtype=np.float32)
                d = (gain.shape[-1] - withsky.shape[-1]) // 2
                gain_act = np.ascontiguousarray(gain[d:-d, d:-d] if d > 0 else gain)
                t_gain_act, t_withsky = up(gain_act), up(withsky)
                w_all, has_all, endslice = ramp_weight_vectors(pinfo, G)
                t_endslice = up(endslice)
        if "O" in cmd:
            # pseudo-Poisson layer (gen_noise_image.py:173-240): the moment ratios per end slice on the host, one Pearson deviate per
            # pixel of that end slice on the device, scaled by the pixel's gain * rate
            t_fr = l2["roman"]["meta"]["exposure"]["frame_time"]
            gI = t_gain_act.to(torch.float64) * t_withsky.to(torch.float64) if t_gain_act.dtype == torch.float64 else \
                (t_gain_act * t_withsky).to(torch.float64)
            start = 1 if pinfo["exclude_first"] else 0
            rp_l2 = pinfo["meta"].get("read_pattern", read_pattern)
            a_beta = np.array([rp_l2[k][0] for k in range(G)], dtype=int)
            N_beta = np.array([len(rp_l2[k]) for k in range(G)], dtype=int)
            noise_array = torch.zeros(na, dtype=torch.float32, device=dev)
            for k in range(start + 1, G):
                tilnu21, tilnu31, tilnu41, _tilnu42 = get_tilde_nus(N_beta, a_beta, w_all[k])
                tilnu21 *= t_fr
                tilnu31 *= t_fr**2
                tilnu41 *= t_fr**3
                sel = torch.nonzero(t_endslice == k, as_tuple=True)
                npx = int(sel[0].numel())
                print("n pix", npx, "tilnus", tilnu21, tilnu31, tilnu41)
                sys.stdout.flush()
                if npx:
                    t_I = gI[sel].contiguous()
                    t_draw = torch.empty_like(t_I)
                    tsync()
                    ctx.check(ctx.lib.rip_stage_pearson(
                        ctx.h, npx, t_I.data_ptr(), float(tilnu21), float(tilnu31), float(tilnu41),
                        draw_seed(np.random.default_rng([int(seed) & 0xFFFFFFFF, i_noise, k])), (100 * (i_noise + 1) + k) & 0xFFFFFFFF,
                        t_draw.data_ptr(), None, None))
                    noise_array[sel] = t_draw.to(torch.float32)
            diff = (diff + noise_array / t_gain_act).to(torch.float32)   # numpy: f32 array += (f32 / gain dtype), cast back
        if "P" in cmd:
            noiseflags = _get_subscript(cmd, "P")
            t_fr = roman["meta"]["exposure"]["frame_time"]
            if "b" in noiseflags:   # background only: the low-order sky model (one per order for the whole list)
                sky_order = int("0" + _get_subscript(noiseflags.upper(), "B"))
                if sky_order not in sky_models:
                    sky_models[sky_order] = up(sky.medfit(DevArray(t_withsky), order=sky_order, ctx=ctx)[1])
                skylevel = sky_models[sky_order]
            else:
                skylevel = t_withsky.clone()
            if "r" in noiseflags:
                diff = diff.contiguous()
                tsync()
                poisson_resample(DevArray(diff), DevArray(skylevel), DevArray(t_gain_act), t_fr, read_pattern, w_all, has_all,
                                 DevArray(t_endslice), seed=seed, layer=1000 + i_noise, ctx=ctx)
        if "S" in cmd:
            sky_order = int("0" + _get_subscript(cmd, "S"))
            diff = diff.contiguous()
            tsync()
            sky.medfit(DevArray(diff), order=sky_order, subtract=True, ctx=ctx, want_model=False)
        noiseimage[i_noise] = diff.cpu().numpy()
    return noiseimage


def make_noise_cube(config, rng=None):
    """The noise realisations listed in ``config["NOISE"]["LAYER"]``: array (N_noise, ny_active, nx_active) f32."""
    layers = config["NOISE"]["LAYER"]
    files = _Files()
    if _device_path_applies(config, rng):
        with calio.open_tree(config["IN"]) as f_in:
            base_tree_ = calio._materialise(f_in if isinstance(f_in, dict) else dict(f_in))
        seed_ = config["NOISE"].get("SEED", 0) if rng is None else int(rng)
        got = _make_noise_cube_device(config, seed_, files, base_tree_)
        if got is not None:
            return got
    synth_dev = None   # from_sim.sim_to_isim.L1Synth of this CALDIR set: reference pixels and correlated noise on the device
    host_rng = rng if isinstance(rng, np.random.Generator) else None
    seed = config["NOISE"].get("SEED", 0) if (rng is None or host_rng is not None) else int(rng)
    nb = pars.nborder
    noiseimage = None
    # config["NOISE"]["IN_MEMORY"] (default true): the noise-injected exposures and their L2 images stay in memory instead of going
    # through the TEMP files of the reference (same arithmetic; a layer then costs its two chain runs and the noise generation,
    # not four full-frame ASDF round trips); the dark-based reference image of the layers without 'a' is computed once
    in_memory = bool(config["NOISE"].get("IN_MEMORY", True))
    with calio.open_tree(config["IN"]) as f_in:
        base_tree = calio._materialise(f_in if isinstance(f_in, dict) else dict(f_in))
    orig_data = np.asarray(files.roman(config["OUT"])["data"])
    dark_ref = None   # L2 "data" of the dark cube itself
    for i_noise, cmd in enumerate(layers):
        mytree = {k: v for k, v in base_tree.items()}
        mytree["roman"] = {k: (v.copy() if k in ("data", "amp33") and isinstance(v, np.ndarray) else v)
                           for k, v in base_tree["roman"].items()}
        diff = np.zeros_like(orig_data)
        if noiseimage is None:
            noiseimage = np.zeros((len(layers),) + diff.shape, dtype=np.float32)
        read_pattern = mytree["roman"]["meta"]["exposure"]["read_pattern"]

        if "R" in cmd:
            noiseflags = _get_subscript(cmd, "R")
            ref_data = orig_data
            if "a" not in noiseflags:  # start from the dark instead of the data
                dark = np.asarray(files.roman(config["CALDIR"]["dark"])["data"])
                de = dark.shape[0] - np.shape(mytree["roman"]["data"])[0]
                if de not in [0, 1]:
                    raise ValueError("Dark date cube has the wrong shape.")
                mytree["roman"]["data"] = dark.astype(mytree["roman"]["data"].dtype)[de:, :, :]
                if in_memory:
                    if dark_ref is None:
                        dark_ref = np.asarray(calibrateimage(dict(config, IN=mytree, OUT=None))["roman"]["data"])
                    ref_data = dark_ref
                else:
                    calio.write_asdf(config["NOISE"]["TEMP"], mytree)
                    config3 = deepcopy(config)
                    config3["IN"] = config["NOISE"]["TEMP"]
                    config3["OUT"] = config["NOISE"]["TEMP"][:-5] + "_refL2.asdf"
                    calibrateimage(config3)
                    with calio.open_tree(config3["OUT"]) as f_ref:
                        ref_data = np.asarray(f_ref["roman"]["data"])
            read = np.asarray(files.roman(config["CALDIR"]["read"])["data"], dtype=np.float32)
            data = np.ascontiguousarray(mytree["roman"]["data"])
            normals = None
            if host_rng is not None:  # one draw per group, in the reference's order
                na = (data.shape[1] - 2 * nb, data.shape[2] - 2 * nb)
                normals = np.stack([host_rng.standard_normal(na, dtype=np.float32) for _ in range(data.shape[0])])
            mytree["roman"]["data"] = inject_read_noise(data, read, read_pattern, nb=nb, normals=normals, seed=seed,
                                                        layer=i_noise)
            # correlated noise: fresh reference pixels, reference output and 1/f noise (sim_to_isim.fill_in_refdata_and_1f), on the
            # device (from_sim.sim_to_isim.L1Synth.fill: the white deviates of a host generator are handed in in the reference's
            # order, otherwise everything is drawn there)
            if config["NOISE"].get("CORRELATED", True):
                import torch

                from ..from_sim.sim_to_isim import L1Synth

                if synth_dev is None:
                    cal_fill = {k: files.roman(config["CALDIR"][k]) for k in ("read", "gain", "dark")}
                    synth_dev = L1Synth(cal_fill, read_pattern, 1.0, nb=nb)
                cube = np.ascontiguousarray(mytree["roman"]["data"])
                a33 = mytree["roman"].get("amp33")
                t_cube = torch.from_numpy(cube.view(np.int16)).to(synth_dev.dev)
                t_a33 = None if a33 is None else torch.from_numpy(np.ascontiguousarray(a33).view(np.int16)).to(synth_dev.dev)
                normals = white33 = None
                if host_rng is not None:
                    normals = host_rng.standard_normal((cube.shape[0] + 1,) + cube.shape[1:], dtype=np.float32)
                    if t_a33 is not None:
                        white33 = host_rng.standard_normal(tuple(t_a33.shape), dtype=np.float32)
                synth_dev.fill(t_cube, t_a33, (int(seed) + 7919 * (i_noise + 1)) & (2**64 - 1), banding=True, normals=normals,
                               white33=white33)
                synth_dev.ctx.synchronize()
                mytree["roman"]["data"] = t_cube.cpu().numpy().view(np.uint16)
                if t_a33 is not None:
                    mytree["roman"]["amp33"] = t_a33.cpu().numpy().view(np.uint16)
            if in_memory:
                diff = np.asarray(calibrateimage(dict(config, IN=mytree, OUT=None))["roman"]["data"]) - ref_data
            else:
                calio.write_asdf(config["NOISE"]["TEMP"], mytree)
                config2 = deepcopy(config)
                config2["IN"] = config["NOISE"]["TEMP"]
                config2["OUT"] = config["NOISE"]["TEMP"][:-5] + "_L2.asdf"
                calibrateimage(config2)
                with calio.open_tree(config2["OUT"]) as f_out:
                    diff = np.asarray(f_out["roman"]["data"]) - ref_data
            if "z" in noiseflags:
                zclip = float(_get_subscript(noiseflags.upper(), "Z"))
                p25, med, p75 = sky.nanpercentiles(diff, [25.0, 50.0, 75.0])
                iqr = p75 - p25
                print("***", noiseflags, zclip, iqr, med)
                diff = np.clip(diff, med - zclip * iqr / 1.34896, med + zclip * iqr / 1.34896)
        if "O" in cmd:
            # pseudo-Poisson layer (gen_noise_image.py:173-240): per end slice, the moment ratios of the ramp-fit slope under
            # Poisson noise (host), then one Pearson-family deviate per pixel (device) scaled by the pixel's gain * rate
            gain = np.clip(np.asarray(files.roman(config["CALDIR"]["gain"])["data"]), 1e-4, 1e4)
            f_L2 = files.tree(config["OUT"])
            withsky = np.asarray(f_L2["roman"]["data_withsky"])
            pinfo = f_L2["processinfo"]
            t_fr = f_L2["roman"]["meta"]["exposure"]["frame_time"]
            d = (gain.shape[-1] - withsky.shape[-1]) // 2
            if d > 0:
                gain = gain[d:-d, d:-d]
            gI = gain * withsky
            ngrp_o = len(read_pattern)
            w, _has, endslice = ramp_weight_vectors(pinfo, ngrp_o)
            start = 1 if pinfo["exclude_first"] else 0
            rp_l2 = pinfo["meta"].get("read_pattern", read_pattern)   # from the L2 file, as gen_noise_image.py:208-212 reads it
            a_beta = np.array([rp_l2[k][0] for k in range(ngrp_o)], dtype=int)
            N_beta = np.array([len(rp_l2[k]) for k in range(ngrp_o)], dtype=int)
            noise_array = np.zeros(endslice.shape, dtype=np.float32)
            for k in range(start + 1, ngrp_o):
                tilnu21, tilnu31, tilnu41, _tilnu42 = get_tilde_nus(N_beta, a_beta, w[k])
                tilnu21 *= t_fr          # e/frame -> e/s
                tilnu31 *= t_fr**2
                tilnu41 *= t_fr**3
                pixels = np.where(endslice == k)
                print("n pix", len(pixels[0]), "tilnus", tilnu21, tilnu31, tilnu41)
                sys.stdout.flush()
                if len(pixels[0]):
                    noise_array[pixels] = draw_from_Pearson(
                        tilnu21, tilnu31, tilnu41, gI[pixels],
                        rng=host_rng if host_rng is not None else np.random.default_rng([int(seed) & 0xFFFFFFFF, i_noise, k]),
                        stream=100 * (i_noise + 1) + k)
            diff = np.asarray(diff, dtype=np.float32).copy()
            diff[:, :] += noise_array / gain
        if "P" in cmd:
            noiseflags = _get_subscript(cmd, "P")
            f_L2 = files.tree(config["OUT"])
            withsky = np.asarray(f_L2["roman"]["data_withsky"], dtype=np.float32)
            pinfo = f_L2["processinfo"]
            t_fr = mytree["roman"]["meta"]["exposure"]["frame_time"]
            if "b" in noiseflags:  # background only: the low-order sky model
                sky_order = int("0" + _get_subscript(noiseflags.upper(), "B"))
                skylevel = sky.medfit(withsky, order=sky_order)[1]
            else:
                skylevel = withsky.copy()
            if "r" in noiseflags:
                gain = np.clip(np.asarray(files.roman(config["CALDIR"]["gain"])["data"]), 1e-4, 1e4)
                d = (gain.shape[-1] - skylevel.shape[-1]) // 2
                if d > 0:
                    gain = gain[d:-d, d:-d]
                w, has, endslice = ramp_weight_vectors(pinfo, len(read_pattern))
                samples = None
                if host_rng is not None:
                    e = np.clip(skylevel * gain * t_fr, 0.0, None)
                    samples = np.stack([host_rng.poisson(e.astype(np.float64)).astype(np.float64)
                                        for _ in range(int(read_pattern[-1][-1]) + 1)])
                diff = np.ascontiguousarray(diff, dtype=np.float32)
                poisson_resample(diff, skylevel, np.ascontiguousarray(gain), t_fr, read_pattern, w, has, endslice, samples=samples,
                                 seed=seed, layer=1000 + i_noise)
        if "S" in cmd:
            sky_order = int("0" + _get_subscript(cmd, "S"))
            diff = diff - sky.medfit(diff, order=sky_order)[1]
        noiseimage[i_noise, :, :] = diff
    return noiseimage


def generate_all_noise(config):
    """Driver (gen_noise_image.py:334-389): ``config["NOISE"]`` holds LAYER (list of directives), TEMP (scratch file), SEED
    and OUT; the configuration must have been run through ``calibrateimage`` already."""
    noiseimage = make_noise_cube(config, None)
    print(np.shape(noiseimage))
    print("percentiles:")
    per_layer = [sky.nanpercentiles(layer, [5.0, 25.0, 50.0, 75.0, 95.0]) for layer in noiseimage]   # selection on the device
    for i, q in enumerate([5, 25, 50, 75, 95]):
        print(q, np.array([p[i] for p in per_layer]))
    if "NOISE_PRECISION" in config:
        if config["NOISE_PRECISION"] == 16:
            noiseimage = noiseimage.astype(np.float16)
        if config["NOISE_PRECISION"] not in [16, 32]:
            raise ValueError("Unsupported noise precision.")
    calio.write_asdf(config["NOISE"]["OUT"], {"config": config, "noise": noiseimage})
    if config.get("FITSOUT", False):
        raise NotImplementedError("FITS output needs astropy")


if __name__ == "__main__":
    import yaml

    with open(sys.argv[1]) as f:
        cfg = yaml.safe_load(f)
    calibrateimage(cfg | {"SLICEOUT": True})
    generate_all_noise(cfg)
