"""Statistics over many noise realisations of one ramp -- this package's counterpart of the reference's
``validation_tests/many_realizations.py:47-106`` (BASELINE config 5).

For seeds SEED+10, SEED+20, ... one noisy L1 ramp of a fixed ideal-slope scene is generated (own generator:
``synth.make_ramp`` with a fixed ``rate``; the reference uses romanisim + galsim, not available offline), run through
the GPU chain, and kept in HBM as one layer of four stacks: ``diffs`` = L1[-1] - L1[1], the L2 ``images`` and ``err``
embedded in the 4096 x 4096 frame, and ``good`` = not ``PixelMask1.build(dq)``.  The eight output planes follow the
reference's arithmetic exactly (``rip_stats_reduce``):

    0 ideal   1 median(diffs)   2 median(images)   3 N   4 mean   5 std   6 mean - ideal   7 median(err)

with N, mean, std from f32 sums in realisation order (``:78-89``; -1000 where N = 0, ``:87``; zero border).  The
reference maps ``images`` and ``err`` onto the same file (``:58-59``), so as written its plane 2 is median(err);
``reference_alias=True`` (default) reproduces that, ``False`` gives the median of the images.

Several GPUs: realisations are shared round robin; before the statistics the stacks are transposed over RCCL from
"my realisations, all rows" to "all realisations, my rows" (``sharding.seeds_to_rows``), which keeps the sums in
realisation order -- the result is bit-identical for any number of ranks -- and rank 0 collects the planes.
"""

import numpy as np

from .. import _native, pars, sharding, synth
from ..utils import maskhandling

NPLANES = 8


class SeedStacks:
    """The four device stacks for ``nlocal`` realisations of an (ny, nx) frame."""

    def __init__(self, nlocal, ny, nx, device, nb=pars.nborder, mask=maskhandling.PixelMask1, ctx=None):
        import torch

        self.ctx = ctx or _native.default_context(device.index if device.index is not None else 0)
        self.ny, self.nx, self.nb, self.grow = ny, nx, nb, np.ascontiguousarray(mask.array, dtype=np.uint8)
        self.diffs = torch.empty((nlocal, ny, nx), dtype=torch.float32, device=device)
        self.images = torch.empty_like(self.diffs)
        self.err = torch.empty_like(self.diffs)
        self.good = torch.empty((nlocal, ny, nx), dtype=torch.uint8, device=device)

    def push(self, k, cube, slope, err_read, err_poisson, pixeldq):
        """Store realisation ``k`` from device tensors: the u16 L1 cube (ngrp, ny, nx) and the chain's four planes
        (``many_realizations.py:69-77``)."""
        c, ny, nx = self.ctx, self.ny, self.nx
        ngrp = cube.shape[0]
        c.check(c.lib.rip_stats_l1_diff(c.h, cube.data_ptr(), ngrp, ny, nx, ngrp - 1, 1, self.diffs[k].data_ptr()))
        c.check(c.lib.rip_stats_l2_pack(c.h, slope.data_ptr(), err_read.data_ptr(), err_poisson.data_ptr(), pixeldq.data_ptr(),
                                        ny, nx, self.nb, self.grow.ctypes.data, self.images[k].data_ptr(),
                                        self.err[k].data_ptr(), self.good[k].data_ptr()))


def reduce_rows(diffs, images, err, good, ideal_rows, y0, ny, nb=pars.nborder, reference_alias=True, ctx=None):
    """The eight planes (8, nrows, nx) for rows [y0, y0+nrows) from stacks (nseeds, nrows, nx) holding every
    realisation in order (``many_realizations.py:78-101``)."""
    import torch

    ctx = ctx or _native.default_context(diffs.device.index or 0)
    nseeds, nrows, nx = diffs.shape
    for t in (diffs, images, err, good, ideal_rows):
        assert t.is_contiguous() and t.device == diffs.device
    out = torch.empty((NPLANES, nrows, nx), dtype=torch.float32, device=diffs.device)
    torch.cuda.synchronize(diffs.device)
    ctx.check(ctx.lib.rip_stats_reduce(ctx.h, nseeds, diffs.data_ptr(), images.data_ptr(), err.data_ptr(), good.data_ptr(),
                                       ideal_rows.data_ptr(), y0, nrows, ny, nx, nb, 1 if reference_alias else 0,
                                       out.data_ptr()))
    ctx.synchronize()
    return out


def ideal_slope(cal, rate, nb=pars.nborder):
    """The scene's slope in the L2 image's units (DN/s after the flat), zero in the border."""
    ny, nx = rate.shape
    ideal = np.zeros((ny, nx), np.float32)
    flat = np.clip(cal["flat"]["data"].astype(np.float64), 0.1, 10)
    ideal[nb:-nb, nb:-nb] = (rate / flat)[nb:-nb, nb:-nb]
    return ideal


def run(calibrator, slot, cal, nseeds=256, seed0=100, read_pattern=None, device=None, reference_alias=True, generator="host",
        timings=None, rate=None, l1synth=None, stack_capacity=None):
    """Generate and calibrate ``nseeds`` realisations (this rank's share of them), exchange, reduce.  Returns the
    (8, ny, nx) f32 planes as a numpy array on rank 0, None on the other ranks.

    ``generator``: "host" = ``synth.make_ramp`` (numpy; minutes per full frame), "device" = ``synth_gpu.RampFactory`` (the same
    recipe in torch on the GPU: about half a second per full frame, nothing crosses PCIe -- BASELINE config 5 at full size),
    "hip" = the reference's own synthesis path on the device (``from_sim.sim_to_isim.L1Synth``: Poisson totals apportioned to
    the reads, ``make_l1_fullcal``, ``fill_in_refdata_and_1f`` as HIP kernels; dq-init and saturation flagging by the
    calibration call itself).
    ``timings``: optional dict that receives the seconds spent generating, calibrating + stacking, and reducing.
    ``rate``: the scene, (ny, nx) DN/s (default: ``synth.make_rate_image(ny, nx, seed0)``, 0.6 s of numpy for a full frame).
    ``stack_capacity``: allocate the stacks for this many realisations on this rank (at least its share of ``nseeds``): a warm-up
    run with the capacity of the job that follows leaves the 56 GB with torch's caching allocator, and the job does not wait
    0.3-1.9 s for the driver to map them.
    ``l1synth``: generator "hip" -- a ``sim_to_isim.L1Synth`` of this calibration set and read pattern whose arrays are on the
    device already (the synthesis side's counterpart of ``calibrator.load_caldir``: 1.5 GB of uploads for a full frame)."""
    import time

    import torch

    t_start = time.perf_counter()
    device = device or torch.device("cuda", calibrator.ctx.device)
    rp = synth.READ_PATTERN_8 if read_pattern is None else read_pattern
    ny, nx = cal["gain"]["data"].shape
    nb = pars.nborder
    if rate is None:
        rate = synth.make_rate_image(ny, nx, seed0)
    seeds = sharding.scatter_items([seed0 + 10 * (j + 1) for j in range(nseeds)], device=device)
    pid, _ = calibrator.plan_for(rp, synth.FRAME_TIME)
    # the four stacks (13 B per pixel and realisation: 56 GB for 256 realisations of an SCA) are allocated by a helper thread while
    # the first exposure is being made -- the driver maps a fresh allocation of that size in 0.3-1.9 s, and the first exposure
    # spends 1.3 s of host time in the transform library's plan
    import threading

    st_box = {}

    def _alloc_stacks():
        try:
            st_box["st"] = SeedStacks(max(len(seeds), int(stack_capacity or 0)), ny, nx, device, nb=nb, ctx=calibrator.ctx)
        except BaseException as exc:   # re-raised by the main thread
            st_box["exc"] = exc

    st_thread = threading.Thread(target=_alloc_stacks, daemon=True)
    st_thread.start()

    join_wait = [0.0]

    def stacks():
        if st_thread.is_alive():
            tj = time.perf_counter()
            st_thread.join()
            join_wait[0] += time.perf_counter() - tj   # reported on its own ("stack_allocation_wait_s"), not as calibration time
        if "exc" in st_box:
            raise st_box["exc"]
        return st_box["st"]
    slope = torch.empty((ny, nx), dtype=torch.float32, device=device)
    er, ep = torch.empty_like(slope), torch.empty_like(slope)
    pdq = torch.empty((ny, nx), dtype=torch.int32, device=device)
    factory, rate_t = None, None
    if generator == "device":
        from .. import synth_gpu

        factory = synth_gpu.RampFactory(cal, rp, device=calibrator.ctx.device)
        rate_t = torch.from_numpy(np.ascontiguousarray(rate)).to(device)
    hip_synth, counts_mean, t_pdq_hip = None, None, None
    if generator == "hip":
        from ..from_sim import sim_to_isim

        hip_synth = l1synth or sim_to_isim.L1Synth(cal, rp, synth.FRAME_TIME, ctx=calibrator.ctx, nb=nb)
        act = (slice(nb, ny - nb), slice(nb, nx - nb))
        per_s = (np.asarray(rate, dtype=np.float64)[act] + cal["dark"]["dark_slope"][act]) * cal["gain"]["data"][act]   # e/s
        counts_mean = torch.from_numpy(np.clip(per_s * hip_synth.t_reads[-1], 0.0, None).astype(np.float32)).to(device)
        t_pdq_hip = torch.from_numpy(np.array(cal["mask"]["dq"], dtype=np.uint32).view(np.int32)).to(device)
    elif generator not in ("host", "device"):
        raise ValueError(f"generator {generator!r}: host, device or hip")
    t_gen = t_cal = 0.0
    torch.cuda.synchronize(device)
    t_setup = time.perf_counter() - t_start   # rate image, generator state (the stacks are still being allocated by the helper thread)
    for k, sd in enumerate(seeds):
        t0 = time.perf_counter()
        if hip_synth is not None:
            # The synthesis runs on ITS context's streams, the calibration + stacking of the exposure on the calibrator's: with
            # two contexts (``l1synth`` made on its own) exposure k is calibrated (HBM-bound kernels, 1.6 ms) beside the inverse
            # linearity of exposure k+1 (f64 arithmetic); with one context everything is in stream order as before.  The tensors
            # of exposure k stay referenced until the calibrator has been waited for, an exposure later.
            cube, a33 = hip_synth.make(counts_mean, sd, poisson=True)   # (returns with the exposure complete)
            t1 = time.perf_counter()
            calibrator.synchronize()   # calibration + stacking of the exposure before: done beside this make(), or long ago
            held = (cube, a33)         # noqa: F841 -- replaces (frees) the exposure before
            calibrator.calibrate_device(slot, pid, len(rp), cube.data_ptr(), True, a33.data_ptr(), None, t_pdq_hip.data_ptr(),
                                        slope.data_ptr(), er.data_ptr(), ep.data_ptr(), pdq.data_ptr(), flag_saturation=True,
                                        read_pattern=rp)   # the saturation rule calibrateimage applies (gen_cal_image.py:172-185)
            stacks().push(k, cube, slope, er, ep, pdq)
            if hip_synth.ctx is calibrator.ctx or k + 1 == len(seeds):
                calibrator.synchronize()
            t_gen += t1 - t0
            t_cal += time.perf_counter() - t1
            continue
        if factory is not None:
            cube, a33, t_gdq, t_pdq = factory.make(sd, rate_t, poisson=True)
        else:
            ramp = synth.make_ramp(cal, read_pattern=rp, seed=sd, rate=rate)
            gdq = ramp["groupdq"].copy()
            gdq[0] |= 1
            cube = torch.from_numpy(ramp["data"].view(np.int16)).to(device)
            a33 = torch.from_numpy(ramp["amp33"].view(np.int16)).to(device)
            t_gdq = torch.from_numpy(gdq).to(device)
            t_pdq = torch.from_numpy(ramp["pixeldq"].view(np.int32)).to(device)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        calibrator.calibrate_device(slot, pid, len(rp), cube.data_ptr(), True, a33.data_ptr(), t_gdq.data_ptr(),
                                    t_pdq.data_ptr(), slope.data_ptr(), er.data_ptr(), ep.data_ptr(), pdq.data_ptr())
        stacks().push(k, cube, slope, er, ep, pdq)
        calibrator.synchronize()
        t_gen += t1 - t0
        t_cal += time.perf_counter() - t1
    st = stacks()
    t_red = time.perf_counter()
    ideal = torch.from_numpy(ideal_slope(cal, rate, nb)).to(device)
    rows = []
    for stack in (st.diffs, st.images, st.err, st.good):
        t, y0 = sharding.seeds_to_rows(stack[:len(seeds)], nseeds)
        rows.append(t)
    nrows = rows[0].shape[1]
    planes = reduce_rows(*rows, ideal[y0:y0 + nrows].contiguous(), y0, ny, nb=nb, reference_alias=reference_alias,
                         ctx=calibrator.ctx)
    full = sharding.gather_rows(planes, ny)
    if timings is not None:
        timings.update({"setup_s": t_setup, "generate_s": t_gen, "calibrate_and_stack_s": t_cal - join_wait[0],
                        "stack_allocation_wait_s": join_wait[0],
                        "exchange_and_reduce_s": time.perf_counter() - t_red, "realisations_on_this_rank": len(seeds)})
    return None if full is None else full.cpu().numpy()
