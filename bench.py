#!/usr/bin/env python3
"""Throughput of the L1->L2 detector-calibration chain on MI355X: SCA ramps per second.

Contract (see the task description): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it runs one rank per GPU
under ``torch.distributed.run`` -- started that way by the driver, or by bench.py itself: a plain ``python bench.py --gpus N``
(no WORLD_SIZE in the environment) re-launches itself under ``torch.distributed.run --nproc-per-node N`` before anything
touches the GPU and passes the children's output and exit code on.  A "step" is one pass of the whole chain
(reference pixels, bias, Legendre linearity, IPC deconvolution, ramp fit with jump detection, dark rate,
flat) over one 4096 x 4096 x 8-group ramp whose inputs are already resident in HBM.  Ramps are
independent, so ranks share nothing on the data path: rank 0 scatters the work-item indices (RCCL
broadcast of an int32 list) and every rank processes its own items (weak scaling).
Defaults: 0.5 s of untimed calls first (``--clock-ramp-s``: an idle MI355X needs a few tenths of a second of load to reach
its steady clock; recorded in ``config.clock_ramp_s_before_warmup``), then W = 100 warm-up steps and K = 1000 timed steps.

Workloads (``--workload``):
  single   (default) BASELINE config 2: one seeded NON-PERIODIC full frame (sky + 25 Gaussian sources through IPC
           and the inverted linearity curve, read noise, cosmic rays, saturation; generated on the device by synth_gpu) per
           rank, calibrated K times against one resident CALDIR set.
  batch72  BASELINE config 4: 18 CALDIR sets (SCAs) resident at once, 72 (filter, SCA) ramps with seed 1000 * filter + sca,
           item i on rank i mod N; a step is one item, the ranks walk their items round and round.

One JSON line is printed by rank 0 with the metric of BASELINE.json plus
  roofline     : dominant kernel, algorithmic bytes / measured kernel time (HIP events on the library's stream)
  cpu_baseline : the numpy oracle timed on a bounded sample of the same workload on this host (N = 1 only)
and, at N = 1 (skipped with --no-extras): the exclusive time of the reference-pixel pre-pass, the production-representative
variants (f64 ipc4d, 16 groups, P_ORDER 10) and the 18-slot batch of config 4.
"""

import argparse
import json
import os
import platform
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# HIP events round the kernels of every PROFILE_EVERY-th timed step (the kernel durations of the roofline block are their averages).
# With events round every kernel of EVERY call the event packets themselves cost 0.02-0.03 ms of wall time per ramp (same-box A/B,
# profiles/r04_summary.md): every 15th step keeps >= 60 samples of the default 1000 steps and the wall time clean (an ODD period:
# consecutive calls alternate between two sets of table / flag buffers, an even period would only ever see one of them).
PROFILE_EVERY = max(1, int(os.environ.get("BENCH_PROFILE_EVERY", "15")))
# vector-pipe floors of the fused kernel's instantiations (ms per 4096 x 4096 ramp, P_ORDER 8): SQ_INSTS_VALU of the kernel at 2
# cycles per f32-rate and 4 per f64-rate instruction on 1024 SIMDs at 2.1 GHz (profiles/r03_variant_counters.txt).  Where this floor
# lies above the HBM floor (16 groups) the HBM fraction alone misstates the bound: the variants carry both.
VALU_FLOOR_MS = {("f32", 8): 0.407, ("f64", 8): 0.520, ("f32", 16): 0.787, ("f64", 16): 1.024}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def alg_bytes(G, ny, nx, nb, nplanes, gain_size=4, ipc_size=4, data_size=2):
    """Algorithmic HBM bytes of one ramp (every distinct array crosses HBM once) and the
    per-kernel split of the current kernel chain."""
    npix, na = ny * nx, (ny - 2 * nb) * (nx - 2 * nb)
    inputs = (G * npix * data_size + G * ny * 128 * 2 + G * npix + npix * 4  # cube, amp33, groupdq, pixeldq
              + G * npix * 4 + ny * 128 * 4                                   # dark.data, amp33.med
              + G * na * 4                                                   # biascorr
              + nplanes * npix * 4 + 3 * npix * 4 + npix * 4                 # linearity planes, Smin/Smax/Sref, dq
              + 9 * na * ipc_size                                            # ipc4d
              + npix * gain_size + 3 * npix * 4)                             # gain, read, dark_slope, flat
    outputs = 4 * npix * 4 + G * npix
    per_kernel = {
        "cube_stage": npix * (G * (data_size + 4 + 4 + 1 + 4) + 4 * (nplanes + 3) + 4 + 4 + 4),
        "ipc": npix * (G * 8 + 9 * ipc_size + gain_size),
        "rampfit": npix * (G * 6 + 4 + gain_size + 4 + 4 + 4 + 4 + 16),
        "refpix_prepass": G * ny * 128 * 2 * 4 + ny * 128 * 4 * 4 + 2 * 8 * nx * G * (data_size + 4),
    }
    per_kernel["chain_fused"] = inputs + outputs - G * ny * 128 * 2 - ny * 128 * 4  # everything but the reference-output block
    return inputs + outputs, per_kernel


def kernel_source_hash():
    """sha256 (first 16 hex digits) of the sources the fused kernel is compiled from: a committed PMC profile is quoted in the
    bench line only while it belongs to the kernel that ran."""
    import hashlib

    h = hashlib.sha256()
    names = ["chain2_kernel.h", "chain_common.h", "device_rampfit.h", "rip_common.h", "Makefile", "chain.hip"]
    names += [f"chain_np{n}{k}.hip" for n in (4, 9, 11) for k in ("", "_k64")]   # they select the instantiation that runs
    for name in names:
        with open(os.path.join(REPO, "romanimpreprocess_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# --------------------------------------------------------------------------------------------- CPU baseline
def _cpu_info():
    model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return {"cpu_model": model, "cpus_available": ncpu, "numpy": np.__version__,
            "threads_env": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}}


def _cpu_strip(cal, ramp, rows, timings=None):
    import oracle

    nb = 4
    sl = slice(0, rows + 2 * nb)
    na = rows
    sub_cal = {
        "dark": {k: (v[:, sl] if v.ndim == 3 else v[sl]) for k, v in cal["dark"].items()},
        "gain": {"data": cal["gain"]["data"][sl]},
        "ipc4d": {"data": np.ascontiguousarray(cal["ipc4d"]["data"][:, :, :na])},
        "linearitylegendre": {k: (v[:, sl] if v.ndim == 3 else v[sl]) for k, v in cal["linearitylegendre"].items()},
        "flat": {"data": cal["flat"]["data"][sl]},
        "read": {"anc": cal["read"]["anc"], "data": cal["read"]["data"][sl],
                 "amp33": {**cal["read"]["amp33"], "med": cal["read"]["amp33"]["med"][sl],
                           "std": cal["read"]["amp33"]["std"][sl]}},
        "biascorr": {"data": np.ascontiguousarray(cal["biascorr"]["data"][:, :na])},
    }
    sub_ramp = {"data": ramp["data"][:, sl], "amp33": ramp["amp33"][:, sl], "groupdq": ramp["groupdq"][:, sl],
                "pixeldq": ramp["pixeldq"][sl], "read_pattern": ramp["read_pattern"], "frame_time": ramp["frame_time"]}
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        oracle.calibrate_arrays(sub_ramp, sub_cal, timings=timings)
    dt = time.perf_counter() - t0
    frac = (rows + 2 * nb) / ramp["data"].shape[1]
    return frac, dt, rows + 2 * nb


def cpu_baseline(cal, ramp, target_s=4.0, repeats=3):
    """numpy oracle (bit-identical to the reference by tests/golden) on a strip of the same ramp sized for about
    `target_s` seconds of CPU work (a 128-row probe first, then the sized sample), best of `repeats`: once with the
    thread variables of numpy's libraries as they are (normally unset), once in a child process with them set to 1."""
    ny = ramp["data"].shape[1]
    probe = _cpu_strip(cal, ramp, min(128, ny - 8))
    per_row = probe[1] / probe[2]
    rows = int(max(128, min(ny - 8, target_s / per_row)))
    best = None
    times = []
    for _ in range(repeats):
        stages = {}
        frac, dt, nrows = _cpu_strip(cal, ramp, rows, timings=stages)
        times.append(dt)
        if best is None or dt < best[1]:
            best = (frac, dt, nrows, stages)
    frac, dt, nrows, stages = best
    out = {"value": frac / dt, "unit": "ramps/s", "cores": 1, "kind": "port",
           "sample": f"{nrows}x{ramp['data'].shape[2]}x{ramp['data'].shape[0]} strip of the same ramp "
                     f"({frac:.4f} ramp) through the numpy oracle, best of {repeats} ({', '.join(f'{t:.2f}' for t in times)} s), "
                     f"thread variables as found (see threads_env)",
           "per_stage_s_per_ramp": {k: v / frac for k, v in stages.items()}}
    out.update(_cpu_info())
    out["threads_1"] = cpu_threads_one(nrows - 8, ramp["data"].shape[0], repeats)
    return out


def _one_thread_worker(args):
    rows, groups, repeats = args
    import numpy as _np

    import oracle
    from romanimpreprocess_amd import synth

    rp = synth.READ_PATTERN_8 if groups == 8 else synth.READ_PATTERN_16
    cal = synth.make_caldir(rows + 8, 4096, read_pattern=rp, p_order=8, seed=1777)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=77)
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        with _np.errstate(all="ignore"):
            oracle.calibrate_arrays(ramp, cal)
        ts.append(time.perf_counter() - t0)
    return ts


def cpu_threads_one(rows, groups, repeats):
    """the same oracle on a numpy-made strip of the same size in a child process started with OMP / OPENBLAS / MKL_NUM_THREADS = 1"""
    import multiprocessing as mp

    names = ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")
    saved = {k: os.environ.get(k) for k in names}
    try:
        os.environ.update({k: "1" for k in names})
        with mp.get_context("spawn").Pool(1) as pool:
            ts = pool.map(_one_thread_worker, [(rows, groups, repeats)])[0]
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    frac = (rows + 8) / 4096.0
    return {"value": frac / min(ts), "unit": "ramps/s", "cores": 1,
            "sample": f"{rows + 8}x4096x{groups} numpy-made strip ({frac:.4f} ramp), best of {repeats} "
                      f"({', '.join(f'{t:.2f}' for t in ts)} s), OMP/OPENBLAS/MKL_NUM_THREADS=1"}


def _replica_worker(args):
    """One process of the replica run: its own strip of a seeded synthetic ramp through the numpy oracle."""
    rows, groups, seed, p_order, ipc64, barrier = args
    import numpy as _np

    import oracle
    from romanimpreprocess_amd import synth

    rp = synth.READ_PATTERN_8 if groups == 8 else synth.READ_PATTERN_16
    cal = synth.make_caldir(rows + 8, 4096, read_pattern=rp, p_order=p_order, seed=1000 + seed,
                            ipc_dtype=_np.float64 if ipc64 else _np.float32)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=seed)
    barrier.wait()
    t0 = time.perf_counter()
    with _np.errstate(all="ignore"):
        oracle.calibrate_arrays(ramp, cal)
    return time.perf_counter() - t0


def cpu_replicas(rows, groups, p_order, ipc64, max_procs=16):   # 16 = the CPU share of a one-GPU box of the pool
    """The same oracle as independent single-thread processes, one strip each, started together: ramps/s of C host cores
    (numpy's elementwise work does not thread, so this is what the host can do with the reference's kind of code)."""
    import multiprocessing as mp

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(max_procs, ncpu))
    ctx = mp.get_context("spawn")
    with ctx.Manager() as man:
        barrier = man.Barrier(procs)
        with ctx.Pool(procs) as pool:
            t0 = time.perf_counter()
            dts = pool.map(_replica_worker, [(rows, groups, 100 + i, p_order, ipc64, barrier) for i in range(procs)])
            wall = time.perf_counter() - t0
    frac = procs * (rows + 8) / 4096.0
    return {"value": frac / max(dts), "unit": "ramps/s", "cores": procs,
            "sample": f"{procs} processes x {rows + 8} rows of 4096 ({frac:.3f} ramp in all), slowest {max(dts):.1f} s "
                      f"(wall incl. generating the strips {wall:.1f} s)"}


# --------------------------------------------------------------------------------------------- device side
class Resident:
    """One ramp's inputs resident in HBM + preallocated outputs."""

    def __init__(self, torch, dev, ramp, outs=None):
        g = ramp["groupdq"].copy()
        g[0] |= 1
        self.G, self.ny, self.nx = ramp["data"].shape
        self.data = torch.from_numpy(ramp["data"].view(np.int16)).to(dev)
        self.a33 = torch.from_numpy(ramp["amp33"].view(np.int16)).to(dev)
        self.gdq = torch.from_numpy(g).to(dev)
        self.pdq = torch.from_numpy(ramp["pixeldq"].view(np.int32)).to(dev)
        self.outs = outs or Outputs(torch, dev, self.G, self.ny, self.nx)


class Outputs:
    def __init__(self, torch, dev, G, ny, nx):
        self.slope = torch.empty((ny, nx), dtype=torch.float32, device=dev)
        self.er = torch.empty_like(self.slope)
        self.ep = torch.empty_like(self.slope)
        self.pdq = torch.empty((ny, nx), dtype=torch.int32, device=dev)
        self.gdq = torch.empty((G, ny, nx), dtype=torch.uint8, device=dev)


def run_steps(cb, calls, warmup, steps, fence, ramp_s=0.0):
    """`calls`: list of zero-argument callables walked round and round.  Returns (elapsed s, per-stage ms sums, calls).
    ``ramp_s``: seconds of untimed calls BEFORE the warm-up steps (setup, like generating the inputs): an idle MI355X needs a few
    tenths of a second of load to reach its steady clock -- the same kernel takes 0.98 ms right after idle and 0.92 ms from then
    on (profiles/r02_summary.md) -- and the metric is steady-state throughput."""
    n = len(calls)
    if ramp_s > 0:
        t_end = time.perf_counter() + ramp_s
        while time.perf_counter() < t_end:
            for i in range(16):
                calls[i % n]()
            fence()
    for i in range(warmup):
        calls[i % n]()
    fence()
    # HIP events round the kernels of every PROFILE_EVERY-th step of the timed region (each recorded event is a packet the
    # command processor has to work through between two kernels: with events round every kernel of every call the event
    # traffic itself costs wall time)
    cb.ctx.profile(False)
    cb.ctx.profile_read()
    t0 = time.perf_counter()
    for i in range(steps):
        if i % PROFILE_EVERY == 0:
            cb.ctx.profile(True)
            calls[(warmup + i) % n]()
            cb.ctx.profile(False)
        else:
            calls[(warmup + i) % n]()
    fence()
    elapsed = time.perf_counter() - t0
    ms, ncalls = cb.ctx.profile_read()
    return elapsed, ms, ncalls


def host_path(cb, ramp, N, G, batch=8):
    """ramps/s through the host-array entry points (rip_calibrate / rip_calibrate_batch with location = RIP_HOST) on CALDIR slot 0."""
    def best_ms(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return 1e3 * min(ts)

    shapes = {"slope": ((N, N), np.float32), "err_read": ((N, N), np.float32), "err_poisson": ((N, N), np.float32),
              "pixeldq": ((N, N), np.uint32), "groupdq": ((G, N, N), np.uint8)}
    h = {k: np.ascontiguousarray(ramp[k]) for k in ("data", "amp33", "groupdq", "pixeldq")}
    host_ramp = dict(ramp, **h)
    out_pg = {k: np.zeros(sh, dt) for k, (sh, dt) in shapes.items()}   # pageable, touched once (steady state of a caller's loop)
    in_bytes = sum(v.nbytes for v in h.values())
    out_bytes = sum(v.nbytes for v in out_pg.values())
    res = {"pcie_bytes_in": in_bytes, "pcie_bytes_out": out_bytes, "pcie_bytes_out_without_groupdq": out_bytes - out_pg["groupdq"].nbytes}
    for want_gdq in (True, False):
        ms = best_ms(lambda: cb.calibrate(0, host_ramp, want_groupdq=want_gdq, out=out_pg), 5)
        res["pageable" + ("" if want_gdq else "_no_groupdq_out")] = {"ms_per_ramp": ms, "ramps_per_s": 1e3 / ms}
    # the default call: no `out`, the result arrays come from the Calibrator's pool of memory touched before (pipeline.ResultPool;
    # fresh np.empty arrays cost 33 ms of first-touch page faults per ramp: 20 ramps/s)
    ms = best_ms(lambda: cb.calibrate(0, host_ramp, want_groupdq=True), 4)
    res["pageable_default_call"] = {"ms_per_ramp": ms, "ramps_per_s": 1e3 / ms}
    pin = {k: cb.pinned_empty(v.shape, v.dtype) for k, v in h.items()}
    for k in pin:
        pin[k][...] = h[k]
    pin_ramp = dict(ramp, **pin)
    outs = [{k: cb.pinned_empty(sh, dt) for k, (sh, dt) in shapes.items()} for _ in range(batch)]
    ms = best_ms(lambda: cb.calibrate(0, pin_ramp, want_groupdq=True, out=outs[0]), 5)
    res["page_locked"] = {"ms_per_ramp": ms, "ramps_per_s": 1e3 / ms}
    for k in ("slope", "pixeldq", "groupdq"):
        if not np.array_equal(outs[0][k], out_pg[k], equal_nan=True):
            raise SystemExit(f"bench: host path, {k} differs between pageable and page-locked arrays")
    ms = best_ms(lambda: cb.calibrate_many(0, [pin_ramp] * batch, want_groupdq=True, out=outs), 3) / batch
    res["batch_page_locked"] = {"ramps": batch, "ms_per_ramp": ms, "ramps_per_s": 1e3 / ms}
    res["note"] = ("wall time of Calibrator.calibrate / calibrate_many on numpy arrays, best of 3-5, results into preallocated arrays; "
                   "PCIe Gen5 x16: 63 GB/s per direction (spec)")
    return res


def gather_elapsed(torch, dist, elapsed, world, cdev):
    """(maximum over the ranks, list of every rank's elapsed seconds)"""
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world == 1:
        return elapsed, [elapsed]
    all_t = [torch.empty_like(t_el) for _ in range(world)]
    dist.all_gather(all_t, t_el)
    per_rank = [float(t.item()) for t in all_t]
    return max(per_rank), per_rank


def run_realizations(args, cb, torch, dist, fence, rank, world, local_rank, cdev):
    """--workload realizations: BASELINE config 5 through harness/many_realizations.py -- `--realizations` noise seeds of one scene
    generated on the device (the reference's synthesis steps as HIP kernels), calibrated, stacked in HBM, exchanged (all-to-all:
    realisations -> rows) and reduced to the eight statistics planes on rank 0.  Strong scaling: the seeds are shared by the ranks."""
    from romanimpreprocess_amd import synth, synth_gpu
    from romanimpreprocess_amd.harness import many_realizations as mr

    N, n = args.side, args.realizations
    rp = synth.READ_PATTERN_8 if args.groups == 8 else synth.READ_PATTERN_16
    cal = synth_gpu.make_caldir(N, N, read_pattern=rp, p_order=args.p_order, seed=1001,
                                ipc_dtype=np.float64 if args.ipc_dtype == "f64" else np.float32, device=local_rank)
    cb.load_caldir(0, cal)
    # inputs of the job, resident before the clock starts like the calibration set of the chain: the scene (synthetic, numpy) and
    # the synthesis side's copy of the calibration arrays
    from romanimpreprocess_amd.from_sim import sim_to_isim

    scene = synth.make_rate_image(N, N, 100)
    from romanimpreprocess_amd import _native

    # (a context of its own: the calibration of exposure k runs beside the synthesis of exposure k + 1, many_realizations.run)
    l1s = sim_to_isim.L1Synth(cal, rp, synth.FRAME_TIME, ctx=_native.Context(cb.ctx.device))
    # warm-up: plans, workspaces, clocks -- and the allocator: two realisations with the stacks of the job that follows, which then
    # takes them from torch's caching allocator (a fresh 56 GB allocation is 0.3-1.9 s of driver time, box to box)
    mr.run(cb, 0, cal, nseeds=max(world, 2), seed0=900, read_pattern=rp, generator="hip", l1synth=l1s,
           stack_capacity=(n + world - 1) // world)
    fence()
    tm = {}
    t0 = time.perf_counter()
    planes = mr.run(cb, 0, cal, nseeds=n, seed0=100, read_pattern=rp, generator="hip", timings=tm, rate=scene, l1synth=l1s)
    fence()
    elapsed, per_rank = gather_elapsed(torch, dist, time.perf_counter() - t0, world, cdev)
    if rank == 0:
        good = planes[3][4:-4, 4:-4]
        ok = good > n // 2
        if not (ok.mean() > 0.5 and np.isfinite(planes[6][4:-4, 4:-4][ok]).all()):
            raise SystemExit("bench sanity check failed: statistics planes of the realisations")
        print(json.dumps({
            "metric": f"noise realisations/s ({N}x{N}x{len(rp)}grp: generate + calibrate + stack + exchange + reduce)",
            "value": n / elapsed, "unit": "realisations/s", "n_gpus": world, "steps": n, "warmup": max(world, 2),
            "ms_per_step": 1e3 * elapsed / n, "elapsed_s_per_rank": per_rank, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"many_realizations: {n} noise seeds of one {N}x{N}x{len(rp)}-group scene, seed j on rank j mod {world}; "
                                   "stacks (13 B per pixel and realisation) resident in HBM, one all-to-all per stack, planes gathered on rank 0; "
                                   "in place before the clock: the calibration set (the chain's and the synthesis side's device copies), the scene, "
                                   "and -- through a warm-up of two realisations at the job's stack capacity -- the allocator's memory for the stacks",
                       "sharding": ("RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal)")
                                   + ("; ranks share one GPU (rehearsal, not a scaling measurement)" if args.share_gpu else "")},
            "phases_rank0_s": tm,
            "median_bias_DN_per_s": float(np.median(planes[6][4:-4, 4:-4][ok])),
            "median_scatter_over_pipeline_error": float(np.median((planes[5] / np.maximum(planes[7], 1e-9))[4:-4, 4:-4][ok]))}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def partition(n_items, rank, world):
    """BASELINE config 4 / SURVEY 8e: item i belongs to rank i mod world."""
    return [i for i in range(n_items) if i % world == rank]


def spawn_ranks(n):
    """`python bench.py --gpus N` started plainly: run the same command line as N ranks of one node (child process: nothing
    in this process has touched the GPU, and no exec from a GPU-initialised process happens).  Returns the exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def dry_run(args, rank, world):
    """--dry-run: the N-rank plumbing of this file (rendezvous, item scatter, barriers round the timed region, MAX all-reduce,
    one JSON line from rank 0) with no GPU and no library: a step is a 1 ms sleep.  A rehearsal, never a measurement."""
    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n_items = 72 if args.workload == "batch72" else (args.steps + args.warmup) * world
    items = torch.arange(n_items, dtype=torch.int32) if rank == 0 else torch.empty(n_items, dtype=torch.int32)
    if world > 1:
        dist.broadcast(items, src=0)
    mine = items[rank::world].tolist()
    assert len(mine) >= 1 and mine == partition(n_items, rank, world)

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.001)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    fence()
    elapsed, per_rank = gather_elapsed(torch, dist, time.perf_counter() - t0, world, "cpu")
    all_items = [None] * world
    if world > 1:
        dist.all_gather_object(all_items, mine)
    else:
        all_items = [mine]
    if rank == 0:
        print(json.dumps({
            "metric": "SCA ramps/sec (dry run: no device work)", "value": world * args.steps / elapsed, "unit": "ramps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "elapsed_s_per_rank": per_rank,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none", "dry_run": True,
            "config": {"workload": f"dry run of --workload {args.workload}: a step is a 1 ms sleep",
                       "sharding": f"items round-robin over {world} rank(s), index list broadcast over gloo",
                       "items_per_rank": all_items}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--clock-ramp-s", type=float, default=0.5,
                    help="seconds of untimed calls before the warm-up steps, so that the timed steps run at the steady clock")
    ap.add_argument("--workload", default="single", choices=("single", "batch72", "realizations"))
    ap.add_argument("--realizations", type=int, default=256,
                    help="--workload realizations (BASELINE config 5): noise realisations of one scene in all, shared by the ranks")
    ap.add_argument("--groups", type=int, default=8, choices=(8, 16))
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--ipc-dtype", default="f32", choices=("f32", "f64"),
                    help="dtype of the ipc4d coefficients (the reference's production writer stores f64)")
    ap.add_argument("--p-order", type=int, default=8, choices=(3, 8, 10), help="Legendre order of the linearity file")
    ap.add_argument("--tiled", action="store_true", help="the round-1 input: a 128-row strip repeated down the frame (numpy)")
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=VALUE",
                    help="rip_set_option before the run (A/B timing: overlap=0, prepass_form=0, ...)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip pre-pass exclusive timing, variants and the 18-slot batch")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="collective backend of the N > 1 run (nccl = RCCL over xGMI; gloo: rehearsals on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: rank r uses GPU r mod (GPUs present) instead of GPU r (needs --dist-backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearsal of the N-rank plumbing without a GPU: item scatter, barriers, MAX all-reduce and the JSON line run "
                         "as they are, a step is a 1 ms sleep; the line carries dry_run: true and is not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and not args.share_gpu and not args.dry_run:
        # one GPU per rank: checked before any rendezvous (counting devices does not initialise the GPU on this image), by the
        # spawning process and by every rank a launcher started
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, {have} visible here "
                             "(rehearsal on fewer GPUs: --dist-backend gloo --share-gpu; no GPU at all: --dry-run)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_gpu and args.dist_backend != "gloo":
        raise SystemExit("--share-gpu needs --dist-backend gloo (RCCL wants one GPU per rank)")
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist

    if args.share_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from romanimpreprocess_amd import pipeline, synth, synth_gpu

    N, nb = args.side, 4

    def make_inputs(groups, p_order, ipc_dtype, cal_seed, ramp_seed, want_cal=True, cal=None):
        rp_ = synth.READ_PATTERN_8 if groups == 8 else synth.READ_PATTERN_16
        kdt_ = np.float64 if ipc_dtype == "f64" else np.float32
        if args.tiled:
            return rp_, *synth.make_tiled_inputs(N, N, read_pattern=rp_, p_order=p_order, seed=ramp_seed, strip_rows=128, ipc_dtype=kdt_)
        if cal is None:
            cal = synth_gpu.make_caldir(N, N, read_pattern=rp_, p_order=p_order, seed=cal_seed, ipc_dtype=kdt_, device=local_rank)
        ramp = synth_gpu.make_ramp(cal, read_pattern=rp_, seed=ramp_seed, device=local_rank)
        return rp_, cal, ramp

    cb = pipeline.Calibrator(device=local_rank)
    for kv in args.set:
        k_, v_ = kv.split("=")
        cb.ctx.set_option(k_, int(v_))

    def fence():
        cb.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.workload == "realizations":
        return run_realizations(args, cb, torch, dist, fence, rank, world, local_rank, cdev)

    def call_for(slot, pid, G, res):
        o = res.outs
        return lambda: cb.calibrate_device(slot, pid, G, res.data.data_ptr(), True, res.a33.data_ptr(), res.gdq.data_ptr(),
                                           res.pdq.data_ptr(), o.slope.data_ptr(), o.er.data_ptr(), o.ep.data_ptr(),
                                           o.pdq.data_ptr(), o.gdq.data_ptr(), inputs_complete=True)  # resident since the last sync

    def sane(res):
        good = (res.outs.pdq[nb:-nb, nb:-nb] == 0)
        frac_good = float(good.float().mean().item())
        finite = bool(torch.isfinite(res.outs.slope[nb:-nb, nb:-nb][good]).all().item())
        if not (frac_good > 0.5 and finite):
            raise SystemExit(f"bench sanity check failed: good fraction {frac_good}, finite {finite}")
        return frac_good

    def kernel_report(G, p_order, ipc_dtype, ms, ncalls):
        total, per_kernel = alg_bytes(G, N, N, nb, p_order + 1, ipc_size=8 if ipc_dtype == "f64" else 4)
        fused = ms[2] < 0.05 * max(ncalls, 1) and ms[3] < 0.05 * max(ncalls, 1)  # only event gaps
        names = ["refpix_prepass", "chain_fused" if fused else "cube_stage", "ipc", "rampfit"]
        avg_ms = {n: ms[i] / max(ncalls, 1) for i, n in enumerate(names) if not (fused and i >= 2)}
        dom = "chain_fused" if fused else max((n for n in avg_ms if n != "refpix_prepass"), key=avg_ms.get)
        ach = per_kernel[dom] / (avg_ms[dom] * 1e-3) / 1e9
        return total, per_kernel, avg_ms, dom, ach

    # ------------------------------------------------------------------------------------------ the headline workload
    G = args.groups
    extras = {}
    if args.workload == "single":
        rp, cal, ramp = make_inputs(G, args.p_order, args.ipc_dtype, 1000 + 1 + rank, 1 + rank)
        cb.load_caldir(0, cal)
        pid, _meta = cb.plan_for(rp, ramp["frame_time"])
        res = Resident(torch, dev, ramp)
        calls = [call_for(0, pid, G, res)]
        n_items = (args.steps + args.warmup) * world
        workload = (f"single {N}x{N}x{G}-group ramp, full CALDIR (linearitylegendre P_ORDER {args.p_order} + ipc4d + biascorr + dark + "
                    f"read + flat), u16 cube resident in HBM, f32 gain / {args.ipc_dtype} ipc4d; "
                    + ("128-row strip repeated down the frame" if args.tiled else
                       "non-periodic frame (sky + 25 Gaussian sources, read noise, cosmic rays, saturation; seeded, generated on the device)"))
    else:
        # BASELINE config 4: 18 SCAs x 4 filters; item i = (filter i // 18, sca 1 + i % 18), seed 1000 * filter + sca
        rp = synth.READ_PATTERN_8 if G == 8 else synth.READ_PATTERN_16
        pid = None
        my_items = partition(72, rank, world)
        slots = sorted({1 + i % 18 for i in my_items})
        cals = {}
        t_setup = time.perf_counter()
        for sca in slots:
            c = synth_gpu.make_caldir(N, N, read_pattern=rp, p_order=args.p_order, seed=5000 + sca,
                                      ipc_dtype=np.float64 if args.ipc_dtype == "f64" else np.float32, device=local_rank)
            cb.load_caldir(sca, c)
            cals[sca] = c
        calls, resident = [], []
        outs = Outputs(torch, dev, G, N, N)  # one set of result planes: a consumer would take them before the next item
        for i in my_items:
            filt, sca = i // 18, 1 + i % 18
            r = synth_gpu.make_ramp(cals[sca], read_pattern=rp, seed=1000 * filt + sca, device=local_rank)
            if pid is None:
                pid, _meta = cb.plan_for(rp, r["frame_time"])
            rr_ = Resident(torch, dev, r, outs)
            resident.append(rr_)
            calls.append(call_for(sca, pid, G, rr_))
        del cals
        res = resident[-1]
        ramp, cal = None, None
        n_items = 72
        extras["setup_s"] = time.perf_counter() - t_setup
        workload = (f"batch of 72 (filter, SCA) ramps {N}x{N}x{G} groups (seed 1000*filter+sca), {len(slots)} CALDIR sets resident per rank "
                    f"({len(slots) * 3.1:.0f} GB), item i on rank i mod {world}; a step is one item, the rank's items are walked round and round")

    # work-item scatter: rank 0 owns the list of (exposure, SCA) indices; item i goes to rank i % world
    items = torch.arange(n_items, dtype=torch.int32, device=cdev) if rank == 0 else torch.empty(n_items, dtype=torch.int32, device=cdev)
    if world > 1:
        dist.broadcast(items, src=0)
    mine = items[rank::world].tolist()
    assert len(mine) >= 1 and mine == partition(n_items, rank, world)

    elapsed, ms, ncalls = run_steps(cb, calls, args.warmup, args.steps, fence, ramp_s=args.clock_ramp_s)
    elapsed, per_rank = gather_elapsed(torch, dist, elapsed, world, cdev)
    frac_good = sane(res)

    if rank == 0:
        total, per_kernel, avg_ms, dom, ach = kernel_report(G, args.p_order, args.ipc_dtype, ms, ncalls)
        wall_ms = 1e3 * elapsed / args.steps  # per ramp and GPU; the pre-pass of ramp n+1 overlaps the chain of ramp n
        # HBM traffic of the dominant kernel: PMC measurement of the SAME command, committed under profiles/ (not measured by
        # this run: counters need rocprofv3); null when no profile of this configuration and kernel build is there
        traffic, traffic_source = None, None
        import glob

        if dom == "chain_fused" and (G, N, args.p_order, args.ipc_dtype, args.workload) == (8, 4096, 8, "f32", "single"):
            for tpath in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_hbm_traffic.json")), reverse=True):   # newest round first
                with open(tpath) as tf:
                    tj = json.load(tf)
                name = os.path.relpath(tpath, REPO)
                if tj.get("kernel_source_sha16") == kernel_source_hash():
                    traffic = tj.get("traffic_bytes_per_launch")
                    traffic_source = (f"{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                      f"(tools/profile_round.sh), kernel {tj.get('kernel')}, kernel sources {tj.get('kernel_source_sha16')}; "
                                      "not re-measured by this run")
                    break
                traffic_source = (f"{name} belongs to other kernel sources "
                                  f"({tj.get('kernel_source_sha16')} != {kernel_source_hash()}): traffic not quoted")
        out = {
            "metric": "SCA ramps/sec (4096x4096x8grp full L1->L2 chain)" if (G, N) == (8, 4096) else f"SCA ramps/sec ({N}x{N}x{G}grp full L1->L2 chain)",
            "value": world * args.steps / elapsed,
            "unit": "ramps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "elapsed_s_per_rank": per_rank,   # the timed region of every rank (value uses the maximum): imbalance is visible
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.ipc_dtype == "f32" else "f32 (IPC stage in f64)",
            "data": "synthetic",
            "config": {"workload": workload, "ramps_per_step_per_gpu": 1, "clock_ramp_s_before_warmup": args.clock_ramp_s,
                       "sharding": f"ramps round-robin over {world} GPU(s), index list broadcast over "
                                   + ("RCCL" if args.dist_backend == "nccl" else "gloo (rehearsal)")
                                   + ("; ranks share one GPU (rehearsal, not a scaling measurement)" if args.share_gpu else ""),
                       "items_rank0": mine[:8]},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "kernel_ms_samples": ncalls, "kernel_ms_sampling": f"HIP events round the kernels of every {PROFILE_EVERY}th timed step",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "alg_bytes_kernel": per_kernel[dom], "kernel_ms": avg_ms[dom],
                         "kernel_form": {0: "stage kernels",
                                         2: "wave-specialised fused (chain2_kernel.h)"}.get(cb.ctx.last_chain_form())},
            "chain": {"alg_bytes_per_ramp": total, "kernel_ms": avg_ms,
                      "wall_ms_per_ramp": wall_ms, "achieved_GBs": total / (wall_ms * 1e-3) / 1e9,
                      "frac_of_peak": total / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "good_pixel_fraction": frac_good},
        }
        out["chain"]["kernel_ms"]["refpix_prepass_note"] = ("event-to-event on the second stream while the previous ramp's chain "
                                                           "kernel holds the CUs: not a duration of exclusive use; see refpix_prepass_exclusive")
        out.update(extras)

        if world == 1 and not args.no_extras and args.workload == "single":
            # ---- pre-pass alone (overlap off: every kernel of a call runs on one stream, so event gaps are exclusive times)
            cb.ctx.set_option("overlap", 0)
            el0, ms0, nc0 = run_steps(cb, calls, 2, 30, fence, ramp_s=0.3)
            cb.ctx.set_option("overlap", 1)
            ex = ms0[0] / max(nc0, 1)
            out["chain"]["kernel_ms"]["refpix_prepass_exclusive"] = ex
            out["chain"]["kernel_ms"]["chain_fused_no_overlap"] = ms0[1] / max(nc0, 1)
            out["chain"]["kernel_ms_sum_exclusive"] = ex + ms0[1] / max(nc0, 1)
            out["chain"]["frac_of_peak_sum_of_kernels"] = total / ((ex + ms0[1] / max(nc0, 1)) * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["chain"]["wall_ms_per_ramp_no_overlap"] = 1e3 * el0 / 30

            # ---- the host-array boundary (what a drop-in user of calibrateimage sees: numpy arrays in, numpy arrays out over
            # PCIe; never `value`): one ramp from pageable numpy arrays, from page-locked arrays, and a batch of 8 page-locked
            # ramps through rip_calibrate_batch
            out["host_path"] = host_path(cb, ramp, N, G)

            # ---- production-representative variants (measured by the driver's run, not only by the builder)
            variants = {}
            del res, calls
            for vname, vg, vp, vk in (("ipc4d_f64", 8, 8, "f64"), ("groups16", 16, 8, "f32"), ("p_order10", 8, 10, "f32"),
                                      ("ipc4d_f64_groups16", 16, 8, "f64")):
                rp_v, cal_v, ramp_v = make_inputs(vg, vp, vk, 2000 + vg + vp, 3)
                cb.load_caldir(1, cal_v)
                pid_v, _m = cb.plan_for(rp_v, ramp_v["frame_time"])
                res_v = Resident(torch, dev, ramp_v)
                el_v, ms_v, nc_v = run_steps(cb, [call_for(1, pid_v, vg, res_v)], 5, 100, fence, ramp_s=0.3)
                sane(res_v)
                tot_v, pk_v, avg_v, dom_v, ach_v = kernel_report(vg, vp, vk, ms_v, nc_v)
                hbm_floor = pk_v[dom_v] / (HBM_PEAK_GBS * 1e9) * 1e3
                valu_floor = VALU_FLOOR_MS.get((vk, vg)) if vp == 8 else None
                floor = max(hbm_floor, valu_floor or 0.0)
                variants[vname] = {"ramps_per_s": 100 / el_v, "kernel": dom_v, "kernel_ms": avg_v[dom_v], "alg_bytes_kernel": pk_v[dom_v],
                                   "frac": ach_v / HBM_PEAK_GBS, "hbm_floor_ms": hbm_floor, "valu_floor_ms": valu_floor,
                                   "frac_of_max_floor": floor / avg_v[dom_v], "kernel_form": cb.ctx.last_chain_form(),
                                   "config": f"{N}x{N}x{vg} groups, P_ORDER {vp}, {vk} ipc4d"}
                cb.ctx.drop_caldir(1)
                del res_v, cal_v, ramp_v
            out["variants"] = variants

            # ---- BASELINE config 4 on one GPU: 18 CALDIR slots resident, 72 items walked in (filter, SCA) order.  To keep the
            # default run short, 3 distinct CALDIR sets are uploaded 6 times each and 8 distinct ramps per set are reused;
            # `--workload batch72` builds all 18 sets and 72 ramps.
            t_b = time.perf_counter()
            rp8 = synth.READ_PATTERN_8
            sets, calls_b, keep = [], [], []
            outs_b = Outputs(torch, dev, 8, N, N)
            for k in range(3):
                c = synth_gpu.make_caldir(N, N, read_pattern=rp8, p_order=8, seed=5000 + k, device=local_rank)
                for j in range(6):
                    cb.load_caldir(2 + k * 6 + j, c)
                rs = []
                for q in range(4):
                    r = synth_gpu.make_ramp(c, read_pattern=rp8, seed=1000 * q + k, device=local_rank)
                    rs.append(Resident(torch, dev, r, outs_b))
                sets.append(rs)
                del c
            pid_b, _m = cb.plan_for(rp8, synth.FRAME_TIME)
            for i in range(72):
                filt, sca = i // 18, i % 18
                rr_ = sets[sca // 6][filt]
                keep.append(rr_)
                calls_b.append(call_for(2 + sca, pid_b, 8, rr_))
            el_b, ms_b, nc_b = run_steps(cb, calls_b, 4, 72 * 4, fence, ramp_s=0.3)
            sane(keep[-1])
            single_ms = 1e3 * elapsed / args.steps
            n_b = 72 * 4   # the 72 items walked four times
            out["batch72"] = {"ramps_per_s": n_b / el_b, "ms_per_ramp": 1e3 * el_b / n_b, "slots_resident": 18,
                              "caldir_bytes_resident": 18 * 3.1e9, "items": 72, "timed_items": n_b,
                              "slot_switch_cost_ms": 1e3 * el_b / n_b - single_ms, "setup_s": time.perf_counter() - t_b,
                              "note": "every item runs against another CALDIR slot than the item before; 3 distinct sets x 6 uploads and "
                                      "12 distinct ramps stand in for 18 sets / 72 ramps (--workload batch72 builds them all)"}
            for s_ in range(2, 20):
                cb.ctx.drop_caldir(s_)
            del sets, calls_b, keep

        if world == 1 and not args.no_cpu_baseline and cal is not None:
            out["cpu_baseline"] = cpu_baseline(cal, ramp)
            if N == 4096:
                out["cpu_baseline"]["replicas"] = cpu_replicas(248, G, args.p_order, args.ipc_dtype == "f64")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
