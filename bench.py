#!/usr/bin/env python3
"""Throughput of the L1->L2 detector-calibration chain on MI355X: SCA ramps per second.

Contract (see the task description): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is
launched by ``torch.distributed.run`` with one rank per GPU.  A "step" is one pass of the whole chain
(reference pixels, bias, Legendre linearity, IPC deconvolution, ramp fit with jump detection, dark rate,
flat) over one 4096 x 4096 x 8-group ramp whose inputs are already resident in HBM.  Ramps are
independent, so ranks share nothing on the data path: rank 0 scatters the work-item indices (RCCL
broadcast of an int32 list) and every rank processes its own items (weak scaling).

One JSON line is printed by rank 0 with the metric of BASELINE.json plus
  roofline     : dominant kernel, algorithmic bytes / measured kernel time (HIP events on the library's stream)
  cpu_baseline : the numpy oracle timed on a bounded sample of the same workload on this host (N = 1 only)
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

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
    return inputs + outputs, per_kernel


def cpu_baseline(cal, ramp, target_s=7.0):
    """numpy oracle (bit-identical to the reference by tests/golden) on a strip of the same ramp sized for about
    `target_s` seconds of single-thread CPU work (a 128-row probe first, then the sized sample)."""
    ny = ramp["data"].shape[1]
    probe = _cpu_strip(cal, ramp, min(128, ny - 8))
    per_row = probe[1] / probe[2]
    rows = int(max(128, min(ny - 8, target_s / per_row)))
    frac, dt, nrows = _cpu_strip(cal, ramp, rows)
    return {"value": frac / dt, "unit": "ramps/s", "cores": 1, "kind": "port",
            "sample": f"{nrows}x{ramp['data'].shape[2]}x{ramp['data'].shape[0]} strip of the same ramp "
                      f"({frac:.4f} ramp) through the numpy oracle, {dt:.1f} s, single thread"}


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
    oracle.calibrate_arrays(ramp, cal)
    return time.perf_counter() - t0


def cpu_replicas(rows, groups, p_order, ipc64, max_procs=16):
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


def _cpu_strip(cal, ramp, rows):
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
    oracle.calibrate_arrays(sub_ramp, sub_cal)
    dt = time.perf_counter() - t0
    frac = (rows + 2 * nb) / ramp["data"].shape[1]
    return frac, dt, rows + 2 * nb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--groups", type=int, default=8, choices=(8, 16))
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--ipc-dtype", default="f32", choices=("f32", "f64"),
                    help="dtype of the ipc4d coefficients (the reference's production writer stores f64)")
    ap.add_argument("--p-order", type=int, default=8, choices=(3, 8, 10), help="Legendre order of the linearity file")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from romanimpreprocess_amd import pipeline, synth

    rp = synth.READ_PATTERN_8 if args.groups == 8 else synth.READ_PATTERN_16
    G, N, nb, p_order = len(rp), args.side, 4, args.p_order
    kdt = np.float64 if args.ipc_dtype == "f64" else np.float32
    # synthetic CALDIR + ramp (seeded; a strip repeated down the frame so that the host prepares it in seconds)
    cal, ramp = synth.make_tiled_inputs(N, N, read_pattern=rp, p_order=p_order, seed=1 + rank, strip_rows=128, ipc_dtype=kdt)
    cb = pipeline.Calibrator(device=local_rank)
    cb.load_caldir(0, cal)
    pid, meta = cb.plan_for(rp, ramp["frame_time"])

    # inputs resident in HBM before the timed region; outputs preallocated
    gdq_host = ramp["groupdq"].copy()
    gdq_host[0] |= 1
    t_data = torch.from_numpy(ramp["data"].view(np.int16)).to(dev)
    t_a33 = torch.from_numpy(ramp["amp33"].view(np.int16)).to(dev)
    t_gdq = torch.from_numpy(gdq_host).to(dev)
    t_pdq = torch.from_numpy(ramp["pixeldq"].view(np.int32)).to(dev)
    o_slope = torch.empty((N, N), dtype=torch.float32, device=dev)
    o_er = torch.empty_like(o_slope)
    o_ep = torch.empty_like(o_slope)
    o_pdq = torch.empty((N, N), dtype=torch.int32, device=dev)
    o_gdq = torch.empty((G, N, N), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    # work-item scatter: rank 0 owns the list of (exposure, SCA) indices; item i goes to rank i % world
    n_items = (args.steps + args.warmup) * world
    items = torch.arange(n_items, dtype=torch.int32, device=dev) if rank == 0 else torch.empty(n_items, dtype=torch.int32, device=dev)
    if world > 1:
        dist.broadcast(items, src=0)
    mine = items[rank::world].tolist()
    assert len(mine) == args.steps + args.warmup

    def step():
        cb.calibrate_device(0, pid, G, t_data.data_ptr(), True, t_a33.data_ptr(), t_gdq.data_ptr(), t_pdq.data_ptr(),
                            o_slope.data_ptr(), o_er.data_ptr(), o_ep.data_ptr(), o_pdq.data_ptr(), o_gdq.data_ptr())

    def fence():
        cb.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in mine[: args.warmup]:
        step()
    fence()
    cb.ctx.profile(True)
    cb.ctx.profile_read()
    t0 = time.perf_counter()
    for _ in mine[args.warmup:]:
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ms, ncalls = cb.ctx.profile_read()
    cb.ctx.profile(False)

    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
    elapsed = float(t_el.item())

    # sanity: the result of the last step is a real calibration (most active pixels carry no flag, slopes finite)
    good = (o_pdq[nb:-nb, nb:-nb] == 0)
    frac_good = float(good.float().mean().item())
    finite = bool(torch.isfinite(o_slope[nb:-nb, nb:-nb][good]).all().item())
    if not (frac_good > 0.5 and finite):
        raise SystemExit(f"bench sanity check failed: good fraction {frac_good}, finite {finite}")

    if rank == 0:
        total, per_kernel = alg_bytes(G, N, N, nb, p_order + 1, ipc_size=8 if args.ipc_dtype == "f64" else 4)
        fused = ms[2] < 0.05 * max(ncalls, 1) and ms[3] < 0.05 * max(ncalls, 1)  # only event gaps
        names = ["refpix_prepass", "chain_fused" if fused else "cube_stage", "ipc", "rampfit"]
        per_kernel["chain_fused"] = total - G * N * 128 * 2 - N * 128 * 4  # everything but the reference-output block
        avg_ms = {n: ms[i] / max(ncalls, 1) for i, n in enumerate(names) if not (fused and i >= 2)}
        # dominant kernel: the fused chain (the pre-pass runs on a second stream UNDER it, so its own event-to-event time is
        # stretched by the overlap and is not a duration of exclusive use); on the stage path the slowest stage kernel
        dom = "chain_fused" if fused else max((n for n in avg_ms if n != "refpix_prepass"), key=avg_ms.get)
        chain_ms = sum(avg_ms.values())
        ach = per_kernel[dom] / (avg_ms[dom] * 1e-3) / 1e9
        wall_ms = 1e3 * elapsed / args.steps  # per ramp and GPU; the pre-pass of ramp n+1 overlaps the chain of ramp n
        # HBM traffic of the dominant kernel: PMC measurement of the same command, committed under profiles/
        traffic = None
        tpath = os.path.join(REPO, "profiles", "r01_hbm_traffic.json")
        if dom == "chain_fused" and (G, N, p_order, args.ipc_dtype) == (8, 4096, 8, "f32") and os.path.exists(tpath):
            with open(tpath) as tf:
                traffic = json.load(tf).get("traffic_bytes_per_launch")
        out = {
            "metric": "SCA ramps/sec (4096x4096x8grp full L1->L2 chain)" if (G, N) == (8, 4096) else f"SCA ramps/sec ({N}x{N}x{G}grp full L1->L2 chain)",
            "value": world * args.steps / elapsed,
            "unit": "ramps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.ipc_dtype == "f32" else "f32 (IPC stage in f64)",
            "data": "synthetic",
            "config": {"workload": f"single {N}x{N}x{G}-group ramp, full CALDIR (linearitylegendre P_ORDER {p_order} + ipc4d + "
                                   f"biascorr + dark + read + flat), u16 cube resident in HBM, f32 gain / {args.ipc_dtype} ipc4d",
                       "ramps_per_step_per_gpu": 1, "sharding": f"ramps round-robin over {world} GPU(s), index list broadcast over RCCL"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_kernel": per_kernel[dom], "kernel_ms": avg_ms[dom]},
            "chain": {"alg_bytes_per_ramp": total, "kernel_ms": avg_ms, "kernel_ms_sum": chain_ms,
                      "wall_ms_per_ramp": wall_ms, "achieved_GBs": total / (wall_ms * 1e-3) / 1e9,
                      "frac_of_peak": total / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "good_pixel_fraction": frac_good},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cal, ramp)
            if N == 4096:
                out["cpu_baseline"]["replicas"] = cpu_replicas(248, G, p_order, args.ipc_dtype == "f64")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
