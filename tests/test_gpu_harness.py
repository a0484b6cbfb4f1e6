"""Statistics over many realisations on the GPU against the goldens made by executing the reference's
``validation_tests/many_realizations.py`` (tools/make_goldens.py harness), and the harness end to end."""

import numpy as np
import pytest
import torch
from conftest import assert_same_bits, gpu_context, load_golden
from test_oracle_golden import harness_check

import golden_cases as gc
from romanimpreprocess_amd import pipeline, synth
from romanimpreprocess_amd.harness import many_realizations as mr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _t(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint16:
        a = a.view(np.int16)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).to(DEV)


def _ideal(g, c):
    big = np.zeros((4096, 4096), np.float32)
    big[4:-4, 4:-4] = gc.harness_truth(c["seed"]) / float(gc.HARNESS_EXPTIME) / float(g["g_ideal"])
    return np.ascontiguousarray(gc.harness_orient(big, c["scanum"]))


def _fill(name):
    """Device stacks of a golden case, filled through rip_stats_l1_diff / rip_stats_l2_pack from full-frame planes as
    the chain delivers them (border pixels carry values and flags that must not leak into the statistics)."""
    g, c = load_golden(name), gc.HARNESS_CASES[name]
    ideal = _ideal(g, c)
    st = mr.SeedStacks(c["nrun"], 4096, 4096, DEV, ctx=gpu_context())
    rng = np.random.default_rng(5)
    for j in range(c["nrun"]):
        r = gc.harness_realisation(c["seed"], j, ideal[4:-4, 4:-4])
        cube = np.stack([r["l1_first"] + 1, r["l1_first"], r["l1_last"]])
        slope = rng.standard_normal((4096, 4096)).astype(np.float32)
        er = np.full((4096, 4096), 3.0, np.float32)
        pdq = np.full((4096, 4096), 0x80000001, np.uint32)
        slope[4:-4, 4:-4], er[4:-4, 4:-4], pdq[4:-4, 4:-4] = r["data"], r["err"], r["dq"]
        st.push(j, _t(cube), _t(slope), _t(er), _t(np.zeros((4096, 4096), np.float32)), _t(pdq))
        gpu_context().synchronize()
    return g, st, _t(ideal)


@pytest.mark.parametrize("name", ["harness_a", "harness_b"])
def test_statistics_match_the_reference_script(name):
    g, st, ideal = _fill(name)
    out = mr.reduce_rows(st.diffs, st.images, st.err, st.good, ideal, 0, 4096, ctx=gpu_context()).cpu().numpy()
    harness_check(out, g, "device")
    assert np.isnan(out[7, 504, 604]) and (out[4:6, 104:112, 204:220] == -1000.0).all()
    # row ranges as the ranks of a multi-GPU run own them give the same planes
    parts = []
    for y0, y1 in ((0, 1365), (1365, 2730), (2730, 4096)):
        sl = [t[:, y0:y1].contiguous() for t in (st.diffs, st.images, st.err, st.good)]
        parts.append(mr.reduce_rows(*sl, ideal[y0:y1].contiguous(), y0, 4096, ctx=gpu_context()).cpu().numpy())
    assert_same_bits(np.concatenate(parts, axis=1), out, "row ranges")
    # the median of the images proper (the reference's stacks alias: its plane 2 is the median of err)
    img = mr.reduce_rows(st.diffs, st.images, st.err, st.good, ideal, 0, 4096, reference_alias=False, ctx=gpu_context())
    want = np.median(st.images.cpu().numpy(), axis=0)
    assert_same_bits(img[2].cpu().numpy(), want, "median(images)")
    assert_same_bits(img[[0, 1, 3, 4, 5, 6, 7]].cpu().numpy(), out[[0, 1, 3, 4, 5, 6, 7]], "other planes")


@pytest.mark.parametrize("nseeds", [1, 2, 255, 256, 300, 600])
def test_medians_for_many_realisations(nseeds):
    """LDS-resident columns (<= 512 realisations) and the re-reading fallback, odd and even counts, ties, signed zeros."""
    rng = np.random.default_rng(nseeds)
    ny, nx = 6, 50
    stack = (np.round(rng.standard_normal((nseeds, ny, nx)) * 4) / 4).astype(np.float32)
    stack[:, 0, 0] = np.float32(1.5)
    stack[::2, 0, 1] = np.float32(-0.0)
    stack[nseeds // 2, 1, 1] = np.nan
    good = (rng.random((nseeds, ny, nx)) < 0.8).astype(np.uint8)
    ideal = rng.standard_normal((ny, nx)).astype(np.float32)
    out = mr.reduce_rows(_t(stack), _t(stack), _t(stack), _t(good), _t(ideal), 0, ny, nb=1, reference_alias=False,
                         ctx=gpu_context()).cpu().numpy()
    with np.errstate(invalid="ignore"):
        want = np.median(stack, axis=0)
    assert np.array_equal(out[1] == 0, want == 0)
    np.testing.assert_array_equal(out[1], want)
    assert np.isnan(out[1, 1, 1])
    # moments in realisation order, f32
    n, s1, s2 = (np.zeros((ny, nx), np.float32) for _ in range(3))
    for j in range(nseeds):
        w = good[j] != 0
        n += np.where(w, 1, 0.0)
        s1 += np.where(w, stack[j], 0.0)
        s2 += np.where(w, stack[j] ** 2, 0.0)
    with np.errstate(invalid="ignore"):
        m1, m2 = s1 / (n + 1e-25), s2 / (n + 1e-25)
        sd = np.sqrt(np.clip(m2 - m1 ** 2, 0, None))
    m1, sd = np.where(n > 0.1, m1, -1000.0), np.where(n > 0.1, sd, -1000.0)
    inner = (slice(1, -1), slice(1, -1))
    for plane, ref in ((3, n), (4, m1), (5, sd), (6, (m1 - ideal).astype(np.float32))):
        np.testing.assert_array_equal(out[plane][inner], ref.astype(np.float32)[inner])
    assert (out[3:6, 0] == 0).all() and np.array_equal(out[6, 0], (0 - ideal[0]).astype(np.float32))


def test_harness_end_to_end_small_frame():
    """Generator -> chain -> stacks -> statistics on a small frame: planes are consistent with the per-realisation results."""
    rp = synth.READ_PATTERN_8
    ny, nx, nseeds = 64, 256, 5
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=3, seed=11)
    cb = pipeline.Calibrator(ctx=gpu_context())
    cb.load_caldir(0, cal)
    out = mr.run(cb, 0, cal, nseeds=nseeds, seed0=100, read_pattern=rp, device=DEV, reference_alias=False)
    assert out.shape == (8, ny, nx) and out.dtype == np.float32
    rate = synth.make_rate_image(ny, nx, 100)
    slopes, errs, goods, diffs = [], [], [], []
    from romanimpreprocess_amd.utils import maskhandling
    for j in range(nseeds):
        ramp = synth.make_ramp(cal, read_pattern=rp, seed=100 + 10 * (j + 1), rate=rate)
        res = cb.calibrate(0, ramp, want_groupdq=False)
        im, er = np.zeros((ny, nx), np.float32), np.zeros((ny, nx), np.float32)
        im[4:-4, 4:-4] = res["slope"][4:-4, 4:-4]
        er[4:-4, 4:-4] = np.sqrt(res["err_read"][4:-4, 4:-4] ** 2 + res["err_poisson"][4:-4, 4:-4] ** 2)
        gd = np.zeros((ny, nx), bool)
        gd[4:-4, 4:-4] = ~maskhandling.PixelMask1.build(res["pixeldq"][4:-4, 4:-4], ctx=gpu_context())
        slopes.append(im), errs.append(er), goods.append(gd)
        diffs.append(ramp["data"][-1].astype(np.float32) - ramp["data"][1].astype(np.float32))
    assert_same_bits(out[1], np.median(np.stack(diffs), axis=0), "median(diffs)")
    assert_same_bits(out[2], np.median(np.stack(slopes), axis=0), "median(images)")
    assert_same_bits(out[7], np.median(np.stack(errs), axis=0), "median(err)")
    assert_same_bits(out[3], np.sum(goods, axis=0).astype(np.float32), "N")
    ok = out[3] >= 3
    assert ok.mean() > 0.5
    # the chain recovers the scene: mean within a few standard errors of the ideal slope for most pixels
    z = np.abs(out[6][ok]) / (out[5][ok] / np.sqrt(out[3][ok]) + 1e-3)
    assert np.median(z) < 3.0


def test_harness_with_the_device_generator():
    """The realisations generated ON the device (synth_gpu.RampFactory: what makes BASELINE config 5 feasible at full size) go
    through the same chain and statistics: planes consistent with recomputing every realisation one by one, the scatter over
    realisations equal to the pipeline's own error estimate, the mean unbiased."""
    from romanimpreprocess_amd import synth_gpu
    from romanimpreprocess_amd.utils import maskhandling

    rp = synth.READ_PATTERN_8
    ny, nx, nseeds = 72, 256, 24
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=13)
    cb = pipeline.Calibrator(ctx=gpu_context())
    cb.load_caldir(0, cal)
    tm = {}
    out = mr.run(cb, 0, cal, nseeds=nseeds, seed0=100, read_pattern=rp, device=DEV, reference_alias=False, generator="device",
                 timings=tm)
    assert out.shape == (8, ny, nx) and tm["realisations_on_this_rank"] == nseeds
    # recompute: the factory is seeded, so the same realisations come out again
    rate = synth.make_rate_image(ny, nx, 100)
    f = synth_gpu.RampFactory(cal, rp, device=0)
    slopes, goods = [], []
    for j in range(nseeds):
        data, a33, gq, _m = f.make(100 + 10 * (j + 1), rate, poisson=True)
        ramp = {"data": data.cpu().numpy().view(np.uint16), "amp33": a33.cpu().numpy().view(np.uint16), "groupdq": gq.cpu().numpy(),
                "pixeldq": cal["mask"]["dq"].copy(), "read_pattern": rp, "frame_time": synth.FRAME_TIME}
        res = cb.calibrate(0, ramp, want_groupdq=False)
        im = np.zeros((ny, nx), np.float32)
        im[4:-4, 4:-4] = res["slope"][4:-4, 4:-4]
        gd = np.zeros((ny, nx), bool)
        gd[4:-4, 4:-4] = ~maskhandling.PixelMask1.build(res["pixeldq"][4:-4, 4:-4], ctx=gpu_context())
        slopes.append(im), goods.append(gd)
    assert_same_bits(out[2], np.median(np.stack(slopes), axis=0), "median(images)")
    assert_same_bits(out[3], np.sum(goods, axis=0).astype(np.float32), "N")
    ok = out[3] >= nseeds - 2
    assert ok.mean() > 0.5
    ratio = out[5][ok] / np.maximum(out[7][ok], 1e-6)   # std over realisations / median of the pipeline's error
    assert 0.85 < np.median(ratio) < 1.15, np.median(ratio)
    z = out[6][ok] / (out[5][ok] / np.sqrt(out[3][ok]) + 1e-6)     # (mean - ideal) in standard errors
    assert abs(np.median(z)) < 0.5 and np.median(np.abs(z)) < 1.5


_TWO_RANK_WORKER = """
import os, sys
sys.path.insert(0, {repo!r})
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
from romanimpreprocess_amd import pipeline, synth
from romanimpreprocess_amd.harness import many_realizations as mr
rp = synth.READ_PATTERN_8
cal = synth.make_caldir(64, 256, read_pattern=rp, p_order=3, seed=11)
cb = pipeline.Calibrator(device=0)
cb.load_caldir(0, cal)
out = mr.run(cb, 0, cal, nseeds=5, seed0=100, read_pattern=rp, device=torch.device("cuda", 0), reference_alias=False)
if dist.get_rank() == 0:
    np.save({out!r}, out)
else:
    assert out is None
dist.barrier()
dist.destroy_process_group()
print("rank", os.environ["RANK"], "ok")
"""


def test_two_ranks_give_the_single_process_planes(tmp_path):
    """Two processes (gloo rendezvous, both on this GPU) share five realisations 3 + 2, exchange realisations for rows and
    gather: the eight planes equal the single-process result bit for bit (sums stay in realisation order)."""
    import os
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rp = synth.READ_PATTERN_8
    cal = synth.make_caldir(64, 256, read_pattern=rp, p_order=3, seed=11)
    cb = pipeline.Calibrator(ctx=gpu_context())
    cb.load_caldir(0, cal)
    single = mr.run(cb, 0, cal, nseeds=5, seed0=100, read_pattern=rp, device=DEV, reference_alias=False)
    out = str(tmp_path / "planes.npy")
    script = tmp_path / "worker.py"
    script.write_text(_TWO_RANK_WORKER.format(repo=repo, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-2000:]
    assert_same_bits(np.load(out), single, "two ranks vs one")


def test_harness_with_the_reference_synthesis_path_on_the_device():
    """generator="hip": every realisation made by the reference's own synthesis steps as HIP kernels (Poisson totals, binomial
    shares per read, make_l1_fullcal, fill_in_refdata_and_1f with 1/f noise and reference output), flagged for saturation and
    calibrated by the chain: the scatter over realisations equals the pipeline's error estimate and the mean is unbiased."""
    rp = synth.READ_PATTERN_8
    ny, nx, nseeds = 72, 256, 32
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=13)
    cb = pipeline.Calibrator(ctx=gpu_context())
    cb.load_caldir(0, cal)
    tm = {}
    out = mr.run(cb, 0, cal, nseeds=nseeds, seed0=100, read_pattern=rp, device=DEV, reference_alias=False, generator="hip", timings=tm)
    assert out.shape == (8, ny, nx) and tm["realisations_on_this_rank"] == nseeds
    ok = out[3] >= nseeds - 2
    assert ok[4:-4, 4:-4].mean() > 0.5
    ratio = out[5][ok] / np.maximum(out[7][ok], 1e-6)   # std over realisations / median of the pipeline's error
    assert 0.85 < np.median(ratio) < 1.15, np.median(ratio)
    z = out[6][ok] / (out[5][ok] / np.sqrt(out[3][ok]) + 1e-6)     # (mean - ideal) in standard errors
    assert abs(np.median(z)) < 0.5 and np.median(np.abs(z)) < 1.5, (np.median(z), np.median(np.abs(z)))
    # a second run gives the same planes: the device generators are keyed by the seeds
    again = mr.run(cb, 0, cal, nseeds=nseeds, seed0=100, read_pattern=rp, device=DEV, reference_alias=False, generator="hip")
    assert np.array_equal(out, again)
    # the synthesis on a context of its own (exposure k is calibrated and stacked beside the synthesis of exposure k+1), the scene
    # handed in, the stacks larger than the run needs: the same planes, bit for bit
    from romanimpreprocess_amd import _native
    from romanimpreprocess_amd.from_sim import sim_to_isim

    own = _native.Context(cb.ctx.device)
    l1s = sim_to_isim.L1Synth(cal, rp, synth.FRAME_TIME, ctx=own)
    piped = mr.run(cb, 0, cal, nseeds=nseeds, seed0=100, read_pattern=rp, device=DEV, reference_alias=False, generator="hip",
                   l1synth=l1s, rate=synth.make_rate_image(ny, nx, 100), stack_capacity=nseeds + 5)
    assert np.array_equal(out, piped)
    del l1s
    own.close()
