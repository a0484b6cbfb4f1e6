"""Level-1 synthesis on the device against the fixture made by EXECUTING the reference's
``make_l1_fullcal`` / ``fill_in_refdata_and_1f`` (tests/golden/l1sim.npz, tools/make_goldens.py case ``l1sim``) and against the
oracle restatement (oracle/l1sim.py); the random parts by their distributions."""

import numpy as np
import pytest
import torch  # noqa: F401  (before the library: both bring a HIP runtime)
from conftest import assert_same_bits, l1sim_golden_cal, load_golden

from romanimpreprocess_amd import _native, synth
from romanimpreprocess_amd.from_sim import sim_to_isim

pytestmark = pytest.mark.gpu


def _u16(t):
    return t.cpu().numpy().view(np.uint16)


@pytest.fixture(scope="module")
def golden():
    g = load_golden("l1sim")
    cal, rp = l1sim_golden_cal(g)
    return g, cal, rp


def test_make_l1_fullcal_and_fill_match_the_reference_functions(golden):
    g, cal, rp = golden
    s = sim_to_isim.L1Synth(cal, rp, float(g["read_time"]), channelwidth=16)
    reads_e = torch.from_numpy(g["reads_e"]).to(s.dev)
    out = s.resultants(reads_e, 1, normals_reset=g["normals_reset"], normals_read=g["normals_read"], want_resultants=True,
                       want_start=True)
    s.ctx.synchronize()
    assert_same_bits(out["resultants"].cpu().numpy(), g["resultants"], "resultants")
    cube = out["cube"]
    assert np.array_equal(_u16(cube), g["im_before"])
    amp33 = torch.zeros((s.ngrp, s.ny, s.cw), dtype=torch.int16, device=s.dev)
    s.fill(cube, amp33, 1, banding=True, normals=g["normals_fill"], frames=g["frames"], white33=g["white33"])
    s.ctx.synchronize()
    assert np.array_equal(_u16(cube), g["im_after"])
    assert np.array_equal(_u16(amp33), g["amp33_after"])


def test_fill_without_banding_and_extract_ref(golden):
    from oracle import l1sim

    g, cal, rp = golden
    s = sim_to_isim.L1Synth(cal, rp, float(g["read_time"]), channelwidth=16)
    tij = l1sim.read_pattern_to_tij(rp, float(g["read_time"]))
    want = g["im_before"].copy()
    l1sim.fill_in_refdata_and_1f(want, cal, tij, g["normals_fill"], frames=None)
    cube = torch.from_numpy(g["im_before"].view(np.int16).copy()).to(s.dev)
    s.fill(cube, None, 1, banding=False, normals=g["normals_fill"])
    s.ctx.synchronize()
    assert np.array_equal(_u16(cube), want)
    ref_want, rest_want = l1sim.extract_ref(want, 1000)
    ref, rest = s.extract_ref(cube, 1000)
    s.ctx.synchronize()
    assert np.array_equal(_u16(ref), ref_want) and np.array_equal(_u16(rest.contiguous()), rest_want)
    tree = {"data": want.copy(), "amp33": g["amp33_after"].copy(), "meta": {"exposure": {"read_pattern": [list(r) for r in rp]}}}
    sim_to_isim.extract_ref(tree, {"EXTRACT_REF": {"data_encoding_offset": 1000}})
    assert np.array_equal(tree["data"], rest_want) and np.array_equal(tree["reference_read"], ref_want)
    assert tree["meta"]["exposure"]["read_pattern"] == [list(r) for r in rp][1:] and tree["amp33"].shape[0] == len(rp) - 1


@pytest.mark.parametrize("kdt,gdt,np_order", [(np.float64, np.float32, 8), (np.float32, np.float64, 5), (None, np.float32, 3)])
def test_resultants_other_dtypes_against_the_oracle(kdt, gdt, np_order):
    """f64 ipc4d / f64 gain / no ipc4d, other Legendre orders, 6 groups: device == oracle restatement on handed-in deviates"""
    from oracle import l1sim

    ny, nx, nb = 32, 512, 4
    rp = synth.READ_PATTERN_6
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=np_order, seed=77, gain_dtype=gdt, ipc_dtype=kdt or np.float32,
                            bias_amplitude=1.5)
    if kdt is None:
        del cal["ipc4d"]
    rng = np.random.default_rng(5)
    counts = rng.poisson(rng.uniform(100, 40000, size=(ny - 2 * nb, nx - 2 * nb))).astype(np.float32)
    tij = l1sim.read_pattern_to_tij(rp, 3.04)
    reads_e = l1sim.binomial_shares(counts, tij, rng)
    n_reset = rng.standard_normal(counts.shape).astype(np.float32)
    n_read = rng.standard_normal((len(rp),) + counts.shape).astype(np.float32)
    want, start = l1sim.make_l1_fullcal(counts, rp, cal, 3.04, n_reset, reads_e, n_read)
    s = sim_to_isim.L1Synth(cal, rp, 3.04, channelwidth=16)
    out = s.resultants(torch.from_numpy(reads_e).to(s.dev), 3, normals_reset=n_reset, normals_read=n_read, want_resultants=True,
                       want_start=True)
    s.ctx.synchronize()
    assert_same_bits(out["start_e"].cpu().numpy(), start, "reset-noise image")
    assert_same_bits(out["resultants"].cpu().numpy(), want, "resultants")
    assert np.array_equal(_u16(out["cube"]), l1sim.embed(want, ny, nx))


def test_apportioning_is_binomial_and_ends_at_the_counts():
    ny, nx, nb = 32, 512, 4
    rp = synth.READ_PATTERN_8
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=3, seed=78)
    s = sim_to_isim.L1Synth(cal, rp, 3.04, channelwidth=16)
    na = (ny - 2 * nb, nx - 2 * nb)

    def shares(counts, seed, poisson=False):   # the entry is asynchronous on the context's stream, torch's copy is not on it
        t = s.apportion(counts, seed=seed, poisson=poisson)
        s.ctx.synchronize()
        return t.cpu().numpy()

    for total in (7, 300, 60000):
        counts = np.full(na, float(total), dtype=np.float32)
        e = shares(counts, 11 + total).astype(np.float64)
        assert np.all(np.diff(e, axis=0) >= 0) and np.all(e[-1] == total) and np.all(e[0] == 0)   # read 0 is at t = 0
        t = s.t_reads
        # electrons up to read r: Binomial(total, t_r / t_end)
        for r in (3, 12, 30):
            p = t[r] / t[-1]
            n = e[r].size
            assert abs(e[r].mean() - total * p) < 5 * np.sqrt(total * p * (1 - p) / n) + 1e-9
            assert abs(e[r].var() / (total * p * (1 - p)) - 1) < 0.06
        # increments of disjoint intervals are anticorrelated as a multinomial's: cov = -total p1 p2
        d1, d2 = e[10] - e[5], e[30] - e[20]
        p1, p2 = (t[10] - t[5]) / t[-1], (t[30] - t[20]) / t[-1]
        cov = np.mean((d1 - d1.mean()) * (d2 - d2.mean()))
        assert abs(cov + total * p1 * p2) < 6 * total * np.sqrt(p1 * p2) / np.sqrt(d1.size) + 0.02 * total * p1 * p2
    # Poisson totals drawn on the device: mean and variance of the last read
    lam = np.full(na, 2500.0, dtype=np.float32)
    e = shares(lam, 5, True).astype(np.float64)
    assert abs(e[-1].mean() - 2500) < 5 * 50 / np.sqrt(e[-1].size) and abs(e[-1].var() / 2500 - 1) < 0.06
    # small means per read (the sky: a few electrons per read -- the f32 inversion branch of the device generator): totals and
    # per-read increments are Poisson (mean, variance, probability of zero), increments of different reads uncorrelated
    nr = len(s.t_reads) - 1   # read 0 at t = 0 receives nothing; the others share the total equally
    for total in (4.0, 70.0, 330.0):
        e = shares(np.full(na, total, dtype=np.float32), 40 + int(total), True).astype(np.float64)
        n = e[-1].size
        assert abs(e[-1].mean() - total) < 5 * np.sqrt(total / n) and abs(e[-1].var() / total - 1) < 0.06
        d = np.diff(e, axis=0)
        per = total / nr
        assert np.all(d >= 0) and abs(d[5].mean() - per) < 5 * np.sqrt(per / n) and abs(d[20].var() / per - 1) < 0.08
        assert abs(np.mean(d[11] == 0) - np.exp(-per)) < 5 * np.sqrt(np.exp(-per) * (1 - np.exp(-per)) / n) + 1e-4
        assert abs(np.corrcoef(d[7].ravel(), d[8].ravel())[0, 1]) < 5 / np.sqrt(n)
    # larger means per read (transformed rejection, its acceptance test in f32 first with an f64 re-test inside the error band):
    # the histogram of 4 x 10^5 increments against the Poisson probabilities (chi-square), and the third central moment
    from scipy import stats

    for total in (400.0, 2500.0, 40000.0):
        e = shares(np.full(na, total, dtype=np.float32), 70 + int(total), True).astype(np.float64)
        d = np.diff(e, axis=0).ravel()
        per = total / nr
        n = d.size
        assert abs(d.mean() - per) < 5 * np.sqrt(per / n) and abs(d.var() / per - 1) < 0.02
        assert abs(np.mean((d - per) ** 3) / per - 1) < 12 * np.sqrt(6.0 * per * per * per / n) / per + 0.02   # mu_3 = lam
        lo, hi = int(stats.poisson.ppf(1e-4, per)), int(stats.poisson.ppf(1 - 1e-4, per))
        ks = np.arange(lo, hi + 1)
        obs = np.array([np.sum(d == k) for k in ks], dtype=np.float64)
        obs = np.concatenate([[np.sum(d < lo)], obs, [np.sum(d > hi)]])
        exp = n * np.concatenate([[stats.poisson.cdf(lo - 1, per)], stats.poisson.pmf(ks, per), [stats.poisson.sf(hi, per)]])
        chi2 = np.sum((obs - exp) ** 2 / exp)
        assert chi2 < stats.chi2.ppf(1 - 1e-6, len(obs) - 1), (total, chi2, len(obs))
    # two seeds differ, one seed repeats
    a, b, c = (shares(lam, sd, True) for sd in (5, 6, 5))
    assert np.array_equal(a, c) and not np.array_equal(a, b)


def test_device_generated_exposure_has_the_statistics_of_the_model(golden):
    """the whole path with device deviates on the fixture's calibration set: reference pixels and amp33 around their model
    means with the model's scatter, active pixels close to a handed-in-deviate run"""
    g, cal, rp = golden
    s = sim_to_isim.L1Synth(cal, rp, float(g["read_time"]), channelwidth=16)
    cube, amp33 = s.make(g["counts"], seed=21)
    s.ctx.synchronize()
    im, a33 = _u16(cube).astype(np.float64), _u16(amp33).astype(np.float64)
    dark = cal["dark"]["data"].astype(np.float64)
    nreads = np.array([len(r) for r in rp], dtype=np.float64)
    for j in range(len(rp)):
        top = im[j, :4] - dark[j, :4]
        expect = np.sqrt(np.mean(cal["read"]["data"][:4].astype(np.float64) ** 2) / nreads[j] + np.mean(cal["read"]["resetnoise"][:4].astype(np.float64) ** 2))
        assert abs(top.mean()) < 5 * expect / np.sqrt(top.size) + 1.0 and 0.8 < top.std() / expect < 1.25
        assert abs(np.mean(a33[j] - cal["read"]["amp33"]["med"])) < 2.0
    # active region: same counts, other deviates -> differences of the order of read + reset + shot noise, no bias
    d = im[:, 4:-4, 4:-4] - g["im_after"][:, 4:-4, 4:-4].astype(np.float64)
    assert abs(np.median(d[-1])) < 3.0 and 5.0 < np.std(d[-1][np.abs(d[-1]) < 500]) < 120.0
    # the reset noise is common to all groups of a reference pixel: group-to-group differences lose it
    dd = (im[1, :4] - dark[1, :4]) - (im[2, :4] - dark[2, :4])
    assert dd.std() < 0.8 * (im[1, :4] - dark[1, :4]).std()


def test_reference_signatures_numpy_in_and_out(golden):
    g, cal, rp = golden
    caldir = {k: {"roman": v} for k, v in cal.items()}
    l1, dq = sim_to_isim.make_l1_fullcal(g["counts"], rp, caldir, rng=np.random.default_rng(1), read_time=float(g["read_time"]))
    assert l1.shape == g["resultants"].shape and l1.dtype == np.float32 and np.array_equal(dq, g["dq"])
    assert np.all(l1 == np.round(l1)) and abs(np.median(l1[-1] - g["resultants"][-1])) < 3.0
    with pytest.raises(ValueError, match="integers"):
        sim_to_isim.make_l1_fullcal(g["counts"] + 0.5, rp, caldir, rng=1)
    with pytest.raises(ValueError, match="rng"):
        sim_to_isim.make_l1_fullcal(g["counts"], rp, caldir)
    # fill_in_refdata_and_1f with the reference's signature: in place on numpy arrays, statistics of the border as in the model
    im = g["im_before"].copy()
    a33 = np.zeros(g["amp33_after"].shape, dtype=np.uint16)
    tij = sim_to_isim.read_pattern_to_tij(rp, float(g["read_time"]))
    sim_to_isim.fill_in_refdata_and_1f(im, caldir, np.random.default_rng(2), tij, fill_in_banding=True, amp33=a33)
    assert np.all(im[:, :4] > 0) and abs(np.mean(a33.astype(np.float64) - cal["read"]["amp33"]["med"])) < 2.0
    assert np.max(np.abs(im[:, 4:-4, 4:-4].astype(np.int32) - g["im_before"][:, 4:-4, 4:-4].astype(np.int32))) < 40
    frame = sim_to_isim.noise_1f_frame(7)
    assert frame.shape == (4096, 128) and frame.dtype == np.float32 and abs(frame.mean()) < 1e-3 and 1.5 < frame.std() < 5.0


def test_calibrate_right_behind_the_synthesis_without_a_sync():
    """rip_calibrate's pre-pass runs on a second stream; it must still see an exposure that rip_synth_* kernels are only about
    to write on the main stream (the library orders it behind its own device-pointer entry points): no synchronisation between
    making the exposure and calibrating it, same results as with one."""
    from romanimpreprocess_amd import pipeline

    rp = synth.READ_PATTERN_8
    ny, nx, nb = 520, 1024, 4
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=5)
    cb = pipeline.Calibrator(ctx=_native.default_context(0))
    cb.load_caldir(12, cal)
    s = sim_to_isim.L1Synth(cal, rp, synth.FRAME_TIME, ctx=cb.ctx)
    pid, _ = cb.plan_for(rp, synth.FRAME_TIME)
    dev = s.dev
    counts = torch.full((ny - 2 * nb, nx - 2 * nb), 3000.0, dtype=torch.float32, device=dev)
    mask = torch.from_numpy(np.array(cal["mask"]["dq"], dtype=np.uint32).view(np.int32)).to(dev)
    torch.cuda.synchronize()

    def run(seed, sync):
        out = [torch.zeros((ny, nx), dtype=torch.float32, device=dev) for _ in range(3)] + [torch.zeros((ny, nx), dtype=torch.int32, device=dev)]
        torch.cuda.synchronize()
        reads_e = s.apportion(counts, seed, poisson=True)
        cube = s.resultants(reads_e, seed)["cube"]
        a33 = torch.zeros((len(rp), ny, s.cw), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()   # the zero fill above is torch's; everything below is the library's
        s.fill(cube, a33, seed)
        if sync:
            cb.synchronize()
        cb.calibrate_device(12, pid, len(rp), cube.data_ptr(), True, a33.data_ptr(), None, mask.data_ptr(), out[0].data_ptr(),
                            out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), flag_saturation=True)
        cb.synchronize()
        return [o.cpu().numpy() for o in out]

    # interleave so that a previous call's main kernel is in flight when the next exposure is made
    try:
        want = [run(31 + i, True) for i in range(3)]
        got = [run(31 + i, False) for i in range(3)]
    finally:
        cb.ctx.drop_caldir(12)
    for a, b in zip(want, got):
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.isfinite(want[0][0]).mean() > 0.9 and abs(np.median(want[0][0][8:-8, 8:-8])) < 50


def test_frames_made_ahead_give_the_same_exposure():
    """L1Synth.make starts the 1/f frames of its fill on the second stream (rip_synth_frames_ahead) beside the apportioning and
    the inverse linearity; the exposure must equal, bit for bit, the one whose fill makes its frames in stream order -- also
    when exposures follow each other without a pause, with another seed in between, and with a mismatching seed."""
    rp = synth.READ_PATTERN_8
    ny, nx, nb = 264, 512, 4
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=9)
    ctx = _native.default_context(0)
    s = sim_to_isim.L1Synth(cal, rp, synth.FRAME_TIME, ctx=ctx)
    dev = s.dev
    counts = torch.full((ny - 2 * nb, nx - 2 * nb), 2500.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def in_order(seed):
        reads_e = s.apportion(counts, seed, poisson=True)
        cube = s.resultants(reads_e, seed)["cube"]
        a33 = torch.zeros((len(rp), ny, s.cw), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        s.fill(cube, a33, seed)
        ctx.synchronize()
        return cube.cpu().numpy(), a33.cpu().numpy()

    want = {sd: in_order(sd) for sd in (11, 12, 13)}
    for sd in (11, 12, 13, 12):
        cube, a33 = s.make(counts, sd, poisson=True)
        assert np.array_equal(cube.cpu().numpy(), want[sd][0]) and np.array_equal(a33.cpu().numpy(), want[sd][1]), sd
    # frames made ahead for ANOTHER seed are not taken by the fill (it makes its own, after waiting for the stray ones)
    ctx.check(ctx.lib.rip_synth_frames_ahead(ctx.h, ny, s.cw, len(rp) * (nx // s.cw + 2), 999))
    got = in_order(13)
    assert np.array_equal(got[0], want[13][0]) and np.array_equal(got[1], want[13][1])
    # and the exposures do differ by seed
    assert not np.array_equal(want[11][0], want[12][0])


def test_one_over_f_calls_on_both_streams_do_not_share_scratch_unordered():
    """The 1/f transforms keep one plan and one set of scratch buffers per context.  An asynchronous rip_synth_noise_1f (device
    output, main stream) followed WITHOUT a host synchronisation by rip_synth_frames_ahead (second stream, another seed) must
    leave both frame sets as serial runs make them (round-3 advisor: the second call used to start while the first was still
    in the scratch)."""
    rp = synth.READ_PATTERN_8
    ny, nx, nb = 264, 512, 4
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=9)
    ctx = _native.default_context(0)
    s = sim_to_isim.L1Synth(cal, rp, synth.FRAME_TIME, ctx=ctx)
    dev = s.dev
    nfr = len(rp) * (nx // s.cw + 2)
    counts = torch.full((ny - 2 * nb, nx - 2 * nb), 2500.0, dtype=torch.float32, device=dev)

    def frames(seed):
        out = torch.empty((nfr, ny, s.cw), dtype=torch.float32, device=dev)
        ctx.check(ctx.lib.rip_synth_noise_1f(ctx.h, ny, s.cw, nfr, seed, 7, out.data_ptr()))
        return out

    def exposure(seed, ahead):
        reads_e = s.apportion(counts, seed, poisson=True)
        cube = s.resultants(reads_e, seed)["cube"]
        a33 = torch.zeros((len(rp), ny, s.cw), dtype=torch.int16, device=dev)
        torch.cuda.current_stream(dev).synchronize()   # torch's zero fill runs on torch's stream (the context's streams are not waited for)
        if ahead:
            ahead()
        s.fill(cube, a33, seed)
        ctx.synchronize()
        return cube.cpu().numpy(), a33.cpu().numpy()

    ctx.synchronize()
    torch.cuda.synchronize()
    want_t = frames(21)
    ctx.synchronize()   # (the entry is asynchronous on the context's stream; torch's copy below is not ordered behind it)
    want_frames = want_t.cpu().numpy()
    want_exp = exposure(22, None)
    for _ in range(3):
        holder = {}

        def both():
            holder["f"] = frames(21)                                                  # main stream, asynchronous
            ctx.check(ctx.lib.rip_synth_frames_ahead(ctx.h, ny, s.cw, nfr, 22))       # second stream, straight away

        got_exp = exposure(22, both)
        assert np.array_equal(holder["f"].cpu().numpy().view(np.uint32), want_frames.view(np.uint32))
        assert np.array_equal(got_exp[0], want_exp[0]) and np.array_equal(got_exp[1], want_exp[1])


def test_one_over_f_frames_do_not_depend_on_the_transform_batch():
    """More frames than one transform buffer holds (64 of 2^20 points) are transformed in EQUAL chunks (70 -> 2 x 35, a full
    4096-wide exposure's 272 -> 5 x 55); the device generator's streams stay laid out in blocks of 64 frames: frame 64 + k of a
    long call with stream id s is frame k of a call with stream id s + 64, whatever the chunking."""
    ctx = _native.default_context(0)
    dev = torch.device("cuda", ctx.device)
    rows, width = 4096, 128

    def frames(n, sid):
        out = torch.empty((n, rows, width), dtype=torch.float32, device=dev)
        ctx.check(ctx.lib.rip_synth_noise_1f(ctx.h, rows, width, n, 31, sid, out.data_ptr()))
        ctx.synchronize()
        return out

    long = frames(70, 7)
    head, tail = frames(3, 7), frames(6, 7 + 64)
    assert torch.equal(long[:3], head)
    assert torch.equal(long[64:], tail)
    assert not torch.equal(long[0], long[64])
    assert abs(float(long.double().std()) / float(head.double().std()) - 1.0) < 0.05
