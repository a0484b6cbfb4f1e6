"""Noise layers: read-noise injection kernel and the gen_noise_image driver on the GPU chain."""

import numpy as np
import pytest
import torch
from conftest import assert_same_bits, gpu_context

from oracle import noise as onoise
from romanimpreprocess_amd import calio, pipeline, synth
from romanimpreprocess_amd.L1_to_L2 import gen_cal_image, gen_noise_image

pytestmark = pytest.mark.gpu


def test_get_subscript():
    assert gen_noise_image._get_subscript("RS2Pg4", "S") == "2"
    assert gen_noise_image._get_subscript("RS2Pg4", "P") == "g4"
    assert gen_noise_image._get_subscript("Raz3.5S1", "R") == "az3.5"


def test_injection_is_exact_given_the_normals():
    rng = np.random.default_rng(7)
    rp = synth.READ_PATTERN_8
    G, ny, nx = len(rp), 40, 200
    data = rng.integers(0, 65536, size=(G, ny, nx), dtype=np.uint16)
    data[:, 10, 10:14] = [0, 1, 65534, 65535]                 # clipping at both ends
    read = (6 + 5 * rng.random((ny, nx))).astype(np.float32)
    normals = rng.standard_normal((G, ny - 8, nx - 8)).astype(np.float32)
    normals[:, 3, 3] = np.float32(0.5) / (read[7, 7] / np.sqrt(1.0)).astype(np.float32)   # a tie for round-half-even in group 0
    got = gen_noise_image.inject_read_noise(data, read, rp, normals=normals, ctx=gpu_context())
    assert_same_bits(got, onoise.inject_read_noise(data, read, rp, normals), "injected cube")
    assert np.array_equal(got[:, :4], data[:, :4]) and np.array_equal(got[:, :, -4:], data[:, :, -4:])


def test_device_deviates_are_standard_normal_and_reproducible():
    rp = synth.READ_PATTERN_8
    G, ny, nx = len(rp), 264, 520
    data = np.full((G, ny, nx), 30000, np.uint16)
    read = np.full((ny, nx), 64.0, np.float32)                 # large sigma: rounding to integers does not matter
    a = gen_noise_image.inject_read_noise(data, read, rp, seed=5, layer=0, ctx=gpu_context())
    b = gen_noise_image.inject_read_noise(data, read, rp, seed=5, layer=0, ctx=gpu_context())
    c = gen_noise_image.inject_read_noise(data, read, rp, seed=5, layer=1, ctx=gpu_context())
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    act = (slice(None), slice(4, -4), slice(4, -4))
    for k in range(G):
        z = (a[act][k].astype(np.float64) - 30000.0) / (64.0 / np.sqrt(len(rp[k])))
        n = z.size
        assert abs(z.mean()) < 5 / np.sqrt(n) + 0.02 and abs(z.std() - 1.0) < 0.02
        assert abs(np.mean(z**3)) < 0.05 and abs(np.mean(z**4) - 3.0) < 0.1
    # groups, layers and neighbouring pixels are uncorrelated
    z0 = a[act][0].astype(np.float64) - 30000.0
    z1 = a[act][1].astype(np.float64) - 30000.0
    zc = c[act][0].astype(np.float64) - 30000.0
    for u, v in ((z0, z1), (z0, zc), (z0[:, 1:], z0[:, :-1]), (z0[1:], z0[:-1])):
        assert abs(np.corrcoef(u.ravel(), v.ravel())[0, 1]) < 0.01


def _write_inputs(tmp_path, rp, ny, nx):
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=3, seed=31, bias_amplitude=1.0)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=32, cr_frac=0.0)
    caldir = {}
    names = {"dark": "dark", "read": "read", "gain": "gain", "linearitylegendre": "linearitylegendre", "ipc4d": "ipc4d",
             "flat": "pflat", "biascorr": "biascorr", "mask": "mask", "saturation": "saturation"}
    for key, fname in names.items():
        path = tmp_path / f"roman_wfi_{fname}_TEST_SCA04.asdf"
        calio.write_asdf(str(path), {"roman": cal[key]})
        caldir[key] = str(path)
    l1 = {"roman": {"data": ramp["data"], "amp33": ramp["amp33"],
                    "meta": {"exposure": {"frame_time": synth.FRAME_TIME, "read_pattern": rp},
                             "instrument": {"detector": "WFI04"}}}}
    calio.write_asdf(str(tmp_path / "l1.asdf"), l1)
    config = {"IN": str(tmp_path / "l1.asdf"), "OUT": str(tmp_path / "l2.asdf"), "CALDIR": caldir, "SLICEOUT": True,
              "NOISE": {"LAYER": ["Ra", "R", "RaS2", "Raz2", "Ccomment", "Pr", "Pb2r", "OS2"], "TEMP": str(tmp_path / "tmp.asdf"), "SEED": 11,
                        "OUT": str(tmp_path / "noise.asdf")}, "NOISE_PRECISION": 32}
    return cal, ramp, config


def test_noise_layers_end_to_end(tmp_path):
    """calibrateimage, then generate_all_noise: every layer is the difference of two chain runs; its scatter is the read
    noise of the slope, the sky-mode layer has no low-order content, the clipped layer is bounded."""
    rp = synth.READ_PATTERN_8
    cal, ramp, config = _write_inputs(tmp_path, rp, 72, 256)
    cb = pipeline.Calibrator(ctx=gpu_context())
    gen_cal_image.calibrateimage(config, verbose=False, calibrator=cb)
    gen_noise_image.generate_all_noise(config)
    out = calio.read_asdf(config["NOISE"]["OUT"])
    noise = np.asarray(out["noise"])
    l2 = calio.read_asdf(config["OUT"])
    assert noise.shape == (8,) + np.asarray(l2["roman"]["data"]).shape and noise.dtype == np.float32
    good = np.asarray(l2["roman"]["dq"]) == 0
    err_read = np.sqrt(np.asarray(l2["roman"]["var_rnoise"]))[good]
    for i in (0, 1, 2):
        ratio = np.std(noise[i][good]) / np.sqrt(np.mean(err_read**2))
        print('R layer', i, 'ratio', ratio)
        # an extra read-noise realisation: the white part alone is the slope's read-noise error times the noise gain of the IPC
        # deconvolution (checked below with CORRELATED off: +-3 %); fresh reference pixels and 1/f noise add the rest
        assert 0.95 < ratio < 1.25, (i, ratio)
        # correlated (1/f) noise leaves a small common offset after the reference-pixel correction
        assert abs(np.mean(noise[i][good])) < 0.3 * np.std(noise[i][good])
    assert not np.array_equal(noise[0], noise[1])
    # 'S2': the quadratic sky model of the layer is gone
    from romanimpreprocess_amd.utils import sky
    coef, _ = sky.medfit(noise[2], order=2, ctx=gpu_context())
    coef0, _ = sky.medfit(noise[0], order=2, ctx=gpu_context())
    assert np.max(np.abs(coef)) < 0.2 * np.max(np.abs(coef0)) + 1e-4
    # 'z2': clipped at 2 sigma-equivalents of the interquartile range
    p25, med, p75 = np.percentile(noise[3], [25, 50, 75])
    assert noise[3].max() - noise[3].min() < 2 * 2.2 * (p75 - p25) / 1.34896
    assert np.count_nonzero(noise[4]) == 0               # a comment-only layer
    # same seed, same layers
    again = gen_noise_image.make_noise_cube(config)
    assert_same_bits(again, noise, "layers from the same seed")
    # the HBM-resident layer loop (default) and the host-array loop drive the same kernels with the same seeds: every layer of the
    # list -- read noise with correlated noise, clipping, sky model, resampled Poisson, pseudo-Poisson -- bit for bit
    host_loop = gen_noise_image.make_noise_cube(dict(config, NOISE=dict(config["NOISE"], DEVICE_RESIDENT=False)))
    assert_same_bits(host_loop, noise, "host-array layer loop vs the HBM-resident one")
    # ... also where a clipped read-noise layer goes on through the sky model, the resampled Poisson or the pseudo-Poisson step in
    # the SAME layer (the production list's shape, 'Rz4PbrS2' / 'Rz4OS2': the clip bounds are float32 on both sides)
    prod = dict(config["NOISE"], LAYER=["Rz3S1", "Rz4Pb1rS2", "Rz4OS2", "RaOS1"])
    dev_loop = gen_noise_image.make_noise_cube(dict(config, NOISE=prod))
    host_loop = gen_noise_image.make_noise_cube(dict(config, NOISE=dict(prod, DEVICE_RESIDENT=False)))
    assert_same_bits(host_loop, dev_loop, "host-array layer loop vs the HBM-resident one (clip + sky model + Poisson in one layer)")
    assert np.std(dev_loop[1][good]) > 0 and not np.array_equal(dev_loop[1], dev_loop[2])
    # the exposures kept in memory (default) or sent through the TEMP files as the reference does: the same layers
    one = dict(config["NOISE"], LAYER=["R", "Raz3S1"])
    mem = gen_noise_image.make_noise_cube(dict(config, NOISE=one))
    files = gen_noise_image.make_noise_cube(dict(config, NOISE=dict(one, IN_MEMORY=False)))
    assert_same_bits(mem, files, "in-memory vs temp-file layers")
    white = gen_noise_image.make_noise_cube(dict(config, NOISE=dict(config["NOISE"], LAYER=["R"], CORRELATED=False)))
    ratio = np.std(white[0][good]) / np.sqrt(np.mean(err_read**2))
    # white noise per pixel goes through the order-2 IPC deconvolution, which amplifies it: sum of the squared taps of the
    # inverse kernel (mean coefficients of this CALDIR set: centre 0.936 -> 1.068, variance factor 1.14); the chain's read-noise
    # error follows the reference's formula, which does not know about it
    from scipy.signal import convolve2d

    kmean = np.asarray(cal["ipc4d"]["data"], dtype=np.float64)[:, :, 8:-8, 8:-8].mean(axis=(2, 3))
    delta = np.zeros((9, 9))
    delta[4, 4] = 1.0
    taps = delta.copy()
    for _ in range(2):
        taps = taps + delta - convolve2d(taps, kmean, mode="same")
    gain_ipc = np.sqrt(np.sum(taps**2))
    print('white R layer ratio', ratio, 'deconvolution factor', gain_ipc)
    assert 1.05 < gain_ipc < 1.09 and 0.97 < ratio / gain_ipc < 1.03, (ratio, gain_ipc)
    # host deviates in the reference's order give different, equally valid layers
    other = gen_noise_image.make_noise_cube(dict(config, NOISE=dict(config["NOISE"], LAYER=["Ra"])), np.random.default_rng(3))
    assert 0.8 < np.std(other[0][good]) / np.sqrt(np.mean(err_read**2)) < 1.25
    # resampled Poisson layers, from first principles: a Poisson increment of e electrons per frame enters the slope through
    # the coefficients c_j of get_tilde_nus, so a layer's variance is sum_j c_j^2 * e / gain^2 with e = skylevel * gain * t_frame
    # (skylevel = the L2 image with sky for 'Pr', its order-2 model for 'Pb2r'); pixels with the full ramp only.  The slope's own
    # err_poisson is NOT the yardstick: it also holds the dark current's shot noise, which a sky layer does not resample.
    from romanimpreprocess_amd.L1_to_L2.GalPoisson.find_tilnus import get_tilde_nus

    k2 = get_tilde_nus([len(r) for r in rp], [r[0] for r in rp], np.asarray(l2["processinfo"]["weights"], dtype=np.float64))[0]
    withsky = np.asarray(l2["roman"]["data_withsky"], dtype=np.float32)
    gain_act = np.clip(cal["gain"]["data"][4:-4, 4:-4].astype(np.float64), 1e-4, 1e4)
    full = good & (np.asarray(l2["processinfo"]["endslice"]) <= 0)
    err_p = np.sqrt(np.asarray(l2["roman"]["var_poisson"]))[good]
    for i, level in ((5, withsky), (6, sky.medfit(withsky, order=2, ctx=gpu_context())[1]), (7, withsky)):   # 7: the 'O' layer, same variance
        predicted = np.sqrt(k2 * np.clip(level, 0.0, None) * synth.FRAME_TIME / gain_act)
        use = full & (predicted > 0)
        ratio = np.std(noise[i][use] / predicted[use])
        print('P layer', i, 'scatter / prediction', ratio, 'scatter / err_poisson', np.std(noise[i][good]) / np.sqrt(np.mean(err_p**2)))
        assert 0.95 < ratio < 1.05, (i, ratio)
    # pseudo-Poisson layer ('O': Pearson-family deviates with the slope's second to fourth moments under Poisson noise): its
    # scatter is the Poisson error of the slope as well, its mean is zero
    ratio = np.std(noise[7][good]) / np.sqrt(np.mean(err_p**2))
    print('O layer ratio', ratio)
    assert 0.8 < ratio < 1.25, ratio
    assert abs(np.mean(noise[7][good])) < 0.05 * np.std(noise[7][good])


@pytest.mark.parametrize("gdt", [np.float32, np.float64])
def test_poisson_resampling_is_exact_given_the_deviates(gdt):
    rng = np.random.default_rng(17)
    rp = synth.READ_PATTERN_8
    ngrp, ny, nx = len(rp), 30, 70
    sky_ = (0.4 + 3.0 * rng.random((ny, nx)) ** 4).astype(np.float32)
    sky_[3, 3] = -0.2                                       # negative sky: clipped to zero electrons
    sky_[4, 4] = 4000.0
    gain = np.clip((1.5 + 0.1 * rng.standard_normal((ny, nx))).astype(gdt), 1e-4, 1e4)
    t_fr = 3.04
    tbar = [t_fr * np.mean(g) for g in rp]
    pinfo = {"meta": {"tbar": tbar}, "weights": rng.standard_normal(ngrp).astype(np.float32), "exclude_first": True,
             "endslice": rng.integers(-1, ngrp, size=(ny, nx)).astype(np.int8)}
    w, has, endslice = gen_noise_image.ramp_weight_vectors(pinfo, ngrp)
    assert has.tolist() == [0, 0, 1, 1, 1, 1, 1, 1] and endslice.min() >= 1
    e = np.clip(sky_ * gain * t_fr, 0.0, None)
    samples = np.stack([rng.poisson(e.astype(np.float64)).astype(np.float64) for _ in range(rp[-1][-1] + 1)])
    start = (0.01 * rng.standard_normal((ny, nx))).astype(np.float32)
    start[0, 0] = -0.0
    want = onoise.poisson_resample(start.copy(), sky_, gain, t_fr, rp, w, has, endslice, samples)
    got = gen_noise_image.poisson_resample(start.copy(), sky_, gain, t_fr, rp, w, has, endslice, samples=samples, ctx=gpu_context())
    assert_same_bits(got, want, "resampled Poisson layer")


def test_device_poisson_deviates():
    """Mean and variance of the device generator across the inversion / PTRS switch: a single read group, weight 1, so
    that the layer IS the re-centred deviate in DN."""
    rp = [[0]]
    n = 400_000
    for lam in (0.05, 0.7, 3.0, 9.5, 10.5, 40.0, 900.0, 20000.0):
        sky_ = np.full((1, n), lam, np.float32)
        gain = np.ones((1, n), np.float32)
        diff = np.zeros((1, n), np.float32)
        gen_noise_image.poisson_resample(diff, sky_, gain, 1.0, rp, np.ones((1, 1), np.float32), np.ones(1, np.uint8),
                                         np.zeros((1, n), np.int8), seed=3, layer=int(lam * 10), ctx=gpu_context())
        k = diff.astype(np.float64) + lam
        assert np.all(np.abs(k - np.round(k)) < 1e-2 * max(1.0, lam / 1000)) and k.min() >= -1e-3
        se = np.sqrt(lam / n)
        assert abs(k.mean() - lam) < 5 * se, (lam, k.mean())
        assert abs(k.var() - lam) < 5 * lam * np.sqrt(2.0 / n + 1.0 / (lam * n)), (lam, k.var())
        if lam < 5:
            p0 = np.mean(np.round(k) == 0)
            assert abs(p0 - np.exp(-lam)) < 5 * np.sqrt(np.exp(-lam) / n) + 1e-4


@pytest.mark.parametrize("rows,width,form", [(60, 128, -1), (5, 7, -1), (64, 128, 0), (64, 64, -1), (16, 8, -1), (8, 8, -1), (8192, 128, -1)])
def test_1f_frames_of_other_lengths_against_numpy_fft(rows, width, form):
    """Frame lengths that are no power of two (and, option pink_form = 0, any length) go through the library's transform, powers
    of two from 2^7 points through the hand-written one (pink_fft.h: 8 x 8 up to 1024 x 1024, odd and even log2); a frame of
    2^6 points is below it.  Odd frame sizes take the unpaired stores of the last kernel."""
    ctx = gpu_context()
    rng = np.random.default_rng(5)
    normals = rng.standard_normal((3, 4 * rows * width))
    ctx.set_option("pink_form", form)
    try:
        got = gen_noise_image.noise_1f_frames(3, rows=rows, width=width, normals=normals, ctx=ctx)
    finally:
        ctx.set_option("pink_form", -1)
    for f in range(3):
        want = onoise.noise_1f_frame(normals[f], rows, width)
        np.testing.assert_allclose(got[f], want, rtol=0, atol=2e-6 * np.abs(want).max())
        assert np.mean(got[f] != want) < 0.01


@pytest.mark.parametrize("rows,width", [(64, 128), (4096, 128)])
def test_1f_frames_against_numpy_fft(rows, width):
    rng = np.random.default_rng(23)
    normals = rng.standard_normal((3, 4 * rows * width))
    got = gen_noise_image.noise_1f_frames(3, rows=rows, width=width, normals=normals, ctx=gpu_context())
    for f in range(3):
        want = onoise.noise_1f_frame(normals[f], rows, width)
        # another FFT than numpy's pocketfft: equal to rounding (the f32 cast hides almost all of it)
        np.testing.assert_allclose(got[f], want, rtol=0, atol=2e-6 * np.abs(want).max())
        assert np.mean(got[f] != want) < 0.01
    # device deviates: same spectrum -- variance per octave of the row-major time stream is constant (1/f)
    dev = gen_noise_image.noise_1f_frames(4, rows=rows, width=width, seed=9, stream=5, ctx=gpu_context())
    assert dev.shape == (4, rows, width) and not np.array_equal(dev[0], dev[1])
    again = gen_noise_image.noise_1f_frames(4, rows=rows, width=width, seed=9, stream=5, ctx=gpu_context())
    assert np.array_equal(dev, again)
    ref = np.stack([onoise.noise_1f_frame(rng.standard_normal(4 * rows * width), rows, width) for _ in range(4)])

    def octave_power(x):
        p = np.abs(np.fft.rfft(x.reshape(x.shape[0], -1).astype(np.float64), axis=1)) ** 2
        n = p.shape[1]
        edges = 2 ** np.arange(7, int(np.log2(n)))   # octaves with at least 128 modes: the estimate is good to a few %
        return np.array([p[:, a:b].sum(axis=1).mean() for a, b in zip(edges[:-1], edges[1:])])

    ratio = octave_power(dev) / octave_power(ref)
    assert np.all((ratio > 0.8) & (ratio < 1.25)), ratio
    assert abs(dev.std() / ref.std() - 1) < 0.15


def test_injection_and_resampling_against_the_executed_reference_loop(golden):
    """rip_stage_noise_inject and rip_stage_poisson_resample against arrays made by EXECUTING the reference's make_noise_cube with
    recorded deviates (tests/golden/noise_arith.npz, tools/make_goldens.py noise_arith): injected cubes and layers bit for bit, with
    host arrays and with the planes resident in HBM."""
    import json

    from romanimpreprocess_amd.devarray import DevArray

    g = golden("noise_arith")
    ctx = gpu_context()
    rp = json.loads(str(g["read_pattern"]))
    G, nb, ft = len(rp), 4, float(g["frame_time"])
    gain_act = np.ascontiguousarray(np.clip(g["gain"], 1e-4, 1e4)[nb:-nb, nb:-nb])
    pinfo = {"meta": {"tbar": g["tbar"]}, "weights": g["weights"], "exclude_first": True, "endslice": g["endslice"]}
    w, has, endslice = gen_noise_image.ramp_weight_vectors(pinfo, G)
    inj_a = gen_noise_image.inject_read_noise(g["cube"], g["read"], rp, normals=g["normals"][0], ctx=ctx)
    assert_same_bits(inj_a, g["injected_Ra"], "cube after the injection ('Ra')")
    dark_as_data = np.ascontiguousarray(g["dark"].astype(np.uint16)[1:])
    inj_d = gen_noise_image.inject_read_noise(dark_as_data, g["read"], rp, normals=g["normals"][1], ctx=ctx)
    assert_same_bits(inj_d, g["injected_R"], "dark cube after the injection ('R')")
    zero = np.zeros_like(g["withsky"])
    lay2 = gen_noise_image.poisson_resample(zero.copy(), g["withsky"], gain_act, ft, rp, w, has, endslice, samples=g["poisson"][0], ctx=ctx)
    assert_same_bits(lay2, g["noise"][2], "layer 'Pr'")
    lay3 = gen_noise_image.poisson_resample(zero.copy(), g["sky_b1"], gain_act, ft, rp, w, has, endslice, samples=g["poisson"][1], ctx=ctx)
    assert_same_bits(lay3, g["noise"][3], "layer 'Pb1r'")
    # the same with every plane resident in HBM (the layer loop's form)
    dev = torch.device("cuda", ctx.device)
    t_diff = torch.zeros(g["withsky"].shape, dtype=torch.float32, device=dev)
    t_sky, t_gain, t_end = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (g["withsky"], gain_act, endslice))
    torch.cuda.synchronize()
    gen_noise_image.poisson_resample(DevArray(t_diff), DevArray(t_sky), DevArray(t_gain), ft, rp, w, has, DevArray(t_end),
                                     samples=g["poisson"][0], ctx=ctx)
    assert_same_bits(t_diff.cpu().numpy(), g["noise"][2], "layer 'Pr' (HBM-resident planes)")
    t_cube = torch.from_numpy(g["cube"].view(np.int16)).to(dev)
    t_read = torch.from_numpy(g["read"]).to(dev)
    nreads = np.array([len(r) for r in rp], dtype=np.int32)
    nrm = np.ascontiguousarray(g["normals"][0], dtype=np.float32)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.rip_stage_noise_inject(ctx.h, t_cube.data_ptr(), G, 64, 64, nb, t_read.data_ptr(), nreads.ctypes.data,
                                             nrm.ctypes.data, 0, 0, t_cube.data_ptr()))
    assert_same_bits(t_cube.cpu().numpy().view(np.uint16), g["injected_Ra"], "injection in place in HBM")


def test_1f_frames_from_both_transforms_agree():
    """The same device deviates through the hand-written two-pass transform and through hipFFT (option pink_form = 0): frames of
    2^20 points equal to rounding -- a handful of f32 samples in 10^7 differ by one unit in the last place."""
    ctx = gpu_context()
    rows, width, n = 4096, 128, 12
    dev = torch.device("cuda", ctx.device)
    outs = []
    try:
        for form in (0, -1):
            ctx.set_option("pink_form", form)
            out = torch.empty((n, rows, width), dtype=torch.float32, device=dev)
            ctx.check(ctx.lib.rip_synth_noise_1f(ctx.h, rows, width, n, 123, 9, out.data_ptr()))
            ctx.synchronize()
            outs.append(out)
    finally:
        ctx.set_option("pink_form", -1)
    a, b = outs
    scale = float(a.abs().max())
    assert float((a - b).abs().max()) <= 4e-7 * scale
    assert float((a != b).double().mean()) < 1e-4
    assert 0.5 < float(a.std()) < 50.0 and not torch.equal(a[0], a[1])
