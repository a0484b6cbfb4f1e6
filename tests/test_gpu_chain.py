"""The whole chain (rip_calibrate) on the GPU vs the CPU oracle on seeded synthetic ramps."""

import numpy as np
import pytest
import torch  # before libromanhip is loaded: torch brings its own copy of the HIP runtime, and the first one loaded must be the one both use
from conftest import assert_same_bits, gpu_context

import oracle
from oracle import saturation
from romanimpreprocess_amd import pipeline, synth

pytestmark = pytest.mark.gpu

CASES = [
    # name, (ny, nx), read pattern, p_order, gain dtype, ipc dtype
    ("g8_f32", (64, 256), synth.READ_PATTERN_8, 8, np.float32, np.float32),
    ("g6_prod_dtypes", (48, 384), synth.READ_PATTERN_6, 3, np.float32, np.float64),
    ("g16_test_dtypes", (40, 128), synth.READ_PATTERN_16, 10, np.float64, np.float32),
]


def _oracle_lines(out, G, nch):
    """(G, nch, 2) LAPACK (m, c) the oracle used for the science channels."""
    lines = np.zeros((G, nch, 2))
    for g in range(G):
        lines[g] = out["refpix_diag"][g]["channels"][:nch, 2:4]
    return lines


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("name,shape,rp,p,gdt,kdt", CASES)
def test_chain_vs_oracle(name, shape, rp, p, gdt, kdt, fused):
    """fused = 1: the single fused kernel (chain.hip) where the configuration allows it; 0: stage kernels."""
    ny, nx = shape
    gpu_context().set_option("fused", fused)
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=p, seed=77, gain_dtype=gdt, ipc_dtype=kdt,
                            bias_amplitude=2.0, bad_lin_frac=0.01)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=78, cr_frac=0.02)
    area = 1.0 + 0.01 * np.cos(np.arange(ny * nx, dtype=np.float64).reshape(ny, nx) / 50.0)
    ref = oracle.calibrate_arrays(ramp, cal, area_factor=area)

    cb = pipeline.Calibrator(ctx=gpu_context())
    cb.load_caldir(3, cal)
    # (1) with LAPACK's channel lines handed in: everything bit-identical to the oracle
    lines = _oracle_lines(ref, len(rp), nx // 128)
    got = cb.calibrate(3, ramp, area_factor=area, want_cube=True, channel_lines=lines)
    assert_same_bits(got["K"], ref["K"], "K")
    assert_same_bits(got["cube"], ref["data"], "corrected cube", zero_sign_ok=True)
    assert_same_bits(got["groupdq"], ref["groupdq"], "groupdq")
    assert_same_bits(got["pixeldq"], ref["pixeldq"], "pixeldq")
    for k in ("slope", "err_read", "err_poisson"):
        assert_same_bits(got[k], ref[k], k, zero_sign_ok=True)
    assert np.count_nonzero(got["pixeldq"] & 4) > 5 and np.count_nonzero(got["pixeldq"] & 2) > 5

    # (2) channel lines fitted on the device: DQ identical, floats within the north-star tolerance
    got2 = cb.calibrate(3, ramp, area_factor=area)
    assert_same_bits(got2["groupdq"], ref["groupdq"], "groupdq (device lines)")
    assert_same_bits(got2["pixeldq"], ref["pixeldq"], "pixeldq (device lines)")
    # tolerance: 1e-5 relative on the slope (BASELINE.json north star), errors relative to the total error
    np.testing.assert_allclose(got2["slope"], ref["slope"], rtol=1e-5, atol=1e-7)
    tot = np.hypot(ref["err_read"], ref["err_poisson"])
    assert np.all(np.abs(got2["err_read"] - ref["err_read"]) <= 1e-5 * tot + 1e-12)
    assert np.all(np.abs(got2["err_poisson"] - ref["err_poisson"]) <= 1e-5 * tot + 1e-12)
    cb.ctx.drop_caldir(3)
    gpu_context().set_option("fused", 1)


def test_stage_subsets_and_f32_input():
    """stage mask: reduced chain of BASELINE config 1 (ramp fit + dark rate on an already-corrected f32 cube)."""
    rp = synth.READ_PATTERN_8
    ny, nx = 40, 128
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=3, seed=5)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=6, cr_frac=0.02)
    full = oracle.calibrate_arrays(ramp, cal)
    from oracle import finish, rampfit
    meta = rampfit.ma_table_meta(rp, synth.FRAME_TIME)
    meta["nborder"] = 4
    meta["K"] = full["K"]
    rdq = ramp["groupdq"].copy()
    pdq = ramp["pixeldq"].copy()
    s, er, ep = rampfit.ramp_fit(full["data"], rdq, pdq, cal["gain"]["data"], cal["read"]["data"], meta, True, None)
    dark_rate = finish.dark_rate_deconvolved(cal["dark"]["dark_slope"], cal["ipc4d"]["data"], cal["gain"]["data"])
    s, er, ep = finish.finish(s, er, ep, pdq, 4, dark_rate, None, None, None)

    cb = pipeline.Calibrator(ctx=gpu_context())
    cb.load_caldir(0, cal)
    r2 = dict(ramp)
    r2["data"] = full["data"]  # f32 cube, already corrected
    got = cb.calibrate(0, r2, stages=pipeline.STAGE_RAMPFIT | pipeline.STAGE_DARK)
    assert_same_bits(got["pixeldq"], pdq, "pixeldq")
    assert_same_bits(got["groupdq"], rdq, "groupdq")
    assert_same_bits(got["slope"], s, "slope", zero_sign_ok=True)
    assert_same_bits(got["err_read"], er, "err_read", zero_sign_ok=True)
    assert_same_bits(got["err_poisson"], ep, "err_poisson", zero_sign_ok=True)
    # cube-only sub-chain: refpix + bias + linearity, no fit
    lines = _oracle_lines(full, len(rp), nx // 128)
    part = cb.calibrate(0, ramp, stages=pipeline.STAGE_REFPIX | pipeline.STAGE_BIAS | pipeline.STAGE_LIN, want_cube=True,
                        channel_lines=lines)
    assert "slope" not in part
    ref_lin = oracle.calibrate_arrays(ramp, {k: v for k, v in cal.items() if k != "ipc4d"})
    assert_same_bits(part["cube"], ref_lin["data"], "cube after linearity", zero_sign_ok=True)


def test_bad_arguments():
    cb = pipeline.Calibrator(ctx=gpu_context())
    rp = synth.READ_PATTERN_6
    cal = synth.make_caldir(32, 128, read_pattern=rp, p_order=3, seed=1)
    cb.load_caldir(1, cal)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=2)
    bad = dict(ramp)
    bad["data"] = ramp["data"][:, :, :64]
    with pytest.raises(ValueError):
        cb.calibrate(1, bad)
    with pytest.raises(ValueError):
        cb.ctx.drop_caldir(9)
    with pytest.raises(KeyError):
        cb.calibrate(7, ramp)  # never loaded


def test_fused_exact_everywhere_gives_same_flags():
    """guard = inf makes the fused kernel re-evaluate EVERY jump significance in the reference's exact order;
    the default (approximate evaluation + error band + exact inside the band) must give identical outputs."""
    rp = synth.READ_PATTERN_8
    cal = synth.make_caldir(64, 256, read_pattern=rp, p_order=8, seed=21, bias_amplitude=2.0)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=22, cr_frac=0.05)
    ctx = gpu_context()
    cb = pipeline.Calibrator(ctx=ctx)
    cb.load_caldir(2, cal)
    a = cb.calibrate(2, ramp)
    ctx.set_option_f64("guard_band", float("inf"))
    try:
        b = cb.calibrate(2, ramp)
    finally:
        ctx.set_option_f64("guard_band", 1e-5)
    for k in ("slope", "err_read", "err_poisson", "pixeldq", "groupdq"):
        assert_same_bits(a[k], b[k], k)
    assert np.count_nonzero(a["pixeldq"] & 4) > 100
    cb.ctx.drop_caldir(2)


def test_calibrateimage_files_end_to_end(tmp_path):
    """config dict -> ASDF L1 + CALDIR files -> calibrateimage -> ASDF L2, against the oracle chain."""
    from romanimpreprocess_amd import calio
    from romanimpreprocess_amd.L1_to_L2 import gen_cal_image

    rp = synth.READ_PATTERN_6
    ny, nx = 48, 256
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=3, seed=31, bias_amplitude=1.0)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=32, cr_frac=0.02)
    caldir = {}
    names = {"dark": "dark", "read": "read", "gain": "gain", "linearitylegendre": "linearitylegendre", "ipc4d": "ipc4d",
             "flat": "pflat", "biascorr": "biascorr", "mask": "mask", "saturation": "saturation"}
    for key, fname in names.items():
        path = tmp_path / f"roman_wfi_{fname}_TEST_SCA04.{'npz' if key == 'biascorr' else 'asdf'}"
        if key == "biascorr":
            calio.save_npz_tree(str(path), {"roman": cal[key]})
        else:
            calio.write_asdf(str(path), {"roman": cal[key]})
        caldir[key] = str(path)
    l1 = {"roman": {"data": ramp["data"], "amp33": ramp["amp33"],
                    "meta": {"exposure": {"frame_time": synth.FRAME_TIME, "read_pattern": rp},
                             "instrument": {"detector": "WFI04"}}}}
    calio.write_asdf(str(tmp_path / "l1.asdf"), l1)
    config = {"IN": str(tmp_path / "l1.asdf"), "OUT": str(tmp_path / "l2.asdf"), "CALDIR": caldir,
              "JUMP_DETECT_PARS": {"SthreshA": 5.0, "IthreshB": 800.0}, "SLICEOUT": True}
    cb_files = pipeline.Calibrator(ctx=gpu_context())
    gen_cal_image.calibrateimage(config, verbose=False, calibrator=cb_files)
    out = calio.read_asdf(config["OUT"])

    # oracle on the same inputs (dq-init + this package's saturation flagging are host steps shared by both)
    r0 = {"data": ramp["data"], "amp33": ramp["amp33"], "groupdq": np.zeros(ramp["data"].shape, np.uint8),
          "pixeldq": cal["mask"]["dq"].copy(), "read_pattern": rp, "frame_time": synth.FRAME_TIME}
    r0["groupdq"][0] |= 1
    saturation.flag_saturation(r0, cal["saturation"]["data"], backup=1, skip_firstn=1, sat_dq=cal["saturation"]["dq"],
                               read_pattern=rp)   # the reference hands the read pattern to stcal (gen_cal_image.py:172-185)
    ref = oracle.calibrate_arrays(r0, cal, jump_pars=config["JUMP_DETECT_PARS"])
    act = (slice(4, -4), slice(4, -4))
    assert_same_bits(out["roman"]["dq"], ref["pixeldq"][act], "L2 dq")
    np.testing.assert_allclose(out["roman"]["data"], ref["slope"][act], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out["roman"]["var_poisson"], ref["err_poisson"][act] ** 2, rtol=3e-5, atol=1e-12)
    assert out["processinfo"]["exclude_first"] is True and "endslice" in out["processinfo"]
    assert_same_bits(np.asarray(out["processinfo"]["weights"], dtype=np.float32), ref["K"], "weights")
    with pytest.raises(KeyError):
        gen_cal_image.calibrateimage({"IN": config["IN"], "OUT": config["OUT"], "CALDIR": {"gain": caldir["gain"]}},
                                     verbose=False)
    # post-path reductions: sky mode of the masked, binned image; endslice; optional sky model
    from romanimpreprocess_amd.utils import maskhandling, sky
    pi = out["processinfo"]
    m = maskhandling.PixelMask1.build(ref["pixeldq"], ctx=gpu_context())
    want_sky, _ = sky.smooth_mode(sky.binkxk(ref["slope"], 4, mask=m, ctx=gpu_context()), ctx=gpu_context())
    np.testing.assert_allclose(pi["medsky"], want_sky, rtol=1e-4)
    assert pi["skyorder"] == -1 and np.asarray(pi["skycoefs"]).size == 0
    assert np.asarray(pi["endslice"]).dtype == np.int8 and np.asarray(pi["endslice"]).shape == out["roman"]["data"].shape
    np.testing.assert_allclose(pi["medgain"], np.median(cal["gain"]["data"]), rtol=0, atol=0)
    cfg2 = dict(config, SKYORDER=2, OUT=str(tmp_path / "l2_sky.asdf"))
    gen_cal_image.calibrateimage(cfg2, verbose=False, calibrator=cb_files)
    out2 = calio.read_asdf(cfg2["OUT"])
    assert out2["processinfo"]["skyorder"] == 2 and np.asarray(out2["processinfo"]["skycoefs"]).shape == (6,)
    assert_same_bits(np.asarray(out2["roman"]["data_withsky"]), np.asarray(out["roman"]["data"]), "data before the sky model")
    coef, model = sky.medfit(np.asarray(out2["roman"]["data_withsky"]), order=2, ctx=gpu_context())
    assert_same_bits(np.asarray(out2["roman"]["data"]), np.asarray(out2["roman"]["data_withsky"]) - model, "sky-subtracted data",
                     zero_sign_ok=True)
    with pytest.raises(NotImplementedError):
        gen_cal_image.calibrateimage(dict(config, romancal_ramp_fit=True), verbose=False)


# ---- the fused kernel (chain2_kernel.h): every instantiation the dispatcher can pick (256-column form, narrow forms), and the
# seams between their column strips / row ranges


def _set_form(ctx, form):
    """2: the fused kernel, 0: stage kernels"""
    ctx.set_option("fused", 1 if form else 0)
    ctx.set_option("chain2", 1 if form >= 2 else 0)


def _default_form(ctx):
    """the library's defaults: fused, the specialised kernels where they apply"""
    ctx.set_option("fused", 1)
    ctx.set_option("chain2", 1)


SPECIALISED = [
    # name, (ny, nx), read pattern, p_order (NP = p + 1 planes), exclude_first, ipc4d dtype
    ("g6_np9_start1", (48, 384), synth.READ_PATTERN_6, 8, True, np.float32),
    ("g6_np4_start0", (40, 256), synth.READ_PATTERN_6, 3, False, np.float32),
    ("g8_np11_start1", (40, 256), synth.READ_PATTERN_8, 10, True, np.float32),
    ("g8_np9_start0", (56, 256), synth.READ_PATTERN_8, 8, False, np.float32),
    ("g8_np4_start1", (40, 128), synth.READ_PATTERN_8, 3, True, np.float32),
    ("g16_np9_start1", (40, 256), synth.READ_PATTERN_16, 8, True, np.float32),
    ("g16_np4_start0", (32, 128), synth.READ_PATTERN_16, 3, False, np.float32),
    # f64 ipc4d (the reference's production writer): f64 Neumann iterates
    ("g8_np9_start1_k64", (56, 384), synth.READ_PATTERN_8, 8, True, np.float64),
    ("g8_np11_start0_k64", (40, 256), synth.READ_PATTERN_8, 10, False, np.float64),
    ("g6_np4_start1_k64", (48, 128), synth.READ_PATTERN_6, 3, True, np.float64),
    ("g6_np9_start0_k64", (40, 256), synth.READ_PATTERN_6, 8, False, np.float64),
    # f64 ipc4d x 16 groups (narrow form, two workgroups per CU)
    ("g16_np9_start1_k64", (40, 256), synth.READ_PATTERN_16, 8, True, np.float64),
    ("g16_np11_start0_k64", (32, 256), synth.READ_PATTERN_16, 10, False, np.float64),
]


@pytest.mark.parametrize("name,shape,rp,p,exclude_first,kdt", SPECIALISED)
def test_specialised_kernel_vs_oracle(name, shape, rp, p, exclude_first, kdt):
    ny, nx = shape
    form = 2
    ctx = gpu_context()
    _set_form(ctx, 2)
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=p, seed=91, bias_amplitude=2.0, bad_lin_frac=0.01,
                            ipc_dtype=kdt)
    # degenerate gains: the waves holding them leave the shared-reciprocal division for the division operator
    cal["gain"]["data"][20, 30] = 0.0
    cal["gain"]["data"][21, 40] = 1e-25
    cal["gain"]["data"][22, 50] = -1.5
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=92, cr_frac=0.03)
    with np.errstate(all="ignore"):
        ref = oracle.calibrate_arrays(ramp, cal, exclude_first=exclude_first)
    cb = pipeline.Calibrator(ctx=ctx)
    cb.load_caldir(4, cal)
    lines = _oracle_lines(ref, len(rp), nx // 128)
    got = cb.calibrate(4, ramp, exclude_first=exclude_first, want_cube=True, channel_lines=lines)
    assert ctx.last_chain_form() == form, "the requested fused kernel did not run"
    _default_form(ctx)
    assert_same_bits(got["cube"], ref["data"], "corrected cube", zero_sign_ok=True)
    assert_same_bits(got["groupdq"], ref["groupdq"], "groupdq")
    assert_same_bits(got["pixeldq"], ref["pixeldq"], "pixeldq")
    for k in ("slope", "err_read", "err_poisson"):
        assert_same_bits(got[k], ref[k], k, zero_sign_ok=True)
    assert np.count_nonzero(got["pixeldq"] & 4) > 5
    cb.ctx.drop_caldir(4)


@pytest.mark.parametrize("rp,kdt", [(synth.READ_PATTERN_8, np.float32), (synth.READ_PATTERN_8, np.float64),
                                    (synth.READ_PATTERN_16, np.float32), (synth.READ_PATTERN_16, np.float64)],
                         ids=["g8_f32", "g8_k64", "g16_f32", "g16_k64"])
def test_fused_forms_agree_across_seams(rp, kdt):
    """A frame wider than several column strips and taller than several row ranges: the wave-specialised kernel and the stage
    kernels must give identical bits (halo columns, range boundaries, frame edges) -- the 256-column form and every narrow
    form (f64 ipc4d, 16 groups, both)."""
    ny, nx = 1160, 896
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=31, bias_amplitude=2.0, bad_lin_frac=0.005,
                            ipc_dtype=kdt)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=32, cr_frac=0.02)
    ctx = gpu_context()
    cb = pipeline.Calibrator(ctx=ctx)
    cb.load_caldir(5, cal)
    outs = []
    try:
        for form in (2, 0):
            _set_form(ctx, form)
            outs.append(cb.calibrate(5, ramp, want_cube=True))
            assert ctx.last_chain_form() == form
    finally:
        _default_form(ctx)
    for other, label in ((outs[1], "stage kernels"),):
        for k in ("cube", "slope", "err_read", "err_poisson", "pixeldq", "groupdq"):
            assert_same_bits(outs[0][k], other[k], f"{k}: wave-specialised vs {label}")
    assert np.count_nonzero(outs[0]["pixeldq"] & 4) > 1000 and np.count_nonzero(outs[0]["pixeldq"] & 2) > 100
    cb.ctx.drop_caldir(5)


def test_saturation_flagging_on_device_matches_host_restatement():
    """dq-init + saturation flagging inside rip_calibrate against the numpy restatement
    oracle.saturation.flag_saturation (parity unpinned: stcal's source is not in the reference tree): identical
    flags, hence identical chain outputs.  Thresholds lowered so that many pixels saturate at different groups; some
    pixels are NO_SAT_CHECK / NaN-threshold."""
    from romanimpreprocess_amd.dqflags import pixel

    rp = synth.READ_PATTERN_8
    ny, nx = 72, 256
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=41, bias_amplitude=2.0)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=42, cr_frac=0.02)
    rng = np.random.default_rng(7)
    thr = np.quantile(ramp["data"].astype(np.float32), 0.9, axis=0).astype(np.float32)  # ~10 % of the resultants exceed it
    thr[rng.random((ny, nx)) < 0.97] = 65535.0  # ... at 3 % of the pixels
    thr[3, 5] = np.nan
    sdq = np.zeros((ny, nx), np.uint32)
    sdq[rng.random((ny, nx)) < 0.01] = np.uint32(pixel.NO_SAT_CHECK)
    cal = dict(cal)
    cal["saturation"] = {"data": thr, "dq": sdq}
    mask = ramp["pixeldq"].copy()
    for backup in (1, 2, 0):
        # host restatement
        h = {"data": ramp["data"], "groupdq": np.zeros(ramp["data"].shape, np.uint8), "pixeldq": mask.copy()}
        saturation.flag_saturation(h, thr, backup=backup, skip_firstn=1, n_pix_grow_sat=1, sat_dq=sdq)
        assert 50 < np.count_nonzero(h["groupdq"][-1] & 2) < 0.5 * ny * nx
        cb = pipeline.Calibrator(ctx=gpu_context())
        cb.load_caldir(6, cal)
        r_host = dict(ramp, groupdq=h["groupdq"], pixeldq=h["pixeldq"])
        a = cb.calibrate(6, r_host)
        r_dev = dict(ramp, groupdq=None, pixeldq=mask)
        b = cb.calibrate(6, r_dev, flag_saturation=True, saturation_backup=backup)
        for k in ("groupdq", "pixeldq", "slope", "err_read", "err_poisson"):
            assert_same_bits(a[k], b[k], f"{k} (backup {backup})")
        # read-pattern rule (groups averaging several reads are compared with threshold * mean(reads) / last read): groups whose
        # later reads alone saturate sit between the diluted and the full threshold -- flagged only with the rule on
        h2 = {"data": ramp["data"], "groupdq": np.zeros(ramp["data"].shape, np.uint8), "pixeldq": mask.copy()}
        saturation.flag_saturation(h2, thr, backup=backup, skip_firstn=1, n_pix_grow_sat=1, sat_dq=sdq, read_pattern=rp)
        more = np.count_nonzero(h2["groupdq"] & 2) - np.count_nonzero(h["groupdq"] & 2)
        assert more > 20, "the synthetic thresholds do not exercise partially saturated groups"
        c = cb.calibrate(6, r_dev, flag_saturation=True, saturation_backup=backup, saturation_read_pattern=True)
        d = cb.calibrate(6, dict(ramp, groupdq=h2["groupdq"], pixeldq=h2["pixeldq"]))
        for k in ("groupdq", "pixeldq", "slope", "err_read", "err_poisson"):
            assert_same_bits(c[k], d[k], f"{k} (backup {backup}, read-pattern rule)")
        cb.ctx.drop_caldir(6)


FULL_FRAME = [
    # name, read pattern, ipc4d dtype, P_ORDER, rows the oracle runs on
    ("g8_f32", synth.READ_PATTERN_8, np.float32, 8, 4096),     # BASELINE config 2: the bench workload
    ("g8_k64", synth.READ_PATTERN_8, np.float64, 8, 4096),     # ... with the f64 ipc4d of production CALDIR sets
    # BASELINE config 3 (READS = [0..35], 16 groups): the numpy oracle needs more than 7 minutes for a 16-group full frame
    # (13 truncated refits, O(G^2) variance passes), so it checks a 264-row frame of the full width and the full frame is
    # checked between the fused kernel and the stage kernels
    ("g16_f32", synth.READ_PATTERN_16, np.float32, 8, 264),
    # f64 ipc4d x 16 groups: the 76 KB narrow form at 33 strips of the full width (frame-edge lanes emitting, 2 workgroups per CU)
    ("g16_k64", synth.READ_PATTERN_16, np.float64, 8, 264),
]


@pytest.mark.parametrize("name,rp,kdt,p,oracle_rows", FULL_FRAME)
def test_full_frame_4096x4096_vs_oracle_and_between_forms(name, rp, kdt, p, oracle_rows):
    """BASELINE configs 2 and 3 at their full size on a NON-PERIODIC frame (sky + 25 Gaussian sources, seeded, generated
    on the device by synth_gpu): the numpy oracle on the whole frame (about a minute of CPU) against the default fused kernel,
    bit for bit with LAPACK's channel lines handed in; then the fused kernel against the stage kernels against each other with the lines fitted on the device."""
    from romanimpreprocess_amd import synth_gpu

    n = 4096
    ctx = gpu_context()
    cb = pipeline.Calibrator(ctx=ctx)
    if oracle_rows < n:
        cal_s = synth_gpu.make_caldir(oracle_rows, n, read_pattern=rp, p_order=p, seed=1002, ipc_dtype=kdt)
        ramp_s = synth_gpu.make_ramp(cal_s, read_pattern=rp, seed=2, cr_frac=0.01)
        ref_s = oracle.calibrate_arrays(ramp_s, cal_s)
        cb.load_caldir(6, cal_s)
        _default_form(ctx)
        got_s = cb.calibrate(6, ramp_s, channel_lines=_oracle_lines(ref_s, len(rp), n // 128))
        assert ctx.last_chain_form() == 2
        for k in ("groupdq", "pixeldq", "slope", "err_read", "err_poisson"):
            assert_same_bits(got_s[k], ref_s[k], f"{k} ({oracle_rows}-row frame)", zero_sign_ok=True)
        assert np.count_nonzero(got_s["pixeldq"] & 4) > 1000
        cb.ctx.drop_caldir(6)
        del cal_s, ramp_s, ref_s, got_s
    cal = synth_gpu.make_caldir(n, n, read_pattern=rp, p_order=p, seed=1001, ipc_dtype=kdt)
    ramp = synth_gpu.make_ramp(cal, read_pattern=rp, seed=1)
    cb.load_caldir(6, cal)
    try:
        _default_form(ctx)
        if oracle_rows == n:
            ref = oracle.calibrate_arrays(ramp, cal)
            got = cb.calibrate(6, ramp, channel_lines=_oracle_lines(ref, len(rp), n // 128))
            assert ctx.last_chain_form() == 2   # (f64 ipc4d too: with its f64 chains batched the wave-specialised kernel is the faster one)
            assert_same_bits(got["groupdq"], ref["groupdq"], "groupdq")
            assert_same_bits(got["pixeldq"], ref["pixeldq"], "pixeldq")
            for k in ("slope", "err_read", "err_poisson"):
                assert_same_bits(got[k], ref[k], k, zero_sign_ok=True)
            del ref
        else:
            got = cb.calibrate(6, ramp)
        frac_good = np.mean(got["pixeldq"][4:-4, 4:-4] == 0)
        assert frac_good > 0.9 and np.count_nonzero(got["pixeldq"] & 4) > 10000
        # the sources are there
        assert np.count_nonzero(ramp["rate"][4:-4, 4:-4] > 100.0) > 200
        outs = []
        for form in (2, 0):
            _set_form(ctx, form)
            outs.append(cb.calibrate(6, ramp))
            assert ctx.last_chain_form() == form
        for other, label in ((outs[1], "stage kernels"),):
            for k in ("slope", "err_read", "err_poisson", "pixeldq", "groupdq"):
                assert_same_bits(outs[0][k], other[k], f"{k}: wave-specialised vs {label}")
        # device-fitted lines against LAPACK's: flags identical, slopes within the north-star tolerance
        assert_same_bits(outs[0]["pixeldq"], got["pixeldq"], "pixeldq (device lines)")
        np.testing.assert_allclose(outs[0]["slope"], got["slope"], rtol=1e-5, atol=1e-7)
    finally:
        _default_form(ctx)
        cb.ctx.drop_caldir(6)


def test_preallocated_and_page_locked_host_arrays():
    """``out=`` fills the caller's arrays; page-locked arrays from the library behave like any numpy array."""
    rp = synth.READ_PATTERN_8
    ny, nx = 40, 256
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=3, seed=3)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=4, cr_frac=0.02)
    cb = pipeline.Calibrator(ctx=gpu_context())
    cb.load_caldir(7, cal)
    ref = cb.calibrate(7, ramp)
    pinned = {k: cb.pinned_empty(v.shape, v.dtype) for k, v in ramp.items() if isinstance(v, np.ndarray)}
    for k, v in pinned.items():
        v[...] = ramp[k]
    out = {k: cb.pinned_empty(ref[k].shape, ref[k].dtype) for k in ("slope", "err_read", "err_poisson", "pixeldq", "groupdq")}
    got = cb.calibrate(7, dict(ramp, **pinned), out=out)
    for k in out:
        assert got[k] is out[k]
        assert_same_bits(out[k], ref[k], k)
    # a groupdq without DO_NOT_USE on the first group: the library sets it on its own copy, the caller's array stays
    bare = ramp["groupdq"].copy()
    bare[0] &= np.uint8(0xFE)
    keep = bare.copy()
    got2 = cb.calibrate(7, dict(ramp, groupdq=bare))
    assert np.array_equal(bare, keep)
    assert_same_bits(got2["slope"], ref["slope"], "slope (first group flagged by the library)")
    with pytest.raises(ValueError):
        cb.calibrate(7, ramp, out={"slope": np.empty((ny, nx), np.float64)})
    cb.ctx.drop_caldir(7)


UNUSUAL = [
    # name, read pattern, exclude_first
    ("g3", [[0], [1, 2], [3, 4, 5, 6]], False),          # the shortest ramp the fit accepts without the first group excluded
    ("g4_excl", [[0], [1], [2, 3], [4, 5, 6, 7]], True),
    ("g5_odd", [[0], [1, 2], [3, 4], [5, 6, 7, 8], [9]], True),
    ("g7_odd", [[0], [1], [2, 3], [4, 5, 6], [7, 8, 9, 10], [11, 12], [13]], True),
    ("g12", [[i] for i in range(4)] + [[4 + 2 * i, 5 + 2 * i] for i in range(8)], True),
]


@pytest.mark.parametrize("name,rp,exclude_first", UNUSUAL)
def test_unusual_group_counts_vs_oracle(name, rp, exclude_first):
    """Group counts outside the fused kernel's instantiations run through the stage kernels and still match the oracle bit for
    bit."""
    ny, nx = 40, 256
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=3, seed=61, bias_amplitude=1.0)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=62, cr_frac=0.03)
    ref = oracle.calibrate_arrays(ramp, cal, exclude_first=exclude_first)
    ctx = gpu_context()
    _default_form(ctx)
    cb = pipeline.Calibrator(ctx=ctx)
    cb.load_caldir(8, cal)
    got = cb.calibrate(8, ramp, exclude_first=exclude_first, want_cube=True, channel_lines=_oracle_lines(ref, len(rp), nx // 128))
    assert_same_bits(got["cube"], ref["data"], "corrected cube", zero_sign_ok=True)
    assert_same_bits(got["groupdq"], ref["groupdq"], "groupdq")
    assert_same_bits(got["pixeldq"], ref["pixeldq"], "pixeldq")
    for k in ("slope", "err_read", "err_poisson"):
        assert_same_bits(got[k], ref[k], k, zero_sign_ok=True)
    cb.ctx.drop_caldir(8)


@pytest.mark.parametrize("flag_sat", [False, True])
def test_batch_of_host_ramps_equals_single_calls(flag_sat):
    """rip_calibrate_batch (upload / chain / download of consecutive ramps overlapped) against one call per ramp; with device-side
    saturation flagging also with the read-pattern rule of calibrateimage on (it changes several hundred flags of these ramps)."""
    rp = synth.READ_PATTERN_8
    ny, nx = 72, 256
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=41, bias_amplitude=2.0)
    cb = pipeline.Calibrator(ctx=gpu_context())
    cb.load_caldir(2, cal)
    ramps = []
    for i in range(5):
        r = synth.make_ramp(cal, read_pattern=rp, seed=50 + i, cr_frac=0.02)
        if flag_sat:
            r["groupdq"] = None
            r["pixeldq"] = cal["mask"]["dq"].copy()
        ramps.append(r)
    kept = {}
    for rule in ((False, True) if flag_sat else (False,)):
        singles = [cb.calibrate(2, r, flag_saturation=flag_sat, saturation_read_pattern=rule) for r in ramps]
        pinned_out = [{k: cb.pinned_empty((ny, nx), np.float32) for k in ("slope", "err_read", "err_poisson")} for _ in ramps]
        many = cb.calibrate_many(2, ramps, want_groupdq=True, flag_saturation=flag_sat, out=pinned_out, saturation_read_pattern=rule)
        assert len(many) == len(ramps)
        for i, (a, b) in enumerate(zip(many, singles)):
            assert a["slope"] is pinned_out[i]["slope"]
            for k in ("slope", "err_read", "err_poisson", "pixeldq", "groupdq"):
                assert_same_bits(a[k], b[k], f"ramp {i}: {k} (read-pattern rule {rule})")
        kept[rule] = many[0]["groupdq"].copy()
    if flag_sat:   # the rule does act on these ramps
        assert np.count_nonzero(kept[False] != kept[True]) > 100
    assert cb.calibrate_many(2, []) == []
    cb.ctx.drop_caldir(2)


@pytest.mark.parametrize("inputs_complete", [True, False])
@pytest.mark.parametrize("flag_sat", [False, True])
def test_back_to_back_device_calls_without_sync(flag_sat, inputs_complete):
    """Different device-resident ramps issued back to back with no synchronisation in between (the reference-pixel pre-pass and
    the saturation pass of call n+1 run ahead on the second stream while the chain of call n is still reading ITS tables and
    flag copies: they are double-buffered by call parity, api.hip): results must equal those of synchronised single calls."""
    dev = torch.device("cuda", 0)
    rp = synth.READ_PATTERN_8
    ny, nx = 1024, 1024
    cal, _ = synth.make_tiled_inputs(ny, nx, read_pattern=rp, p_order=8, seed=3, strip_rows=64)
    cal = dict(cal)
    thr = np.full((ny, nx), 50000.0, np.float32)
    cal["saturation"] = {"data": thr, "dq": np.zeros((ny, nx), np.uint32)}
    ctx = gpu_context()
    _default_form(ctx)
    cb = pipeline.Calibrator(ctx=ctx)
    cb.load_caldir(9, cal)
    pid, _meta = cb.plan_for(rp, synth.FRAME_TIME)

    def to_dev(a):
        a = np.ascontiguousarray(a)
        view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32}.get(a.dtype)
        return torch.from_numpy(a.view(view) if view else a).to(dev)

    n = 5
    ramps = []
    for i in range(n):
        _, r = synth.make_tiled_inputs(ny, nx, read_pattern=rp, p_order=8, seed=3, strip_rows=64, ramp_seed=100 + i)
        g = r["groupdq"].copy()
        g[0] |= 1
        # distinct reference-output levels: the row corrections of consecutive ramps differ by much more than rounding
        a33 = (r["amp33"].astype(np.int32) + 40 * i * (np.arange(ny)[None, :, None] % 7)).astype(np.uint16)
        ramps.append([to_dev(r["data"]), to_dev(a33), None if flag_sat else to_dev(g), to_dev(r["pixeldq"])])
    outs = [[torch.empty((ny, nx), dtype=torch.float32, device=dev) for _ in range(3)] +
            [torch.empty((ny, nx), dtype=torch.int32, device=dev), torch.empty((8, ny, nx), dtype=torch.uint8, device=dev)]
            for _ in range(2 * n)]
    torch.cuda.synchronize()

    def call(i, o):
        t = ramps[i]
        cb.calibrate_device(9, pid, 8, t[0].data_ptr(), True, t[1].data_ptr(), None if t[2] is None else t[2].data_ptr(),
                            t[3].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(),
                            flag_saturation=flag_sat, inputs_complete=inputs_complete)   # True: the pre-pass runs ahead

    try:
        for i in range(n):          # reference: one call at a time
            call(i, outs[i])
            cb.synchronize()
        for rep in range(3):        # asynchronous: all calls queued, one synchronisation at the end
            for i in range(n):
                call(i, outs[n + i])
            cb.synchronize()
            for i in range(n):
                for k, name in enumerate(("slope", "err_read", "err_poisson", "pixeldq", "groupdq")):
                    assert torch.equal(outs[i][k], outs[n + i][k]), f"ramp {i} {name}: queued call differs (repeat {rep})"
        # the ramps do differ from one another (otherwise the test could not see a stale table)
        assert not torch.equal(outs[0][0], outs[1][0])
    finally:
        cb.ctx.drop_caldir(9)


def _small_resident_set(slot, n, ny=512, nx=512, sat=True):
    """a CALDIR set in `slot` and n different device-resident ramps (tensors: data, amp33, groupdq, pixeldq)"""
    dev = torch.device("cuda", 0)
    rp = synth.READ_PATTERN_8
    cal, _ = synth.make_tiled_inputs(ny, nx, read_pattern=rp, p_order=8, seed=3, strip_rows=64)
    cal = dict(cal)
    if sat:
        cal["saturation"] = {"data": np.full((ny, nx), 50000.0, np.float32), "dq": np.zeros((ny, nx), np.uint32)}
    ctx = gpu_context()
    _default_form(ctx)
    cb = pipeline.Calibrator(ctx=ctx)
    cb.load_caldir(slot, cal)
    pid, _meta = cb.plan_for(rp, synth.FRAME_TIME)

    def to_dev(a):
        a = np.ascontiguousarray(a)
        view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32}.get(a.dtype)
        return torch.from_numpy(a.view(view) if view else a).to(dev)

    ramps = []
    for i in range(n):
        _, r = synth.make_tiled_inputs(ny, nx, read_pattern=rp, p_order=8, seed=3, strip_rows=64, ramp_seed=200 + i)
        g = r["groupdq"].copy()
        g[0] |= 1
        a33 = (r["amp33"].astype(np.int32) + 40 * i * (np.arange(ny)[None, :, None] % 7)).astype(np.uint16)
        ramps.append([to_dev(r["data"]), to_dev(a33), to_dev(g), to_dev(r["pixeldq"])])

    def outputs():
        return [torch.empty((ny, nx), dtype=torch.float32, device=dev) for _ in range(3)] + [
            torch.empty((ny, nx), dtype=torch.int32, device=dev), torch.empty((8, ny, nx), dtype=torch.uint8, device=dev)]

    torch.cuda.synchronize()
    return cb, pid, ramps, outputs


def test_mixed_overlap_modes_back_to_back():
    """Calls that run their pre-pass / saturation pass on the MAIN stream (overlap off, or a sub-chain without the reference-pixel
    step that flags saturation) queued between overlapped calls, no synchronisation: every call takes a parity of the
    double-buffered tables and flag copies and leaves its completion event, so a following overlapped call can neither reuse
    nor overwrite buffers a queued kernel still reads (round-2 advisor finding on api.hip)."""
    n = 6
    cb, pid, ramps, outputs = _small_resident_set(9, n)
    no_ref = pipeline.STAGE_ALL & ~pipeline.STAGE_REFPIX
    # (overlap option, stage mask, flag saturation on the device) per call
    modes = [(1, pipeline.STAGE_ALL, True), (0, pipeline.STAGE_ALL, True), (1, pipeline.STAGE_ALL, True),
             (1, no_ref, True), (1, pipeline.STAGE_ALL, True), (0, no_ref, False)]

    def call(i, o):
        t = ramps[i]
        ov, stages, fs = modes[i]
        cb.ctx.set_option("overlap", ov)
        cb.calibrate_device(9, pid, 8, t[0].data_ptr(), True, t[1].data_ptr(), None if fs else t[2].data_ptr(), t[3].data_ptr(),
                            o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), stages=stages,
                            flag_saturation=fs, inputs_complete=True)

    ref = [outputs() for _ in range(n)]
    got = [outputs() for _ in range(n)]
    try:
        for i in range(n):
            call(i, ref[i])
            cb.synchronize()
        for rep in range(3):
            for i in range(n):
                call(i, got[i])
            cb.synchronize()
            for i in range(n):
                for k, name in enumerate(("slope", "err_read", "err_poisson", "pixeldq", "groupdq")):
                    assert torch.equal(ref[i][k], got[i][k]), f"call {i} {modes[i]} {name}: queued call differs (repeat {rep})"
        assert not torch.equal(ref[0][0], ref[2][0])
    finally:
        cb.ctx.set_option("overlap", 1)
        cb.ctx.drop_caldir(9)


@pytest.mark.parametrize("producer", ["side_stream_event", "library_stream_default"])
def test_inputs_written_by_queued_work_no_sync(producer):
    """The cube and the reference output of a call are written by kernels that are still QUEUED when rip_calibrate is called:
    on a torch side stream, guarded by rip_ramp_desc::ready_event, or on rip_stream() itself with the default
    (stream-ordered) contract.  No host synchronisation anywhere; the result must be that of the synchronised call -- the
    second-stream pre-pass would otherwise read the stale previous contents and produce other row corrections."""
    cb, pid, ramps, outputs = _small_resident_set(9, 3, sat=False)
    dev = torch.device("cuda", 0)
    ref, got = outputs(), outputs()
    stale, fresh = ramps[0], ramps[1]
    work = [torch.empty_like(stale[0]), torch.empty_like(stale[1])]   # the buffers the calls read
    junk = torch.empty((4096, 4096), dtype=torch.float32, device=dev)
    try:
        def call(o, **kw):
            cb.calibrate_device(9, pid, 8, work[0].data_ptr(), True, work[1].data_ptr(), fresh[2].data_ptr(), fresh[3].data_ptr(),
                                o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), **kw)

        work[0].copy_(fresh[0]); work[1].copy_(fresh[1])
        torch.cuda.synchronize()
        call(ref)
        cb.synchronize()
        for rep in range(3):
            work[0].copy_(stale[0]); work[1].copy_(stale[1])
            torch.cuda.synchronize()
            call(outputs())          # a call in flight in front (its main kernel is what an unordered pre-pass would run beside)
            if producer == "side_stream_event":
                side = torch.cuda.Stream(device=dev)
                ev = torch.cuda.Event()
                with torch.cuda.stream(side):
                    for _ in range(20):
                        junk.normal_()      # ~ a millisecond of queued work in front of the copies
                    work[0].copy_(fresh[0]); work[1].copy_(fresh[1])
                    ev.record(side)
                call(got, ready_event=ev.cuda_event)
            else:
                ext = torch.cuda.ExternalStream(cb.ctx.stream, device=dev)
                with torch.cuda.stream(ext):
                    for _ in range(20):
                        junk.normal_()
                    work[0].copy_(fresh[0]); work[1].copy_(fresh[1])
                call(got)             # default: ordered behind everything queued on rip_stream()
            cb.synchronize()
            torch.cuda.synchronize()
            for k, name in enumerate(("slope", "err_read", "err_poisson", "pixeldq", "groupdq")):
                assert torch.equal(ref[k], got[k]), f"{name}: inputs written by queued work were read too early (repeat {rep})"
    finally:
        cb.ctx.drop_caldir(9)


def test_several_caldir_slots_resident_and_interleaved():
    """BASELINE config 4 in small: several CALDIR sets (SCAs) resident at once, ramps of different (filter, SCA) items issued
    interleaved and back to back, every result checked against the oracle run with that item's own calibration set."""
    dev = torch.device("cuda", 0)
    rp = synth.READ_PATTERN_8
    ny, nx = 64, 256
    ctx = gpu_context()
    _default_form(ctx)
    cb = pipeline.Calibrator(ctx=ctx)
    scas = (3, 7, 12, 18)
    cals = {}
    for sca in scas:
        kdt = np.float64 if sca == 12 else np.float32   # one production-style set with f64 ipc4d among them
        cals[sca] = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=5000 + sca, ipc_dtype=kdt, bias_amplitude=1.0)
        cb.load_caldir(sca, cals[sca])
    pid, _meta = cb.plan_for(rp, synth.FRAME_TIME)

    def to_dev(a):
        a = np.ascontiguousarray(a)
        view = {np.dtype(np.uint16): np.int16, np.dtype(np.uint32): np.int32}.get(a.dtype)
        return torch.from_numpy(a.view(view) if view else a).to(dev)

    items = [(f, s) for f in range(3) for s in scas]    # (filter, SCA), seed 1000 * filter + sca
    order = [items[i] for i in (0, 5, 10, 3, 4, 9, 2, 7, 8, 1, 6, 11)]   # never the same slot twice in a row
    refs, ins, outs = {}, {}, {}
    for f, s in items:
        r = synth.make_ramp(cals[s], read_pattern=rp, seed=1000 * f + s, cr_frac=0.02)
        refs[(f, s)] = oracle.calibrate_arrays(r, cals[s])
        g = r["groupdq"].copy()
        g[0] |= 1
        ins[(f, s)] = [to_dev(r["data"]), to_dev(r["amp33"]), to_dev(g), to_dev(r["pixeldq"])]
        outs[(f, s)] = [torch.empty((ny, nx), dtype=torch.float32, device=dev) for _ in range(3)] + [
            torch.empty((ny, nx), dtype=torch.int32, device=dev), torch.empty((8, ny, nx), dtype=torch.uint8, device=dev)]
    torch.cuda.synchronize()
    try:
        for f, s in order:      # all queued, one synchronisation at the end
            t, o = ins[(f, s)], outs[(f, s)]
            cb.calibrate_device(s, pid, 8, t[0].data_ptr(), True, t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr())
        cb.synchronize()
        for key in items:
            ref, o = refs[key], outs[key]
            # channel lines are fitted on the device here: flags identical, floats within the north-star tolerance
            assert_same_bits(o[3].cpu().numpy().view(np.uint32), ref["pixeldq"], f"pixeldq of item {key}")
            assert_same_bits(o[4].cpu().numpy(), ref["groupdq"], f"groupdq of item {key}")
            np.testing.assert_allclose(o[0].cpu().numpy(), ref["slope"], rtol=1e-5, atol=1e-7)
            tot = np.hypot(ref["err_read"], ref["err_poisson"])
            assert np.all(np.abs(o[1].cpu().numpy() - ref["err_read"]) <= 1e-5 * tot + 1e-12)
            assert np.all(np.abs(o[2].cpu().numpy() - ref["err_poisson"]) <= 1e-5 * tot + 1e-12)
        # the items do differ (a result computed against the wrong slot would not pass the checks above)
        assert not np.array_equal(refs[(0, 3)]["slope"], refs[(0, 7)]["slope"])
    finally:
        for sca in scas:
            cb.ctx.drop_caldir(sca)


@pytest.mark.parametrize("form", [2, 0])
def test_read_file_without_reference_output(form):
    """A read-noise file without ``amp33``: the reference's row step then fits its slope on an all-zero reference block
    (reference_subtraction.py:104-117, np.polyfit on a rank-deficient system: slope 0) and multiplies it by row medians that
    are all zero, i.e. it leaves the image as it is; the channel step still acts.  The oracle runs that branch as written
    (oracle/refpix.row_step with slope None); the device treats the row step as the identity (refpix.hip) -- same bits."""
    rp = synth.READ_PATTERN_8
    ny, nx = 64, 256
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=81, bias_amplitude=1.0)
    cal["read"] = {k: v for k, v in cal["read"].items() if k != "amp33"}
    ramp = synth.make_ramp(dict(cal, read=dict(cal["read"], amp33={"med": np.full((ny, 128), 29000.0, np.float32),
                                                                  "std": np.full((ny, 128), 4.0, np.float32),
                                                                  "M_PINK": 0.8, "RU_PINK": 1.0})), read_pattern=rp, seed=82, cr_frac=0.02)
    ramp["amp33"] = None
    with np.errstate(all="ignore"):
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # numpy's RankWarning of the degenerate fit
            ref = oracle.calibrate_arrays(ramp, cal)
    ctx = gpu_context()
    _set_form(ctx, form)
    cb = pipeline.Calibrator(ctx=ctx)
    cb.load_caldir(10, cal)
    try:
        got = cb.calibrate(10, ramp, want_cube=True, channel_lines=_oracle_lines(ref, len(rp), nx // 128))
        assert_same_bits(got["cube"], ref["data"], "corrected cube", zero_sign_ok=True)
        assert_same_bits(got["pixeldq"], ref["pixeldq"], "pixeldq")
        for k in ("slope", "err_read", "err_poisson"):
            assert_same_bits(got[k], ref[k], k, zero_sign_ok=True)
    finally:
        _default_form(ctx)
        cb.ctx.drop_caldir(10)


@pytest.mark.parametrize("clash", [False, True])
def test_flat_flags_and_dark_dq_reach_pixeldq_through_the_merged_flag_word(clash):
    """The wave-specialised kernel reads ONE flag word per pixel (linearity dq merged with the flat flags and the dark dq on the
    active region, built at rip_caldir_upload) instead of three planes.  A dark file with dq bits and a flat with out-of-range
    pixels must give the oracle's pixeldq bit for bit; when an added word carries NO_LIN_CORR / REFERENCE_PIXEL (it would change
    the linearity test) the set is not mergeable and another kernel form runs -- same results."""
    rp = synth.READ_PATTERN_8
    ny, nx = 48, 384
    cal = synth.make_caldir(ny, nx, read_pattern=rp, p_order=8, seed=91, bias_amplitude=1.0, bad_lin_frac=0.01)
    rng = np.random.default_rng(92)
    ddq = np.where(rng.random((ny, nx)) < 0.05, np.uint32(1) << rng.integers(3, 18, size=(ny, nx)).astype(np.uint32), 0).astype(np.uint32)
    ddq[0, 0] = ddq[1, 200] = 1 << 9          # on the reference-pixel border: the dark step never reaches it
    if clash:
        ddq[20, 50] |= np.uint32(1 << 20)     # NO_LIN_CORR in the dark dq
    cal["dark"] = dict(cal["dark"], dq=ddq)
    flat = cal["flat"]["data"].copy()
    flat[10, 10], flat[11, 300], flat[30, 77] = 0.01, 50.0, 0.05      # NO_FLAT_FIELD
    cal["flat"] = dict(cal["flat"], data=flat)
    gain = cal["gain"]["data"].copy()
    gain[12, 12] = 0.05                                               # NO_GAIN_VALUE
    cal["gain"] = dict(cal["gain"], data=gain)
    ramp = synth.make_ramp(cal, read_pattern=rp, seed=93, cr_frac=0.02)
    with np.errstate(all="ignore"):
        ref = oracle.calibrate_arrays(ramp, cal)
    ctx = gpu_context()
    _default_form(ctx)
    cb = pipeline.Calibrator(ctx=ctx)
    cb.load_caldir(11, cal)
    try:
        got = cb.calibrate(11, ramp, channel_lines=_oracle_lines(ref, len(rp), nx // 128))
        form = ctx.last_chain_form()
        assert (form != 2) if clash else (form == 2), form
        assert_same_bits(got["pixeldq"], ref["pixeldq"], "pixeldq")
        assert_same_bits(got["groupdq"], ref["groupdq"], "groupdq")
        for k in ("slope", "err_read", "err_poisson"):
            assert_same_bits(got[k], ref[k], k, zero_sign_ok=True)
        # the flags are there: dark dq on active pixels only, flat and gain flags
        assert (got["pixeldq"][4:-4, 4:-4] & ddq[4:-4, 4:-4] == ddq[4:-4, 4:-4]).all() and not (got["pixeldq"][0, 0] & (1 << 9))
        assert got["pixeldq"][10, 10] & (1 << 18) and got["pixeldq"][12, 12] & (1 << 19)
        # a sub-chain without the flat stage: the flat flags must not appear (another merged word, or none)
        no_flat = cb.calibrate(11, ramp, stages=pipeline.STAGE_ALL & ~pipeline.STAGE_FLAT, channel_lines=_oracle_lines(ref, len(rp), nx // 128))
        assert not (no_flat["pixeldq"][10, 10] & (1 << 18)) and (no_flat["pixeldq"][4:-4, 4:-4] & ddq[4:-4, 4:-4] == ddq[4:-4, 4:-4]).all()
        no_dark = cb.calibrate(11, ramp, stages=pipeline.STAGE_ALL & ~pipeline.STAGE_DARK, channel_lines=_oracle_lines(ref, len(rp), nx // 128))
        assert no_dark["pixeldq"][10, 10] & (1 << 18)
        sel = (ddq[4:-4, 4:-4] != 0) & ((ref["pixeldq"][4:-4, 4:-4] & ~ddq[4:-4, 4:-4]) == (no_dark["pixeldq"][4:-4, 4:-4]))
        assert sel.sum() > 10     # pixels whose only extra flags were the dark's: gone without the dark stage
    finally:
        cb.ctx.drop_caldir(11)
