"""The CPU oracle must reproduce the reference's own outputs bit for bit (CPU-only test).

Fixtures: tests/golden/*.npz, written by tools/make_goldens.py from the reference's unmodified
utils/fitting.py, utils/ipc_linearity.py, utils/flatutils.py, utils/reference_subtraction.py.
"""

import hashlib
import json
import os

import numpy as np
import pytest
from conftest import assert_same_bits, load_golden

import golden_cases as gc
import oracle
from oracle import finish, ipc, linearity, rampfit, refpix
from romanimpreprocess_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lin_known_answer():
    # literal vector of the reference's tests/romanimpreprocess/test_linutils.py:14-48
    g = load_golden("lin_known_answer")
    phi, ex = linearity.legendre_series(g["z"], g["coefs"])
    assert_same_bits(phi, g["phi"], "phi")
    assert np.array_equal(ex, g["ex"])
    assert np.all(np.abs(phi - g["literal"]) < 1e-6)


@pytest.mark.parametrize("name", ["multilin_p3_g6", "multilin_p8_g8", "multilin_p8_g8_flagfirst", "multilin_p10_g16"])
def test_multilin(name):
    g = load_golden(name)
    ac = linearity.attempt_corr_from_groupdq(g["groupdq"]) if bool(g["use_attempt_corr"]) else None
    phi, dq = linearity.multilin(g["S"], g["coefs"], g["Smin"], g["Smax"], g["Sref"], g["lin_dq"],
                                 do_not_flag_first=bool(g["do_not_flag_first"]), attempt_corr=ac)
    assert_same_bits(phi, g["phi"], "phi")
    assert_same_bits(dq, g["dq"], "dq")
    assert np.count_nonzero(dq & linearity.NO_LIN_CORR) > 10  # the case does exercise the flag


@pytest.mark.parametrize("name", ["ipc_f32", "ipc_k64", "ipc_g64", "ipc_g64_k64"])
def test_ipc(name):
    g = load_golden(name)
    act = g["cube"][0, 4:-4, 4:-4]
    gact = g["gain"][4:-4, 4:-4]
    assert_same_bits(ipc.ipc_fwd(act, g["K"]), g["fwd"], "fwd")
    assert_same_bits(ipc.ipc_fwd(act, g["K"], gain=gact), g["fwd_g"], "fwd_g")
    assert_same_bits(ipc.ipc_rev(act, g["K"]), g["rev"], "rev")
    assert_same_bits(ipc.ipc_rev(act, g["K"], gain=gact), g["rev_g"], "rev_g")
    assert_same_bits(ipc.correct_cube(g["cube"].copy(), g["K"], g["gain"]), g["cube_gain"], "cube_gain")
    assert_same_bits(ipc.correct_cube(g["cube"].copy(), g["K"], None), g["cube_nogain"], "cube_nogain")


def test_weights():
    g = load_golden("weights")
    for tag, rp in (("g6", synth.READ_PATTERN_6), ("g8", synth.READ_PATTERN_8), ("g16", synth.READ_PATTERN_16)):
        meta = rampfit.ma_table_meta(rp, synth.FRAME_TIME)
        assert_same_bits(meta["tbar"], g[f"{tag}_tbar"], "tbar")
        assert_same_bits(meta["tau"], g[f"{tag}_tau"], "tau")
        assert_same_bits(meta["N"], g[f"{tag}_N"], "N")
        for ef in (True, False):
            for utag, u in (("udef", 0.4 / 1.8 / 6.5**2), ("ubig", 0.05)):
                K = rampfit.construct_weights(u, meta, exclude_first=ef)
                assert_same_bits(K, g[f"{tag}_{'ex' if ef else 'in'}_{utag}"], f"K {tag} {ef} {utag}")
    # SURVEY.md appendix A.5 known answer (probe of the reference in the survey session)
    meta = rampfit.ma_table_meta(synth.READ_PATTERN_8, 3.04)
    K = rampfit.construct_weights(0.4 / 1.8 / 6.5**2, meta, True)
    np.testing.assert_allclose(K, [0, -2.1932600e-03, -3.6439309e-03, -6.6932663e-03, 3.2127209e-10, 6.6932654e-03,
                                   3.6439314e-03, 2.1932600e-03], rtol=2e-7, atol=1e-15)


RAMPFIT = ["rampfit_g8", "rampfit_g6_custom", "rampfit_g16", "rampfit_g8_include_first", "rampfit_g8_gain64",
           "rampfit_g4"]


@pytest.mark.parametrize("name", RAMPFIT)
def test_rampfit(name):
    g = load_golden(name)
    rp = json.loads(str(g["read_pattern"]))
    ef = bool(g["exclude_first"])
    jp = json.loads(str(g["jump_pars"]))
    meta = rampfit.ma_table_meta(rp, synth.FRAME_TIME)
    meta["nborder"] = 4
    meta["K"] = rampfit.construct_weights(0.4 / 1.8 / 6.5**2, meta, ef)
    assert_same_bits(meta["K"], g["K"], "K")
    # single pass
    loc = np.zeros_like(g["groupdq"])
    s0, er0, ep0, smap = rampfit.fit_and_flag(g["data"], loc, g["gain"], g["read"], meta, 4, ef, None, jp)
    assert_same_bits(s0, g["jd_slope"], "jd_slope")
    assert_same_bits(er0, g["jd_err_read"], "jd_err_read")
    assert_same_bits(ep0, g["jd_err_poisson"], "jd_err_poisson")
    assert_same_bits(smap, g["jd_smap"], "jd_smap")
    assert_same_bits(loc, g["jd_flags"], "jd_flags")
    # full ramp_fit
    rdq, pdq = g["groupdq"].copy(), g["pixeldq"].copy()
    slope, er, ep = rampfit.ramp_fit(g["data"], rdq, pdq, g["gain"], g["read"], meta, ef, jp)
    assert_same_bits(slope, g["slope"], "slope")
    assert_same_bits(er, g["err_read"], "err_read")
    assert_same_bits(ep, g["err_poisson"], "err_poisson")
    assert_same_bits(rdq, g["groupdq_out"], "groupdq")
    assert_same_bits(pdq, g["pixeldq_out"], "pixeldq")
    assert np.count_nonzero(rdq & 4) >= 10


@pytest.mark.parametrize("name", ["flat_f32", "flat_g64"])
def test_flat(name):
    g = load_golden(name)
    pdq = g["pixeldq"].copy()
    out = finish.get_flat(g["flat"], g["gain"], g["K"], 4, pdq)
    assert_same_bits(out, g["flat_out"], "flat")
    assert_same_bits(pdq, g["pixeldq_out"], "pdq")
    with np.errstate(divide="ignore", invalid="ignore"):
        assert_same_bits(finish.get_flat(g["flat"], g["gain"], g["K"], 4, None), g["flat_out_nopdq"], "flat nopdq")
    assert_same_bits(finish.get_flat(g["flat"], g["gain"], g["K"], 4, g["pixeldq"].copy(), ipc_deconvolve=False),
                     g["flat_out_noipc"], "flat noipc")


@pytest.mark.parametrize("name", ["refpix_full_a", "refpix_full_b"])
def test_refpix_fullframe(name):
    g = load_golden(name)
    c = gc.refpix_fullframe_inputs(int(g["seed"]))
    slope = refpix.optimal_refout_slope(c["M_PINK"], c["RU_PINK"], c["C_PINK"], c["std"])
    assert slope == float(g["slope"])
    out, diag = refpix.correct_group(c["data"], c["dark"], c["amp33"], c["med"], slope)
    assert_same_bits(diag["amp33_median"], g["amp33_median"], "amp33 median")
    assert_same_bits(diag["ref_med"], g["ref_med"], "ref_med")
    assert_same_bits(np.float32(diag["ctr"]), g["ctr"], "ctr")
    assert_same_bits(diag["channels"][:, :2].astype(np.float32), g["bottom_top"], "channel medians")
    assert hashlib.sha256(out.tobytes()).hexdigest() == str(g["data_sha256"])
    assert_same_bits(out[::257], g["sample_rows"], "rows")
    assert_same_bits(out[:, ::331], g["sample_cols"], "cols")


def test_refpix_row_polyfit():
    g = load_golden("refpix_row_polyfit")
    c = gc.refpix_fullframe_inputs(int(g["seed"]))
    image = np.zeros((4096, 4224), dtype=np.float32)
    image[:, :4096] = c["data"] - c["dark"]
    image, _, _ = refpix.row_step(image, 4096, False, None)
    assert hashlib.sha256(image.tobytes()).hexdigest() == str(g["image_sha256"])


@pytest.mark.parametrize("name", ["chain_g8", "chain_g6_prod_dtypes"])
def test_chain_composition(name):
    """bias -> multilin -> correct_cube -> ramp_fit -> dark-rate deconvolution -> get_flat, composed by the
    oracle's chain, against the same composition of the reference's functions."""
    g = load_golden(name)
    rp = json.loads(str(g["read_pattern"]))
    ny, nx = g["gain"].shape
    cal = {
        "dark": {"data": np.zeros((len(rp), ny, nx), np.float32), "dark_slope": g["dark_slope"],
                 "dq": np.zeros((ny, nx), np.uint32)},
        "read": {"data": g["read"]},
        "gain": {"data": g["gain"]},
        "linearitylegendre": {"data": g["lin_data"], "Smin": g["Smin"], "Smax": g["Smax"], "Sref": g["Sref"],
                              "dq": g["lin_dq"]},
        "ipc4d": {"data": g["K4d"]}, "flat": {"data": g["flat"]}, "biascorr": {"data": g["biascorr"]},
    }
    ramp = {"data": g["data_u16"], "amp33": None, "groupdq": g["groupdq"], "pixeldq": g["pixeldq"],
            "read_pattern": rp, "frame_time": synth.FRAME_TIME}
    out = oracle.calibrate_arrays(ramp, cal, stages={"refpix": False})
    assert_same_bits(out["data"], g["cube_out"], "cube")
    assert_same_bits(out["K"], g["K"], "K")
    assert_same_bits(out["groupdq"], g["groupdq_out"], "groupdq")
    assert_same_bits(out["pixeldq"], g["pixeldq_out"], "pixeldq")
    assert_same_bits(out["flat"], g["flat_out"], "flat")
    # finish algebra from the reference's ramp-fit planes
    nb = 4
    s, er, ep = finish.finish(g["slope"].copy(), g["err_read"].copy(), g["err_poisson"].copy(),
                              g["pixeldq_before_flat"].copy(), nb, g["dark_slope_ipc"], None, g["flat_out"], None)
    assert_same_bits(out["slope"], s, "slope")
    assert_same_bits(out["err_read"], er, "err_read")
    assert_same_bits(out["err_poisson"], ep, "err_poisson")


def test_post_path_reductions_against_reference_goldens():
    """oracle/post.py (numpy restatement of maskhandling.PixelMask1.build and sky.binkxk / smooth_mode / medfit) against
    the goldens made by the reference's own modules."""
    from oracle import post
    g = load_golden("post_mask")
    assert_same_bits(post.build_mask(g["dq"]).astype(np.uint8), g["mask"], "PixelMask1.build")
    s = load_golden("post_sky")
    binned = post.binkxk(np.where(s["mask"].astype(bool), np.nan, s["img"]), 4)
    assert_same_bits(binned, s["binned"], "binkxk")
    ctr, width = post.smooth_mode(s["binned"])
    assert_same_bits(np.array([ctr, width], dtype=np.float64), s["mode"], "smooth_mode")
    for order in (1, 2, 3):
        coef, model = post.medfit(s["withnan"], order=order)
        assert_same_bits(np.asarray(coef, np.float64), s[f"coef{order}"], f"medfit coefficients, order {order}")
        assert_same_bits(model, s[f"model{order}"], f"medfit model, order {order}")
    coef, model = post.medfit(s["img"], N=4, order=2)
    assert_same_bits(model, s["model_n4"], "medfit model (N=4)")


def test_numpy_percentile_arithmetic_of_the_sky_mirror():
    """utils/sky.py reproduces np.nanpercentile on float32 data from two order statistics: index and weight arithmetic
    checked here with a sort standing in for the device selection."""
    from romanimpreprocess_amd.utils import sky
    rng = np.random.default_rng(2)
    for _ in range(200):
        n = int(rng.integers(1, 5000))
        a = (rng.standard_normal(n) * rng.choice([1.0, 100.0, 1e-3])).astype(np.float32)
        v = np.sort(a)
        for q in (0.0, 10.0, 25.0, 50.0, 75.0, 99.9, 100.0, float(rng.uniform(0, 100))):
            p, nx_, w = sky._linear_index(n, q)
            assert np.float32(sky._lerp32(v[p], v[nx_], w)).tobytes() == np.float32(np.nanpercentile(a, q)).tobytes()


def harness_check(out, g, what):
    """Compare eight (4096, 4096) planes with a harness fixture: per-plane SHA-256 plus the stored samples."""
    for i, plane in enumerate(out):
        assert_same_bits(plane[list(gc.HARNESS_ROWS)][:, ::29], g["sample"][i], f"{what}: plane {i} rows")
        assert_same_bits(plane[96:112, 196:220], g["block"][i], f"{what}: plane {i} block")
        assert hashlib.sha256(np.ascontiguousarray(plane)).hexdigest() == str(g["plane_sha256"][i]), f"{what}: plane {i}"


def test_many_realizations_statistics_against_the_reference_script():
    """oracle/harness.py against the output of the reference's own many_realizations.py (one case: full 4096 x 4096 frames
    are the script's fixed geometry)."""
    from oracle import harness
    g = load_golden("harness_b")
    c = gc.HARNESS_CASES["harness_b"]
    ideal = harness.ideal_slope(gc.harness_truth(c["seed"]), gc.HARNESS_EXPTIME, float(g["g_ideal"]), c["scanum"])
    rs = (gc.harness_realisation(c["seed"], j, ideal[4:-4, 4:-4]) for j in range(c["nrun"]))
    out = harness.statistics(rs, ideal)
    assert np.isnan(g["nanpix"][7]) and np.isnan(g["nanpix"][2]) and not np.isnan(g["nanpix"][4])
    harness_check(out, g, "oracle")


@pytest.mark.parametrize("name", list(gc.IL_CASES))
def test_inverse_linearity_and_il_apply(name):
    """oracle.linearity.invlinearity / il_apply against the reference's ipc_linearity.invlinearity and IL.apply."""
    g = load_golden(name)
    with np.errstate(all="ignore"):
        sl = (slice(4, -4), slice(4, -4))
        blk = (slice(6, 40), slice(3, 60))
        z = -1 + 2 * (g["raw"] - g["Smin"][blk]) / (g["Smax"][blk] - g["Smin"][blk])
        phi, exf = linearity.legendre_series(z, g["coefs"][(slice(None),) + blk])
        assert_same_bits(phi, g["lin_phi"], "linearity (one image)")
        assert_same_bits(g["lin_dq"][blk] | np.where(exf, np.uint32(2**20), np.uint32(0)), g["lin_out_dq"], "linearity dq")
        S, ex = linearity.invlinearity(g["counts"], g["coefs"][(slice(None),) + sl], g["Smin"][sl], g["Smax"][sl])
        assert_same_bits(S, g["inv_S"], "invlinearity")
        assert_same_bits(ex.astype(np.uint8), g["inv_ex"], "exflag")
        args = (g["K"], g["gain"], g["coefs"], g["Smin"], g["Smax"], g["Sref"])
        assert_same_bits(linearity.il_apply(g["counts"], *args), g["apply_dn"], "IL.apply DN -> DN")
        assert_same_bits(linearity.il_apply(g["counts"], *args, electrons=True), g["apply_e_in"], "IL.apply e -> DN")
        assert_same_bits(linearity.il_apply(g["counts"], *args, start_e=g["start_e"], electrons=True, electrons_out=True),
                         g["apply_e_both"], "IL.apply e -> e")
        assert_same_bits(linearity.il_apply(g["counts"], None, *args[1:], start_e=25.0, electrons_out=True),
                         g["apply_noipc"], "IL.apply without IPC")
    # the bisection ends inside the well: out-of-range targets saturate at the ends of [Smin, Smax]
    lo, hi = np.minimum(g["Smin"][sl], g["Smax"][sl]), np.maximum(g["Smin"][sl], g["Smax"][sl])
    assert np.all((g["inv_S"] >= lo - 1) & (g["inv_S"] <= hi + 1))


def il_example_inputs(g):
    """The two count images of the reference's il_example (test_workflow.py:409-420) on the fixture's block."""
    y0, x0 = (int(v) for v in g["block_origin"])
    ny, nx = g["gain"].shape
    ne1 = np.zeros((ny, nx), np.float32)
    ne2 = np.zeros((ny, nx), np.float32)
    yy, xx = np.mgrid[y0:y0 + ny, x0:x0 + nx]
    ne2[(yy % 3 == 0) & (xx % 3 == 0)] = 2.0e3
    return ne1, ne2


def test_il_known_answers_of_the_reference_workflow_test():
    """The reference's own literals for IL.apply (test_workflow.py:402-407, tolerance 0.002 as there) and the reference
    class's output on the surrounding block, bit for bit (one pixel in from the block's edge: the IPC footprint)."""
    g = load_golden("il_example")
    for ne, target, ref_out in zip(il_example_inputs(g), (g["target1"], g["target2"]), (g["ref_out1"], g["ref_out2"])):
        conv = ipc.ipc_fwd(ne + 0.0, g["K"])
        S, _ = linearity.invlinearity(conv / g["gain"], g["coefs"], g["Smin"], g["Smax"])
        assert np.all(np.abs(S[10:12, 10:13] - target) < 0.002)
        assert_same_bits(S[1:-1, 1:-1], ref_out[1:-1, 1:-1], "IL.apply block")


def test_get_tilde_nus_matches_the_reference(golden):
    """The host mirror of GalPoisson/find_tilnus.get_tilde_nus against values computed by the reference's own module."""
    from romanimpreprocess_amd.L1_to_L2.GalPoisson.find_tilnus import get_tilde_nus

    g = golden("pearson_params")
    names = sorted({k[3:-4] for k in g if k.startswith("tn_") and k.endswith("_out")})
    assert len(names) >= 8
    for name in names:
        got = np.array(get_tilde_nus(g[f"tn_{name}_N"], g[f"tn_{name}_a"], g[f"tn_{name}_W"]), dtype=np.float64)
        # f64 sums of a few dozen terms; the order of the additions inside numpy's dot / sum is not part of the contract
        np.testing.assert_allclose(got, g[f"tn_{name}_out"], rtol=1e-12, atol=1e-18, err_msg=f"tilde nus of {name}")


def test_noise_1f_frame_matches_the_reference(golden):
    """oracle.noise.noise_1f_frame against the reference's own function (taken from sim_to_isim.py with ast and executed by
    tools/make_goldens.py noise1f): the same normals give the same frame, bit for bit (SHA-256 of the 4096 x 128 f32 frame)."""
    import hashlib

    from oracle import noise as onoise

    g = golden("noise_1f_frame")
    for seed in g["seeds"]:
        normals = np.random.default_rng(int(seed)).standard_normal(4 * 4096 * 128)
        frame = onoise.noise_1f_frame(normals, 4096, 128)
        assert_same_bits(frame[::257], g[f"s{seed}_rows"], f"sampled rows, seed {seed}")
        assert hashlib.sha256(np.ascontiguousarray(frame)).hexdigest() == str(g[f"s{seed}_sha256"])


def test_l1_synthesis_matches_the_reference_functions():
    """oracle/l1sim.py against make_l1_fullcal and fill_in_refdata_and_1f executed from the reference's file (fixture l1sim)"""
    from conftest import l1sim_golden_cal
    from oracle import l1sim

    g = load_golden("l1sim")
    cal, rp = l1sim_golden_cal(g)
    read_time = float(g["read_time"])
    res, start = l1sim.make_l1_fullcal(g["counts"], rp, cal, read_time, g["normals_reset"], g["reads_e"], g["normals_read"])
    assert_same_bits(res, g["resultants"], "rounded resultants of make_l1_fullcal")
    tij = l1sim.read_pattern_to_tij(rp, read_time)
    # the apportioned electrons are cumulative, end at the counts, and restating the draw with the same generator gives them back
    assert np.array_equal(g["reads_e"][-1], g["counts"].astype(np.int32)) and np.all(np.diff(g["reads_e"], axis=0) >= 0)
    assert np.array_equal(l1sim.binomial_shares(g["counts"], tij, np.random.default_rng(903 + 1)), g["reads_e"])
    im = l1sim.embed(res, 32, 512)
    assert np.array_equal(im, g["im_before"])
    amp33 = np.zeros(g["amp33_after"].shape, dtype=np.uint16)
    l1sim.fill_in_refdata_and_1f(im, cal, tij, g["normals_fill"], frames=g["frames"], white33=g["white33"], amp33=amp33,
                                 channelwidth=16)
    assert np.array_equal(im, g["im_after"]), "cube after fill_in_refdata_and_1f"
    assert np.array_equal(amp33, g["amp33_after"]), "reference output after fill_in_refdata_and_1f"
    # reference pixels really changed, the active region moved by the 1/f noise only
    assert np.all(g["im_before"][:, :4] == 0) and np.all(im[:, :4] > 0)
    assert np.max(np.abs(im[:, 4:-4, 4:-4].astype(np.int32) - g["im_before"][:, 4:-4, 4:-4].astype(np.int32))) < 40


def _refpix_variants_base(seed):
    c = gc.refpix_fullframe_inputs(seed)
    n = 4096
    base = np.zeros((n, n + 128), dtype=np.float32)
    base[:, :n] = c["data"] - c["dark"]
    base[:, -128:] = c["amp33"] - c["med"]
    base[:, -128:] -= np.median(base[:, -128:])
    return base


def reference_test_row_image():
    """the artificial image of the reference's unit test, tests/romanimpreprocess/test_ref.py:10-15"""
    im = np.zeros((4096, 4224), dtype=np.float32)
    im[:, :] = np.cos(np.linspace(0, 2000, 4096))[:, None]
    im[:, -128:] *= 2.0
    for x in range(4224):
        im[:, x] += np.sin(0.1 * x) * np.sin(np.linspace(0, 2000, 4096)) ** 3
    im[:, :-128] += 1.0
    return im


REFPIX_VARIANTS = {
    # tag: (function, keyword arguments of the reference's call)
    "row_refout_fit": ("row", dict(use_ref_channel=True, slope=None)),
    "row_border_pyfloat": ("row", dict(use_ref_channel=False, slope=0.7)),
    "row_refout_f32": ("row", dict(use_ref_channel=True, slope=np.float32(0.83))),
    "chan_32": ("chan", dict()),
    "chan_window": ("chan", dict(channel_start=4, channel_end=124, use_ref_channel=True)),
    "chan_overlap": ("chan", dict(channel_start=0, channel_end=192, use_ref_channel=False)),
}


@pytest.mark.parametrize("tag", list(REFPIX_VARIANTS) + ["test_row"])
def test_refpix_variants(tag):
    """ref_subtraction_row / ref_subtraction_channel with the arguments calibrateimage does not use, and the image of the
    reference's own unit test (test_ref.py) with its assertions"""
    g = load_golden("refpix_variants")
    if tag == "test_row":
        im = reference_test_row_image()
        old_std = np.std(im)
        im, _, _ = refpix.row_step(im, 4096, False, None)
        assert np.std(im) < 0.75 * old_std and 0.4 < np.std(im[:, :-128]) < 0.5 and 0.99 < np.mean(im[:, :-128]) < 1.01
    else:
        kind, kw = REFPIX_VARIANTS[tag]
        im = _refpix_variants_base(int(g["seed"]))
        if kind == "row":
            im, _, _ = refpix.row_step(im, 4096, kw["use_ref_channel"], kw["slope"])
        else:
            im, _ = refpix.channel_step(im, 4096, kw.get("use_ref_channel", False), kw.get("channel_start", 0),
                                        kw.get("channel_end", 128))
    assert_same_bits(im[::257], g[f"{tag}_rows"], f"{tag}: sampled rows")
    assert hashlib.sha256(im.tobytes()).hexdigest() == str(g[f"{tag}_sha256"])


JD_TRUNC = [("rampfit_g8", 4), ("rampfit_g8", 6), ("rampfit_g8", 8), ("rampfit_g8_include_first", 3),
            ("rampfit_g8_include_first", 7), ("rampfit_g16", 5), ("rampfit_g16", 12)]


@pytest.mark.parametrize("name,t", JD_TRUNC)
def test_jump_detect_truncated(name, t):
    """fitting.jump_detect(truncate_ramp=t): two-point weights over groups [0, t) (fitting.py:162-167)"""
    g, gt = load_golden(name), load_golden("jump_detect_trunc")
    rp = json.loads(str(g["read_pattern"]))
    ef = bool(g["exclude_first"])
    meta = rampfit.ma_table_meta(rp, synth.FRAME_TIME)
    meta["nborder"] = 4
    meta["K"] = rampfit.construct_weights(0.4 / 1.8 / 6.5**2, meta, ef)
    loc = np.zeros_like(g["groupdq"])
    s, er, ep, smap = rampfit.fit_and_flag(g["data"], loc, g["gain"], g["read"], meta, 4, ef, t, None)
    for k, v in (("slope", s), ("err_read", er), ("err_poisson", ep), ("smap", smap), ("flags", loc)):
        assert_same_bits(v, gt[f"{name}_t{t}_{k}"], f"{name} truncate {t}: {k}")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference tree is only present in the build container")
def test_golden_recipe_reproduces_the_fixtures():
    """tools/make_goldens.py --check on the fast cases, several of them in ONE process (the cases that execute reference
    scripts leave stand-in packages in sys.modules, which the recipe now restores): regenerated arrays == committed ones."""
    import subprocess
    import sys
    cases = ["weights", "lin_known_answer", "flat", "il_example", "pearson", "noise1f", "jump_detect_trunc"]
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "make_goldens.py"), "--check"] + cases,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "identical" in out.stdout.splitlines()[-1]


def _noise_arith_inputs(g):
    rp = json.loads(str(g["read_pattern"]))
    G = len(rp)
    nb = 4
    gain_act = np.clip(g["gain"], 1e-4, 1e4)[nb:-nb, nb:-nb]
    # ramp-fit weights by end slice, as gen_noise_image.py:248-262 builds them
    rows = np.zeros((G, G), np.float32)
    has = np.zeros(G, bool)
    rows[-1], has[-1] = g["weights"], True
    start = 1
    for iend in range(start + 2, G):
        rows[iend - 1, iend - 1] = 1.0 / (g["tbar"][iend - 1] - g["tbar"][start])
        rows[iend - 1, start] = -rows[iend - 1, iend - 1]
        has[iend - 1] = True
    endslice = np.where(g["endslice"] > 0, g["endslice"], G - 1)
    return rp, G, nb, gain_act, rows, has, endslice


def _stand_in_l2(cube, nb=4):
    """the L2 'data' the golden run's calibrateimage stand-in answers with (tools/make_goldens.py: case_noise_arith)"""
    act = cube[:, nb:-nb, nb:-nb].astype(np.float32)
    return (act[-1] - act[1]) / np.float32(7.0)


def test_noise_layer_arithmetic_against_the_reference_loop():
    """oracle/noise.py against arrays produced by EXECUTING the reference's make_noise_cube (gen_noise_image.py:60-331) with
    recording stand-ins for its deviates, files and chain runs: the injected cubes and every layer, bit for bit."""
    from oracle import noise as onoise

    g = load_golden("noise_arith")
    rp, G, nb, gain_act, rows, has, endslice = _noise_arith_inputs(g)
    ft = float(g["frame_time"])
    assert json.loads(str(g["layers"])) == ["Ra", "R", "Pr", "Pb1r", "RaPr"]
    inj_a = onoise.inject_read_noise(g["cube"], g["read"], rp, g["normals"][0], nb)
    assert_same_bits(inj_a, g["injected_Ra"], "cube after the injection ('Ra')")
    dark_as_data = g["dark"].astype(np.uint16)[1:]
    assert_same_bits(dark_as_data, g["dark_as_data"], "dark cube taken as data ('R' without 'a')")
    inj_d = onoise.inject_read_noise(dark_as_data, g["read"], rp, g["normals"][1], nb)
    assert_same_bits(inj_d, g["injected_R"], "dark cube after the injection ('R')")
    assert_same_bits(_stand_in_l2(inj_a) - _stand_in_l2(g["cube"]), g["noise"][0], "layer 'Ra'")
    assert_same_bits(_stand_in_l2(inj_d) - _stand_in_l2(dark_as_data), g["noise"][1], "layer 'R'")
    zero = np.zeros_like(g["withsky"])
    lay2 = onoise.poisson_resample(zero.copy(), g["withsky"], gain_act, ft, rp, rows, has, endslice, g["poisson"][0])
    assert_same_bits(lay2, g["noise"][2], "layer 'Pr'")
    lay3 = onoise.poisson_resample(zero.copy(), g["sky_b1"], gain_act, ft, rp, rows, has, endslice, g["poisson"][1])
    assert_same_bits(lay3, g["noise"][3], "layer 'Pb1r'")
    inj_c = onoise.inject_read_noise(g["cube"], g["read"], rp, g["normals"][2], nb)
    assert_same_bits(inj_c, g["injected_RaPr"], "cube after the injection ('RaPr')")
    start4 = _stand_in_l2(inj_c) - _stand_in_l2(g["cube"])
    lay4 = onoise.poisson_resample(start4.copy(), g["withsky"], gain_act, ft, rp, rows, has, endslice, g["poisson"][2])
    assert_same_bits(lay4, g["noise"][4], "layer 'RaPr'")
    assert np.count_nonzero(inj_a != g["cube"]) > 1000 and np.std(lay2) > 0
