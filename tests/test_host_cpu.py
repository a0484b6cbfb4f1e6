"""CPU-only tests of the host side: C-ABI binding, plan scalars, file formats, sharding over gloo."""

import ctypes as C
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest
from conftest import REPO, assert_same_bits, load_golden

from romanimpreprocess_amd import _native, calio, plan as planmod, synth
from romanimpreprocess_amd.dqflags import group, pixel


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "romanhip.h")).read()
    declared = set(re.findall(r"\b(rip_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rip_ctx", "rip_status", "rip_dtype", "rip_location"}
    lib = _native.load_library()  # binds every entry of SYMBOLS or raises
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.rip_version() == 100


def test_ctypes_structs_match_the_header(tmp_path):
    """sizeof/offsetof of the ABI structs as gcc sees them == the ctypes mirrors."""
    src = tmp_path / "abi.c"
    fields = {
        "rip_caldir_desc": ["ny", "dark_data", "refout_slope", "gain", "lin_coefs", "ipc4d", "flat", "biascorr", "saturation",
                            "saturation_dq"],
        "rip_plan_desc": ["ngrp", "tbar", "nreads", "K", "nvariants", "variant_coef", "sthresh_a", "ithresh_b"],
        "rip_ramp_desc": ["location", "data", "data_dtype", "amp33", "area_factor", "channel_lines", "flag_saturation",
                          "sat_skip_firstn", "sat_dilution", "inputs_ready", "ready_event", "or_first_group"],
        "rip_outputs": ["location", "slope", "pixeldq", "groupdq", "cube"],
        "rip_synth_cal": ["ny", "channelwidth", "amp33_valid", "gain", "dark", "smax", "ipc4d", "biascorr", "tbias", "amp33_std",
                          "m_pink", "c_pink"],
    }
    body = "".join(f'printf("{s} %zu\\n", sizeof({s}));\n' + "".join(
        f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));\n' for f in fl) for s, fl in fields.items())
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "romanhip.h"\nint main(void){\n' + body + "return 0;}\n")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    mirror = {"rip_caldir_desc": _native.CaldirDesc, "rip_plan_desc": _native.PlanDesc,
              "rip_ramp_desc": _native.RampDesc, "rip_outputs": _native.Outputs, "rip_synth_cal": _native.SynthCal}
    for s, fl in fields.items():
        assert int(got[s]) == C.sizeof(mirror[s]), s
        for f in fl:
            assert int(got[f"{s}.{f}"]) == getattr(mirror[s], f).offset, f"{s}.{f}"


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device|rip_ctx_create"):
        _native.Context(0)


def test_meta_and_weights_match_reference_goldens():
    g = load_golden("weights")
    for tag, rp in (("g6", synth.READ_PATTERN_6), ("g8", synth.READ_PATTERN_8), ("g16", synth.READ_PATTERN_16)):
        meta = planmod.exposure_meta(rp, synth.FRAME_TIME)
        assert_same_bits(meta["tbar"], g[f"{tag}_tbar"], "tbar")
        assert_same_bits(meta["tau"], g[f"{tag}_tau"], "tau")
        assert_same_bits(meta["N"], g[f"{tag}_N"], "N")
        for ef in (True, False):
            for utag, u in (("udef", planmod.ramp_opt_u(None)), ("ubig", 0.05)):
                assert_same_bits(planmod.construct_weights(u, meta, ef), g[f"{tag}_{'ex' if ef else 'in'}_{utag}"], "K")


def test_plan_desc_scalars_match_the_oracle():
    from oracle import rampfit
    for rp, ef in ((synth.READ_PATTERN_8, True), (synth.READ_PATTERN_6, False), (synth.READ_PATTERN_16, True)):
        meta = planmod.exposure_meta(rp, synth.FRAME_TIME)
        K = planmod.construct_weights(planmod.ramp_opt_u(None), meta, ef)
        d = planmod.plan_desc(meta, K, ef, True, {"SthreshA": 5.0})
        start = 1 if ef else 0
        G = len(rp)
        ends = [G] + list(range(G - 1, 2 + start, -1))
        assert d.nvariants == len(ends) and d.sthresh_a == 5.0 and d.sthresh_b == 4.5
        om = rampfit.ma_table_meta(rp, synth.FRAME_TIME)
        om["K"] = K
        for v, g in enumerate(ends):
            k = K[:g] if v == 0 else rampfit.two_point_weights(om, g, start)
            assert d.variant_g[v] == g
            assert np.float32(d.variant_coef[v]) == np.float32(rampfit.poisson_coef(k, om, g, start))
            assert np.float32(d.variant_rfac[v]) == np.float32(rampfit.read_factor(k, om, g))
    with pytest.raises(ValueError, match="too many groups"):
        rp = [[i] for i in range(70)]
        m = planmod.exposure_meta(rp, 3.04)
        planmod.plan_desc(m, np.zeros(70, np.float32))


def test_asdf_and_npz_round_trip(tmp_path):
    tree = {"roman": {"data": np.arange(24, dtype=np.float32).reshape(2, 3, 4), "dq": np.full((3, 4), 2**31, np.uint32),
                      "anc": {"C_PINK": 0.8, "U_PINK": 0.4}, "amp33": {"valid": True, "M_PINK": 0.8,
                                                                       "med": np.ones((3, 128), np.float32)},
                      "meta": {"exposure": {"frame_time": 3.04, "read_pattern": [[0], [1, 2], [3]]}},
                      "k64": np.linspace(0, 1, 5)}}
    p = tmp_path / "x.asdf"
    calio.write_asdf(str(p), tree)
    with_md5 = tmp_path / "x_md5.asdf"
    calio.write_asdf(str(with_md5), tree, checksum=True)   # same file but for the optional block checksums
    assert p.stat().st_size == with_md5.stat().st_size and p.read_bytes() != with_md5.read_bytes()
    back = calio.read_asdf(str(p))
    r = back["roman"]
    assert_same_bits(r["data"], tree["roman"]["data"], "data")
    assert_same_bits(r["dq"], tree["roman"]["dq"], "dq")
    assert_same_bits(r["k64"], tree["roman"]["k64"], "k64")
    assert r["anc"] == {"C_PINK": 0.8, "U_PINK": 0.4} and r["amp33"]["valid"] is True
    assert r["meta"]["exposure"]["read_pattern"] == [[0], [1, 2], [3]]
    with calio.open_tree(str(p)) as f:
        assert "amp33" in f["roman"] and f["roman"]["anc"]["C_PINK"] == 0.8
    q = tmp_path / "x.npz"
    calio.save_npz_tree(str(q), tree)
    with calio.open_tree(str(q)) as f:
        assert_same_bits(np.asarray(f["roman"]["amp33"]["med"]), tree["roman"]["amp33"]["med"], "med")
        assert f["roman"]["anc"]["C_PINK"] == 0.8
    with calio.open_tree({"data": 1}) as f:
        assert f["roman"]["data"] == 1
    with pytest.raises(ValueError):
        (tmp_path / "bad.asdf").write_bytes(b"not asdf")
        calio.read_asdf(str(tmp_path / "bad.asdf"))


def test_flag_saturation_semantics():
    from oracle.saturation import flag_saturation
    G, n = 6, 9
    data = np.zeros((G, n, n), np.float32)
    data[3:, 4, 4] = 100.0       # saturates from group 3 on
    data[0, 1, 1] = 100.0        # group 0 is never checked (skip_firstn = 1)
    thr = np.full((n, n), 50.0, np.float32)
    thr[7, 7] = np.nan
    data[2:, 7, 7] = 1e6         # NaN threshold: not checked
    ramp = {"data": data, "groupdq": np.zeros((G, n, n), np.uint8), "pixeldq": np.zeros((n, n), np.uint32)}
    flag_saturation(ramp, thr, backup=1, skip_firstn=1)
    sat = (ramp["groupdq"] & np.uint8(group.SATURATED)) != 0
    assert not sat[:2].any() and not sat[:, 7, 7].any() and not sat[:, 1, 1].any()
    assert sat[2:, 3:6, 3:6].all()           # backup by one group, grown by one pixel, sticky
    assert not sat[2:, 0, 0].any()
    assert (ramp["pixeldq"][3:6, 3:6] & np.uint32(pixel.SATURATED)).all() and ramp["pixeldq"][0, 0] == 0


def test_synthetic_inputs_shapes_and_tiling():
    cal, ramp = synth.make_tiled_inputs(160, 256, read_pattern=synth.READ_PATTERN_6, p_order=3, seed=3, strip_rows=32)
    assert ramp["data"].shape == (6, 160, 256) and ramp["data"].dtype == np.uint16
    assert cal["ipc4d"]["data"].shape == (3, 3, 152, 248) and cal["biascorr"]["data"].shape == (6, 152, 248)
    assert cal["linearitylegendre"]["data"].shape == (4, 160, 256) and ramp["amp33"].shape == (6, 160, 128)
    assert (cal["mask"]["dq"][:4] & np.uint32(pixel.REFERENCE_PIXEL)).all()
    assert not (cal["mask"]["dq"][4:-4, 4:-4] & np.uint32(pixel.REFERENCE_PIXEL)).any()
    assert (ramp["groupdq"][0] & 1).all()


_GLOO_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {repo!r})
    import numpy as np, torch, torch.distributed as dist
    from romanimpreprocess_amd import sharding
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = sharding.scatter_items(list(range(100, 172)) if rank == 0 else [])
    assert mine == list(range(100, 172))[rank::world], mine
    # many-realisations exchange: realisations round robin -> all realisations of this rank's rows, in order
    for nseeds, ny, nx, dtype in ((7, 9, 5, np.float32), (4, 10, 3, np.uint8), (1, 4, 6, np.float32)):
        full = (np.arange(nseeds * ny * nx).reshape(nseeds, ny, nx) % 251).astype(dtype)
        rows, y0 = sharding.seeds_to_rows(torch.from_numpy(full[rank::world].copy()), nseeds)
        b = sharding.row_bounds(ny, world)
        assert y0 == b[rank] and rows.shape == (nseeds, b[rank + 1] - b[rank], nx), (y0, rows.shape)
        assert np.array_equal(rows.numpy(), full[:, b[rank]:b[rank + 1]])
        # per-row-range results collected on rank 0
        planes = torch.from_numpy(np.stack([full[:, b[rank]:b[rank + 1]].sum(axis=0), full[0, b[rank]:b[rank + 1]]]).astype(np.float32))
        got = sharding.gather_rows(planes, ny)
        if rank == 0:
            assert np.array_equal(got.numpy(), np.stack([full.sum(axis=0), full[0]]).astype(np.float32))
        else:
            assert got is None
    assert sharding.max_over_ranks(1.0 + rank) == float(world)
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_sharding_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER.format(repo=REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o




def test_flag_saturation_read_pattern_rule():
    """Groups averaging several reads are compared with threshold * mean(reads) / last read (partial saturation of the later
    reads): a group value between the diluted and the full threshold is flagged only when the read pattern is handed over."""
    from oracle.saturation import flag_saturation, read_pattern_dilution
    rp = [[0], [1], [2, 3], [4, 5, 6, 7, 8, 9]]
    dil = read_pattern_dilution(rp)
    assert np.isnan(dil[0]) and dil[1] == 1.0 and dil[2] == 2.5 / 3.0 and dil[3] == 6.5 / 9.0
    n = 7
    thr = np.full((n, n), 1000.0, np.float32)
    data = np.zeros((4, n, n), np.float32)
    data[3, 3, 3] = 800.0        # below 1000, above 1000 * 6.5 / 9 = 722.2: its last reads saturated
    data[2, 5, 5] = 800.0        # group [2, 3]: 1000 * 2.5 / 3 = 833.3 > 800: not flagged either way
    for with_rp, expect in ((False, False), (True, True)):
        ramp = {"data": data, "groupdq": np.zeros((4, n, n), np.uint8), "pixeldq": np.zeros((n, n), np.uint32)}
        flag_saturation(ramp, thr, backup=0, skip_firstn=1, read_pattern=rp if with_rp else None)
        sat = (ramp["groupdq"] & np.uint8(group.SATURATED)) != 0
        assert bool(sat[3, 3, 3]) is expect and not sat[:, 5, 5].any() and not sat[:3].any()


def test_level1_synthesis_host_helpers():
    """from_sim/sim_to_isim.py, the parts that need no GPU: EXTRACT_REF on an L1 tree against the oracle restatement, seeds from the
    generator kinds the mirror accepts, the read times"""
    from conftest import load_golden
    from oracle import l1sim
    from romanimpreprocess_amd.from_sim import sim_to_isim

    g = load_golden("l1sim")
    cube, a33 = g["im_after"], g["amp33_after"]
    rp = [[0], [1, 2], [3, 4, 5, 6], [7]]
    tree = {"data": cube.copy(), "amp33": a33.copy(), "meta": {"exposure": {"read_pattern": [list(r) for r in rp]}}}
    sim_to_isim.extract_ref(tree, {"EXTRACT_REF": {"data_encoding_offset": 500}})
    ref, rest = l1sim.extract_ref(cube, 500)
    ref33, rest33 = l1sim.extract_ref(a33, 500)
    assert np.array_equal(tree["reference_read"], ref) and np.array_equal(tree["data"], rest)
    assert np.array_equal(tree["reference_amp33"], ref33) and np.array_equal(tree["amp33"], rest33)
    assert tree["meta"]["instrument"]["data_encoding_offset"] == 500 and tree["meta"]["exposure"]["read_pattern"] == rp[1:]
    assert rest.dtype == np.uint16 and rest.min() >= 0 and np.all(np.abs(rest[0].astype(int) - (cube[1].astype(int) - cube[0] + 500)) == 0)
    # without a reference output in the tree
    t2 = {"data": cube.copy()}
    sim_to_isim.extract_ref(t2, {"EXTRACT_REF": {}})
    assert "reference_amp33" not in t2 and np.array_equal(t2["data"], l1sim.extract_ref(cube, 0)[1])
    assert sim_to_isim._seed_of(17) == 17 and sim_to_isim._seed_of(np.int64(5)) == 5
    assert sim_to_isim._seed_of(np.random.default_rng(3)) == sim_to_isim._seed_of(np.random.default_rng(3))
    with pytest.raises(ValueError):
        sim_to_isim._seed_of(None)
    with pytest.raises(TypeError):
        sim_to_isim._seed_of("seed")
    tij = sim_to_isim.read_pattern_to_tij(rp, 3.04)
    assert [len(t) for t in tij] == [1, 2, 4, 1] and tij[2][-1] == 3.04 * 6
    assert all(np.array_equal(a, b) for a, b in zip(tij, l1sim.read_pattern_to_tij(rp, 3.04)))


def _bench_json(cmd, env=None, timeout=600):
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines


def test_bench_gpus_n_spawns_its_own_ranks():
    """`python bench.py --gpus 2` started plainly (no WORLD_SIZE) launches two ranks before anything touches a GPU; the
    rehearsal mode runs the item scatter (i mod R), the barriers, the MAX all-reduce and the single JSON line with no device."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out, lines = _bench_json([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "batch72",
                              "--steps", "4", "--warmup", "1"], env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["dry_run"] is True and d["scaling"] == "weak"
    per_rank = d["config"]["items_per_rank"]
    assert per_rank == [[i for i in range(72) if i % 2 == r] for r in range(2)]
    # two ranks sleeping 4 x 1 ms side by side: the job's time is the slower rank's, not the sum
    assert 4.0 <= d["ms_per_step"] * 4 < 400.0 and abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    # a single process that is told --gpus 2 but runs as a world of one must refuse, not report n_gpus 1
    out1, lines1 = _bench_json([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"],
                               env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert out1.returncode != 0 and not lines1 and "WORLD_SIZE=1" in (out1.stderr + out1.stdout)


def test_bench_refuses_more_ranks_than_gpus_before_any_rendezvous():
    """`--gpus N` with fewer than N GPUs visible (and no --share-gpu rehearsal) ends with ONE clear line -- from the spawning process
    and from a rank a launcher started -- instead of a rendezvous that fails in every rank."""
    import json  # noqa: F401
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out, lines = _bench_json([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env)
    assert out.returncode != 0 and not lines
    msg = out.stderr + out.stdout
    assert "--gpus 2 needs 2 GPUs" in msg and msg.count("needs 2 GPUs") == 1
    out, lines = _bench_json([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                             env=dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"))
    assert out.returncode != 0 and not lines and "--gpus 2 needs 2 GPUs" in (out.stderr + out.stdout)
    # the dry run carries every rank's elapsed time
    out, lines = _bench_json([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"], env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert len(d["elapsed_s_per_rank"]) == 2 and max(d["elapsed_s_per_rank"]) * 1e3 / 3 == pytest.approx(d["ms_per_step"])


def test_result_pool_recycles_memory_only_when_every_view_is_gone():
    """Result arrays of Calibrator.calibrate come from memory touched before (no first-touch page faults per call); a block goes
    back to the pool when the last array or view referring to it dies, never earlier."""
    import gc

    from romanimpreprocess_amd.pipeline import ResultPool

    pool = ResultPool(keep=2)
    a = pool.empty((1024, 1024), np.float32)
    assert a.flags.c_contiguous and a.flags.writeable and a.dtype == np.float32 and a.shape == (1024, 1024)
    a[...] = 3.0
    addr = a.ctypes.data
    view = a[10:20]
    del a
    gc.collect()
    assert not pool.free.get(4 << 20)                      # the view keeps the block
    assert view[0, 0] == 3.0
    del view
    gc.collect()
    assert len(pool.free[4 << 20]) == 1
    b = pool.empty((1024, 1024), np.uint32)                # same size, other dtype: the same memory
    assert b.ctypes.data == addr
    small = pool.empty((8, 8), np.float32)                 # small arrays are plain numpy arrays
    assert small.base is None
    blocks = [pool.empty((1024, 1024), np.float32) for _ in range(4)]
    del blocks, b
    gc.collect()
    assert len(pool.free[4 << 20]) == 2                    # at most `keep` spare blocks per size
