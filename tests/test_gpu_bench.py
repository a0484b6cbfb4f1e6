"""bench.py's own N > 1 path on the card: two ranks (gloo rendezvous, both on GPU 0) through the BASELINE config 4 workload."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_two_ranks_batch72_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--workload", "batch72",
           "--side", "256", "--steps", "4", "--warmup", "1", "--clock-ramp-s", "0", "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and "dry_run" not in d
    assert d["config"]["items_rank0"] == [0, 2, 4, 6, 8, 10, 12, 14]          # item i on rank i mod 2
    assert d["value"] > 0 and abs(d["value"] - 2 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert d["roofline"]["kernel_ms"] > 0 and d["chain"]["good_pixel_fraction"] > 0.5
    assert len(d["elapsed_s_per_rank"]) == 2 and max(d["elapsed_s_per_rank"]) * 1e3 / 4 == pytest.approx(d["ms_per_step"])


def test_bench_two_ranks_realizations_on_one_gpu():
    """--workload realizations (BASELINE config 5 through harness/many_realizations.py): two ranks share the seeds, the stacks go
    through the all-to-all, rank 0 reduces; gloo rendezvous, both ranks on GPU 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--share-gpu", "--workload",
           "realizations", "--realizations", "6", "--side", "256"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "strong" and d["unit"] == "realisations/s"
    assert len(d["elapsed_s_per_rank"]) == 2 and d["value"] > 0
    assert d["phases_rank0_s"]["realisations_on_this_rank"] == 3
    assert abs(d["median_bias_DN_per_s"]) < 1.0
