"""Time rip_stats_reduce (many-realisations statistics) on device-generated stacks: python tools/gpu_checks/stats_timing.py [S]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from romanimpreprocess_amd import _native  # noqa: E402
from romanimpreprocess_amd.harness import many_realizations as mr  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 4096
dev = torch.device("cuda", 0)
ctx = _native.default_context(0)
g = torch.Generator(device=dev).manual_seed(1)
diffs = torch.round(torch.randn((S, N, N), device=dev, generator=g) * 30)
images = torch.randn((S, N, N), device=dev, generator=g)
err = torch.rand((S, N, N), device=dev, generator=g)
good = (torch.rand((S, N, N), device=dev, generator=g) < 0.95).to(torch.uint8)
ideal = torch.zeros((N, N), device=dev)
for alias in (True, False):
    mr.reduce_rows(diffs, images, err, good, ideal, 0, N, reference_alias=alias, ctx=ctx)
    t0 = time.perf_counter()
    out = mr.reduce_rows(diffs, images, err, good, ideal, 0, N, reference_alias=alias, ctx=ctx)
    dt = time.perf_counter() - t0
    print(f"S={S} alias={alias}: reduce {dt*1e3:.1f} ms ({dt*1e3/S:.3f} ms per realisation); stacks {13*S*N*N/1e9:.1f} GB")
chk = torch.median(images[:, 100, :64], dim=0).values if S % 2 else None
print("median(err) sample", out[7, 100, :4].tolist(), "N", out[3, 100, 100:104].tolist())
