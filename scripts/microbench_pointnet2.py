"""Per-kernel timings of the `_ext` operators on one MI355X (HIP events)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from unopose_amd.pointnet2 import _ext  # noqa: E402
from helpers import object_cloud  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    g = torch.Generator().manual_seed(0)
    tem = torch.stack([object_cloud(g, 5000) for _ in range(B)]).cuda()
    x = torch.stack([object_cloud(g, 2048) for _ in range(B)])
    c = x.mean(1, keepdim=True)
    x = (x / (x - c).norm(dim=2).max(1)[0].reshape(-1, 1, 1)).contiguous().cuda()
    res = {}
    t = timeit(lambda: _ext.furthest_point_sampling(tem, 2048), 5, 1)
    res["fps_5000_2048"] = dict(s=t, us_per_iter=t / 2047 * 1e6)
    t = timeit(lambda: _ext.furthest_point_sampling(x, 196))
    res["fps_2048_196"] = dict(s=t, us_per_iter=t / 195 * 1e6)
    for r, ns in ((0.1, 64), (0.2, 256)):
        t = timeit(lambda: _ext.ball_query(x, x, r, ns))
        res[f"ball_query_ns{ns}"] = dict(s=t, GBps=4 * B * (6 * 2048 + 2048 * ns) / t / 1e9)
        idx = _ext.ball_query(x, x, r, ns)
        xt = x.transpose(1, 2).contiguous()
        t = timeit(lambda: _ext.group_points(xt, idx))
        byts = 4 * B * (2048 * ns + 3 * 2048 * ns + 3 * 2048)
        res[f"group_points_ns{ns}"] = dict(s=t, GBps=byts / t / 1e9, frac_of_8TBps=byts / t / 8e12)
    f = torch.randn(B, 256, 5000, device="cuda")
    idx = torch.randint(0, 5000, (B, 2048), dtype=torch.int32, device="cuda")
    t = timeit(lambda: _ext.gather_points(f, idx))
    res["gather_points_256x5000_2048"] = dict(s=t, GBps=4 * B * (2048 + 2 * 256 * 2048) / t / 1e9)
    print(json.dumps(dict(B=B, **res), indent=1))


if __name__ == "__main__":
    main()
