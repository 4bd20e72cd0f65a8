"""bf16 pipelined step (depth 2, stage mode) against the host run-ahead:  python scripts/ubench/run_ahead_ab.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.pipeline import PipelinedForward
from unopose_amd.synthetic import make_batch, trained_like_
torch.set_grad_enabled(False)
dev = torch.device("cuda")
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).to(dev).eval()
batch, _, _ = make_batch(32, 2048, 5000, 518, seed=100, device=dev)
batch["coarse_rand"] = torch.rand(32, 18000, device=dev)
if len(sys.argv) > 1:  # burn <n> streams of torch's pool first: does the step depend on WHICH pool streams the pipeline gets?
    BURN = [torch.cuda.Stream() for _ in range(int(sys.argv[1]))]
    p = PipelinedForward(model, depth=2, autocast_dtype=torch.bfloat16)
    for _ in range(4): r = p.submit(dict(batch)).result
    r(); p.drain(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40): r = p.submit(dict(batch)).result
    r(); p.drain(); torch.cuda.synchronize()
    print(f"burned {sys.argv[1]} pool streams first: {(time.perf_counter() - t) / 40 * 1e3:6.2f} ms per step", flush=True)
    sys.exit(0)
for rnd in range(2):
    for ra in (1, 0, 2, 3):
        p = PipelinedForward(model, depth=2, autocast_dtype=torch.bfloat16, run_ahead=ra)
        for _ in range(4): r = p.submit(dict(batch)).result
        r(); p.drain(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(40): r = p.submit(dict(batch)).result
        r(); p.drain(); torch.cuda.synchronize()
        print(f"run_ahead {ra}: {(time.perf_counter() - t) / 40 * 1e3:6.2f} ms per step", flush=True)
        p.close()
