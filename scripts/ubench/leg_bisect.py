"""Why is the fp32 forward slower after pipelined bf16 steps in the same process?  Times fp32 forwards before / after bf16 phases of
different kinds, then after allocator / stream clean-ups.   python scripts/ubench/leg_bisect.py"""
import gc, os, sys, time
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
def fp32(tag, n=6):
    p = PipelinedForward(model, depth=1, autocast_dtype=None)
    for _ in range(2): p.submit(dict(batch)).result()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = p.submit(dict(batch)).result
    r(); torch.cuda.synchronize()
    print(f"{tag:58s} fp32 {(time.perf_counter() - t) / n * 1e3:7.2f} ms   reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    p.close()
def bf16(depth, stages, n=8, timing=False):
    p = PipelinedForward(model, depth=depth, autocast_dtype=torch.bfloat16, stages=stages, timing=timing)
    for _ in range(n): r = p.submit(dict(batch)).result
    r(); torch.cuda.synchronize(); p.close()
if len(sys.argv) > 1 and sys.argv[1] == "bf16first":
    bf16(2, "auto", 25, True); fp32("bf16 depth 2 stage mode FIRST in the process, then")
    fp32("... again"); gc.collect(); torch.cuda.empty_cache(); fp32("... after empty_cache()")
    sys.exit(0)
fp32("fresh process")
bf16(1, False); fp32("after bf16 depth 1")
bf16(2, False); fp32("after bf16 depth 2, whole forwards on two streams")
bf16(2, True); fp32("after bf16 depth 2, stage mode")
bf16(2, "auto", 25, True); fp32("after bf16 depth 2, stages auto, timing, 25 steps")
ep = dict(batch)
with torch.autocast("cuda", dtype=torch.bfloat16, enabled=False):
    for _ in range(3): o = model(ep := dict(batch))
torch.cuda.synchronize(); t = time.perf_counter()
with torch.autocast("cuda", dtype=torch.bfloat16, enabled=False):
    for _ in range(6): o = model(dict(batch))
torch.cuda.synchronize(); print(f"direct model(ep) under autocast(enabled=False): {(time.perf_counter() - t) / 6 * 1e3:.2f} ms", flush=True)
gc.collect(); torch.cuda.empty_cache(); fp32("... after empty_cache()")
model.__dict__.pop("_side_streams", None); fp32("... after dropping the model's side-stream pool")
