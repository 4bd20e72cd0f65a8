"""Durations of the two halves of the pipelined forward in steady state (stage mode, bench shape): features half (ViT) on its stream, matching half on
its stream, per step -- which half bounds the step period."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.pipeline import PipelinedForward
from unopose_amd.synthetic import make_batch, trained_like_
torch.set_grad_enabled(False)
dev = torch.device("cuda"); B, S = 32, int(os.environ.get("SD_S", 518))
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=S)))).to(dev).eval()
b, _, _ = make_batch(B, 2048, 5000, S, seed=100, device=dev); b["coarse_rand"] = torch.rand(B, 18000, device=dev)
pipe = PipelinedForward(model, depth=2, timing=True, stages=True)
marks = []
orig_f, orig_m = model.forward_features, model.forward_matching
def ff(ep):
    a = torch.cuda.Event(enable_timing=True); a.record()
    r = orig_f(ep)
    z = torch.cuda.Event(enable_timing=True); z.record(); marks.append(("f", a, z)); return r
def fm(ep, feats):
    a = torch.cuda.Event(enable_timing=True); a.record()
    r = orig_m(ep, feats)
    z = torch.cuda.Event(enable_timing=True); z.record(); marks.append(("m", a, z)); return r
model.forward_features, model.forward_matching = ff, fm
for _ in range(5): pipe.submit(dict(b))
pipe.drain(); torch.cuda.synchronize(); marks.clear()
t0 = time.perf_counter()
for _ in range(20): pipe.submit(dict(b))
pipe.drain(); torch.cuda.synchronize()
step = (time.perf_counter() - t0) / 20 * 1e3
f = sorted(a.elapsed_time(z) for k, a, z in marks if k == "f"); m = sorted(a.elapsed_time(z) for k, a, z in marks if k == "m")
print(f"step period {step:.2f} ms; features half median {f[len(f) // 2]:.2f} ms (min {f[0]:.2f}, max {f[-1]:.2f}); matching half median {m[len(m) // 2]:.2f} ms (min {m[0]:.2f}, max {m[-1]:.2f})")
