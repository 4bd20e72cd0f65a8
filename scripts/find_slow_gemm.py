import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from unopose_amd import _lib
import unopose_amd.ops as ops
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
torch.set_grad_enabled(False)
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).cuda().eval()
ep, _, _ = make_batch(32, 2048, 5000, 518, device="cuda")
ep["coarse_rand"] = torch.rand(32, 18000, device="cuda")
with torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(ep))
orig = _lib.call
rec = []
def timed(name, *a):
    if not name.startswith("unopose_linear"):
        return orig(name, *a)
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig(name, *a); torch.cuda.synchronize()
    rec.append(((time.perf_counter() - t0) * 1e6, name, [int(x) if isinstance(x, int) else None for x in a]))
    return r
_lib.call = timed
for fam in (ops.dense, ops.attention, ops.geometry, ops.sampling, ops.pose, ops.train):  # (each family module binds `call` at import)
    fam.call = timed
import unopose_amd.model.modules as mm
for m in (mm,):
    if hasattr(m, "call"): m.call = timed
with torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(ep))
rec.sort(key=lambda r: -r[0])
for r in rec[:12]: print(round(r[0]), r[1], [x for x in r[2] if x is not None])
