"""Every aten op one autocast forward dispatches, by Python call site: calls and bytes written (TorchDispatchMode; run on the GPU box).
Shows where the torch glue around the HIP kernels comes from (cat / cast / fill / copy / elementwise)."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd.model import UNOPose, default_model_cfg  # noqa: E402
from unopose_amd.synthetic import make_batch, trained_like_  # noqa: E402

img = int(sys.argv[1]) if len(sys.argv) > 1 else 518
fp32 = len(sys.argv) > 2 and sys.argv[2] == "fp32"  # the reference's default precision (no autocast)
import contextlib
ac = contextlib.nullcontext if fp32 else (lambda: torch.autocast("cuda", dtype=torch.bfloat16))
torch.set_grad_enabled(False)
dev = torch.device("cuda")
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=img)))).to(dev).eval()
batch, _, _ = make_batch(32, 2048, 5000, img, seed=1, device=dev)
batch["coarse_rand"] = torch.rand(32, 18000, device=dev)
with ac():
    model(dict(batch))
sites = collections.defaultdict(lambda: [0, 0, set()])
VIEW = ("aten.view", "aten.reshape", "aten.expand", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.transpose", "aten.permute",
        "aten.t.default", "aten.alias", "aten.detach", "aten._unsafe_view", "aten.as_strided", "aten.unbind", "aten.split", "aten.sym_", "aten.is_",
        "aten.empty", "aten._local_scalar", "aten.record_stream", "aten.unfold", "aten.lift_fresh")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(v in name for v in VIEW):
            outs = out if isinstance(out, (tuple, list)) else (out,)
            nbytes = sum(o.numel() * o.element_size() for o in outs if torch.is_tensor(o) and o.is_cuda)
            st = [fr for fr in traceback.extract_stack()[:-1] if "unopose_amd" in fr.filename]
            where = f"{st[-1].filename.split('unopose_amd/')[-1]}:{st[-1].lineno}" if st else "?"
            s = sites[(where, name)]
            s[0] += 1
            s[1] += nbytes
            s[2].update(f"{tuple(o.shape)}:{str(o.dtype)[6:]}" for o in outs if torch.is_tensor(o) and o.is_cuda and o.numel() > 1e5)
        return out


with ac(), Log():
    model(dict(batch))
torch.cuda.synchronize()
tot = sum(v[0] for v in sites.values())
print(f"{tot} aten ops that launch work, {sum(v[1] for v in sites.values()) / 1e6:.0f} MB written")
for (where, name), (n, b, shp) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{b / 1e6:9.1f} MB {n:4d} x  {name:42s} {where}  {' '.join(sorted(shp))}")
print("---- by call count")
for (where, name), (n, b, shp) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{b / 1e6:9.1f} MB {n:4d} x  {name:42s} {where}")
