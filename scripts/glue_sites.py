"""Call sites of the data-movement glue in one forward (cat / dtype casts / contiguous copies / relu):
bytes moved per site, found by wrapping the torch entry points (no timing; the profiler in this image
returns no Python stacks)."""
import collections
import os
import sys
import traceback

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd.model import UNOPose, default_model_cfg  # noqa: E402
from unopose_amd.synthetic import make_batch, trained_like_  # noqa: E402

img = int(sys.argv[1]) if len(sys.argv) > 1 else 224
torch.set_grad_enabled(False)
dev = torch.device("cuda")
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=img)))).to(dev).eval()
batch, _, _ = make_batch(32, 2048, 5000, img, seed=1, device=dev)
batch["coarse_rand"] = torch.rand(32, 18000, device=dev)
with torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(batch))
sites = collections.defaultdict(lambda: [0, 0])
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "unopose_amd" in fr.filename:
            return f"{fr.filename.split('unopose_amd/')[-1]}:{fr.lineno} {fr.line[:70]}"
    return "?"


def note(kind, nbytes):
    if ON[0]:
        s = sites[(kind, site())]
        s[0] += nbytes
        s[1] += 1


def nb(t):
    return t.numel() * t.element_size()


_cat, _to, _contig, _relu, _float, _clone = torch.cat, torch.Tensor.to, torch.Tensor.contiguous, F.relu, torch.Tensor.float, torch.Tensor.clone


def cat(ts, *a, **k):
    out = _cat(ts, *a, **k)
    if out.is_cuda:
        note("cat", 2 * nb(out))
    return out


def to(self, *a, **k):
    out = _to(self, *a, **k)
    if out.is_cuda and out.data_ptr() != self.data_ptr():
        note("to", nb(self) + nb(out))
    return out


def fl(self):
    out = _float(self)
    if out.is_cuda and out.data_ptr() != self.data_ptr():
        note("float", nb(self) + nb(out))
    return out


def contig(self, *a, **k):
    out = _contig(self, *a, **k)
    if out.is_cuda and out.data_ptr() != self.data_ptr():
        note("contiguous", 2 * nb(out))
    return out


def relu(x, *a, **k):
    out = _relu(x, *a, **k)
    if out.is_cuda:
        note("relu", 2 * nb(out))
    return out


def clone(self, *a, **k):
    out = _clone(self, *a, **k)
    if out.is_cuda:
        note("clone", 2 * nb(out))
    return out


torch.cat, torch.Tensor.to, torch.Tensor.contiguous, F.relu, torch.Tensor.float, torch.Tensor.clone = cat, to, contig, relu, fl, clone
ON[0] = True
with torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(batch))
ON[0] = False
tot = sum(v[0] for v in sites.values())
print(f"total glue traffic {tot/1e6:.0f} MB (~{tot/5e9:.2f} ms at 5 TB/s)")
for (kind, s), (b, n) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{b/1e6:8.1f} MB {n:3d}x {kind:10s} {s}")
