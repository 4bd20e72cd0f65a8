"""Which bf16 GEMM shapes does one forward launch, and what does each cost under the current shape policy of csrc/gemm.hip?
    python scripts/gemm_shapes.py [img]            -> one JSON line {"shape": [count, us], ...}
The policy is read from the environment once per process (UNOPOSE_GEMM_SMALL_TILES), so an A/B runs this
script once per setting (scripts/gemm_policy_ab.sh)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import _lib
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch

torch.set_grad_enabled(False)
img = int(sys.argv[1]) if len(sys.argv) > 1 else 518
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=img)))).cuda().eval()
ep, _, _ = make_batch(32, 2048, 5000, img, device="cuda")
ep["coarse_rand"] = torch.rand(32, 18000, device="cuda")
with torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(ep))  # warm caches
seen = {}
orig = _lib.call
def logged(name, *a):
    if name == "unopose_linear_bf16":
        key = ("lin", int(a[4]), int(a[5]), int(a[6]), int(a[7]))
    elif name == "unopose_linear_add_layernorm_bf16":
        key = ("lin_ln", int(a[8]), 256, int(a[9]), 3)
    elif name == "unopose_linear_bf16_ld":
        key = ("lin_ld", int(a[7]), int(a[8]), int(a[9]), int(a[10]))
    else:
        return orig(name, *a)
    seen[key] = seen.get(key, 0) + 1
    return orig(name, *a)
import unopose_amd.ops as ops
_lib.call = logged
FAMS = (ops.dense, ops.attention, ops.geometry, ops.sampling, ops.pose, ops.train)  # (each family module binds `call` at import)
for fam in FAMS:
    fam.call = logged
with torch.autocast("cuda", dtype=torch.bfloat16):
    model(dict(ep))
_lib.call = orig
for fam in FAMS:
    fam.call = orig

def timeit(f, n=20):
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

st = _lib.stream_ptr()
out = {}
for (kind, M, N, K, epi), cnt in sorted(seen.items()):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda"); c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if kind == "lin_ln":
        r = torch.randn(M, 256, device="cuda").bfloat16(); lw = torch.rand(256, device="cuda") + .5; lb = torch.randn(256, device="cuda")
        f = lambda: orig("unopose_linear_add_layernorm_bf16", _lib.ptr(a), _lib.ptr(w), _lib.ptr(b), _lib.ptr(r), _lib.ptr(lw), _lib.ptr(lb), 1e-5, _lib.ptr(c), M, K, st)
    else:
        f = lambda: orig("unopose_linear_bf16", _lib.ptr(a), _lib.ptr(w), _lib.ptr(b), _lib.ptr(c), M, N, K, epi, st)
    out[f"{kind} M={M} N={N} K={K} epi={epi}"] = [cnt, round(timeit(f), 2)]
print(json.dumps(out))
