import os, sys, torch, json
sys.path.insert(0, os.getcwd())
torch.set_grad_enabled(False)
from unopose_amd import ops, _lib
import bench
M = 64 * 1374
stream = torch.cuda.current_stream()
mode = sys.argv[1]
if mode == "model":  # allocate the model + a batch first, like bench.py
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import trained_like_, make_batch
    model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).cuda().eval()
    ep, _, _ = make_batch(32, 2048, 5000, 518, device="cuda")
    ep["coarse_rand"] = torch.rand(32, 18000, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        model(dict(ep)); model(dict(ep))
    torch.cuda.synchronize()
res = []
for name, K_, N_, gelu in (("qkv", 768, 2304, False), ("proj", 768, 768, False), ("fc1", 768, 3072, True), ("fc2", 3072, 768, False)):
    a = torch.randn(M, K_, device="cuda").bfloat16(); w = (torch.randn(N_, K_, device="cuda") / K_ ** 0.5).bfloat16(); bias = torch.randn(N_, device="cuda")
    t = bench.hip_event_time(lambda: ops.linear_bf16_hip(a, w, bias, gelu), 20, stream, warm=3)
    out = torch.empty(M, N_, device="cuda", dtype=torch.bfloat16); st = _lib.stream_ptr()
    t2 = bench.hip_event_time(lambda: _lib.call("unopose_linear_bf16", _lib.ptr(a), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), M, N_, K_, 1 if gelu else 0, st), 20, stream, warm=3)
    res.append((name, round(2.0 * M * K_ * N_ / t / 1e12), round(2.0 * M * K_ * N_ / t2 / 1e12)))
print(mode, res, "(TF: fresh output per call, same output)")
