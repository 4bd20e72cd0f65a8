"""Which kernels change their results beside the token attention kernel (same-CU co-residency)?"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unopose_amd import ops
from unopose_amd._lib import call, ptr, stream_ptr
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
torch.set_grad_enabled(False)
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).cuda().eval()
ep, _, _ = make_batch(32, S=518, seed=50, device="cuda")
ep2, _, _ = make_batch(32, S=518, seed=51, device="cuda")
g = torch.Generator().manual_seed(0)
yq = torch.randn(64, 197, 1280, generator=g).cuda().bfloat16(); vt = torch.randn(64, 256, 256, generator=g).cuda().bfloat16()
Eb = torch.randn(64, 197, 197, 256, generator=g).cuda().bfloat16(); outa = torch.empty(64, 197, 256, device="cuda", dtype=torch.bfloat16)
def attn():
    for _ in range(12):
        call("unopose_token_attention", ptr(yq), 1280, ctypes.c_void_p(yq.data_ptr() + 768 * 2), 1280, ptr(vt), ctypes.c_void_p(yq.data_ptr() + 256 * 2), 1280,
             ptr(Eb), 64, 197, 197, 0.125, ptr(outa), stream_ptr())
pts = ep["pts"]; tem = ep["tem1_pts"]
c = pts.mean(1, keepdim=True); pn = ((pts - c) / (pts - c).norm(dim=2).max(1)[0].reshape(-1, 1, 1)).contiguous()
w = torch.rand(32, 2048, generator=g).cuda(); src = torch.randn(32, 2048, 3, generator=g).cuda(); ref = torch.randn(32, 2048, 3, generator=g).cuda()
lrf197 = torch.cat([torch.ones(64, 1, 3), torch.rand(64, 196, 3, generator=g) * 1.2 - 0.6], 1).cuda()
victims = {
    "lrf_global(2048)": lambda: ops.lrf_global(pts),
    "lrf_global(5000)": lambda: ops.lrf_global(tem),
    "fps 5000->2048": lambda: ops.furthest_point_sample(tem, 2048),
    "fps 2048->196": lambda: ops.furthest_point_sample(pn, 196),
    "weighted_procrustes": lambda: torch.cat([t.reshape(32, -1) for t in ops.weighted_procrustes(src, ref, w, 0.001)], 1),
    "geo_embedding": lambda: ops.geo_embedding(lrf197, model.geo_embedding).float(),
    "ball_query": lambda: __import__("unopose_amd.pointnet2._ext", fromlist=["x"]).ball_query(pn, pn, 0.2, 64),
}
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()
vit = model.feature_extraction.rgb_net.vit
for name, fn in victims.items():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref_out = fn().clone(); torch.cuda.synchronize()
        nbad = 0
        for it in range(10):
            with torch.cuda.stream(s3):
                vit((ep2["rgb"], ep2["tem1_rgb"]), taps_side_by_side=True)
            with torch.cuda.stream(s2):
                attn()
            out = fn(); torch.cuda.synchronize()
            nbad += int((out != ref_out).any())
    print(f"victim {name}: {nbad} of 10 runs differ beside token attention (+ ViT)", flush=True)
