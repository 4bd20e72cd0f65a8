"""Which coarse-stage kernel perturbs the PE kernel running beside it?  PE (32 reference clouds, split output) on one stream, ONE kind of
coarse-stage kernel in a loop on a second stream, the ViT of another batch on a third."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
torch.set_grad_enabled(False)
img = 518
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=img)))).cuda().eval()
PE = model.fine_point_matching.PE
eps = []
for i in range(2):
    ep, _, _ = make_batch(32, S=img, seed=50 + i, device="cuda"); ep["coarse_rand"] = torch.rand(32, 18000).cuda(); eps.append(ep)
with torch.autocast("cuda", dtype=torch.bfloat16):
    feats = model.forward_features(dict(eps[0]))
torch.cuda.synchronize()
pts = feats[2].float().contiguous()
def pe():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        buf = torch.empty(64, 2048, 512, dtype=torch.bfloat16, device="cuda")
        return PE.groups_split(pts, buf, 32)[32:].clone()
ref = pe(); torch.cuda.synchronize()
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()
vit = model.feature_extraction.rgb_net.vit
g = torch.Generator().manual_seed(0)
lrf = torch.cat([torch.ones(64, 1, 3), torch.rand(64, 196, 3, generator=g) * 1.2 - 0.6], 1).cuda()
x197 = torch.randn(64, 197, 256, generator=g).cuda()
layer = model.coarse_point_matching.transformers[0].layers[0]
cross = model.coarse_point_matching.transformers[0].layers[1]
E = None
def geo():
    global E
    E = ops.geo_embedding(lrf, model.geo_embedding)
def rpe():
    for _ in range(6): layer(x197, None, E)
def crs():
    for _ in range(6): cross(x197[:32], x197[32:])
sim = torch.randn(32, 197, 197, generator=g).cuda() * 3; sc = torch.rand(32, 392, generator=g).cuda()
p1 = torch.rand(32, 196, 3, generator=g).cuda(); p2 = torch.rand(32, 196, 3, generator=g).cuda(); rnd = torch.rand(32, 18000, generator=g).cuda()
def cpose():
    for _ in range(3): ops.coarse_pose(sim, sc, p1, p2, rnd)
def fsim():
    for _ in range(20): ops.feature_similarity(x197[:32], x197[32:], 0.1)
def fps196():
    for _ in range(2): ops.furthest_point_sample(pts, 196)
with torch.autocast("cuda", dtype=torch.bfloat16):
    geo(); torch.cuda.synchronize()
import ctypes
from unopose_amd._lib import call, ptr, stream_ptr
yq = torch.randn(64, 197, 1280, generator=g).cuda().bfloat16(); vt = torch.randn(64, 256, 256, generator=g).cuda().bfloat16(); Eb = E.to(torch.bfloat16).contiguous()
outa = torch.empty(64, 197, 256, device="cuda", dtype=torch.bfloat16)
def attn_only():
    for _ in range(12):
        call("unopose_token_attention", ptr(yq), 1280, ctypes.c_void_p(yq.data_ptr() + 768 * 2), 1280, ptr(vt), ctypes.c_void_p(yq.data_ptr() + 256 * 2), 1280,
             ptr(Eb), 64, 197, 197, 0.125, ptr(outa), stream_ptr())
w5 = torch.randn(1280, 256, generator=g).cuda().bfloat16(); b5 = torch.zeros(1280, device="cuda"); xin = torch.randn(64 * 197, 256, generator=g).cuda().bfloat16()
def gemm_small():
    for _ in range(12): ops.linear_bf16_hip(xin, w5, b5)
lin = layer.attention.linear; nrm = layer.attention.norm
def gemm_ln():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(12): ops.linear_add_layernorm(xin.reshape(64, 197, 256), lin, xin.reshape(64, 197, 256), nrm)
def tpad():
    for _ in range(12): call("unopose_transpose_pad_bf16", ctypes.c_void_p(yq.data_ptr() + 1024 * 2), 1280, 64, 197, 256, 256, ptr(vt), stream_ptr())
def victim_pe(): return pe()
def victim_qlg(): return ops.query_lrf_group(pts[:8], PE.r1, PE.ns1).clone()
for vname, victim in (("PE", victim_pe), ("query_lrf_group", victim_qlg)):
    vref = victim(); torch.cuda.synchronize()
    for name, fn in {"token_attention kernel only": attn_only, "small GEMM (N=1280)": gemm_small, "GEMM + residual + LN epilogue": gemm_ln, "transpose_pad": tpad}.items():
        nbad = 0
        for it in range(12):
            with torch.cuda.stream(s3), torch.autocast("cuda", dtype=torch.bfloat16):
                vit((eps[1]["rgb"], eps[1]["tem1_rgb"]), taps_side_by_side=True)
            with torch.cuda.stream(s2):
                fn()
            out = victim(); torch.cuda.synchronize()
            nbad += int((vref != out).any())
        print(f"victim {vname}; beside {name} (+ ViT): {nbad} of 12 runs differ", flush=True)
from unopose_amd._lib import lib
dbg = torch.zeros(8, 2048, 24, device="cuda")
assert lib().unopose_lrf_debug_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
vref = victim_qlg(); torch.cuda.synchronize(); dref = dbg.clone()
NAMES = ["a00","a01","a02","a11","a12","a22","z0x","z0y","z0z","vote","vx","vy","vz","xpx","xpy","xpz","l0","l1","l2","cnt","e0x","e1x","cx","inv_s"]
for it in range(4):
    with torch.cuda.stream(s3), torch.autocast("cuda", dtype=torch.bfloat16):
        vit((eps[1]["rgb"], eps[1]["tem1_rgb"]), taps_side_by_side=True)
    with torch.cuda.stream(s2):
        attn_only()
    out = victim_qlg(); torch.cuda.synchronize()
    dd = (dbg != dref)
    bad = dd.any(-1).nonzero()
    print("run", it, "centres with differing intermediates:", len(bad))
    for b, n in bad[:5].tolist():
        cols = dd[b, n].nonzero().flatten().tolist()
        print("   centre", b, n, "first differing:", [(NAMES[c], dref[b, n, c].item(), dbg[b, n, c].item()) for c in cols[:4]], "all:", [NAMES[c] for c in cols])
