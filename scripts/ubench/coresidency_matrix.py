"""The discriminating control for the co-residency finding: one victim kernel at a
time beside SYNTHETIC neighbours that stress one resource each (scripts/ubench/aggressors.hip), beside the real token attention
(positive control) and beside nothing.  20 runs per cell; a cell counts the runs whose output differs in any bit from the isolated
result and the number of distinct outputs seen.
Build here: hipcc -O3 -fPIC -shared --offload-arch=gfx950 scripts/ubench/aggressors.hip -o scripts/ubench/_aggressors.so"""
import ctypes, hashlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unopose_amd import ops
from unopose_amd._lib import call, ptr, stream_ptr
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import trained_like_, make_batch
from unopose_amd.pointnet2 import _ext
torch.set_grad_enabled(False)
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
A = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_aggressors.so"))
A.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=224)))).cuda().eval()
ep, _, _ = make_batch(32, S=224, seed=50, device="cuda")
g = torch.Generator().manual_seed(0)
yq = torch.randn(64, 197, 1280, generator=g).cuda().bfloat16(); vt = torch.randn(64, 256, 256, generator=g).cuda().bfloat16()
Eb = torch.randn(64, 197, 197, 256, generator=g).cuda().bfloat16(); outa = torch.empty(64, 197, 256, device="cuda", dtype=torch.bfloat16)
big = torch.zeros(2048 * 65536 * 4, device="cuda")  # 2 GiB window for the streaming neighbour
s2 = torch.cuda.Stream()

def token_attn():
    for _ in range(12):
        call("unopose_token_attention", ptr(yq), 1280, ctypes.c_void_p(yq.data_ptr() + 768 * 2), 1280, ptr(vt), ctypes.c_void_p(yq.data_ptr() + 256 * 2), 1280,
             ptr(Eb), 64, 197, 197, 0.125, ptr(outa), stream_ptr())

def synthetic(which, iters):
    def f():
        for _ in range(6):
            assert A.aggressor_launch(which, big.data_ptr(), 2048, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    return f

NEIGHBOURS = [("nothing", None), ("token_attn (real)", token_attn), ("valu 3 VGPRs", synthetic(0, 2000)), ("valu 242 VGPRs", synthetic(1, 1500)),
              ("mfma bf16 chain", synthetic(2, 3000)), ("lds ring 60 KiB", synthetic(3, 600)), ("global stream", synthetic(4, 1500)),
              ("s_barrier loop", synthetic(5, 2000)), ("transcendental chain", synthetic(6, 1500)), ("dpp reductions", synthetic(7, 1500))]
pts = ep["pts"]; tem = ep["tem1_pts"]
c = pts.mean(1, keepdim=True); pn = ((pts - c) / (pts - c).norm(dim=2).max(1)[0].reshape(-1, 1, 1)).contiguous()
w = torch.rand(32, 2048, generator=g).cuda(); src = torch.randn(32, 2048, 3, generator=g).cuda(); ref = torch.randn(32, 2048, 3, generator=g).cuda()
pe = model.fine_point_matching.PE
VICTIMS = {
    "weighted_procrustes": lambda: torch.cat([t.reshape(32, -1) for t in ops.weighted_procrustes(src, ref, w, 0.001)], 1),
    "lrf_global(5000)": lambda: ops.lrf_global(tem),
    "query_lrf_group(S=64)": lambda: ops.query_lrf_group(pn, 0.1, 64),
    "pe_bf16x3(S=256)": lambda: ops.pe_group_mlp_max(pn, pe.r2, pe.ns2, pe.mlp2, bf16x3=True),
    "fps 5000->2048 (control)": lambda: ops.furthest_point_sample(tem, 2048),
}
def digest(t):
    return hashlib.md5(t.contiguous().cpu().numpy().tobytes()).hexdigest()
# how long does each neighbour run (so the victim really sits inside it)?
for name, fn in NEIGHBOURS[1:]:
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    print(f"neighbour {name:24s}: {s.elapsed_time(e):7.2f} ms per burst", flush=True)
for vname, vfn in VICTIMS.items():
    want = digest(vfn()); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); vfn(); e.record(); torch.cuda.synchronize()
    print(f"victim {vname} ({s.elapsed_time(e):.2f} ms alone)", flush=True)
    for nname, nfn in NEIGHBOURS:
        seen, bad = set(), 0
        for it in range(RUNS):
            if nfn is not None:
                with torch.cuda.stream(s2):
                    nfn()
            d = digest(vfn()); torch.cuda.synchronize()
            seen.add(d); bad += d != want
        print(f"    beside {nname:24s}: {bad:2d} of {RUNS} runs differ, {len(seen)} distinct outputs", flush=True)
