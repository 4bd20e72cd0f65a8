"""Run scripts/ubench/sgpr_probe.hip beside the token attention kernel (and beside nothing / a GEMM) and report corrupted registers.
Build here: python scripts/ubench/sgpr_probe.py build ; run on the GPU box: python scripts/ubench/sgpr_probe.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.path.join(HERE, "_sgpr_probe.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(HERE, "sgpr_probe.hip"), "-o", SO]); sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from unopose_amd import ops
from unopose_amd._lib import call, ptr, stream_ptr
L = ctypes.CDLL(SO)
L.sgpr_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
g = torch.Generator().manual_seed(0)
yq = torch.randn(64, 197, 1280, generator=g).cuda().bfloat16(); vt = torch.randn(64, 256, 256, generator=g).cuda().bfloat16()
Eb = torch.randn(64, 197, 197, 256, generator=g).cuda().bfloat16(); outa = torch.empty(64, 197, 256, device="cuda", dtype=torch.bfloat16)
A = torch.randn(87936, 768, device="cuda").bfloat16(); W = torch.randn(3072, 768, device="cuda").bfloat16(); bias = torch.zeros(3072, device="cuda")
def attn():
    for _ in range(12):
        call("unopose_token_attention", ptr(yq), 1280, ctypes.c_void_p(yq.data_ptr() + 768 * 2), 1280, ptr(vt), ctypes.c_void_p(yq.data_ptr() + 256 * 2), 1280,
             ptr(Eb), 64, 197, 197, 0.125, ptr(outa), stream_ptr())
def attn_cross():
    for _ in range(40):
        call("unopose_token_attention", ptr(yq), 1280, ctypes.c_void_p(yq.data_ptr() + 768 * 2), 1280, ptr(vt), None, 1280, None, 64, 197, 197, 0.125, ptr(outa), stream_ptr())
def gemm():
    for _ in range(3): ops.linear_bf16_hip(A, W, bias, gelu=True)
side = torch.cuda.Stream()
for name, load in (("nothing", None), ("own GEMM", gemm), ("token attention (RPE)", attn), ("token attention (cross)", attn_cross)):
    tot = 0; examples = []
    for it in range(10):
        rep = torch.zeros(1024, dtype=torch.int32, device="cuda")
        if load is not None:
            with torch.cuda.stream(side): load()
        assert L.sgpr_probe_launch(ctypes.c_void_p(rep.data_ptr()), 2048, 20000, it * 77 + 5, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        torch.cuda.synchronize()
        r = rep.cpu().numpy().astype("uint32")
        tot += int(r[0])
        if r[0] and len(examples) < 4: examples.append(r[1:7].tolist())
    print(f"beside {name}: {tot} lanes reported corrupted registers over 10 launches; examples (wave, lane, bad sgprs, bad vgprs, first sgpr index, value): {examples}", flush=True)
