"""A/B of the RPE token attention: the LDS-DMA kernel (round 5) against round 4's fragment-load kernel, bit comparison included.
Build the probe library first:  UNOPOSE_EXTRA_HIPCC_FLAGS=-DUNOPOSE_PROBE_BUILD python -m unopose_amd.build --force
Run on the GPU:  python scripts/ubench/ta_ab.py   (spawns itself once with UNOPOSE_TA_OLD=1)"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd._lib import call, ptr, stream_ptr
torch.set_grad_enabled(False)
B2, n = 64, 197
g = torch.Generator(device="cuda").manual_seed(0)
yq = torch.randn(B2, n, 1280, device="cuda", generator=g).bfloat16()
ykv = torch.randn(B2, n, 512, device="cuda", generator=g).bfloat16()
vt = torch.randn(B2, 256, 224, device="cuda", generator=g).bfloat16()
Eb = torch.randn(B2, n, n, 256, device="cuda", generator=g).bfloat16()
oa = torch.empty(B2, n, 256, device="cuda", dtype=torch.bfloat16)
def attn():
    call("unopose_token_attention", ptr(yq), 1280, ptr(ykv), 512, ptr(vt), ctypes.c_void_p(yq.data_ptr() + 256 * 2), 1280, ptr(Eb), B2, n, n, 0.125, ptr(oa), stream_ptr())
attn(); torch.cuda.synchronize()
ts = []
for r in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): attn()
    e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 10 * 1e3)
ts.sort()
tag = "old (fragment loads)" if os.environ.get("UNOPOSE_TA_OLD") else "new (LDS-DMA)"
print(f"{tag:22s} min {ts[0]:7.1f} us  med {ts[3]:7.1f} us  {Eb.numel() * 2 / ts[3] / 1e6:.2f} TB/s of embedding stream   checksum {oa.float().double().sum().item():.6f} {oa.view(torch.int16).long().sum().item()}", flush=True)
if not os.environ.get("UNOPOSE_TA_OLD"):
    subprocess.call([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, UNOPOSE_TA_OLD="1"))
