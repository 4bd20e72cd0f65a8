"""Time of the RPE token attention (unopose_token_attention with E) at the bench shape: 64 clouds x 197 x 197 x 256, against the stream-only floor
of round 5 (232 us) -- and its output against the fp32 composite of the same operands."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd._lib import call, ptr, stream_ptr
torch.set_grad_enabled(False)
dev = torch.device("cuda"); B2, n = 64, 197
g = torch.Generator(device="cuda").manual_seed(0)
yq = (0.5 * torch.randn(B2, n, 1280, device=dev, generator=g)).bfloat16(); ykv = (0.5 * torch.randn(B2, n, 512, device=dev, generator=g)).bfloat16()
Eb = (0.5 * torch.randn(B2, n, n, 256, device=dev, generator=g)).bfloat16()
v = ykv[..., 256:].float()                       # (B2, n, 256)
vt = torch.zeros(B2, 256, 224, device=dev, dtype=torch.bfloat16); vt[:, :, :n] = v.transpose(1, 2).bfloat16()
oa = torch.empty(B2, n, 256, device=dev, dtype=torch.bfloat16)
def attn():
    call("unopose_token_attention", ptr(yq), 1280, ptr(ykv), 512, ptr(vt), ctypes.c_void_p(yq.data_ptr() + 256 * 2), 1280, ptr(Eb), B2, n, n, 0.125, ptr(oa), stream_ptr())
attn(); torch.cuda.synchronize()
# reference: per head h: scores = (q_h k_h^T + qp_h . E) / 8
q = yq[..., :256].float().reshape(B2, n, 4, 64); k = ykv[..., :256].float().reshape(B2, n, 4, 64); qp = yq[..., 256:].float().reshape(B2, n, 4, 256)
errs = []
for b in range(0, B2, 17):
    sc = torch.einsum("nhc,mhc->hnm", q[b], k[b]) + torch.einsum("nhc,nmc->hnm", qp[b], Eb[b].float())
    p = torch.softmax(sc * 0.125, dim=-1)
    ref = torch.einsum("hnm,mhc->nhc", p, v[b].reshape(n, 4, 64)).reshape(n, 256)
    errs.append((oa[b].float() - ref).abs().max().item())
print("max |out - fp32 composite| over sampled clouds:", max(errs))
ts = []
for r in range(7):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): attn()
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10 * 1e3)
ts.sort()
byt = 2.0 * B2 * n * n * 256 + 2.0 * B2 * n * (1280 + 512 + 256)
print(f"{'r5 kernel' if os.environ.get('UNOPOSE_TA_R5') else 'kernel   '}: min {ts[0]:.1f} us, median {ts[3]:.1f} us = {byt / ts[3] / 1e3:.0f} GB/s ({byt / ts[3] / 1e3 / 8000:.3f} of 8 TB/s)")
