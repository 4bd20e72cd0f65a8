"""fp32-class ViT attention at the bench size (64 crops x 12 heads x 1374 tokens): round 3's kernel (fp32 in, split out) vs round 4's
(split in, split out; csrc/vit_attn_f32s.hip).  python scripts/ubench/attn_f32_ss_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unopose_amd import ops
torch.set_grad_enabled(False)
B, T = 64, 1374
qkv = torch.randn(B, T, 2304, device="cuda"); qkv[:, :, :768] *= 2
qs = ops.split_f32(qkv.reshape(B * T, 2304))
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
a = ops.vit_attention_f32_split(qkv, 12); b = ops.vit_attention_f32_ss(qs, B, T, 12)
rec = lambda s: (lambda k: (k[:, :, 0] + k[:, :, 1]).reshape(B * T, 768))(s.reshape(B * T, 24, 2, 32).float())
print("max |new - old|:", float((rec(a) - rec(b)).abs().max()))
t_old = timeit(lambda: ops.vit_attention_f32_split(qkv, 12)); t_new = timeit(lambda: ops.vit_attention_f32_ss(qs, B, T, 12))
fl = 4.0 * B * 12 * T * T * 64
print(f"round 3 kernel {t_old:.0f} us ({fl / t_old / 1e6:.0f} TF fp32-equivalent)   round 4 kernel {t_new:.0f} us ({fl / t_new / 1e6:.0f} TF, {3 * fl / t_new / 1e6:.0f} TF of bf16 MFMA)")
