"""fc1 + GELU: separate exact-erf GELU kernel vs torch._addmm_activation(use_gelu=True) (hipBLASLt epilogue)."""
import torch, torch.nn.functional as F
torch.set_grad_enabled(False)
M, K, N = 64 * 1374, 768, 3072
x = (torch.randn(M, K, device="cuda") * 1.0).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
a = lambda: F.gelu(F.linear(x, w, b))
c = lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True)
r = lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)
print("linear + gelu(erf): %.0f us | _addmm_activation(gelu): %.0f us | _addmm_activation(relu): %.0f us | linear only: %.0f us" % (t(a), t(c), t(r), t(lambda: F.linear(x, w, b))))
ya, yc = a().float(), c().float()
h = F.linear(x, w, b).float()
print("max |fused - erf| = %.3e ; max |tanh-gelu(h) - erf-gelu(h)| in fp32 = %.3e ; fraction of elements that differ: %.4f" % ((ya - yc).abs().max().item(), (F.gelu(h, approximate="tanh") - F.gelu(h)).abs().max().item(), (ya != yc).float().mean().item()))
