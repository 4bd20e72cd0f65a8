import torch, sys, os
sys.path.insert(0, os.getcwd())
from unopose_amd import ops
torch.set_grad_enabled(False)
def timeit(f, n=50):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for rows, C, ab, bb, ld in ((131136, 256, True, True, 256), (87936, 768, False, None, 3072), (12608, 256, True, True, 256)):
    a = torch.randn(rows, C, device="cuda"); a = a.bfloat16() if ab else a
    b = None if bb is None else torch.randn(rows, C, device="cuda").bfloat16()
    ln = torch.nn.LayerNorm(C).cuda()
    wide = torch.empty(rows, ld, dtype=torch.bfloat16, device="cuda")
    out = wide[:, :C] if ld != C else None
    f = (lambda: ops.add_layernorm(a, b, ln, torch.bfloat16)) if out is None else (lambda: ops.add_layernorm(a, b, ln, out=out))
    us = timeit(f)
    byts = rows * C * ((2 if ab else 4) + (2 if bb else 0) + 2)
    print(f"add_layernorm rows={rows} C={C}: {us:.1f} us, {byts/us/1e6:.2f} TB/s")
