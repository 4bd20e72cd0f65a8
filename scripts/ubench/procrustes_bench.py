import torch, sys
sys.path.insert(0, '.')
from unopose_amd import ops
torch.set_grad_enabled(False)
for npts in (3, 196, 700, 2048, 3000):
    Mh = 8192
    src = torch.randn(Mh, npts, 3, device="cuda"); ref = torch.randn(Mh, npts, 3, device="cuda"); w = torch.rand(Mh, npts, device="cuda")
    for _ in range(3): ops.weighted_procrustes(src, ref, w)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.weighted_procrustes(src, ref, w)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print(f"N={npts:5d}: {t*1e6:8.1f} us  {Mh/t/1e6:7.1f} M problems/s  {Mh*npts*28/t/1e12:.2f} TB/s")
