"""Time of the three fine_assign launches at the bench shape (B pairs, 2049 x 2049)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import ops
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
f1 = torch.randn(B, 2049, 256, device="cuda"); f2 = torch.randn(B, 2049, 256, device="cuda")
sc = torch.rand(B, 4096, device="cuda"); x = torch.randn(B, 2048, 3, device="cuda")
def run():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return ops.fine_pose_from_features(f1, f2, 0.1, sc, x, x)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print(f"fine_pose_from_features (normalise + 3 assignment passes + Procrustes + min-dist), B={B}: {s.elapsed_time(e) / 20 * 1e3:.1f} us")
