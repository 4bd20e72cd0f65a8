"""Does the mere existence of a high-priority HIP stream slow kernels on the other streams?  Runs bench.py --dtype fp32 in this process
after creating (and using once) extra streams:  python scripts/ubench/prio_probe.py <normal streams> <priority -1 streams>"""
import os, runpy, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KEEP = []
n_normal, n_high = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for i in range(n_normal + n_high):
    KEEP.append(torch.cuda.Stream(priority=-1 if i >= n_normal else 0))
    with torch.cuda.stream(KEEP[-1]):
        torch.zeros(1024, device="cuda").add_(1)
torch.cuda.synchronize()
sys.argv = ["bench.py", "--dtype", "fp32", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-roofline"]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
