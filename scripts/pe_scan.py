import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
from test_geom_gpu import norm_clouds
torch.set_grad_enabled(False)
m = UNOPose(default_model_cfg()).cuda().eval()
x = norm_clouds(2048, 32, seed=1, repl_every=99).cuda()
mlp = m.fine_point_matching.PE.mlp2
def t(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
for r, S in ((0.05, 32), (0.1, 64), (0.15, 128), (0.2, 256)):
    print(S, "bf16x3 us", round(t(lambda: ops.pe_group_mlp_max(x, r, S, mlp, bf16x3=True))), " query_lrf_group us", round(t(lambda: ops.query_lrf_group(x, r, S))))
