"""A/B: geometric structure embedding, matrix-core kernel vs table-interpolated kernel (bf16 result, 16 x 197 x 197 x 256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd import ops
from unopose_amd.model.unopose import UNOPose
from unopose_amd.model.config import default_model_cfg

torch.manual_seed(0)
model = UNOPose(default_model_cfg()).cuda().eval()
m = model.geo_embedding
for B, n in ((16, 197), (2, 197), (16, 65)):
    pts = torch.cat([torch.ones(B, 1, 3), torch.rand(B, n - 1, 3) * 1.2 - 0.6], 1).cuda()
    ref = ops.geo_embedding(pts, m, out_dtype=torch.float32)
    for table in (False, True):
        ops.GEO_TABLE = table
        for _ in range(3):
            out = ops.geo_embedding(pts, m, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            out = ops.geo_embedding(pts, m, out_dtype=torch.bfloat16)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        err = (out.float() - ref).abs()
        gb = out.numel() * 2 / 1e9
        print(f"B={B} n={n} table={table}: {us:8.1f} us  {gb / us * 1e6:7.1f} GB/s written  max err {err.max().item():.2e} mean {err.mean().item():.2e}", flush=True)
