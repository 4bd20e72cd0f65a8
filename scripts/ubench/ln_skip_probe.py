"""Verdict r05 item 1(a): the ceiling of taking the residual + LayerNorm passes out of the ViT.  Same-box A/B of whole bench steps with
the 23 `scale_residual_layernorm` launches of a forward SKIPPED (wrong results: the LayerNorm output is a stale buffer of unit normal values, the residual
stream is never updated) against the product.  What a perfect fold of those passes into the GEMMs around them could gain AT MOST (it
removes all 810 MB per launch; a real fold still moves 540 MB of them inside the GEMM epilogues).
usage: python scripts/ubench/ln_skip_probe.py [--img 518]"""
import io, json, os, sys
from contextlib import redirect_stdout

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from unopose_amd import ops

real = ops.scale_residual_layernorm_
_stale = {}


def skipped(x, y, gamma, norm):
    k = (tuple(x.shape), x.device)
    if k not in _stale:
        _stale[k] = torch.randn(x.shape, device=x.device).bfloat16()  # (NOT zeros: MFMAs on zero operands draw less power and clock higher)
    return _stale[k]


# one measurement per PROCESS (a long-running process drifts: the product leg of an in-process loop went 31.2 -> 35.4 ms over four
# repetitions while the skipped leg did not), alternating, the driver's own step counts
if len(sys.argv) > 1 and sys.argv[1] in ("product", "ln_skipped"):
    name = sys.argv[1]
    ops.scale_residual_layernorm_ = real if name == "product" else skipped
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-fp32", "--no-extra", "--no-roofline", "--steps", "20", "--warmup", "5"] + sys.argv[2:]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(f"{name:10s}: {line['value']:.1f} {line['unit']}  {line['ms_per_step']:.3f} ms/step  (HIP-event median {line['step_ms_hip_events']['median']:.3f})", flush=True)
else:
    import subprocess
    for rep in range(4):
        for name in ("product", "ln_skipped"):
            subprocess.run([sys.executable, os.path.abspath(__file__), name] + sys.argv[1:], stderr=subprocess.DEVNULL)
