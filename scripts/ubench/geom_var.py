"""Variants of csrc/geom.hip's query_lrf_group beside the MFMA-only neighbour (scripts/ubench/aggressors.hip): which piece of the
kernel makes its results change under co-residency?  
    python scripts/ubench/geom_var.py build            here: one small .so per variant (geom.hip + abi.hip with -D flags)
    python scripts/ubench/geom_var.py [runs]           on the GPU box"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "unopose_amd", "csrc")
VARIANTS = {"base": [], "sum_vgpr": ["-DUNOPOSE_SUM_VGPR=1"], "sum_vote_vgpr": ["-DUNOPOSE_SUM_VGPR=1", "-DUNOPOSE_VOTE_VGPR=1"],
            "vote_vgpr": ["-DUNOPOSE_VOTE_VGPR=1"], "noeig": ["-DUNOPOSE_QLG_NOEIG=1"], "noeig_sum_vote_vgpr": ["-DUNOPOSE_QLG_NOEIG=1", "-DUNOPOSE_SUM_VGPR=1", "-DUNOPOSE_VOTE_VGPR=1"],
            "sum_shfl": ["-DUNOPOSE_SUM_SHFL=1"], "O1": ["-O1"], "O2": ["-O2"], "O3_no_slp": ["-fno-slp-vectorize"],
            "O3_no_slp_sum_vgpr": ["-fno-slp-vectorize", "-DUNOPOSE_SUM_VGPR=1"]}
ONLY = os.environ.get("GEOM_VAR_ONLY")
if ONLY:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in ONLY.split(",")}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for name, flags in VARIANTS.items():
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off"] + flags +
                              [os.path.join(CSRC, "geom.hip"), os.path.join(CSRC, "abi.hip"), "-o", os.path.join(HERE, f"_geomvar_{name}.so")])
        print("built", name, flush=True)
    sys.exit(0)
import torch
sys.path.insert(0, ROOT)
from unopose_amd.synthetic import make_batch
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
A = ctypes.CDLL(os.path.join(HERE, "_aggressors.so"))
A.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
ep, _, _ = make_batch(32, S=224, seed=50, device="cuda")
pts = ep["pts"]; c = pts.mean(1, keepdim=True); pn = ((pts - c) / (pts - c).norm(dim=2).max(1)[0].reshape(-1, 1, 1)).contiguous()
big = torch.zeros(1 << 20, device="cuda"); s2 = torch.cuda.Stream()
def neighbour():
    for _ in range(6):
        assert A.aggressor_launch(2, big.data_ptr(), 2048, 3000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
base_out = None
for name in VARIANTS:
    L = ctypes.CDLL(os.path.join(HERE, f"_geomvar_{name}.so"))
    L.unopose_query_lrf_group.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    def run(S=64, r=0.1):
        out = torch.empty(32, 6, 2048, S, device="cuda")
        assert L.unopose_query_lrf_group(pn.data_ptr(), 32, 2048, r, S, out.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        return out
    want = run().clone(); torch.cuda.synchronize()
    if base_out is None:
        base_out = want
    bad = 0; worst = 0; centres = 0
    for it in range(RUNS):
        with torch.cuda.stream(s2):
            neighbour()
        out = run(); torch.cuda.synchronize()
        d = out != want
        if d.any():
            bad += 1
            worst = max(worst, float((out - want).abs().max()))
            centres = max(centres, int(d.any(dim=3).any(dim=1).sum()))
    same = "bit-identical to base alone" if torch.equal(want, base_out) else f"differs from base alone in {int((want != base_out).sum())} elements (max {float((want - base_out).abs().max()):.2e})"
    print(f"{name:22s}: {bad:2d} of {RUNS} runs differ beside the MFMA chain (max |diff| {worst:.3g}, up to {centres} centres of 65536); alone: {same}", flush=True)
