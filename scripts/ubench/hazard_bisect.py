"""Micro-victims of scripts/ubench/hazard_bisect.hip beside the MFMA-only neighbour of aggressors.hip (and, as controls, beside
nothing and beside the VALU-only neighbour): RUNS launches each, bitwise comparison with the isolated result.
Build here:  hipcc -O3 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off scripts/ubench/hazard_bisect.hip -o scripts/ubench/_hazard_bisect.so"""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
A = ctypes.CDLL(os.path.join(HERE, "_aggressors.so")); V = ctypes.CDLL(os.path.join(HERE, "_hazard_bisect.so"))
A.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
V.hazard_victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
big = torch.zeros(1 << 20, device="cuda")
s2 = torch.cuda.Stream()
BLOCKS = 2048
NAMES = ["vote (v_cmp -> s_bcnt1)", "wave_sum (dpp + readlane)", "fp32 division", "sqrtf", "lds list (ballot/mbcnt)", "eig_sym3 (jacobi)", "readlane",
         "v_cmp -> v_cndmask", "packed fp32 (v_pk_mul/add)", "v_pk_mul_f32 op_sel (self-check)"]
ITERS = [4000, 2000, 2000, 4000, 2000, 300, 4000, 4000, 4000, 4000]
ONLY = [int(v) for v in os.environ.get("HAZARD_ONLY", "").split(",") if v]
def victim(which):
    out = torch.zeros(BLOCKS * 256, device="cuda")
    assert V.hazard_victim_launch(which, out.data_ptr(), BLOCKS, ITERS[which], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    return out
def neighbour(which, iters):
    def f():
        for _ in range(3):
            assert A.aggressor_launch(which, big.data_ptr(), 2048, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    return f
NEIGH = [("nothing", None), ("valu 3 VGPRs", neighbour(0, 2000)), ("mfma bf16 chain", neighbour(2, 3000))]
for w, name in enumerate(NAMES):
    if ONLY and w not in ONLY:
        continue
    want = victim(w).clone(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); victim(w); e.record(); torch.cuda.synchronize()
    line = f"{name:28s} ({s.elapsed_time(e):6.2f} ms alone):"
    for nname, nfn in NEIGH:
        bad = 0; nel = 0
        for it in range(RUNS):
            if nfn is not None:
                with torch.cuda.stream(s2):
                    nfn()
            out = victim(w); torch.cuda.synchronize()
            d = (out != want) & ~(torch.isnan(out) & torch.isnan(want))
            if w == 9:  # self-checking kernel: any non-zero count is a wrong packed product
                d = out != 0
            bad += int(d.any()); nel = max(nel, int(d.sum()))
        line += f"  beside {nname}: {bad}/{RUNS} runs differ (max {nel} of {BLOCKS * 256} elements)"
    print(line, flush=True)
