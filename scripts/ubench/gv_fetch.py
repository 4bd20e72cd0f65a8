"""Run ONE gemm_var.py build (argv[1] = name) on the four ViT linears, REPS launches each in a fixed order: the target of a
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass whose per-dispatch rows scripts/ubench/gv_fetch.sh attributes by launch order."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ORDER = (("qkv", 768, 2304, 0), ("proj", 768, 768, 0), ("fc1", 768, 3072, 1), ("fc2", 3072, 768, 0))
REPS = 3
if __name__ == "__main__":
    L = ctypes.CDLL(os.path.join(HERE, f"_gv_{sys.argv[1]}.so"))
    M = 64 * 1374
    fn = L.unopose_linear_bf16
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for nm, K, N, epi in ORDER:
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        torch.cuda.synchronize()
        for _ in range(REPS):
            fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, st)
        torch.cuda.synchronize()
