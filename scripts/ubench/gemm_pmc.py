"""Run ONE ablation build of csrc/gemm.hip (argv[1] = variant id of gemm_abl.py) a few times: target of rocprofv3 --pmc passes."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
v = int(sys.argv[1]); K = int(sys.argv[2]) if len(sys.argv) > 2 else 3072; N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
L = ctypes.CDLL(os.path.join(HERE, f"_gemm_abl{v}.so"))
M = 64 * 1374
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fn = L.unopose_linear_bf16
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for _ in range(4):
    fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
