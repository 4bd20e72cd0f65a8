"""Per-wave wait anatomy of one tile's K loop (GEMM_STAMP build): compute section / vmcnt wait / barrier wait per K-step."""
import ctypes, os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
SH = {"qkv": (768, 2304, 0), "proj": (768, 768, 0), "fc1": (768, 3072, 1), "fc2": (3072, 768, 0)}
K, N, epi = SH[sys.argv[2]]
L = ctypes.CDLL(os.path.join(HERE, f"_gv_{sys.argv[1]}.so"))
M = 64 * 1374
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fn = L.unopose_linear_bf16
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for _ in range(3):
    fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
st = np.zeros(8 * 64 * 4, dtype=np.uint64)
assert L.unopose_gemm_read_stamps(st.ctypes.data_as(ctypes.c_void_p)) == 0
st = st.reshape(8, 64, 4).astype(np.int64)
nk = K // 64
t0 = st[:, 0, 0].min()
print("per K-step, cycles (s_memtime ticks): [compute = previous barrier release -> end of MFMA issue + reads landed] [vmcnt wait] [barrier wait]")
for wv in range(8):
    row = []
    for kt in range(1, min(nk, 12)):
        comp = st[wv, kt, 0] - st[wv, kt - 1, 2]; vm = st[wv, kt, 1] - st[wv, kt, 0]; bar = st[wv, kt, 2] - st[wv, kt, 1]
        row.append(f"{comp:5d}/{vm:4d}/{bar:4d}")
    print(f"wave {wv}: " + "  ".join(row))
print("K-step period (wave 0):", [int(st[0, kt, 2] - st[0, kt - 1, 2]) for kt in range(1, min(nk, 12))])
