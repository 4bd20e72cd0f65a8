"""scripts/ubench/lds_probe.hip beside a hipBLASLt GEMM on another stream: does the library kernel disturb another
workgroup's LDS?  Build here: python scripts/ubench/lds_probe.py build ; run on the GPU box without arguments."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "_lds_probe.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(HERE, "lds_probe.hip"), "-o", so])
    sys.exit(0)
import torch
lib = ctypes.CDLL(so)
P = ctypes.c_void_p
side = torch.cuda.Stream()
BF = torch.bfloat16
a = torch.randn(8192, 768, device="cuda").to(BF); w = torch.randn(3072, 768, device="cuda").to(BF)
a2 = torch.randn(1566, 768, device="cuda").to(BF); w2 = torch.randn(2304, 768, device="cuda").to(BF)
x = torch.randn(8192, 768, device="cuda")
loads = {"none": lambda: None, "lib gemm 8192x768x3072": lambda: torch.nn.functional.linear(a, w),
         "lib gemm 1566x768x2304": lambda: torch.nn.functional.linear(a2, w2), "elementwise": lambda: x * 2 + 1}
for words in (0, 1024):
    for name, f in loads.items():
        st = torch.zeros(4, dtype=torch.int32, device="cuda")
        for it in range(30):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                lib.run(3, words, 60000 if words == 0 else 3000, P(st.data_ptr()), P(side.cuda_stream))
            for _ in range(40): f()
            torch.cuda.synchronize()
        s_ = st.cpu().tolist()
        print(f"{'hand-off probe' if words == 0 else 'pattern probe '} beside {name:24s}: bad pattern reads {s_[0]}, stale / wrong block sums {s_[1]}")
