"""Which XCD does workgroup b land on?  (512-thread workgroups with 128 KiB of LDS, i.e. one per CU, grid 256 and 2048.)
Build here: python scripts/ubench/xcc_probe.py build ; run on the GPU box: python scripts/ubench/xcc_probe.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = r'''
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(512, 1) void probe(int *out, int spin) {
  __shared__ char big[131072];
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  big[threadIdx.x] = (char)xcc;
  __syncthreads();
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xf) + 0 * big[5];
}
extern "C" void run(int *out, int grid, int spin, void *stream) { hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 0, (hipStream_t)stream, out, spin); }
'''
so = os.path.join(HERE, "_xcc_probe.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = os.path.join(HERE, "_xcc_probe.hip")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", so])
    sys.exit(0)
import torch
lib = ctypes.CDLL(so)
for grid in (256, 2048):
    out = torch.full((grid,), -1, dtype=torch.int32, device="cuda")
    lib.run(ctypes.c_void_p(out.data_ptr()), grid, 20000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.cpu()
    print("grid", grid, "first 24:", o[:24].tolist())
    print("  b % 8 == xcc for", int((o == torch.arange(grid) % 8).sum()), "of", grid, "; per-XCC counts", torch.bincount(o, minlength=8).tolist())
