"""L2 -> LDS (LDS-DMA) and L2 -> VGPR streaming rates of the whole chip, one 512-thread workgroup per CU.
Build here: python scripts/ubench/dma_rate.py build ; run on the GPU box: python scripts/ubench/dma_rate.py"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_dma_rate.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(HERE, "dma_rate.hip"), "-o", SO])
    sys.exit(0)
import torch
L = ctypes.CDLL(SO)
L.dma_rate_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
BIG = 1 << 30
big = torch.randint(0, 2 ** 31 - 1, (BIG // 4,), device="cuda", dtype=torch.int32)
buf = torch.randint(0, 2 ** 31 - 1, (1024 * 1024 * 1024 // 4,), device="cuda", dtype=torch.int32)
sink = torch.zeros(256, device="cuda", dtype=torch.int32)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
MODES = {0: "lds-dma, vmcnt(0)+barrier / tile", 1: "lds-dma, one tile in flight", 2: "global_load -> VGPR", 3: "lds-dma + 24 ds_read_b128",
         4: "1/8 misses, vmcnt(0)", 5: "1/8 misses, one tile in flight", 6: "1/8 misses + warm-ahead", 7: "1/8 misses + reads", 8: "1/8 misses + reads + warm-ahead"}
def run(mode, region, stride, shared, tiles=400, grid=256):
    f = lambda: L.dma_rate_launch(mode, buf.data_ptr(), region, stride, tiles, shared, sink.data_ptr(), grid, st, big.data_ptr(), BIG)
    assert f() == 0; torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        s.record(); f(); e.record(); torch.cuda.synchronize(); best = min(best, s.elapsed_time(e) * 1e-3)
    byt = grid * tiles * 65536
    return byt / best / 1e12, best / tiles * 1e6
print("rates in TB/s chip-wide (us per 64 KiB tile per CU)")
for name, region, stride, shared in (("private 64 KiB contiguous (L2 hits)", 65536, 128, 0),
                                     ("private 64 KiB used, rows 1536 B apart (L2 hits)", 512 * 1536, 1536, 0),
                                     ("private 64 KiB used, rows 6144 B apart (L2 hits)", 512 * 6144 // 4, 6144, 0),
                                     ("XCD-shared 2 MiB contiguous (sharers in lock step)", 2 << 20, 128, 1),
                                     ("XCD-shared 3 MiB, rows 1536 B apart", 4 * 512 * 1536, 1536, 1),
                                     ("private 1 MiB contiguous (256 MiB: MALL / HBM)", 1 << 20, 128, 0),
                                     ("private 2 MiB contiguous (512 MiB: HBM)", 2 << 20, 128, 0)):
    if stride == 6144: region = 512 * 6144
    for mode in ((0, 1, 2, 3, 4, 5, 6, 7, 8) if region <= 4 * 512 * 1536 else (0, 1)):
        tb, us = run(mode, region, stride, shared)
        print(f"{name:55s} {MODES[mode]:34s} {tb:6.2f} TB/s  ({us:5.2f} us/tile)", flush=True)
    print()
