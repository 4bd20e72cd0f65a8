"""Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two ways csrc/ reads HBM: LDS-DMA (`buffer_load_dwordx4 ... lds`, the
GEMMs' operand path) and plain `global_load_dwordx4`.  Each launch of dma_rate.hip reads a 1 GiB buffer (4x the Infinity Cache)
exactly ONCE: 256 workgroups x 64 tiles x 64 KiB, private 4 MiB regions, 16 B per lane.  Run under
`rocprofv3 --pmc FETCH_SIZE --kernel-trace`; FETCH_SIZE (KiB) x 1024 / 2^30 is the fraction of the bytes the counter reports
(0.5 -> the doubling of MI355X_MICROARCH.md applies to that path).  Build first: python scripts/ubench/dma_rate.py build"""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "_dma_rate.so"))
L.dma_rate_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                              ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
BIG = 1 << 30
buf = torch.randint(0, 2 ** 31 - 1, (BIG // 4,), device="cuda", dtype=torch.int32)
sink = torch.zeros(256, device="cuda", dtype=torch.int32)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode in (0, 2, 0, 2, 0, 2):  # 0: LDS-DMA, 2: global_load -> VGPR
    assert L.dma_rate_launch(mode, buf.data_ptr(), 4 << 20, 128, 64, 0, sink.data_ptr(), 256, st, buf.data_ptr(), BIG) == 0
    torch.cuda.synchronize()
print("done: 3 launches per mode, 2^30 bytes read once per launch")
