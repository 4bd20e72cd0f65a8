"""Generator of the hand-scheduled gfx950 instruction stream of the 4-wave (one wave per SIMD, 512 registers) 256 x 256 GEMM
(`csrc/gemm4w.hip`), and a functional emulator of the instruction subset it uses (CPU-side check of addresses, register
allocation, wait counts and barrier placement before a GPU run)."""
