"""GPU checks of the four-wave kernel (moved out of tests/ with the kernel in round 6): run against a variant library built by g4w_var.py:
    python scripts/ubench/gemm4w/g4w_var.py build base && UNOPOSE_LIB=scripts/ubench/gemm4w/_g4w_base.so python -m pytest scripts/ubench/gemm4w/test_gpu.py"""
import pytest
import torch
import torch.nn.functional as F


@torch.no_grad()
@pytest.mark.parametrize("M,K,N,epi", [(256, 512, 256, 0), (904, 512, 1024, 0), (2048, 768, 3072, 1), (5000, 768, 768, 0), (1000, 448, 512, 2),
                                       (64 * 261, 768, 768, 0), (20000, 3072, 768, 0), (64 * 1374, 768, 2304, 0), (40000, 768, 3072, 1)])
def test_four_wave_kernel_vs_fp32_reference_and_8_wave(M, K, N, epi):
    """The four-wave, one-wave-per-SIMD kernel (csrc/gemm4w.hip; a generated instruction stream, tests/test_gemm4w_emu_cpu.py) behind the
    same entry point: against the fp32 reference of the op, and against the 8-wave kernel (same products, same K-tile order: bias-only
    results may differ by the fragment order inside a K-tile; GELU is evaluated on the bf16-rounded pre-activation here)."""
    import ctypes

    from unopose_amd import _lib

    L = _lib.lib()
    g = torch.Generator().manual_seed(M + K + N + epi)
    a = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    ref = a.float() @ w.float().t() + b
    if epi == 1:
        ref = F.gelu(ref.bfloat16().float())
    if epi == 2:
        ref = torch.relu(ref)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = {}
    was = L.unopose_gemm4w_enable(-1)
    try:
        for mode in (2, 0):
            L.unopose_gemm4w_enable(mode)
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            assert L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, epi, st) == 0
            torch.cuda.synchronize()
            outs[mode] = out.float()
    finally:
        L.unopose_gemm4w_enable(was)
    assert not torch.isnan(outs[2]).any()   # every element written (ragged last row panel included)
    err = (outs[2] - ref).abs()
    # GELU acts on the bf16-rounded pre-activation: a pre-activation within an accumulation-order difference of a rounding boundary lands one
    # bf16 step away from the reference's, and its GELU with it (2^-8 relative = half a step for the rest)
    # (a flipped pre-activation moves the result by one bf16 step of the PRE-activation -- up to two steps of the output, GELU'(x) > 1
    #  around x = 2..4 -- on top of the output's own half step: 2^-6 for those, and they must be rare)
    tight = ref.abs() * 2.0 ** -8 + 2e-3
    tol = ref.abs() * 2.0 ** -6 + 2e-3 if epi == 1 else tight
    assert (err <= tol).all(), (err.max().item(), int((err > tol).sum()))
    assert (err > tight).float().mean().item() < 5e-3
    assert (outs[2] - outs[0]).abs().max().item() <= 2.0 ** -7 * max(1.0, ref.abs().max().item())


@torch.no_grad()
def test_four_wave_kernel_many_launches_back_to_back():
    """ticket slots, the re-zeroing by the last workgroup and the prologue's scalar loads under back-to-back launches (the first version
    of the stream wrote an SGPR pair that a pending s_load then overwrote: an intermittent fault in the first launches of a process)"""
    import ctypes

    from unopose_amd import _lib

    L = _lib.lib()
    M, K, N = 30000, 768, 768
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    was = L.unopose_gemm4w_enable(2)
    try:
        outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(2)]
        for i in range(1200):   # more launches than the ring of ticket slots has entries
            L.unopose_linear_bf16(a.data_ptr(), w.data_ptr(), b.data_ptr(), outs[i & 1].data_ptr(), M, N, K, 0, st)
        torch.cuda.synchronize()
    finally:
        L.unopose_gemm4w_enable(was)
    assert torch.equal(outs[0], outs[1])
    ref = a.float() @ w.float().t() + b
    assert ((outs[0].float() - ref).abs() <= ref.abs() * 2.0 ** -8 + 2e-3).all()


