"""The generated gemm4w instruction stream on the functional emulator (gen4w/emu.py): results against numpy,
with LDS-DMA data landing early and late and the waves of a workgroup emulated in both orders (none may matter)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen4w import emu, host, kernel  # noqa: E402


def bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def bf16_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def gelu(x):
    from math import erf
    return np.vectorize(lambda v: 0.5 * v * (1.0 + erf(v / 2 ** 0.5)))(x)


def run_case(M, N, K, grid, epi, dma_late=False, order=(0, 1, 2, 3), seed=0, stores_ooo=False, **gen_kw):
    rng = np.random.default_rng(seed)
    Af = rng.standard_normal((M, K)).astype(np.float32)
    Wf = (rng.standard_normal((N, K)) / K ** 0.5).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    A, W = bf16(Af), bf16(Wf)
    mem = emu.Memory()
    pA, pW, pB = mem.alloc(A), mem.alloc(W), mem.alloc(bias)
    C0 = np.full((M, N), 0x7FC0, np.uint16)   # NaN: every element must be written
    pC = mem.alloc(C0)
    pS = mem.alloc(np.zeros(16, np.uint32))
    args, d = host.fill_args(M, N, K, grid, pA, pW, pB, pC, pS)
    pK = mem.alloc(args)
    g = kernel.Gen(epi=epi, **gen_kw)
    assert d["nk"] >= g.min_nk
    seq = g.build()
    E = emu.Emu(seq, mem, dma_late=dma_late, order=order, stores_ooo=stores_ooo)
    nm = 0
    for b in range(grid):
        def init(w, b=b):
            w.s[kernel.KARG.n], w.s[kernel.KARG.n + 1] = pK & 0xFFFFFFFF, pK >> 32
            w.s[kernel.WAVE.n], w.s[kernel.BID.n] = w.wid, b
        wg = E.run_block(init)
        nm += sum(w.nmfma for w in wg.waves)
    C = bf16_to_f32(mem.get(pC, np.uint16).reshape(M, N))
    ref = bf16_to_f32(A).astype(np.float64) @ bf16_to_f32(W).astype(np.float64).T + bias
    if epi == 1:
        ref = gelu(bf16_to_f32(bf16(ref)).astype(np.float64))
    if epi == 2:
        ref = np.maximum(ref, 0)
    assert not np.isnan(C).any(), "unwritten outputs"
    err = np.abs(C - ref).max()
    assert err < 0.03 * max(1.0, np.abs(ref).max() / 4), err
    sched = mem.get(pS, np.uint32)
    assert (sched[:9] == 0).all(), sched   # the last workgroup re-zeroed the ticket slot
    return err, nm


@pytest.mark.parametrize("late,order,ooo", [(False, (0, 1, 2, 3), False), (True, (3, 2, 1, 0), True)])
def test_two_tiles_per_workgroup_bias(late, order, ooo):
    # 16 tiles on 8 workgroups: every workgroup crosses one tile seam; ragged last row panel (M = 3 * 256 + 136)
    run_case(904, 1024, 512, 8, 0, dma_late=late, order=order, stores_ooo=ooo)


def test_gelu_and_relu():
    run_case(512, 512, 768, 8, 1, stores_ooo=True)
    run_case(256, 512, 512, 8, 2)


def test_wait_counts_do_not_rely_on_store_order():
    """a stream whose counted waits include younger stores fails once acknowledgements overtake loads: the emulator must see that"""
    from gen4w import isa
    seq = [isa.Ins("buffer_load_dwordx4", isa.V(2, 2), isa.S(28, 4), isa.S(91), addr="idxen offen", tag="g"),
           isa.Ins("buffer_store_dwordx4", isa.V(4, 4), isa.V(2, 2), isa.S(36, 4), isa.S(89), addr="idxen offen", tag="st"), isa.wait_vm("g")]
    out, vm, _ = isa.resolve_waits(seq)
    assert out[-1].mods["vmcnt"] == 0 and not any(ld for _, ld in vm), (out[-1].mods, vm)   # the store is not counted: the wait covers it
