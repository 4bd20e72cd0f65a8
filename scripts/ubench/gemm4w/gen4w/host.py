"""Host-side launch parameters of the gemm4w kernel (the Python twin of csrc/gemm4w.hip's `g4w_fill_args`: used by the emulator tests)."""
import numpy as np

from .kernel import KA, KARG_DWORDS


def magic31(d):
    return (2 ** 31 + d - 1) // d


def tile_walk(M, N, K, grid):
    """gemm_kernel.h's tile walk: XCD chunks of a row-major tile sequence, or the column-blocked walk when W exceeds an XCD's L2."""
    tiles_n, tiles_m = N // 256, (M + 255) // 256
    tiles = tiles_m * tiles_n
    nk = K // 64
    cw = (2400 * 1024) // (256 * K * 2) if N * K * 2 > (4 << 20) else 0
    if cw < 3:
        cw = 0
    colwalk = cw > 0 and tiles_n > cw and ((tiles_m & 7) == 0 or tiles_m >= 512)
    rq, rr = tiles_m >> 3, tiles_m & 7
    cq, cr = tiles >> 3, tiles & 7
    d = dict(tiles_n=tiles_n, tiles_m=tiles_m, tiles=tiles, nk=nk, cw=cw if colwalk else 1, colwalk=int(colwalk), nslots=grid // 8)
    d["cbase"] = [x * (cq + 1) if x < cr else cr * (cq + 1) + (x - cr) * cq for x in range(8)]
    rows_x = [rq + (1 if x < rr else 0) for x in range(8)]
    d["rbase"] = [x * (rq + 1) if x < rr else rr * (rq + 1) + (x - rr) * rq for x in range(8)]
    d["clen"] = [rows_x[x] * tiles_n if colwalk else cq + (1 if x < cr else 0) for x in range(8)]
    cwv = d["cw"]
    d["pb"] = [max(1, rows_x[x] * cwv) for x in range(8)]
    ncb = (tiles_n + cwv - 1) // cwv
    d["ncb1"] = ncb - 1
    d["cwl"] = tiles_n - (ncb - 1) * cwv
    return d


def tile_of(d, xcd, ti):
    """reference of the in-kernel scalar code: ticket -> (tm, tn)"""
    if d["colwalk"]:
        cb, v = divmod(ti, d["pb"][xcd])
        cwb = d["cwl"] if cb == d["ncb1"] else d["cw"]
        g, tl = divmod(v, cwb)
        return d["rbase"][xcd] + g, cb * d["cw"] + tl
    t = d["cbase"][xcd] + ti
    return divmod(t, d["tiles_n"])


def fill_args(M, N, K, grid, pA, pW, pB, pC, pS, lda=None, ldw=None, ldc=None):
    d = tile_walk(M, N, K, grid)
    a = np.zeros(KARG_DWORDS, np.uint32)

    def p64(k, v):
        a[KA[k]] = v & 0xFFFFFFFF
        a[KA[k] + 1] = v >> 32
    p64("A", pA), p64("W", pW), p64("bias", pB), p64("C", pC), p64("sched", pS)
    a[KA["M"]], a[KA["N"]], a[KA["K"]] = M, N, K
    a[KA["lda"]], a[KA["ldw"]], a[KA["ldc"]] = 2 * (lda or K), 2 * (ldw or K), 2 * (ldc or N)
    a[KA["nk"]], a[KA["tiles_n"]], a[KA["nslots"]] = d["nk"], d["tiles_n"], d["nslots"]
    a[KA["mg_tn"]], a[KA["mg_ns"]], a[KA["mg_nk"]] = magic31(d["tiles_n"]), magic31(d["nslots"]), magic31(d["nk"])
    a[KA["cw"]], a[KA["mg_cw"]], a[KA["cwl"]], a[KA["mg_cwl"]] = d["cw"], magic31(d["cw"]), d["cwl"], magic31(d["cwl"])
    a[KA["ncb1"]], a[KA["colwalk"]], a[KA["grid"]] = d["ncb1"], d["colwalk"], grid
    for x in range(8):
        a[KA["cbase"] + x], a[KA["clen"] + x], a[KA["rbase"] + x] = d["cbase"][x], d["clen"][x], d["rbase"][x]
        a[KA["pb"] + x], a[KA["mg_pb"] + x] = d["pb"][x], magic31(d["pb"][x])
    return a, d
