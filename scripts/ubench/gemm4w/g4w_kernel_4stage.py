"""Generator of the gemm4w instruction stream (design notes: csrc/gemm4w.hip).

One workgroup = 4 wavefronts, ONE per SIMD, each owning the whole 512-entry register file of its SIMD:
  a[0:255]   the wave's 128 x 128 fp32 accumulator (16 blocks of 32 x 32, block (mb, nb) at a[(4 mb + nb) * 16 ..])
  v[0:127]   P: the PREVIOUS tile's outputs as packed bf16 -- drained out of the accumulators at the tile seam, written out lazily
             (activation, LDS staging, whole-row stores) in the MFMA shadows of the current tile's K loop
  v[128:191] two fragment sets (8 ds_read_b128 each), v[192:207] T (drain / staging temporaries), v[208:227] bias fragments,
  the rest: addresses and lane constants.
The stream is a sequence of BLOCKS, one per K-tile (64 k): `F` (first K-tile of a tile), `mid` blocks, `L` (last K-tile).
A block is a list of MFMAs with FILLERS placed into the gaps between them by `Sched`: one wave issues about 8 instructions per
32-cycle MFMA, of which <= `cap` are planned per gap.

Order of the MFMAs.  mid blocks: k-substep outer (4 substeps of 16 MFMAs; a substep needs 8 fragments: 4 A row blocks + 4 W column
blocks).  F and L: QUADRANT major (8 substeps of 8 MFMAs: quadrant (2 x 2 blocks) x k-pair), so that the 16 accumulator blocks of a tile
finish -- and those of the next tile start -- 16 / 20 MFMAs apart instead of all within 16: the drain of a quadrant (64
v_accvgpr_read + 32 v_cvt_pk per lane) then has a window of ~48 gaps before the next tile's first MFMA on that quadrant.
The bias enters through the matrix pipe: the first MFMA of a block in F multiplies a fragment holding the column's bias as three
bf16 pieces (exact split of the fp32 value) by a fragment of ones, with C = 0."""
from .isa import A, EXEC, Ins, M0, S, V, label, resolve_waits, wait_lgkm, wait_vm

LDS_STAGE = 131072            # 4 waves x 4 KiB
LDS_BIAS = LDS_STAGE + 16384  # 2 x 1 KiB (tile parity)
LDS_MBOX = LDS_BIAS + 2048
LDS_TOTAL = LDS_MBOX + 64

# ---------------------------------------------------------------------------------------------------------------- registers
P = V(0, 128)
FSET = (V(128, 32), V(160, 32))
T = V(192, 16)
BF = V(208, 16)
ONES = V(224, 4)
RA = V(228, 2)   # fragment read bases of the A / W image for the two k-substeps of a stage (ring-pair bit toggled per block)
RW = V(232, 2)
IDXA, IDXW = V(236), V(237)
PE, PO = V(238, 2), V(240, 2)   # [row index, byte offset] pairs of the LDS-DMA source address: even / odd pieces
VRW = V(242)                    # 64 * wave + (lane >> 3): the lane's row within the workgroup's 256-row operand tile
STW, STRD, STIDX = V(243), V(244), V(245)
STP = V(246, 2)
BRD = V(248)
TK = V(249)
X = V(250, 4)
GC2 = V(254)
HM = V(255)
ACC = A(0, 256)

KARG = S(24, 2)
DA, DW, DC, DB = S(28, 4), S(32, 4), S(36, 4), S(40, 4)
A_PTR, W_PTR, B_PTR, C_PTR, SCHED = S(44, 2), S(46, 2), S(48, 2), S(50, 2), S(52, 2)
a_M, a_N, a_K, a_LDA, a_LDW, a_LDC = (S(54 + i) for i in range(6))
a_NK, a_TN, a_NSLOT, a_MGTN, a_MGNS, a_MGNK, a_CW, a_MGCW, a_CWL, a_MGCWL, a_NCB1, a_COLW, a_GRID = (S(60 + i) for i in range(13))
a_NK2 = S(73)      # stages per tile = 2 nk (computed)
TP = S(74, 2)      # 64-bit temporary (kernarg dwords 30, 31 are padding)
CBASE, CLEN, RBASE, PB, MGPB = (S(76 + i) for i in range(5))
XCD5, WAVE, BID = S(81), S(82), S(83)
TI, TIN, CM0, CN0, PM0, PCOL0, PCOL1, KOFF, KTD, DMABW, KLEFT, MORE, BSEL, NM0, NN0, NROT = (S(84 + i) for i in range(16))
T0, T1 = S(26), S(27)
T2, T3, T4, T5 = S(44), S(45), S(46), S(47)   # (the A / W pointers are dead once the descriptors exist)
SCHEDX = S(48, 2)                              # &sched[xcd]  (the bias pointer is dead as well)
T6, T7 = S(50), S(51)

KARG_DWORDS = 72
# kernarg dword offsets (host side: csrc/gemm4w.hip G4wArgs)
KA = dict(A=0, W=2, bias=4, C=6, sched=8, M=10, N=11, K=12, lda=13, ldw=14, ldc=15, nk=16, tiles_n=17, nslots=18, mg_tn=19, mg_ns=20,
          mg_nk=21, cw=22, mg_cw=23, cwl=24, mg_cwl=25, ncb1=26, colwalk=27, grid=28, cbase=32, clen=40, rbase=48, pb=56, mg_pb=64)


class Raw:
    """an inline-asm operand placeholder (%0 ...): text only"""

    def __init__(self, s):
        self.s = s

    def __repr__(self):
        return self.s


def issue_cost(i):
    """issue slots an instruction keeps its wave busy for, in units of a plain VALU / SALU instruction: what a gap's capacity is counted in.
    (measured with 5 fragment reads in one gap: every substep lost ~100 cycles -- a wave-wide ds_read_b128 or LDS-DMA piece is several
    plain slots long, and one wave per SIMD has nobody to issue the next MFMA meanwhile)"""
    k = i.kind
    if k in ("label", "waitvm", "waitlgkm", "wait", "entry"):
        return 0
    if k == "nop":
        return 1 + i.ops[0]
    if k == "ds":
        return 3 if i.op == "ds_read_b128" else 2
    if k in ("dma", "store", "atomic", "gmem", "trans"):
        return 2
    return 1


class Item:
    """A filler: instructions that stay together, a window of gaps [lo, hi], optionally a preferred gap."""
    __slots__ = ("ins", "lo", "hi", "want", "sec", "slots", "heavy")

    def __init__(self, ins, lo=None, hi=None, want=None, sec="body"):
        self.ins = ins if isinstance(ins, list) else [ins]
        self.lo, self.hi, self.want, self.sec = lo, hi, want, sec
        self.slots = sum(issue_cost(i) for i in self.ins)
        self.heavy = sum(1 for i in self.ins if i.kind in ("ds", "dma", "store", "atomic", "gmem"))   # wave-wide memory instructions


class Sched:
    """Gaps -1 .. n-1: gap g follows MFMA g (gap -1 precedes the first MFMA).  Each gap has three sections: pre, body, post."""

    def __init__(self, mfmas, cap, hcap=1):
        self.mfmas, self.n, self.cap, self.hcap = mfmas, len(mfmas), cap, hcap
        self.gaps = {g: {"pre": [], "body": [], "post": []} for g in range(-1, self.n)}
        self.count = {g: 0 for g in range(-1, self.n)}
        self.hcount = {g: 0 for g in range(-1, self.n)}
        self.over = 0

    def fixed(self, g, ins, sec="body"):
        it = ins if isinstance(ins, Item) else Item(ins)
        self.gaps[g][sec].append(it)
        self.count[g] += it.slots
        self.hcount[g] += it.heavy

    def stream(self, items, lo, hi, spread=True):
        """Place the items of one in-order stream between gaps lo and hi: evenly (spread) or as early as the caps allow."""
        n = len(items)
        g_prev = lo
        for i, it in enumerate(items):
            ilo = lo if it.lo is None else max(lo, it.lo)
            ihi = hi if it.hi is None else min(hi, it.hi)
            ihi = max(ihi, ilo)
            want = ilo + (ihi - ilo) * (i + 0.5) / n if spread else ilo
            if it.want is not None:
                want = it.want
            g = min(max(ilo, int(want), g_prev), ihi)
            g = max(g, g_prev)
            while g < ihi and (self.count[g] + it.slots > self.cap or (it.heavy and self.hcount[g] + it.heavy > self.hcap)):
                g += 1
            if self.count[g] + it.slots > self.cap:
                self.over += min(it.slots, self.count[g] + it.slots - self.cap)
            self.gaps[g][it.sec].append(it)
            self.count[g] += it.slots
            self.hcount[g] += it.heavy
            g_prev = g

    def emit(self):
        out = []
        for g in range(-1, self.n):
            if g >= 0:
                out.append(self.mfmas[g])
            for sec in ("pre", "body", "post"):
                for it in self.gaps[g][sec]:
                    out.extend(it.ins)
        return out


def quad_of(q):
    """quadrant q -> (row blocks, column blocks)"""
    mp, np_ = q >> 1, q & 1
    return (2 * mp, 2 * mp + 1), (2 * np_, 2 * np_ + 1)


class Gen:
    def __init__(self, epi=0, nt=False, cap=6, n3_mid=4, n3_seam=4, bar_after=2, uid="0", drain_cap=4, cap_f=None, cap_l=None, cap_pre=None,
                 cap_mid=None, units_per_mid=None, null_desc=1, piece_nop=-1, bar_gap=0, abl=0, hcap=1):
        self.hcap = hcap
        self.null_desc, self.piece_nop, self.bar_gap, self.abl = null_desc, piece_nop, bar_gap, abl   # abl (timing probes, wrong results): 1 no LDS-DMA in the blocks, 2 no MFMAs, 3 no drain / epilogue, 4 no barriers
        self.epi, self.nt, self.cap = epi, nt, cap
        self.cap_by = {"F": cap_f, "L": cap_l, "pre": cap_pre, "mid": cap_mid}
        self.n3_mid, self.n3_seam, self.bar_after = n3_mid, n3_seam, bar_after
        self.uid, self.drain_cap = uid, drain_cap
        self.units_per_mid = (units_per_mid or 2) if epi != 1 else 1
        self.n_units_mid = 8 // self.units_per_mid     # mid blocks that stage epilogue units (the last unit's stores: one block more)
        self.J = max(4, self.n_units_mid + 1)          # unrolled mid blocks (chores: 0 ticket, 1 next-tile parameters, 3 bias fragments)
        self.min_nk = 2 + self.J + 1                   # F + J unrolled + [plain loop] + pre-last + L
        self.over = {}
        self.lbl = 0

    def L(self, name):
        return f".Lg4w{self.uid}_{name}"

    def newlabel(self, stem):
        self.lbl += 1
        return self.L(f"{stem}{self.lbl}")

    # ------------------------------------------------------------------------------------------------------------ pieces
    def acc(self, mb, nb):
        return ACC[(mb * 4 + nb) * 16:(mb * 4 + nb) * 16 + 16]

    def mfma(self, mb, nb, wf, af):
        return Ins("v_mfma_f32_32x32x16_bf16", self.acc(mb, nb), wf, af, self.acc(mb, nb))

    def frag_read(self, dst, is_a, idx, ks, tag):
        """fragment (32-row block idx, k-substep ks of the 64-k block) of the A / W image: ks 0, 1 lie in the block's first stage, 2, 3 in
        its second one (the ring slot 32 KiB further)"""
        return Ins("ds_read_b128", dst, (RA if is_a else RW)[ks & 1], offset=(ks >> 1) * 32768 + idx * 2048, tag=tag)

    def substeps(self, pattern):
        """[(fragment reads [(slot, is_a, idx, ks)], MFMAs [(mb, nb, w slot, a slot)])]"""
        out = []
        if pattern == "ks":
            for ks in range(4):
                reads = [(mb, True, mb, ks) for mb in range(4)] + [(4 + nb, False, nb, ks) for nb in range(4)]
                out.append((reads, [(mb, nb, 4 + nb, mb) for mb in range(4) for nb in range(4)]))
        else:
            for j in range(8):
                mbs, nbs = quad_of(j >> 1)
                kp = j & 1
                reads, mf = [], []
                for kk in range(2):
                    for i in range(2):
                        reads.append((kk * 2 + i, True, mbs[i], 2 * kp + kk))
                    for i in range(2):
                        reads.append((4 + kk * 2 + i, False, nbs[i], 2 * kp + kk))
                for kk in range(2):
                    for i in range(2):
                        for i2 in range(2):
                            mf.append((mbs[i], nbs[i2], 4 + kk * 2 + i2, kk * 2 + i))
                out.append((reads, mf))
        return out

    def read_items(self, reads, fset, tag):
        return [Item(self.frag_read(fset[4 * slot:4 * slot + 4], is_a, idx, ks, tag)) for slot, is_a, idx, ks in reads]

    def dma_piece(self, p, tag):
        """this wave's piece p of an 8-piece group = one STAGE (32 k) of the operand tiles (0..3: A rows 64 w + 16 p .., 4..7: W rows):
        16 rows x 64 B into the ring slot at DMABW"""
        is_a, q = p < 4, p % 4
        pair = PE if q % 2 == 0 else PO
        dst = (0 if is_a else 16384) + q * 1024
        ins = [Ins("s_add_u32", M0, DMABW, dst),
               Ins("v_add_u32", pair[0], 16 * q, IDXA if is_a else IDXW),
               Ins("buffer_load_dwordx4", pair, DA if is_a else DW, KOFF, addr="idxen offen", tag=tag)]
        if self.abl == 1 and tag != "pro":
            ins[2] = Ins("s_nop", 0, tag=tag)
        if self.piece_nop >= 0:
            ins.append(Ins("s_nop", self.piece_nop))
        return Item(ins)

    def group_end(self):
        """after the last piece of a group: the next ring slot, the next stage of the tile the stream is in"""
        return Item([Ins("s_add_u32", DMABW, DMABW, 32768), Ins("s_and_b32", DMABW, DMABW, 0x1FFFF),
                     Ins("s_add_u32", KTD, KTD, 1), Ins("s_cmp_eq_u32", KTD, a_NK2), Ins("s_cselect_b32", KTD, 0, KTD),
                     Ins("s_lshl_b32", KOFF, KTD, 6)])

    def r_toggle(self):
        return [Item(Ins("v_xor_b32", r, 65536, r)) for r in [RA[0], RA[1], RW[0], RW[1]]]

    def drain_block(self, mb, nb):
        b = mb * 4 + nb
        items = []
        for r in range(16):
            items.append(Item(Ins("v_accvgpr_read_b32", T[r], ACC[b * 16 + r])))
        for j in range(8):
            items.append(Item(Ins("v_cvt_pk_bf16_f32", P[b * 8 + j], T[2 * j], T[2 * j + 1])))
        return items

    def gelu_reg(self, preg, t):
        """one P register (two bf16 pre-activations) -> gelu -> packed back; t: 6 temporaries.  x Phi(x), Phi = 1 / (1 + 2^(x p(x^2)))
        (gemm_common.h gelu_bf16_class; on the bf16-rounded pre-activation, as torch's autocast evaluates it); the two elements are
        interleaved so that no transcendental result is consumed by the very next instruction."""
        xl, xh, a, b, c, d = t
        C0, C1 = -2.3011213, -0.10677572
        return [
            Ins("v_lshlrev_b32", xl, 16, preg), Ins("v_and_b32", xh, 0xFFFF0000, preg),
            Ins("v_mul_f32", a, xl, xl), Ins("v_mul_f32", b, xh, xh),
            Ins("v_min_f32", a, 64.0, a), Ins("v_min_f32", b, 64.0, b),
            Ins("v_fmaak_f32", c, GC2, a, C1), Ins("v_fmaak_f32", d, GC2, b, C1),
            Ins("v_fmaak_f32", c, c, a, C0), Ins("v_fmaak_f32", d, d, b, C0),
            Ins("v_mul_f32", c, xl, c), Ins("v_mul_f32", d, xh, d),
            Ins("v_exp_f32", c, c), Ins("v_exp_f32", d, d),
            Ins("v_add_f32", c, 1.0, c), Ins("v_add_f32", d, 1.0, d),
            Ins("v_rcp_f32", c, c), Ins("v_rcp_f32", d, d),
            Ins("v_mul_f32", xl, xl, c), Ins("v_mul_f32", xh, xh, d),
            Ins("v_cvt_pk_bf16_f32", preg, xl, xh),
        ]

    def unit_blocks(self, u):
        """the two accumulator blocks of epilogue unit u = (row block mb, column pair h): 32 rows x 64 columns"""
        mb, h = u >> 1, u & 1
        return [mb * 4 + 2 * h + nbl for nbl in range(2)]

    def epi_stage(self, u, temps=None):
        """phase A of unit u: P -> [activation] -> staging (8 ds_write_b64) -> 4 ds_read_b128 of whole 128-byte row segments into T"""
        items = []
        k = 0
        for nbl, blk in enumerate(self.unit_blocks(u)):
            for g in range(4):
                pr = P[blk * 8 + 2 * g:blk * 8 + 2 * g + 2]
                if self.epi == 1:
                    for preg in (pr[0], pr[1]):
                        tt = [temps[6 * (k % 2) + i] for i in range(6)]
                        k += 1
                        items += [Item(i) for i in self.gelu_reg(preg, tt)]
                if self.epi == 2:   # ReLU on the packed bf16 pair: a negative bf16 is a negative int16
                    items += [Item(Ins("v_pk_max_i16", pr[0], pr[0], 0)), Item(Ins("v_pk_max_i16", pr[1], pr[1], 0))]
                c = nbl * 4 + g
                xa = X[c % 2]
                if c == 0:
                    items.append(Item(Ins("ds_write_b64", STW, pr, tag="ew")))
                else:
                    items.append(Item([Ins("v_xor_b32", xa, c << 4, STW), Ins("ds_write_b64", xa, pr, tag="ew")]))
        for it in range(4):
            items.append(Item(Ins("ds_read_b128", T[4 * it:4 * it + 4], STRD, offset=it * 1024, tag="er")))
        return items

    def epi_store(self, u):
        """phase B of unit u: the 4 stores of 8 rows x 128 B (whole cache lines) out of T"""
        mb, h = u >> 1, u & 1
        items = [Item(wait_lgkm("er"))]
        for it in range(4):
            items.append(Item([Ins("v_add_u32", STP[0], mb * 32 + it * 8, STIDX),
                               Ins("buffer_store_dwordx4", T[4 * it:4 * it + 4], STP, DC, PCOL0 if h == 0 else PCOL1, addr="idxen offen",
                                   nt=self.nt, tag="st")]))
        return items

    def unit_temps(self, u):
        """12 temporaries of unit u's activation: T for a tile's first unit (nothing is pending in T then), afterwards the P registers of
        the unit before (written to the staging area already)"""
        if u == 0:
            return [T[i] for i in range(12)]
        b0, b1 = self.unit_blocks(u - 1)
        return [P[b0 * 8 + i] for i in range(8)] + [P[b1 * 8 + i] for i in range(4)]

    # ------------------------------------------------------------------------------------------------------------ scalar tile parameters
    def tile_params(self, ticket, m0, n0, rot, valid=None):
        """SALU, branch-free: ticket -> (m0, n0, rot) of the tile.  Row-major within the XCD's chunk, or the column-blocked walk
        (gemm_kernel.h tile_params) when a_COLW.  Each Item keeps its SCC producer and consumers together."""
        it = []

        def div(q, n, magic, tmp):   # q = n / d with magic = ceil(2^31 / d): mul_hi(2 n + 1, magic), exact for n d < 2^30 (d = 1 included)
            return [Ins("s_lshl_b32", tmp, n, 1), Ins("s_or_b32", tmp, tmp, 1), Ins("s_mul_hi_u32", q, tmp, magic)]
        # row-major: t = cbase + ticket; tm = t / tiles_n; tn = t - tm * tiles_n        -> T2 (tm), T3 (tn)
        it.append(Item([Ins("s_add_u32", T0, CBASE, ticket)] + div(T2, T0, a_MGTN, T1) + [Ins("s_mul_i32", T1, T2, a_TN), Ins("s_sub_u32", T3, T0, T1)]))
        # column-blocked: cb = ticket / pb; v = ticket - cb * pb; cwb = last block ? cwl : cw; g = v / cwb; tl = v - g * cwb
        it.append(Item(div(T0, ticket, MGPB, T1) + [Ins("s_mul_i32", T1, T0, PB), Ins("s_sub_u32", T1, ticket, T1)]))     # T0 = cb, T1 = v
        it.append(Item([Ins("s_cmp_eq_u32", T0, a_NCB1), Ins("s_cselect_b32", T4, a_CWL, a_CW), Ins("s_cselect_b32", T5, a_MGCWL, a_MGCW)]))
        it.append(Item(div(T5, T1, T5, T6) + [Ins("s_mul_i32", T4, T5, T4), Ins("s_sub_u32", T4, T1, T4)]))              # T5 = g, T4 = tl
        it.append(Item([Ins("s_mul_i32", T0, T0, a_CW), Ins("s_add_u32", T4, T0, T4), Ins("s_add_u32", T5, RBASE, T5)]))           # T4 = tn, T5 = tm
        it.append(Item([Ins("s_cmp_eq_u32", a_COLW, 0), Ins("s_cselect_b32", m0, T2, T5), Ins("s_cselect_b32", n0, T3, T4)]))
        it.append(Item([Ins("s_lshl_b32", m0, m0, 8), Ins("s_lshl_b32", n0, n0, 8)]))
        # K rotation: (xcd * 5 + (ticket / nslots) * 3) % nk
        it.append(Item(div(T0, ticket, a_MGNS, T1) + [Ins("s_mul_i32", T0, T0, 3), Ins("s_add_u32", T0, T0, XCD5)] + div(T1, T0, a_MGNK, T6) +
                       [Ins("s_mul_i32", T1, T1, a_NK), Ins("s_sub_u32", rot, T0, T1)]))
        if valid is not None:   # a ticket past the end of the chunk: harmless values
            it.append(Item([Ins("s_cmp_lt_u32", ticket, CLEN), Ins("s_cselect_b32", valid, 1, 0), Ins("s_cselect_b32", m0, m0, 0),
                            Ins("s_cselect_b32", n0, n0, 0), Ins("s_cselect_b32", rot, rot, 0)]))
        return it

    def stream_switch(self):
        """the LDS-DMA stream moves on to the next tile (NM0, NN0, NROT); no next tile: null descriptors (the loads touch nothing)"""
        ins = [Ins("v_add_u32", IDXA, NM0, VRW), Ins("v_add_u32", IDXW, NN0, VRW), Ins("s_lshl_b32", KTD, NROT, 1), Ins("s_lshl_b32", KOFF, NROT, 7)]
        if self.null_desc:
            ins += [Ins("s_cmp_eq_u32", MORE, 0), Ins("s_cselect_b32", DA[2], 0, DA[2]), Ins("s_cselect_b32", DW[2], 0, DW[2])]
        return Item(ins)   # (without null descriptors the loads past the last tile read tile (0, 0): NM0 = NN0 = NROT = 0 then)

    def tile_rotate(self):
        """at the seam: the tile just computed becomes `previous` (its outputs are written during the next tile), the next one current"""
        s = [Ins("s_mov_b32", PM0, CM0)]
        # column byte offset of the wave's 128 columns: (n0 + wn * 128) * 2
        s += [Ins("s_and_b32", T0, WAVE, 1), Ins("s_lshl_b32", T0, T0, 7), Ins("s_add_u32", T0, T0, CN0), Ins("s_lshl_b32", PCOL0, T0, 1),
              Ins("s_add_u32", PCOL1, PCOL0, 128)]
        # row index of the lane's staging-read row: m0 + wm * 128 + (lane >> 3)
        s += [Ins("s_lshr_b32", T0, WAVE, 1), Ins("s_lshl_b32", T0, T0, 7), Ins("s_add_u32", T0, T0, CM0)] + self.lane_id(X[0]) + \
             [Ins("v_lshrrev_b32", X[0], 3, X[0]), Ins("v_add_u32", STIDX, T0, X[0])]
        s += [Ins("s_mov_b32", DC[2], a_M)]   # stores on (a workgroup's first tile has no predecessor: 0 records until here)
        s += [Ins("s_mov_b32", CM0, NM0), Ins("s_mov_b32", CN0, NN0), Ins("s_mov_b32", TI, TIN), Ins("s_xor_b32", BSEL, BSEL, 1024)]
        return s

    def lane_id(self, dst):
        return [Ins("v_mbcnt_lo_u32_b32", dst, -1, 0), Ins("v_mbcnt_hi_u32_b32", dst, -1, dst)]

    def bias_dma(self, n0, slot_sgpr_expr):
        """the tile's 256 bias values (1 KiB) into the bias slot: every wave issues the same piece (its own vmcnt covers the copy it reads)"""
        return Item(self.lane_id(X[2]) + [Ins("v_lshlrev_b32", X[2], 4, X[2]), Ins("s_lshl_b32", T0, n0, 2)] + slot_sgpr_expr +
                    [Ins("s_add_u32", M0, T1, LDS_BIAS), Ins("s_nop", 0), Ins("buffer_load_dwordx4", X[2], DB, T0, addr="offen", tag="b")])

    def bias_frags(self, slot_imm_or_none, T=T):
        """bias slice (LDS) -> the four bias fragments: lane (n = l31, hi = 0) holds b0 | b1 << 16, b2 (exact 3-way bf16 split), others 0.
        Reads slot BSEL ^ 1024 of the NEXT tile (slot_imm_or_none None) or a fixed slot (prologue)."""
        items = []
        if slot_imm_or_none is None:
            items.append(Item([Ins("s_xor_b32", T0, BSEL, 1024), Ins("v_add_u32", X[3], T0, BRD)]))
            addr, off0 = X[3], 0
        else:
            addr, off0 = BRD, slot_imm_or_none
        for nb in range(4):
            items.append(Item(Ins("ds_read_b32", T[nb], addr, offset=off0 + nb * 128, tag="bf")))
        items.append(Item(wait_lgkm("bf")))
        for nb in range(4):
            b, r1, r2, b0 = T[nb], T[4 + nb], T[8 + nb], T[12 + nb]
            items += [Item(Ins("v_and_b32", b0, 0xFFFF0000, b)), Item(Ins("v_sub_f32", r1, b, b0)),
                      Item(Ins("v_and_b32", b, 0xFFFF0000, r1)),        # b1
                      Item(Ins("v_sub_f32", r2, r1, b)),                 # b2 (<= 8 significant bits)
                      Item(Ins("v_lshrrev_b32", b0, 16, b0)), Item(Ins("v_or_b32", b0, b0, b)), Item(Ins("v_and_b32", BF[4 * nb], HM, b0)),
                      Item(Ins("v_lshrrev_b32", r2, 16, r2)), Item(Ins("v_and_b32", BF[4 * nb + 1], HM, r2))]
        return items

    def wave0_only(self, body):
        skip = self.newlabel("w0")
        return [Ins("s_cmp_lg_u32", WAVE, 0), Ins("s_cbranch_scc1", skip)] + body + [label(skip)]

    def ticket_atomic(self):
        """wave 0, lane 0: the next tile's ticket = sched[xcd]++; the result lands in TK behind the counted waits of the stream"""
        body = [Ins("s_mov_b64", TP, EXEC), Ins("s_mov_b64", EXEC, 1), Ins("v_mov_b32", X[2], 0), Ins("v_mov_b32", X[3], 1),
                Ins("global_atomic_add", TK, X[2], X[3], SCHEDX, sc0=True, tag="tk", uncounted=True), Ins("s_mov_b64", EXEC, TP)]
        return Item(self.wave0_only(body))

    def ticket_post(self):
        """wave 0 (its ticket has landed: an older load than the group just waited for): post it in LDS"""
        body = [Ins("s_waitcnt", kind="wait", vmcnt=0, uncounted=True, tag="tkwait"),   # (count patched: loads issued since the ticket atomic)
                Ins("v_readfirstlane_b32", T0, TK), Ins("v_mov_b32", X[2], LDS_MBOX), Ins("v_mov_b32", X[3], T0),
                Ins("ds_write_b32", X[2], X[3], tag="mbw", uncounted=True), Ins("s_waitcnt", kind="wait", lgkmcnt=0, uncounted=True)]
        return self.wave0_only(body)

    # ------------------------------------------------------------------------------------------------------------ blocks
    def block(self, kind, j, vm_in, lg_in, next_quad, drain=None):
        """One 64-k block = two 32-k STAGES of the 4-slot ring.  kind: 'F' | 'mid' | 'L'; j: index of a mid block (0..J-1 unrolled with
        chores, 'loop', 'pre').  vm_in: tags of the LDS-DMA groups still outstanding at entry, oldest first, relative to this block:
        'A' / 'B' its own stages, 'A2' / 'B2' the next block's, (issued here: 'A4' / 'B4' the stages two blocks ahead).
        Two barriers: X after the last fragment read of the first stage (its slot is refilled with stage A4), Y after the last read
        of the second (refilled with B4; the next block's first stage must have landed).  The waits in front of them leave the
        younger groups in flight: two to three stages (64 - 96 KiB per CU) cross every barrier."""
        quad = kind != "mid"
        subs = self.substeps("quad" if quad else "ks")
        nsub = len(subs)
        mf, sub_first, sub_last = [], [], []
        for i, (reads, ms) in enumerate(subs):
            if kind == "F" and i % 2 == 0:   # the quadrant's bias MFMAs (C = 0) open its accumulation
                mbs, nbs = quad_of(i >> 1)
                for mb in mbs:
                    for nb in nbs:
                        mf.append(Ins("v_mfma_f32_32x32x16_bf16", self.acc(mb, nb), BF[4 * nb:4 * nb + 4], ONES, 0))
            sub_first.append(len(mf))
            fs = FSET[i & 1]
            for mb, nb, ws, as_ in ms:
                mf.append(self.mfma(mb, nb, fs[4 * ws:4 * ws + 4], fs[4 * as_:4 * as_ + 4]))
            sub_last.append(len(mf) - 1)
        n = len(mf)
        sc = Sched(mf, self.cap_by.get("pre" if j == "pre" else kind) or self.cap, self.hcap if not quad else max(2, self.hcap))
        n3 = self.n3_mid if not quad else self.n3_seam
        last = nsub - 1
        ia = 6 if quad else 1                       # the substep whose fragments are the last reads of the block's first stage
        gx = sub_first[ia] + self.bar_after - 1     # barrier X follows MFMA gx, Y follows MFMA gy
        gy = sub_first[last] + self.bar_after - 1
        # ---- entry: this block's first fragments were read by the previous block
        sc.fixed(-1, Item(wait_lgkm("f0")), "post")
        # ---- fragment reads one substep ahead (mid blocks: the second stage's first reads wait for barrier X)
        for i in range(nsub - 1):
            tag = f"f{(i + 1) & 1}"
            lo = sub_first[i] if quad or i != 1 else gx + 1
            sc.stream(self.read_items(subs[i + 1][0], FSET[(i + 1) & 1], tag), lo, max(lo, sub_last[i] - 2), spread=False)
            sc.fixed(sub_first[i + 1] - 1, Item(wait_lgkm(tag)), "post")
        # ---- barriers
        barx = [] if quad else [wait_vm("B")]
        if kind == "mid" and j == 0:
            barx += self.ticket_post()
        barx.append(Ins("s_barrier"))
        sc.fixed(gx, Item(barx), "post")
        sc.fixed(gy, Item([wait_vm(("A2", "B2") if next_quad else ("A2",)), Ins("s_barrier")]), "post")
        sc.stream(self.r_toggle(), sub_first[last - 1] + len(subs[last][0]) + 1, gy, spread=True)
        # ---- the LDS-DMA stream of the block, in order: the rest of the group begun by the block before (B2), the refill of the first
        #      stage's slot behind X (A4), the start of the refill of the second one behind Y (B4)
        have = len([t for t in vm_in if t == "B2"])
        dma = [self.dma_piece(p, "B2") for p in range(have, 8)]
        if have < 8:
            dma.append(self.group_end())
        if kind == "mid" and j == "pre":   # the stream moves on to the next tile between two groups
            dma.append(self.stream_switch())
        a4 = [self.dma_piece(p, "A4") for p in range(8)] + [self.group_end()]
        for it in a4:
            it.lo = gx + 1
        b4 = [self.dma_piece(p, "B4") for p in range(n3)] + ([self.group_end()] if n3 == 8 else [])
        for it in b4:
            it.lo = gy + 1
        # ---- behind Y: the next block's first fragments
        after = self.read_items(self.substeps("quad" if next_quad else "ks")[0][0], FSET[0], "f0")
        sc.stream(after, gy + 1, n - 1, spread=False)
        sc.stream(dma + a4 + b4, 0, n - 1, spread=False)
        if kind == "mid" and j == 0:   # everybody picks the next tile's ticket up behind X
            sc.stream([Item([Ins("v_mov_b32", X[2], LDS_MBOX), Ins("s_nop", 0), Ins("ds_read_b32", TK, X[2], tag="mb")])], gx + 1, gy - 1, spread=False)
        # ---- block-specific work
        if kind == "F":
            chores = [self.ticket_atomic(), Item(Ins("s_sub_u32", KLEFT, a_NK, 2 + self.J + 1))]
            sc.stream(chores, 2, 12, spread=False)
        if kind == "mid" and j == 1:
            ch = [Item([wait_lgkm("mb"), Ins("v_readfirstlane_b32", TIN, TK)])] + self.tile_params(TIN, NM0, NN0, NROT, valid=MORE)
            ch.append(self.bias_dma(NN0, [Ins("s_xor_b32", T1, BSEL, 1024)]))
            sc.stream(ch, 0, gy - 2, spread=True)
        if kind == "mid" and j == 3:
            sc.stream(self.bias_frags(None, T=P[0:16]), gx + 1, gy - 1, spread=True)   # (P[0:16]: unit 0 left them in mid block 0)
        if kind == "mid" and j == "loop":
            sc.stream([Item(Ins("s_sub_u32", KLEFT, KLEFT, 1))], 20, 30, spread=False)
        if kind == "mid" and isinstance(j, int) and j <= self.n_units_mid:
            # (the stores of a block's last unit are issued at the start of the next block: see epi_store)
            U = self.units_per_mid
            ep = []
            if j > 0:
                ep += self.epi_store(j * U - 1)
            if j < self.n_units_mid:
                for u in range(j * U, (j + 1) * U):
                    ep += self.epi_stage(u, self.unit_temps(u))
                    if u != (j + 1) * U - 1:
                        ep += self.epi_store(u)
            if self.abl != 3:
                sc.stream(ep, 1, gy - 1, spread=(self.epi == 1))
        if drain and self.abl != 3:
            sc.stream(drain, 0, n - 1, spread=False)
        self.over[(kind, j)] = sc.over
        seq = sc.emit()
        if self.abl == 2:
            seq = [i for i in seq if i.kind != "mfma"]
        if self.abl == 4:
            seq = [i for i in seq if i.kind != "barrier"]
        seq, vm, lg = resolve_waits(seq, [(t, True) for t in vm_in], lg_in)
        # exit state, relative to the next block
        loads = [t for t, ld in vm if ld]
        if self.abl == 1:
            loads = [t for t in loads if t not in ("A", "B", "A2", "B2", "A4", "B4")] + ([] if next_quad else ["B2"] * 8) + ["A4"] * 8 + ["B4"] * n3
        assert "A" not in loads and "B" not in loads and loads.count("A4") == 8 and loads.count("B4") == n3, (kind, j, loads)
        ren = {"A2": "A", "B2": "B", "A4": "A2", "B4": "B2"}
        return seq, [ren.get(t, t) for t in loads], lg

    # drain schedule over the seam: global gap coordinates (L: 0..63, F: 64 + F gap)
    def drain_plan(self):
        items_l, items_f = [], []
        g = 0.0
        per_gap = self.drain_cap
        for q in range(4):
            mbs, nbs = quad_of(q)
            lo = 16 * q + 15 + 3
            hi = 64 + 20 * q - 1
            g = max(g, lo)
            for mb in mbs:
                for nb in nbs:
                    for it in self.drain_block(mb, nb):
                        gi = int(g)
                        gi = min(gi, hi)
                        if gi < 64:
                            it.lo, it.hi, it.want = lo if lo < 64 else 0, 63, gi
                            items_l.append(it)
                        else:
                            it.lo, it.hi, it.want = max(0, lo - 64), hi - 64, gi - 64
                            items_f.append(it)
                        g += it.slots / per_gap
        return items_l, items_f

    # ------------------------------------------------------------------------------------------------------------ whole kernel
    def prologue(self):
        s = []
        s += [Ins("s_mov_b64", KARG, Raw("%0"), kind="entry"), Ins("s_mov_b32", WAVE, Raw("%1"), kind="entry"), Ins("s_mov_b32", BID, Raw("%2"), kind="entry")]
        s += [Ins("s_load_dwordx16", S(44, 16), KARG, offset=0), Ins("s_load_dwordx16", S(60, 16), KARG, offset=64)]
        # TP (s74:75) is inside the second load's destination range: nothing may write it before the load has returned (an s_load that
        # lands late overwrites whatever a SALU instruction put there in the meantime -- there is no interlock on SMEM destinations)
        s.append(Ins("s_waitcnt", kind="wait", lgkmcnt=0))
        s += [Ins("s_and_b32", T0, BID, 7), Ins("s_mul_i32", XCD5, T0, 5), Ins("s_lshl_b32", T0, T0, 2),
              Ins("s_add_u32", TP[0], KARG[0], T0), Ins("s_addc_u32", TP[1], KARG[1], 0)]
        for reg, key in ((CBASE, "cbase"), (CLEN, "clen"), (RBASE, "rbase"), (PB, "pb"), (MGPB, "mg_pb")):
            s.append(Ins("s_load_dword", reg, TP, offset=4 * KA[key]))
        s += [Ins("s_waitcnt", kind="wait", lgkmcnt=0), Ins("s_lshl_b32", a_NK2, a_NK, 1)]
        # descriptors.  A / W / C: structured (stride = row bytes, records = rows: a row index past the end reads zeros / drops the store)
        FLAGS = 0x00020000
        for d, ptr, ld, rec in ((DA, A_PTR, a_LDA, a_M), (DW, W_PTR, a_LDW, a_N), (DC, C_PTR, a_LDC, None)):
            s += [Ins("s_mov_b32", d[0], ptr[0]), Ins("s_lshl_b32", T0, ld, 16), Ins("s_and_b32", T1, ptr[1], 0xFFFF), Ins("s_or_b32", d[1], T1, T0),
                  Ins("s_mov_b32", d[2], rec if rec is not None else 0), Ins("s_mov_b32", d[3], FLAGS)]
        s += [Ins("s_mov_b32", DB[0], B_PTR[0]), Ins("s_and_b32", DB[1], B_PTR[1], 0xFFFF), Ins("s_lshl_b32", DB[2], a_N, 2), Ins("s_mov_b32", DB[3], FLAGS)]
        # &sched[xcd]
        s += [Ins("s_and_b32", T0, BID, 7), Ins("s_lshl_b32", T0, T0, 2), Ins("s_add_u32", T6, SCHED[0], T0), Ins("s_addc_u32", T7, SCHED[1], 0),
              Ins("s_mov_b32", SCHEDX[0], T6), Ins("s_mov_b32", SCHEDX[1], T7)]
        # ---- lane constants.  T[..] are free temporaries here.
        lane, l31, hi, r, c7, fx, base = (T[i] for i in range(7))
        s += self.lane_id(lane)
        s += [Ins("v_and_b32", l31, 31, lane), Ins("v_lshrrev_b32", hi, 5, lane), Ins("v_lshrrev_b32", r, 3, lane), Ins("v_and_b32", c7, 7, lane)]
        # fragment reads.  LDS image of a stage: [256 rows][64 B], the row's 16-byte chunk c at position c ^ ((row >> 2) & 3) (the XOR is
        # applied to the SOURCE address of the LDS-DMA; every 16-lane group of a ds_read_b128 then covers all 64 banks once).
        # A block (row base a multiple of 32), k-substep ks' of the stage: byte = (base + l31) * 64 + (((2 ks' + hi) ^ ((l31 >> 2) & 3)) << 4)
        s += [Ins("v_lshrrev_b32", fx, 2, l31), Ins("v_and_b32", fx, 3, fx), Ins("v_lshlrev_b32", base, 6, l31)]
        # + wm * 8 KiB (A image of the slot), 16 KiB + wn * 8 KiB (W image)
        s += [Ins("s_lshr_b32", T0, WAVE, 1), Ins("s_lshl_b32", T0, T0, 13), Ins("s_and_b32", T1, WAVE, 1), Ins("s_lshl_b32", T1, T1, 13),
              Ins("s_add_u32", T1, T1, 16384)]
        for ks in range(2):
            s += [Ins("v_or_b32", T[9], 2 * ks, hi), Ins("v_xor_b32", T[9], T[9], fx), Ins("v_lshlrev_b32", T[9], 4, T[9]), Ins("v_add_u32", T[9], T[9], base),
                  Ins("v_add_u32", RA[ks], T0, T[9]), Ins("v_add_u32", RW[ks], T1, T[9])]
        # LDS-DMA source of a piece (16 rows x 64 B): lane -> row (lane >> 2) of the piece, chunk (lane & 3) ^ ((row >> 2) & 3)
        s += [Ins("v_lshrrev_b32", T[9], 2, lane), Ins("s_lshl_b32", T0, WAVE, 6), Ins("v_add_u32", VRW, T0, T[9])]
        s += [Ins("v_lshrrev_b32", T[9], 4, lane), Ins("v_and_b32", T[9], 3, T[9]), Ins("v_and_b32", T[10], 3, lane), Ins("v_xor_b32", T[9], T[9], T[10]),
              Ins("v_lshlrev_b32", PE[1], 4, T[9]), Ins("v_mov_b32", PO[1], PE[1])]
        # staging: write base (row l31, 16-byte slot XOR-swizzled by the row, 8-byte half hi), read base (row lane >> 3, slot lane & 7)
        s += [Ins("s_lshl_b32", T0, WAVE, 12), Ins("s_add_u32", T0, T0, LDS_STAGE)]
        s += [Ins("v_lshlrev_b32", T[9], 7, l31), Ins("v_and_b32", T[10], 7, l31), Ins("v_lshlrev_b32", T[10], 4, T[10]), Ins("v_add_u32", T[9], T[9], T[10]),
              Ins("v_lshlrev_b32", T[10], 3, hi), Ins("v_add_u32", T[9], T[9], T[10]), Ins("v_add_u32", STW, T0, T[9])]
        s += [Ins("v_lshlrev_b32", T[9], 7, r), Ins("v_xor_b32", T[10], c7, r), Ins("v_lshlrev_b32", T[10], 4, T[10]), Ins("v_add_u32", T[9], T[9], T[10]),
              Ins("v_add_u32", STRD, T0, T[9]), Ins("v_lshlrev_b32", STP[1], 4, c7)]
        # bias read address: LDS_BIAS + (wn * 128 + l31) * 4
        s += [Ins("s_and_b32", T0, WAVE, 1), Ins("s_lshl_b32", T0, T0, 9), Ins("s_add_u32", T0, T0, LDS_BIAS), Ins("v_lshlrev_b32", T[9], 2, l31),
              Ins("v_add_u32", BRD, T0, T[9])]
        # HM = all ones in the lanes with hi == 0; fragment of ones (k-slots 0..2); zero halves of the bias fragments; GELU constant
        s += [Ins("v_sub_u32", HM, hi, 1), Ins("v_and_b32", ONES[0], 0x3F803F80, HM), Ins("v_and_b32", ONES[1], 0x00003F80, HM),
              Ins("v_mov_b32", ONES[2], 0), Ins("v_mov_b32", ONES[3], 0), Ins("v_mov_b32", GC2, 1.0142630e-3)]
        for nb in range(4):
            s += [Ins("v_mov_b32", BF[4 * nb + 2], 0), Ins("v_mov_b32", BF[4 * nb + 3], 0)]
        # ---- first ticket
        s += self.wave0_only([Ins("s_mov_b64", TP, EXEC), Ins("s_mov_b64", EXEC, 1), Ins("v_mov_b32", X[2], 0), Ins("v_mov_b32", X[3], 1),
                              Ins("global_atomic_add", TK, X[2], X[3], SCHEDX, sc0=True, tag="tk"), Ins("s_waitcnt", kind="wait", vmcnt=0),
                              Ins("v_mov_b32", X[2], LDS_MBOX), Ins("s_nop", 0), Ins("ds_write_b32", X[2], TK, tag="mbw"),   # (lane 0 only)
                              Ins("s_waitcnt", kind="wait", lgkmcnt=0), Ins("s_mov_b64", EXEC, TP)])
        s += [Ins("s_barrier"), Ins("v_mov_b32", X[2], LDS_MBOX), Ins("s_nop", 0), Ins("ds_read_b32", TK, X[2], tag="mb"),
              Ins("s_waitcnt", kind="wait", lgkmcnt=0), Ins("v_readfirstlane_b32", TI, TK),
              Ins("s_cmp_ge_u32", TI, CLEN), Ins("s_cbranch_scc1", self.L("done"))]
        for it in self.tile_params(TI, CM0, CN0, NROT):
            s += it.ins
        s += [Ins("v_add_u32", IDXA, CM0, VRW), Ins("v_add_u32", IDXW, CN0, VRW), Ins("s_lshl_b32", KTD, NROT, 1), Ins("s_lshl_b32", KOFF, NROT, 7),
              Ins("s_lshl_b32", DMABW, WAVE, 12), Ins("s_mov_b32", BSEL, 0), Ins("s_mov_b32", MORE, 1)]
        s += self.bias_dma(CN0, [Ins("s_mov_b32", T1, 0)]).ins
        abl, self.abl = self.abl, 0   # (the pipeline is always filled with real loads)
        for tag, cnt in (("A", 8), ("B", 8), ("A2", 8), ("B2", self.n3_seam)):   # the first four stages of the stream
            for p in range(cnt):
                s += self.dma_piece(p, tag).ins
            if cnt == 8:
                s += self.group_end().ins
        self.abl = abl
        s += [wait_vm(("A", "B")), Ins("s_barrier")]
        for it in self.bias_frags(0):
            s += it.ins
        for it in self.read_items(self.substeps("quad")[0][0], FSET[0], "f0"):
            s += it.ins
        # the first tile has no predecessor: F's lazy epilogue finds a descriptor with 0 records (set above)
        s += [Ins("s_mov_b32", NM0, CM0), Ins("s_mov_b32", NN0, CN0), Ins("s_mov_b32", TIN, TI), Ins("s_branch", self.L("F_body"))]
        seq, vm, lg = resolve_waits(s)
        loads = [t for t, ld in vm if ld]
        assert loads == ["A2"] * 8 + ["B2"] * self.n3_seam and [t for t in lg if t] == ["f0"] * 8, (loads, lg)
        return seq, loads, ["f0"] * 8

    def build(self):
        self.drain_l, self.drain_f = self.drain_plan()
        out = []
        pro, vm_p, lg_p = self.prologue()
        out += pro
        # ---- tile loop
        out.append(label(self.L("F_rot")))
        out += self.tile_rotate()
        out.append(label(self.L("F_body")))
        seq, vm, lg = self.block("F", None, vm_p, lg_p, False, drain=self.drain_f)
        out += seq
        for j in range(self.J):
            seq, vm, lg = self.block("mid", j, vm, lg, False)
            out += seq
        vm_loop_in, lg_loop_in = list(vm), list(lg)
        out += [label(self.L("loop")), Ins("s_cmp_eq_u32", KLEFT, 0), Ins("s_cbranch_scc1", self.L("pre"))]
        seq, vm2, lg2 = self.block("mid", "loop", vm, lg, False)
        assert vm2 == vm_loop_in and lg2 == lg_loop_in, (vm2, lg2, vm_loop_in, lg_loop_in)
        out += seq
        out += [Ins("s_branch", self.L("loop")), label(self.L("pre"))]
        seq, vm, lg = self.block("mid", "pre", vm, lg, True)
        out += seq
        seq, vm, lg = self.block("L", None, vm, lg, True, drain=self.drain_l)
        out += seq
        assert vm == vm_p and lg == lg_p, (vm, lg, vm_p, lg_p)
        out += [Ins("s_cmp_lg_u32", MORE, 0), Ins("s_cbranch_scc1", self.L("F_rot"))]
        # wave 0 waits for its ticket (drawn in F, posted at mid block 0's barrier X) by an explicit count: the loads it has issued since
        idx = [i for i, ins in enumerate(out) if ins.tag == "tk" and ins.uncounted]
        iw = [i for i, ins in enumerate(out) if ins.tag == "tkwait"]
        assert len(idx) == 1 and len(iw) == 1 and idx[0] < iw[0]
        from .isa import is_load
        out[iw[0]].mods["vmcnt"] = min(63, sum(1 for ins in out[idx[0] + 1:iw[0]] if is_load(ins) and not ins.uncounted))
        # ---- tail: the last tile's remaining drain and its whole epilogue, no MFMAs beside it
        tail = [Ins("s_waitcnt", kind="wait", vmcnt=0, lgkmcnt=0)] + self.tile_rotate() + [Ins("s_nop", 7), Ins("s_nop", 7)]
        for it in self.drain_f:
            tail += it.ins
        for u in range(8):
            for it in self.epi_stage(u, self.unit_temps(u)) + self.epi_store(u):
                tail += it.ins
        tail, _, _ = resolve_waits(tail)
        out += tail
        out.append(label(self.L("done")))
        # the last workgroup to finish re-zeroes the ticket slot (sched[0..8]) for the slot's next launch
        fin = [Ins("s_mov_b64", EXEC, 1), Ins("v_mov_b32", X[2], 32), Ins("v_mov_b32", X[3], 1),
               Ins("global_atomic_add", TK, X[2], X[3], SCHED, sc0=True), Ins("s_waitcnt", kind="wait", vmcnt=0),
               Ins("v_readfirstlane_b32", T0, TK), Ins("s_sub_u32", T1, a_GRID, 1), Ins("s_cmp_lg_u32", T0, T1), Ins("s_cbranch_scc1", self.L("end")),
               Ins("v_mov_b32", X[3], 0)]
        for i in range(9):
            fin += [Ins("v_mov_b32", X[2], 4 * i), Ins("global_store_dword", X[2], X[3], SCHED)]
        out += self.wave0_only(fin) + [label(self.L("end")), Ins("s_endpgm")]
        return out


