"""Functional emulator of the instruction subset the gemm4w stream uses (one workgroup = 4 wavefronts sharing an LDS image).

Not a timing model.  What it checks beyond the arithmetic:
  * asynchronous results: a ds_read / returning atomic destination is "pending" until an s_waitcnt retires it -- any use before that is an error;
  * LDS-DMA: bytes with a DMA in flight may not be read; a DMA landed by wave A is visible to wave B only after a barrier B passed later;
    a DMA may not be issued into bytes another wave read in the current barrier epoch (write-after-read);
    `dma_late` chooses whether data lands at issue or at the retiring wait -- results must not depend on it;
  * software wait states hipcc would insert but an asm author must: MFMA result -> v_accvgpr_read (12 issue slots), SALU write of M0 -> LDS-DMA
    (1), VALU-written SGPR -> VMEM (5), transcendental result -> next VALU (1).
Waves run one after another between barriers (`order` permutes them): results must not depend on the order either."""
import struct

import numpy as np

from .isa import Reg, Special, Ins

U32 = np.uint32
LDS_BYTES = 160 * 1024


def f2u(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def bf16_round(x_f32):
    """fp32 array -> bf16 bits (uint32 array, low 16 bits), round to nearest even."""
    u = x_f32.astype(np.float32).view(U32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (r & 0xFFFF).astype(U32)


class EmuError(Exception):
    pass


class Memory:
    def __init__(self):
        self.bufs = []  # (base, np.uint8 array)
        self.next = 0x7F0000000000

    def alloc(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.next += (len(a) + 0xFFFF) & ~0xFFFF
        self.bufs.append((base, a))
        return base

    def find(self, addr, n):
        for base, a in self.bufs:
            if base <= addr and addr + n <= base + len(a):
                return a, addr - base
        raise EmuError("global access outside any buffer: 0x%x (+%d)" % (addr, n))

    def read(self, addr, n):
        a, o = self.find(addr, n)
        return a[o:o + n]

    def write(self, addr, data):
        a, o = self.find(addr, len(data))
        a[o:o + len(data)] = data

    def get(self, base, dtype):
        for b, a in self.bufs:
            if b == base:
                return a.view(dtype)
        raise KeyError(base)


class Wave:
    def __init__(self, wg, wid):
        self.wg, self.wid = wg, wid
        self.v = np.zeros((256, 64), U32)
        self.a = np.zeros((256, 64), U32)
        self.s = [0] * 128
        self.vcc = 0
        self.exec = (1 << 64) - 1
        self.scc = 0
        self.m0 = 0
        self.pc = 0
        self.tick = 0
        self.vmq = []
        self.lgq = []
        self.pend_v = {}   # vgpr index -> what
        self.sgpr_valu_write = {}
        self.m0_write = -10
        self.acc_ready = np.zeros(256, np.int64)
        self.trans_write = {}
        self.epoch = 0
        self.done = False
        self.nmfma = 0


class Workgroup:
    def __init__(self, prog, labels, mem, dma_late=False):
        self.prog, self.labels, self.mem = prog, labels, mem
        self.lds = np.zeros(LDS_BYTES, np.uint8)
        self.inflight = np.zeros(LDS_BYTES // 16, np.int32)
        self.landed = {}     # 16-byte granule -> (wave, epoch)   (kept per 1 KiB piece start)
        self.last_read = np.full((4, LDS_BYTES // 1024), -1, np.int64)
        self.waves = [Wave(self, i) for i in range(4)]
        self.dma_late = dma_late
        self.trace = None


def lanes_of(mask):
    return np.array([(mask >> i) & 1 for i in range(64)], bool)


class Emu:
    def __init__(self, seq, mem, dma_late=False, order=(0, 1, 2, 3), check=True, stores_ooo=False):
        self.stores_ooo = stores_ooo
        self.prog = [i for i in seq if i.kind != "label"]
        self.labels = {}
        n = 0
        for i in seq:
            if i.kind == "label":
                self.labels[i.op] = n
            else:
                n += 1
        self.mem, self.dma_late, self.order, self.check = mem, dma_late, order, check
        self.max_steps = 50_000_000

    # ---------------------------------------------------------------- operand access
    def rs(self, w, x):
        """scalar source value (uint32)"""
        if isinstance(x, Reg):
            assert x.f == "s" and x.c == 1, x
            if self.check and getattr(w, "pend_s", {}).get(x.n, 0) > 0:
                raise EmuError(f"pc {w.pc}: read of s{x.n} while an s_load into it is pending")
            return w.s[x.n] & 0xFFFFFFFF
        if isinstance(x, Special):
            if x.name == "m0":
                return w.m0
            if x.name == "scc":
                return w.scc
            raise EmuError("scalar read of " + x.name)
        if isinstance(x, float):
            return f2u(x)
        return int(x) & 0xFFFFFFFF

    def rs64(self, w, x):
        if isinstance(x, Reg):
            assert x.f == "s" and x.c == 2
            return (w.s[x.n] & 0xFFFFFFFF) | ((w.s[x.n + 1] & 0xFFFFFFFF) << 32)
        if isinstance(x, Special):
            if x.name == "exec":
                return w.exec
            if x.name == "vcc":
                return w.vcc
        return int(x) & ((1 << 64) - 1)

    def ws(self, w, x, val):
        val &= 0xFFFFFFFF
        if self.check and isinstance(x, Reg) and getattr(w, "pend_s", {}).get(x.n, 0) > 0:
            raise EmuError(f"pc {w.pc}: write of s{x.n} while an s_load into it is pending (the load's data would overwrite it)")
        if isinstance(x, Special):
            assert x.name == "m0", x
            w.m0 = val
            w.m0_write = w.tick
            return
        assert x.f == "s" and x.c == 1
        w.s[x.n] = val

    def rv(self, w, x, need=True):
        """vector source: returns uint32[64]"""
        if isinstance(x, Reg):
            if x.f == "v":
                assert x.c == 1
                if self.check and x.n in w.pend_v:
                    raise EmuError(f"pc {w.pc}: use of v{x.n} while its load is pending ({w.pend_v[x.n]})")
                if self.check and w.trans_write.get(x.n, -10) >= w.tick - 1:
                    raise EmuError(f"pc {w.pc}: v{x.n} written by a transcendental op is used by the next VALU instruction")
                return w.v[x.n]
            if x.f == "s":
                return np.full(64, w.s[x.n] & 0xFFFFFFFF, U32)
            raise EmuError("vector read of " + repr(x))
        if isinstance(x, float):
            return np.full(64, f2u(x), U32)
        return np.full(64, int(x) & 0xFFFFFFFF, U32)

    def wv(self, w, x, val, trans=False):
        assert x.f == "v" and x.c == 1
        m = lanes_of(w.exec)
        if x.n in w.pend_v and self.check:
            raise EmuError(f"pc {w.pc}: write of v{x.n} while a load into it is pending")
        w.v[x.n][m] = np.asarray(val, U32)[m]
        if trans:
            w.trans_write[x.n] = w.tick
        else:
            w.trans_write.pop(x.n, None)

    # ---------------------------------------------------------------- helpers
    def desc(self, w, r):
        assert r.f == "s" and r.c == 4
        d = [w.s[r.n + i] & 0xFFFFFFFF for i in range(4)]
        base = d[0] | ((d[1] & 0xFFFF) << 32)
        stride = (d[1] >> 16) & 0x3FFF
        return base, stride, d[2]

    def check_sgpr_vmem(self, w, regs):
        if not self.check:
            return
        for r in regs:
            if isinstance(r, Reg) and r.f == "s":
                for i in range(r.n, r.n + r.c):
                    if w.sgpr_valu_write.get(i, -100) > w.tick - 6:
                        raise EmuError(f"pc {w.pc}: s{i} written by a VALU instruction {w.tick - w.sgpr_valu_write[i]} slots before a VMEM use (needs 5 wait states)")

    def retire(self, w, q, n):
        """s_waitcnt: wait until at most n operations of the counter are outstanding.  Loads return in issue order; with `stores_ooo`
        store acknowledgements overtake older loads (vmcnt is ONE counter for both: what the count then guarantees about a load is
        only that at most n operations are left, whichever they are)."""
        if q is w.vmq and self.stores_ooo:
            i = 0
            while len(q) > n and i < len(q):
                if getattr(q[i], "is_store", False):
                    q.pop(i)()
                else:
                    i += 1
        while len(q) > n:
            fn = q.pop(0)
            fn()

    # ---------------------------------------------------------------- execution
    def run_block(self, init_wave, lds_init=None):
        """init_wave(w: Wave) sets the entry registers.  Runs one workgroup to completion."""
        wg = Workgroup(self.prog, self.labels, self.mem, self.dma_late)
        self.wg = wg
        for w in wg.waves:
            init_wave(w)
        steps = 0
        while not all(w.done for w in wg.waves):
            arrived = 0
            for wi in self.order:
                w = wg.waves[wi]
                if w.done:
                    continue
                while True:
                    ins = self.prog[w.pc]
                    steps += 1
                    if steps > self.max_steps:
                        raise EmuError("step limit")
                    r = self.step(w, ins)
                    if r == "barrier":
                        arrived += 1
                        break
                    if r == "end":
                        w.done = True
                        break
            live = [w for w in wg.waves if not w.done]
            if live and arrived != len(live) and arrived != 0:
                raise EmuError("barrier reached by %d of %d live waves" % (arrived, len(live)))
            for w in wg.waves:
                w.epoch += 1
        return wg

    def step(self, w, ins):
        op, o, m, k = ins.op, ins.ops, ins.mods, ins.kind
        pc0 = w.pc
        w.pc += 1
        w.tick += 1
        E = self
        if k == "nop":
            w.tick += int(o[0])
            return
        if k == "entry":
            return
        if k == "barrier":
            return "barrier"
        if op == "s_endpgm":
            if w.vmq or w.lgq:
                E.retire(w, w.vmq, 0)
                E.retire(w, w.lgq, 0)
            return "end"
        if k == "wait":
            if "vmcnt" in m:
                E.retire(w, w.vmq, m["vmcnt"])
            if "lgkmcnt" in m:
                E.retire(w, w.lgq, m["lgkmcnt"])
            return
        if k == "branch":
            take = {"s_branch": True, "s_cbranch_scc0": w.scc == 0, "s_cbranch_scc1": w.scc == 1,
                    "s_cbranch_execz": w.exec == 0, "s_cbranch_vccz": w.vcc == 0, "s_cbranch_vccnz": w.vcc != 0}[op]
            if take:
                w.pc = self.labels[o[0]]
            return
        if k == "salu":
            return self.salu(w, op, o)
        if k == "smem":
            base = E.rs64(w, o[1]) + m.get("offset", 0)
            n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[op]
            data = E.mem.read(base, 4 * n).view(U32).copy()
            dst = o[0]

            pend = getattr(w, "pend_s", None)
            if pend is None:
                pend = w.pend_s = {}
            for i in range(n):
                pend[dst.n + i] = pend.get(dst.n + i, 0) + 1

            def land(data=data, dst=dst):
                for i in range(n):
                    w.s[dst.n + i] = int(data[i])
                    w.pend_s[dst.n + i] -= 1
            w.lgq.append(land)
            return
        if k == "mfma":
            return self.mfma(w, ins)
        if k in ("valu", "trans"):
            return self.valu(w, ins)
        if k == "ds":
            return self.ds(w, ins)
        if k == "dma":
            return self.dma(w, ins)
        if k == "store":
            return self.store(w, ins)
        if k == "atomic":
            return self.atomic(w, ins)
        if k == "gmem":
            return self.gmem(w, ins)
        raise EmuError("unhandled " + op)

    def salu(self, w, op, o):
        E = self
        M32 = 0xFFFFFFFF

        def sgn(x):
            return x - (1 << 32) if x & 0x80000000 else x
        if op == "s_mov_b32":
            return E.ws(w, o[0], E.rs(w, o[1]))
        if op == "s_mov_b64":
            v = E.rs64(w, o[1])
            if isinstance(o[0], Special):
                if o[0].name == "exec":
                    w.exec = v
                elif o[0].name == "vcc":
                    w.vcc = v
                return
            w.s[o[0].n] = v & M32
            w.s[o[0].n + 1] = v >> 32
            return
        a = E.rs(w, o[1]) if len(o) > 1 else None
        b = E.rs(w, o[2]) if len(o) > 2 else None
        if op in ("s_add_u32", "s_add_i32"):
            r = a + b
            w.scc = 1 if (r >> 32) else 0
            return E.ws(w, o[0], r)
        if op == "s_addc_u32":
            r = a + b + w.scc
            w.scc = 1 if (r >> 32) else 0
            return E.ws(w, o[0], r)
        if op in ("s_sub_u32", "s_sub_i32"):
            r = a - b
            w.scc = 1 if r < 0 else 0
            return E.ws(w, o[0], r)
        if op == "s_mul_i32":
            return E.ws(w, o[0], (a * b) & M32)
        if op == "s_mul_hi_u32":
            return E.ws(w, o[0], (a * b) >> 32)
        if op == "s_lshl_b32":
            r = (a << (b & 31)) & M32
            w.scc = int(r != 0)
            return E.ws(w, o[0], r)
        if op == "s_lshr_b32":
            r = a >> (b & 31)
            w.scc = int(r != 0)
            return E.ws(w, o[0], r)
        if op in ("s_and_b32", "s_or_b32", "s_xor_b32"):
            r = {"s_and_b32": a & b, "s_or_b32": a | b, "s_xor_b32": a ^ b}[op]
            w.scc = int(r != 0)
            return E.ws(w, o[0], r)
        if op == "s_min_u32":
            w.scc = int(a < b)
            return E.ws(w, o[0], min(a, b))
        if op == "s_max_u32":
            w.scc = int(a > b)
            return E.ws(w, o[0], max(a, b))
        if op == "s_cselect_b32":
            return E.ws(w, o[0], a if w.scc else b)
        if op.startswith("s_cmp_"):
            x, y = E.rs(w, o[0]), E.rs(w, o[1])
            c = op[6:]
            if c.endswith("_i32"):
                x, y = sgn(x), sgn(y)
            c = c[:-4]
            w.scc = int({"eq": x == y, "lg": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}[c])
            return
        if op == "s_bfe_u32":
            off, wd = b & 31, (b >> 16) & 0x7F
            r = (a >> off) & ((1 << wd) - 1)
            w.scc = int(r != 0)
            return E.ws(w, o[0], r)
        if op == "s_setprio":
            return
        raise EmuError("salu " + op)

    def valu(self, w, ins):
        E = self
        op, o = ins.op, ins.ops
        f32 = np.float32

        def F(x):
            return E.rv(w, x).view(f32)
        if op == "v_mov_b32":
            return E.wv(w, o[0], E.rv(w, o[1]))
        if op == "v_accvgpr_read_b32":
            src = o[1]
            assert src.f == "a"
            if E.check and w.acc_ready[src.n] > w.tick:
                raise EmuError(f"pc {w.pc}: v_accvgpr_read of a{src.n} {w.acc_ready[src.n] - w.tick} slots before its MFMA result is readable")
            return E.wv(w, o[0], w.a[src.n])
        if op == "v_accvgpr_write_b32":
            w.a[o[0].n][:] = E.rv(w, o[1])
            return
        if op == "v_readfirstlane_b32":
            v = E.rv(w, o[1])
            lane = (w.exec & -w.exec).bit_length() - 1 if w.exec else 0
            w.s[o[0].n] = int(v[lane])
            w.sgpr_valu_write[o[0].n] = w.tick
            return
        if op == "v_mbcnt_lo_u32_b32":
            mask = E.rs(w, o[1]) if not isinstance(o[1], int) else (o[1] & 0xFFFFFFFF)
            base = E.rv(w, o[2])
            r = np.array([bin(mask & ((1 << min(i, 32)) - 1)).count("1") for i in range(64)], U32) + base
            return E.wv(w, o[0], r)
        if op == "v_mbcnt_hi_u32_b32":
            mask = E.rs(w, o[1]) if not isinstance(o[1], int) else (o[1] & 0xFFFFFFFF)
            base = E.rv(w, o[2])
            r = np.array([bin(mask & ((1 << max(i - 32, 0)) - 1)).count("1") for i in range(64)], U32) + base
            return E.wv(w, o[0], r)
        a = E.rv(w, o[1])
        b = E.rv(w, o[2]) if len(o) > 2 else None
        c = E.rv(w, o[3]) if len(o) > 3 else None
        if op == "v_add_u32":
            return E.wv(w, o[0], a + b)
        if op == "v_sub_u32":
            return E.wv(w, o[0], a - b)
        if op == "v_lshlrev_b32":
            return E.wv(w, o[0], (b.astype(np.uint64) << (a & 31).astype(np.uint64)).astype(U32))
        if op == "v_lshrrev_b32":
            return E.wv(w, o[0], b >> (a & 31))
        if op == "v_and_b32":
            return E.wv(w, o[0], a & b)
        if op == "v_or_b32":
            return E.wv(w, o[0], a | b)
        if op == "v_xor_b32":
            return E.wv(w, o[0], a ^ b)
        if op == "v_lshl_or_b32":
            return E.wv(w, o[0], ((a.astype(np.uint64) << (b & 31).astype(np.uint64)).astype(U32)) | c)
        if op == "v_and_or_b32":
            return E.wv(w, o[0], (a & b) | c)
        if op == "v_lshl_add_u32":
            return E.wv(w, o[0], ((a.astype(np.uint64) << (b & 31).astype(np.uint64)).astype(U32)) + c)
        if op == "v_mad_u32_u24":
            return E.wv(w, o[0], ((a & 0xFFFFFF).astype(np.uint64) * (b & 0xFFFFFF).astype(np.uint64)).astype(U32) + c)
        if op == "v_mul_u32_u24":
            return E.wv(w, o[0], ((a & 0xFFFFFF).astype(np.uint64) * (b & 0xFFFFFF).astype(np.uint64)).astype(U32))
        if op == "v_pk_max_i16":
            lo = np.maximum((a & 0xFFFF).astype(np.uint16).view(np.int16), (b & 0xFFFF).astype(np.uint16).view(np.int16)).view(np.uint16).astype(U32)
            hi = np.maximum((a >> 16).astype(np.uint16).view(np.int16), (b >> 16).astype(np.uint16).view(np.int16)).view(np.uint16).astype(U32)
            return E.wv(w, o[0], lo | (hi << 16))
        if op == "v_min_u32":
            return E.wv(w, o[0], np.minimum(a, b))
        if op == "v_cndmask_b32":
            sel = lanes_of(w.vcc)
            return E.wv(w, o[0], np.where(sel, b, a))
        if op.startswith("v_cmp_"):
            cnd = op[6:]
            x, y = E.rv(w, o[0] if o[0] is not None else 0), a  # (never used with explicit sdst here)
            raise EmuError("v_cmp with explicit operands not supported: use vcmp()")
        if op.startswith("v_cmpx"):
            raise EmuError(op)
        # float ops
        fa = a.view(f32)
        fb = b.view(f32) if b is not None else None
        fc = c.view(f32) if c is not None else None
        with np.errstate(all="ignore"):
            if op == "v_add_f32":
                return E.wv(w, o[0], (fa + fb).view(U32))
            if op == "v_sub_f32":
                return E.wv(w, o[0], (fa - fb).view(U32))
            if op == "v_mul_f32":
                return E.wv(w, o[0], (fa * fb).view(U32))
            if op == "v_fmaak_f32":
                r = (fa.astype(np.float64) * fb.astype(np.float64) + fc.astype(np.float64)).astype(f32)
                return E.wv(w, o[0], r.view(U32))
            if op == "v_fma_f32":
                r = (fa.astype(np.float64) * fb.astype(np.float64) + fc.astype(np.float64)).astype(f32)
                return E.wv(w, o[0], r.view(U32))
            if op == "v_max_f32":
                return E.wv(w, o[0], np.maximum(fa, fb).view(U32))
            if op == "v_min_f32":
                return E.wv(w, o[0], np.minimum(fa, fb).view(U32))
            if op == "v_exp_f32":
                return E.wv(w, o[0], np.exp2(fa).astype(f32).view(U32), trans=True)
            if op == "v_rcp_f32":
                return E.wv(w, o[0], (f32(1.0) / fa).astype(f32).view(U32), trans=True)
            if op == "v_cvt_pk_bf16_f32":
                lo, hi = bf16_round(fa), bf16_round(fb)
                return E.wv(w, o[0], lo | (hi << 16))
        raise EmuError("valu " + op)

    def vcmp(self, w, ins):
        raise NotImplementedError

    def mfma(self, w, ins):
        E = self
        o = ins.ops
        assert ins.op == "v_mfma_f32_32x32x16_bf16"
        d, sa, sb, sc = o
        assert d.f == "a" and d.c == 16 and sa.c == 4 and sb.c == 4

        def frag(r):
            regs = np.stack([E.rv(w, r[i]) for i in range(4)])  # [4][64]
            if E.check:
                for i in range(4):
                    pass
            lo = (regs << 16).view(np.float32)
            hi = (regs & 0xFFFF0000).view(np.float32)
            # element e of lane: e = 2*reg + half
            x = np.empty((64, 8), np.float32)
            x[:, 0::2] = lo.T
            x[:, 1::2] = hi.T
            mat = np.empty((32, 16), np.float32)
            mat[:, 0:8] = x[0:32]
            mat[:, 8:16] = x[32:64]
            return mat
        Am, Bm = frag(sa), frag(sb)  # Am[i][k], Bm[j][k]
        D = Am.astype(np.float64) @ Bm.astype(np.float64).T  # [i][j]
        if isinstance(sc, Reg):
            assert sc.f == "a" and sc.c == 16
            C = np.empty((32, 32), np.float32)
            for r in range(16):
                for h in range(2):
                    i = (r & 3) + 8 * (r >> 2) + 4 * h
                    C[i, :] = w.a[sc.n + r][32 * h:32 * h + 32].view(np.float32)
            D = D + C
        else:
            assert sc == 0
        D = D.astype(np.float32)
        for r in range(16):
            for h in range(2):
                i = (r & 3) + 8 * (r >> 2) + 4 * h
                w.a[d.n + r][32 * h:32 * h + 32] = D[i, :].view(U32)
        issue = max(w.tick, getattr(w, "mfma_free", 0))
        w.mfma_free = issue + 8
        w.tick = issue
        w.acc_ready[d.n:d.n + 16] = issue + 12
        w.nmfma += 1

    # LDS piece bookkeeping -------------------------------------------------------------------------------
    def lds_read_check(self, w, addrs, n):
        if not self.check:
            return
        wg = self.wg
        g0 = addrs // 16
        for g in np.unique(np.concatenate([g0 + i for i in range((n + 15) // 16)])):
            if wg.inflight[g] > 0:
                raise EmuError(f"pc {w.pc} wave {w.wid}: ds_read of LDS byte {g * 16} with an LDS-DMA in flight")
            ld = wg.landed.get(int(g) // 64)
            if ld is not None and ld[0] != w.wid and not (w.epoch > ld[1]):
                raise EmuError(f"pc {w.pc} wave {w.wid}: ds_read of LDS byte {g * 16} landed by wave {ld[0]} with no barrier in between")
        for p in np.unique(addrs // 1024):
            wg.last_read[w.wid, p] = w.epoch

    def ds(self, w, ins):
        E = self
        op, o, m = ins.op, ins.ops, ins.mods
        off = m.get("offset", 0)
        wg = self.wg
        act = lanes_of(w.exec)
        if op.startswith("ds_read"):
            n = {"ds_read_b32": 4, "ds_read_b64": 8, "ds_read_b128": 16}[op]
            addr = (E.rv(w, o[1]).astype(np.int64) + off)
            if (addr[act] % n).any() or (addr[act] + n > LDS_BYTES).any():
                raise EmuError(f"pc {w.pc}: misaligned / out-of-range {op} address")
            E.lds_read_check(w, addr[act], n)
            data = np.zeros((n // 4, 64), U32)
            for l in np.nonzero(act)[0]:
                data[:, l] = wg.lds[addr[l]:addr[l] + n].view(U32)
            dst = o[0]
            assert dst.f == "v" and dst.c == n // 4
            for i in range(dst.c):
                if dst.n + i in w.pend_v and E.check:
                    raise EmuError(f"pc {w.pc}: ds_read into v{dst.n + i} while an earlier load into it is pending")
                w.pend_v[dst.n + i] = ins.tag or op

            def land(data=data, dst=dst, act=act):
                for i in range(dst.c):
                    w.v[dst.n + i][act] = data[i][act]
                    w.pend_v.pop(dst.n + i, None)
            w.lgq.append(land)
            return
        if op.startswith("ds_write"):
            n = {"ds_write_b32": 4, "ds_write_b64": 8, "ds_write_b128": 16}[op]
            addr = (E.rv(w, o[0]).astype(np.int64) + off)
            src = o[1]
            assert src.c == n // 4
            vals = np.stack([E.rv(w, src[i]) for i in range(src.c)])
            if (addr[act] % n).any() or (addr[act] + n > LDS_BYTES).any():
                raise EmuError(f"pc {w.pc}: misaligned / out-of-range {op} address")
            for l in np.nonzero(act)[0]:
                wg.lds[addr[l]:addr[l] + n] = vals[:, l].copy().view(np.uint8)
                if E.check and wg.inflight[addr[l] // 16] > 0:
                    raise EmuError(f"pc {w.pc}: ds_write into bytes with a DMA in flight")
            w.lgq.append(lambda: None)
            return
        raise EmuError(op)

    def buf_addr(self, w, ins, vaddr, rsrc, soff, n):
        """returns (addresses int64[64], ok mask)"""
        E = self
        base, stride, nrec = E.desc(w, rsrc)
        so = E.rs(w, soff)
        mode = ins.mods["addr"]
        E.check_sgpr_vmem(w, [rsrc, soff])
        if mode == "offen":
            vo = E.rv(w, vaddr).astype(np.int64)
            ok = vo + n <= nrec if stride == 0 else vo + n <= nrec * stride
            return base + so + vo, ok
        if mode == "idxen offen":
            assert vaddr.c == 2
            idx = E.rv(w, vaddr[0]).astype(np.int64)
            vo = E.rv(w, vaddr[1]).astype(np.int64)
            assert stride > 0
            ok = idx < nrec
            return base + so + idx * stride + vo, ok
        raise EmuError("addr mode " + mode)

    def dma(self, w, ins):
        E = self
        wg = self.wg
        o = ins.ops
        if E.check and w.m0_write >= w.tick - 1:
            raise EmuError(f"pc {w.pc}: LDS-DMA directly after the M0 write (needs 1 wait state)")
        addr, ok = E.buf_addr(w, ins, o[0], o[1], o[2], 16)
        dst = w.m0
        if dst % 16 or dst + 1024 > LDS_BYTES:
            raise EmuError(f"pc {w.pc}: LDS-DMA destination {dst}")
        data = np.zeros(1024, np.uint8)
        for l in range(64):
            if ok[l]:
                data[16 * l:16 * l + 16] = E.mem.read(int(addr[l]), 16)
        if E.check:
            p = dst // 1024
            pieces = {p, (dst + 1023) // 1024}
            for q in pieces:
                for ow in range(4):
                    if ow != w.wid and wg.last_read[ow, q] >= w.epoch:
                        raise EmuError(f"pc {w.pc} wave {w.wid}: LDS-DMA into piece {q} which wave {ow} read in the same barrier epoch")
            for i in range(dst // 16, dst // 16 + 64):
                if any(True for _ in ()) or False:
                    pass
            if w.pend_v and False:
                pass
        wg.inflight[dst // 16:dst // 16 + 64] += 1
        if not E.dma_late:
            wg.lds[dst:dst + 1024] = data

        def land(data=data, dst=dst):
            if E.dma_late:
                wg.lds[dst:dst + 1024] = data
            wg.inflight[dst // 16:dst // 16 + 64] -= 1
            for q in {dst // 1024, (dst + 1023) // 1024}:
                wg.landed[q] = (w.wid, w.epoch)
        w.vmq.append(land)

    def store(self, w, ins):
        E = self
        o = ins.ops
        data, vaddr, rsrc, soff = o
        n = 4 * data.c
        addr, ok = E.buf_addr(w, ins, vaddr, rsrc, soff, n)
        vals = np.stack([E.rv(w, data[i]) for i in range(data.c)])
        act = lanes_of(w.exec)
        for l in range(64):
            if ok[l] and act[l]:
                E.mem.write(int(addr[l]), vals[:, l].copy().view(np.uint8))

        def ack():
            return None
        ack.is_store = True
        w.vmq.append(ack)

    def atomic(self, w, ins):
        E = self
        o = ins.ops
        assert ins.op == "global_atomic_add"
        rtn = ins.mods.get("sc0")
        if rtn:
            dst, vaddr, vdata, saddr = o
        else:
            vaddr, vdata, saddr = o
            dst = None
        base = E.rs64(w, saddr)
        act = lanes_of(w.exec)
        off = E.rv(w, vaddr)
        dat = E.rv(w, vdata)
        old = np.zeros(64, U32)
        for l in np.nonzero(act)[0]:
            a = base + int(off[l])
            cur = E.mem.read(a, 4).view(U32)[0]
            old[l] = cur
            E.mem.write(a, np.array([(int(cur) + int(dat[l])) & 0xFFFFFFFF], U32).view(np.uint8))
        if dst is not None:
            w.pend_v[dst.n] = "atomic"

            def land(old=old, dst=dst, act=act):
                w.v[dst.n][act] = old[act]
                w.pend_v.pop(dst.n, None)
            w.vmq.append(land)
        else:
            w.vmq.append(lambda: None)

    def gmem(self, w, ins):
        E = self
        o = ins.ops
        if ins.op == "global_store_dword":
            vaddr, vdata, saddr = o
            base = E.rs64(w, saddr) + ins.mods.get("offset", 0)
            off = E.rv(w, vaddr)
            dat = E.rv(w, vdata)
            for l in np.nonzero(lanes_of(w.exec))[0]:
                E.mem.write(base + int(off[l]), np.array([dat[l]], U32).view(np.uint8))
            w.vmq.append(lambda: None)
            return
        raise EmuError(ins.op)
