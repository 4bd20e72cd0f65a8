"""A minimal instruction representation for the generated gfx950 stream: operands, instructions, text rendering."""


class Reg:
    """A register range: file 'v' (arch VGPR), 'a' (AccVGPR), 's' (SGPR); `n` first index, `c` count."""
    __slots__ = ("f", "n", "c")

    def __init__(self, f, n, c=1):
        self.f, self.n, self.c = f, int(n), int(c)

    def __getitem__(self, i):
        if isinstance(i, slice):
            a = i.start or 0
            b = self.c if i.stop is None else i.stop
            assert 0 <= a < b <= self.c, (self, i)
            return Reg(self.f, self.n + a, b - a)
        assert 0 <= i < self.c, (self, i)
        return Reg(self.f, self.n + i, 1)

    def __repr__(self):
        return f"{self.f}{self.n}" if self.c == 1 else f"{self.f}[{self.n}:{self.n + self.c - 1}]"

    def __eq__(self, o):
        return isinstance(o, Reg) and (self.f, self.n, self.c) == (o.f, o.n, o.c)

    def __hash__(self):
        return hash((self.f, self.n, self.c))


def V(n, c=1):
    return Reg("v", n, c)


def A(n, c=1):
    return Reg("a", n, c)


def S(n, c=1):
    return Reg("s", n, c)


class Special:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return self.name


M0, VCC, EXEC, SCC, OFF = Special("m0"), Special("vcc"), Special("exec"), Special("scc"), Special("off")


def fmt_opnd(x):
    if isinstance(x, (Reg, Special)):
        return repr(x)
    if isinstance(x, float):
        import struct
        return "0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0]
    if isinstance(x, int):
        if -16 <= x <= 64:
            return str(x)
        return "0x%x" % (x & 0xffffffff)
    raise TypeError(x)


class Ins:
    """One instruction (or pseudo-instruction).  kind: mfma | valu | trans | salu | ds | dma | store | atomic | smem | wait | barrier |
    branch | label | nop | waitvm | waitlgkm (the last two are pseudo: resolved to counted s_waitcnt by `resolve_waits`)."""
    __slots__ = ("op", "ops", "mods", "kind", "tag", "note", "uncounted")

    def __init__(self, op, *ops, kind=None, tag=None, note=None, uncounted=False, **mods):
        self.op, self.ops, self.mods, self.tag, self.note, self.uncounted = op, list(ops), mods, tag, note, uncounted
        self.kind = kind or classify(op)

    def text(self):
        op, o, m = self.op, self.ops, self.mods
        k = self.kind
        if k == "label":
            return f"{op}:"
        if k == "wait":
            parts = []
            if "vmcnt" in m:
                parts.append(f"vmcnt({m['vmcnt']})")
            if "lgkmcnt" in m:
                parts.append(f"lgkmcnt({m['lgkmcnt']})")
            return "s_waitcnt " + " ".join(parts)
        if k in ("waitvm", "waitlgkm"):
            raise RuntimeError("unresolved pseudo wait " + repr(self.tag))
        if k == "branch":
            return f"{op} {o[0]}"
        if op == "s_nop":
            return f"s_nop {o[0]}"
        if op in ("s_barrier", "s_endpgm"):
            return op
        if k == "entry":
            return f"{op} " + ", ".join(fmt_opnd(x) if not hasattr(x, "s") else x.s for x in o)
        if k == "ds":
            # ds_read_bN vdst, vaddr offset:X   |  ds_write_bN vaddr, vdata offset:X
            s = f"{op} {fmt_opnd(o[0])}, {fmt_opnd(o[1])}"
            if m.get("offset"):
                s += f" offset:{m['offset']}"
            return s
        if k == "dma":
            # buffer_load_dwordx4 vaddr(pair), srsrc, soffset idxen offen lds
            return f"{op} {fmt_opnd(o[0])}, {fmt_opnd(o[1])}, {fmt_opnd(o[2])} {m['addr']} lds"
        if k == "store":
            s = f"{op} {fmt_opnd(o[0])}, {fmt_opnd(o[1])}, {fmt_opnd(o[2])}, {fmt_opnd(o[3])} {m['addr']}"
            if m.get("nt"):
                s += " nt"
            return s
        if k == "atomic":
            # global_atomic_add [vdst,] vaddr, vdata, saddr [sc0]
            s = f"{op} " + ", ".join(fmt_opnd(x) for x in o)
            if m.get("sc0"):
                s += " sc0"
            return s
        if k == "smem":
            return f"{op} {fmt_opnd(o[0])}, {fmt_opnd(o[1])}, 0x{m.get('offset', 0):x}"
        if op.startswith("global_store") or op.startswith("global_load"):
            s = f"{op} " + ", ".join(fmt_opnd(x) for x in o)
            if m.get("offset"):
                s += f" offset:{m['offset']}"
            return s
        return f"{op} " + ", ".join(fmt_opnd(x) for x in o)


TRANS = ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op in TRANS:
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith("buffer_load"):
        return "dma"
    if op.startswith("buffer_store"):
        return "store"
    if op.startswith("global_atomic"):
        return "atomic"
    if op.startswith("global_"):
        return "gmem"
    if op.startswith("s_load"):
        return "smem"
    if op == "s_waitcnt":
        return "wait"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op == "s_nop":
        return "nop"
    if op.startswith("s_"):
        return "salu"
    raise ValueError(op)


def label(name):
    return Ins(name, kind="label")


def wait_vm(tag, note=None):
    """Pseudo: wait until every VMEM operation up to and including the LAST one tagged `tag` has completed."""
    return Ins("waitvm", kind="waitvm", tag=tag, note=note)


def wait_lgkm(tag, note=None):
    return Ins("waitlgkm", kind="waitlgkm", tag=tag, note=note)


VM_KINDS = ("dma", "store", "atomic", "gmem")
LGKM_KINDS = ("ds", "smem")


def is_load(ins):
    """VMEM operations that return data (in issue order among themselves): LDS-DMA pieces, returning atomics.  Stores and
    non-returning atomics only acknowledge -- and an acknowledgement may overtake an older load."""
    return ins.kind == "dma" or (ins.kind == "atomic" and ins.mods.get("sc0"))


def resolve_waits(seq, vm_in=(), lgkm_in=(), merge=True):
    """Replace the pseudo waits of a straight-line sequence by counted s_waitcnt.  `vm_in` / `lgkm_in`: the operations still
    outstanding at entry (oldest first): tags (LGKM) and (tag, is_load) pairs (VM).  Returns (new sequence, vm_out, lgkm_out).
    LGKM (LDS only here) returns in issue order: n = operations issued after the last one carrying the tag.
    VM: vmcnt is ONE counter for loads and stores, loads return in order, store acknowledgements come back whenever they like.  A
    load is therefore known to have landed only when the counter is at or below the number of LOADS issued after it: younger
    stores are not counted (they may be acknowledged first), which makes the wait cover them too."""
    vm, lg = [tuple(x) if isinstance(x, (tuple, list)) else (x, True) for x in vm_in], list(lgkm_in)
    out = []
    for ins in seq:
        k = ins.kind
        if ins.uncounted:   # executed by one wave only (wave-uniform branch): not part of the common count
            out.append(ins)
            continue
        if k in ("waitvm", "waitlgkm"):
            tags = ins.tag if isinstance(ins.tag, (tuple, list)) else (ins.tag,)
            if k == "waitvm":
                idx = max((i for i, t in enumerate(vm) if t[0] in tags), default=-1)
                if idx < 0:
                    continue
                n = min(sum(1 for t in vm[idx + 1:] if t[1]), 63)
                del vm[: idx + 1]
                key = "vmcnt"
            else:
                idx = max((i for i, t in enumerate(lg) if t in tags), default=-1)
                if idx < 0:
                    continue  # nothing outstanding carries the tag
                n = min(len(lg) - 1 - idx, 15)
                del lg[: len(lg) - n]
                key = "lgkmcnt"
            if merge and out and out[-1].kind == "wait" and key not in out[-1].mods and not out[-1].uncounted:
                out[-1].mods[key] = n
            else:
                out.append(Ins("s_waitcnt", kind="wait", note=ins.note, **{key: n}))
            continue
        if k == "wait":
            if "vmcnt" in ins.mods:
                n = ins.mods["vmcnt"]
                assert n == 0, "explicit vmcnt waits other than 0 are not modelled"
                vm.clear()
            if "lgkmcnt" in ins.mods:
                n = ins.mods["lgkmcnt"]
                del lg[: max(0, len(lg) - n)]
        elif k in VM_KINDS:
            vm.append((ins.tag, is_load(ins)))
        elif k in LGKM_KINDS:
            lg.append(ins.tag)
        out.append(ins)
    return out, vm, lg


def render(seq, indent="  "):
    lines = []
    for ins in seq:
        t = ins.text()
        if ins.kind == "label":
            lines.append(t)
        else:
            lines.append(indent + t + (f"  ; {ins.note}" if ins.note else ""))
    return lines
