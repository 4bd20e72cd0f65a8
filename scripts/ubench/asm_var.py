"""Is the packed-fp32 fault a missing WAIT STATE?    The -O3 ISA of query_lrf_group (97 packed fp32
instructions; wrong values beside an MFMA-only neighbour) is re-assembled with s_nop padding around the packed instructions:
    orig            the compiler's ISA, untouched (control: must still fail)
    after1/after7   s_nop 1 / s_nop 7 AFTER every v_pk_* (their results reach consumers 2 / 8 issue slots later)
    before7         s_nop 7 BEFORE every v_pk_* (their sources settle first)
    both7           both
    all_valu7       s_nop 7 after EVERY vector ALU instruction
If padding cures it, the hazard recogniser of this ROCm is missing a wait-state rule for packed fp32 on gfx950; if not, no amount
of software wait states helps and the fault is in how the hardware co-executes them with another wave's MFMAs.
    python scripts/ubench/asm_var.py build     here
    python scripts/ubench/asm_var.py [runs]    on the GPU box"""
import ctypes, os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
KERNEL = "_ZN7unopose22query_lrf_group_kernelEPKfifiiPf"
LLVM = "/opt/rocm/lib/llvm/bin"
VARIANTS = ["orig", "after1", "after7", "before7", "both7", "all_valu7", "split_all", "split_mul", "split_add", "split_plain", "split_modified",
            "keep_mul_opselhi", "keep_mul_opsel", "keep_mul_neg", "keep_mul_plain", "keep_add"] + [f"keep_one_{k}" for k in range(9)]
# keep_one_k: ONLY the k-th v_pk_mul_f32 with op_sel (in program order) stays packed
# keep_X: every packed instruction is split EXCEPT class X, which stays packed -- failing means class X alone is enough:
#   mul_opselhi: v_pk_mul_f32 with op_sel_hi only;  mul_opsel: with op_sel (any);  mul_neg: with neg_lo / neg_hi and no op_sel*;
#   mul_plain: v_pk_mul_f32 without modifiers;  add: every v_pk_add_f32
# split_*: packed instructions REPLACED by their two scalar halves (same registers, same arithmetic: v_pk_mul_f32 v[a:b], X, Y ->
# v_mul_f32 v_a, X.lo, Y.lo ; v_mul_f32 v_b, X.hi, Y.hi with op_sel / op_sel_hi / neg_lo / neg_hi honoured): all of them, only the
# multiplies, only the adds, only the plain VGPR-pair forms, only the forms with operand-select / negate modifiers or SGPR sources.


def _elem(op, k):
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", op)
    if m:
        return f"{m.group(1)}{int(m.group(2)) + k}"
    return op  # inline constant / literal: the same value in both halves


def split_packed(ln):
    """One v_pk_{mul,add}_f32 line -> two scalar VOP3 lines, or None when a half would read a register the other half has
    already overwritten (none in this kernel)."""
    m = re.match(r"\s*v_pk_(mul|add)_f32\s+(v\[\d+:\d+\]),\s*([^,]+),\s*([^\s]+)(.*)", ln)
    op, dst, s0, s1, mods = m.group(1), m.group(2), m.group(3).strip(), m.group(4).strip(), m.group(5)

    def mod(name, default):
        mm = re.search(name + r":\[(\d),(\d)\]", mods)
        return [int(mm.group(1)), int(mm.group(2))] if mm else default
    op_sel, op_sel_hi, neg_lo, neg_hi = mod("op_sel", [0, 0]), mod("op_sel_hi", [1, 1]), mod("neg_lo", [0, 0]), mod("neg_hi", [0, 0])
    d = [_elem(dst, 0), _elem(dst, 1)]
    lo = [("-" if neg_lo[0] else "") + _elem(s0, op_sel[0]), ("-" if neg_lo[1] else "") + _elem(s1, op_sel[1])]
    hi = [("-" if neg_hi[0] else "") + _elem(s0, op_sel_hi[0]), ("-" if neg_hi[1] else "") + _elem(s1, op_sel_hi[1])]
    lo_line = f"\tv_{op}_f32_e64 {d[0]}, {lo[0]}, {lo[1]}\n"
    hi_line = f"\tv_{op}_f32_e64 {d[1]}, {hi[0]}, {hi[1]}\n"
    reads = lambda ops_, r: any(o.lstrip("-") == r for o in ops_)  # noqa: E731
    if not reads(hi, d[0]):
        return [lo_line, hi_line]
    if not reads(lo, d[1]):
        return [hi_line, lo_line]
    strip = lambda ops_: [o.lstrip("-") for o in ops_]  # noqa: E731
    if sorted(strip(lo)) == sorted(strip(hi)) and lo[0][:1] != "-" and lo[1][:1] != "-" and hi[0][:1] != "-" and hi[1][:1] != "-":
        return [lo_line, f"\tv_mov_b32_e32 {d[1]}, {d[0]}\n"]  # both halves are the same commutative expression
    if not reads(lo, d[0]) and not reads(hi, d[1]):
        # crossed halves: each result is computed in the OTHER destination register first, then the two are swapped
        return [f"\tv_{op}_f32_e64 {d[0]}, {hi[0]}, {hi[1]}\n", f"\tv_{op}_f32_e64 {d[1]}, {lo[0]}, {lo[1]}\n", f"\tv_swap_b32 {d[0]}, {d[1]}\n"]
    return None


def wants_split(ins_line, name):
    is_mul = "v_pk_mul" in ins_line
    modified = bool(re.search(r"op_sel|neg_|\bs\[", ins_line))
    if name.startswith("keep_"):
        has_sel, has_selhi, has_neg = "op_sel:" in ins_line, "op_sel_hi:" in ins_line, "neg_" in ins_line
        cls = "add" if not is_mul else ("mul_opsel" if has_sel else ("mul_opselhi" if has_selhi else ("mul_neg" if has_neg else "mul_plain")))
        return cls != name[5:]
    return {"split_all": True, "split_mul": is_mul, "split_add": not is_mul, "split_plain": not modified, "split_modified": modified}.get(name, False)


def transform(lines, name):
    out, inside = [], False
    nth = [-1]
    for ln in lines:
        if ln.startswith(KERNEL + ":"):
            inside = True
        ins = ln.strip().split()[0] if ln.strip() else ""
        pk = inside and re.match(r"v_pk_(mul|add|fma)_f32", ins)
        valu = inside and ins.startswith("v_") and not ins.startswith("v_mfma")
        if pk and name.startswith("keep_one_"):
            if "v_pk_mul" in ln and "op_sel:" in ln:
                nth[0] += 1
                if nth[0] == int(name[9:]):
                    print("   kept packed:", ln.strip(), flush=True)
                    out.append(ln)
                    continue
            two = split_packed(ln)
            if two is not None:
                out += two
                continue
        if pk and name.startswith(("split", "keep_")) and wants_split(ln, name):
            two = split_packed(ln)
            if two is not None:
                out += two
                continue
        if pk and name in ("before7", "both7"):
            out.append("\ts_nop 7\n")
        out.append(ln)
        if pk and name in ("after7", "both7"):
            out.append("\ts_nop 7\n")
        if pk and name == "after1":
            out.append("\ts_nop 1\n")
        if valu and name == "all_valu7":
            out.append("\ts_nop 7\n")
        if inside and ins == "s_endpgm":
            inside = False
    return out


if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = os.path.join(HERE, "_asmvar_geom_O3.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "unopose_amd", "csrc", "geom.hip"), "-o", src])
    lines = open(src).readlines()
    for name in VARIANTS:
        s = os.path.join(HERE, f"_asmvar_{name}.s")
        open(s, "w").writelines(transform(lines, name))
        subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", s[:-2] + ".o"])
        subprocess.check_call([f"{LLVM}/ld.lld", "-shared", s[:-2] + ".o", "-o", os.path.join(HERE, f"_asmvar_{name}.hsaco")])
        n_pk = sum(1 for ln in open(s) if re.match(r"\s*v_pk_(mul|add|fma)_f32", ln))
        print("built", name, n_pk, "packed fp32 instructions,", sum(1 for ln in open(s) if ln.strip().startswith("s_nop")), "s_nop", flush=True)
    sys.exit(0)

import torch
sys.path.insert(0, ROOT)
from unopose_amd.synthetic import make_batch
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
hip = ctypes.CDLL("libamdhip64.so")
A = ctypes.CDLL(os.path.join(HERE, "_aggressors.so"))
A.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
ep, _, _ = make_batch(32, S=224, seed=50, device="cuda")
pts = ep["pts"]; c = pts.mean(1, keepdim=True); pn = ((pts - c) / (pts - c).norm(dim=2).max(1)[0].reshape(-1, 1, 1)).contiguous()
big = torch.zeros(1 << 20, device="cuda"); s2 = torch.cuda.Stream()
B, N, S, radius, cpw = 32, 2048, 64, 0.1, 8


def neighbour():
    for _ in range(6):
        assert A.aggressor_launch(2, big.data_ptr(), 2048, 3000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0


base = None
ONLY = os.environ.get("ASM_VAR_ONLY")
for name in VARIANTS:
    if ONLY and name not in ONLY.split(","):
        continue
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipModuleLoad(ctypes.byref(mod), os.path.join(HERE, f"_asmvar_{name}.hsaco").encode()) == 0
    assert hip.hipModuleGetFunction(ctypes.byref(fn), mod, KERNEL.encode()) == 0

    def run():
        out = torch.empty(B, 6, N, S, device="cuda")
        args = [ctypes.c_void_p(pn.data_ptr()), ctypes.c_int(N), ctypes.c_float(radius), ctypes.c_int(S), ctypes.c_int(cpw), ctypes.c_void_p(out.data_ptr())]
        arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
        rc = hip.hipModuleLaunchKernel(fn, (N + 4 * cpw - 1) // (4 * cpw), B, 1, 256, 1, 1, (3 * N + 4 * S) * 4, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), arr, None)
        assert rc == 0, rc
        return out

    want = run().clone(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(); e.record(); torch.cuda.synchronize()
    if base is None:
        base = want
    bad = 0; centres = 0
    for it in range(RUNS):
        with torch.cuda.stream(s2):
            neighbour()
        out = run(); torch.cuda.synchronize()
        d = out != want
        if d.any():
            bad += 1; centres = max(centres, int(d.any(dim=3).any(dim=1).sum()))
    print(f"{name:10s} ({s.elapsed_time(e):5.2f} ms alone, {'bit-identical to orig alone' if torch.equal(want, base) else 'DIFFERS from orig alone'}): "
          f"{bad:2d} of {RUNS} launches differ beside the MFMA chain (up to {centres} centres)", flush=True)
