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
VARIANTS = ["orig", "after1", "after7", "before7", "both7", "all_valu7"]


def transform(lines, name):
    out, inside = [], False
    for ln in lines:
        if ln.startswith(KERNEL + ":"):
            inside = True
        ins = ln.strip().split()[0] if ln.strip() else ""
        pk = inside and re.match(r"v_pk_(mul|add|fma)_f32", ins)
        valu = inside and ins.startswith("v_") and not ins.startswith("v_mfma")
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
for name in VARIANTS:
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
