"""Round 6: the residual + LayerNorm fold (gemm_kernel.h EPI 5 / 6 / 7) against today's chain, on the ViT's shapes at M = 64 x 1374.
Checks the numbers against an fp32 torch composite and times, same box, interleaved:
   today : proj (EPI 0) -> scale_residual_layernorm -> qkv | fc1+GELU (EPI 0 / 1)      [and fc2 -> LN -> qkv]
   fold  : proj' (EPI 5: residual in the epilogue)  -> qkv' | fc1' (EPI 6 / 7: LayerNorm in the epilogue)
usage (GPU box): python scripts/ubench/lnfold_ab.py"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from unopose_amd._lib import call, ptr, stream_ptr
from unopose_amd import ops
torch.set_grad_enabled(False)
dev = torch.device("cuda")
M = int(os.environ.get("GV_M", 64 * 1374))
C = 768
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(f, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


class LN:  # what ops.scale_residual_layernorm_ expects
    def __init__(self):
        self.weight = (1 + 0.3 * rn(C)).contiguous(); self.bias = (0.2 * rn(C)).contiguous(); self.eps = 1e-6


for prod_name, Kp in (("proj", 768), ("fc2", 3072)):
    for cons_name, Nc, gelu in (("qkv", 2304, 0), ("fc1+gelu", 3072, 1)):
        a = rn(M, Kp).bfloat16()
        Wp = rn(C, Kp) / Kp ** 0.5; bp = 0.1 * rn(C); gamma = 0.05 + 0.45 * torch.rand(C, device=dev, generator=g)
        Wc = rn(Nc, C) / C ** 0.5; bc = 0.1 * rn(Nc)
        ln = LN()
        x0 = (rn(M, C) * 2 + 0.3 * rn(1, C)).contiguous()   # residual stream: per-channel offsets like a trained ViT's
        # ---- today's chain
        Wp_b, Wc_b = Wp.bfloat16().contiguous(), Wc.bfloat16().contiguous()
        y = torch.empty(M, C, device=dev, dtype=torch.bfloat16); out_t = torch.empty(M, Nc, device=dev, dtype=torch.bfloat16)
        x_t = x0.clone()
        def today():
            call("unopose_linear_bf16", ptr(a), ptr(Wp_b), ptr(bp), ptr(y), M, C, Kp, 0, stream_ptr())
            n = ops.scale_residual_layernorm_(x_t, y, gamma, ln)
            call("unopose_linear_bf16", ptr(n), ptr(Wc_b), ptr(bc), ptr(out_t), M, Nc, C, gelu, stream_ptr())
        # ---- the fold
        Wp_f = (gamma[:, None] * Wp).bfloat16().contiguous(); bp_f = (gamma * bp).contiguous()
        Wc_f = (Wc * ln.weight[None, :]).bfloat16().contiguous()
        cvec = Wc_f.float().sum(1).contiguous(); dvec = (Wc @ ln.bias + bc).contiguous()
        Mp = (M + 255) // 256 * 256
        stats = torch.zeros(Mp, 3, 2, device=dev); xb = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        out_f = torch.empty(M, Nc, device=dev, dtype=torch.bfloat16)
        x_f = x0.clone()
        def prod(): call("unopose_linear_bf16_residual", ptr(a), ptr(Wp_f), ptr(bp_f), ptr(x_f), ptr(xb), ptr(stats), M, C, Kp, stream_ptr())
        def cons(): call("unopose_linear_bf16_lnfold", ptr(xb), ptr(Wc_f), ptr(dvec), ptr(cvec), ptr(stats), 3, 1e-6, ptr(out_f), M, Nc, C, gelu, stream_ptr())
        def fold(): prod(); cons()
        # ---- numbers (rows of the first and the last, ragged, tile)
        today(); fold(); torch.cuda.synchronize()
        for sl in (slice(0, 2048), slice(M - 300, M)):
            xr = x0[sl] + gamma * ((a[sl].float() @ Wp.T) + bp)
            ref = torch.nn.functional.layer_norm(xr, (C,), ln.weight, ln.bias, 1e-6) @ Wc.T + bc
            if gelu: ref = torch.nn.functional.gelu(ref)
            et, ef = (out_t[sl].float() - ref).abs(), (out_f[sl].float() - ref).abs()
            ex_t, ex_f = (x_t[sl] - xr).abs().max().item(), (x_f[sl] - xr).abs().max().item()
            st = stats[sl].sum(1)
            es = max((st[:, 0] - x_f[sl].sum(1)).abs().max().item() / C, ((st[:, 1] - (x_f[sl] ** 2).sum(1)).abs() / (x_f[sl] ** 2).sum(1)).max().item())
            print(f"{prod_name}->{cons_name} rows {sl.start}..: out err vs fp32  today max {et.max().item():.4f} mean {et.mean().item():.5f} | fold max {ef.max().item():.4f} mean {ef.mean().item():.5f}"
                  f" | x err today {ex_t:.2e} fold {ex_f:.2e} | xb==bf16(x) {torch.equal(xb[sl], x_f[sl].bfloat16())} | stats err {es:.2e}")
        # ---- time
        tt, tf, tp, tc = [], [], [], []
        lin = lambda: call("unopose_linear_bf16", ptr(a), ptr(Wp_b), ptr(bp), ptr(y), M, C, Kp, 0, stream_ptr())
        lnk = lambda: ops.scale_residual_layernorm_(x_t, y, gamma, ln)
        con = lambda: call("unopose_linear_bf16", ptr(y), ptr(Wc_b), ptr(bc), ptr(out_t), M, Nc, C, gelu, stream_ptr())
        parts = {"prod0": [], "ln": [], "cons0": [], "prod5": [], "cons67": []}
        for r in range(5):
            tt.append(timeit(today)); tf.append(timeit(fold))
            for k, f in (("prod0", lin), ("ln", lnk), ("cons0", con), ("prod5", prod), ("cons67", cons)):
                parts[k].append(timeit(f))
        med = lambda v: sorted(v)[len(v) // 2]
        from unopose_amd._lib import lib
        sw = {}
        for stg in (0, 210, 420, 630, 840, 1260, 1680):
            was = lib().unopose_gemm_fold_stagger(stg)
            sw[stg] = med([timeit(prod) for _ in range(5)])
            lib().unopose_gemm_fold_stagger(was)
        print("   producer EPI5 by stagger (1/8 ticks per K-tile): " + "  ".join(f"{k}: {v:.1f}" for k, v in sw.items()))
        print(f"{prod_name}->{cons_name}: today {med(tt):7.1f} us  fold {med(tf):7.1f} us   | prod EPI0 {med(parts['prod0']):6.1f} + LN {med(parts['ln']):6.1f} + cons {med(parts['cons0']):6.1f}"
              f"   vs   prod EPI5 {med(parts['prod5']):6.1f} + cons EPI6/7 {med(parts['cons67']):6.1f}", flush=True)
        del a, x0, x_t, x_f, y, out_t, out_f, xb
