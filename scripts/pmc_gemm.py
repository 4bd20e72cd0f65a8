"""Target of the rocprofv3 --pmc passes for the `roofline` kernel of bench.py: the four linears of one ViT-B block on
csrc/gemm.hip at the bench row count (M = 64 images x 1374 tokens), REPS launches each in a fixed order
(qkv, proj, fc1 + GELU, fc2) so that scripts/pmc_summary.py can attribute dispatches to shapes by position; then the
fp32-class kernel (csrc/gemm_f32.hip) on qkv and fc1.  Nothing else runs in the process.
Round 6 (FOLD): the four shapes in the form the step runs them -- proj / fc2 with the residual epilogue (EPI 5), qkv / fc1 with LayerNorm
applied in the epilogue (EPI 6 / 7); the consumer's row partials are made up with torch (no extra GEMM dispatch)."""
import os
import sys

ORDER = (("qkv", 768, 2304, False), ("proj", 768, 768, False), ("fc1", 768, 3072, True), ("fc2", 3072, 768, False))
REPS = 3
FOLD = True
if __name__ == "__main__":
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from unopose_amd import ops

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda")
    M = 2 * B * 1374
    for name, K, N, gelu in ORDER:
        a = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        if FOLD and ops.ln_fold_ok(M, 768):
            import torch.nn as nn

            lin = nn.Linear(K, N).to(dev)
            if N == 768:
                xres = torch.randn(M, N, device=dev)
                gamma = nn.Parameter(torch.rand(768, device=dev))
                for _ in range(REPS):
                    ops.linear_residual_(xres, a, lin, gamma)
            else:
                norm = nn.LayerNorm(768, eps=1e-6).to(dev)
                st = torch.rand((M + 255) // 256 * 256, 3, 2, device=dev) * 256 + 300
                for _ in range(REPS):
                    ops.linear_lnfold(a, st, lin, norm, gelu=gelu)
        else:
            for _ in range(REPS):
                ops.linear_bf16_hip(a, w, bias, gelu)
        torch.cuda.synchronize()
        del a, w, bias
    for name, K, N, gelu in (ORDER[0], ORDER[2]):
        a_s = ops.split_f32(torch.randn(M, K, device=dev))
        w_s = ops.split_f32(torch.randn(N, K, device=dev) / K ** 0.5)
        bias = torch.randn(N, device=dev)
        for _ in range(REPS):
            ops.linear_f32x3(a_s, w_s, bias, M, N, K, gelu=gelu, out="split")
        torch.cuda.synchronize()
        del a_s, w_s, bias
    print("done")
