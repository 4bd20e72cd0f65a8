"""csrc/fineassign.hip: the fine-stage soft assignment with the similarity recomputed on the matrix cores instead of
stored -- against the oracle's compute_feature_similarity + compute_fine_Rt_overlap (oracle/unopose_ref.py:298, 356, the
restatement of model_utils.py:260-282, 527-566) on the same bf16-rounded features, and against the streaming kernels of
csrc/posehead.hip on the materialised matrix."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
TEMP = 0.1


def constructed_features(B, N1, N2, gen, n_bg=40, noise=0.35):
    """Features (B,N1+1,256), (B,N2+1,256) whose cosine similarity matches row 1+i with column 1+perm[i] (rows past
    N1 - n_bg and unmatched columns look like the background token instead), overlap scores, and congruent clouds."""
    from helpers import random_rotation
    D = 256
    f1 = torch.randn(B, N1 + 1, D, generator=gen)
    f2 = noise * torch.randn(B, N2 + 1, D, generator=gen)
    score = 0.8 + 0.19 * torch.rand(B, N1 + N2, generator=gen)
    p2 = torch.rand(B, N2, 3, generator=gen) - 0.5
    p1 = torch.zeros(B, N1, 3)
    Rg = torch.stack([random_rotation(gen) for _ in range(B)])
    tg = 0.1 * torch.randn(B, 3, generator=gen)
    for b in range(B):
        perm = torch.randperm(N2, generator=gen)
        nm = min(N1 - n_bg, N2)
        f2[b, 0] += F.normalize(f1[b, 0], dim=0) * 4
        matched = torch.zeros(N2, dtype=torch.bool)
        for i in range(N1):
            if i < nm:
                j = int(perm[i])
                f2[b, 1 + j] += F.normalize(f1[b, 1 + i], dim=0) * 4
                matched[j] = True
                p1[b, i] = Rg[b] @ p2[b, j] + tg[b]
            else:  # a background row: looks like the background column
                f1[b, 1 + i] = f2[b, 0] * 4 + noise * torch.randn(D, generator=gen)
                score[b, i] = 0.05 + 0.1 * torch.rand((), generator=gen)
                p1[b, i] = torch.rand(3, generator=gen) * 3 + 2
        for j in range(N2):
            if not matched[j]:
                f2[b, 1 + j] = F.normalize(f1[b, 0], dim=0) * 4 + noise * torch.randn(D, generator=gen)
                score[b, N1 + j] = 0.05 + 0.1 * torch.rand((), generator=gen)
    return f1, f2, score, p1, p2, Rg, tg


def bf16_operands(f1, f2):
    """The operands both sides multiply: L2-normalised, 1/temp on the left, rounded to bf16 (ops.feature_similarity)."""
    a = (F.normalize(f1.float(), p=2, dim=2) / TEMP).to(torch.bfloat16)
    b = F.normalize(f2.float(), p=2, dim=2).to(torch.bfloat16)
    return a, b


@torch.no_grad()
@pytest.mark.parametrize("B,N1,N2", [(3, 300, 417), (2, 2048, 2048), (2, 7, 5), (1, 513, 256)])
def test_fused_fine_pose_vs_oracle(B, N1, N2):
    from oracle import unopose_ref as R
    from unopose_amd import ops

    gen = torch.Generator().manual_seed(100 + N1)
    f1, f2, score, p1, p2, Rg, tg = constructed_features(B, N1, N2, gen, n_bg=min(40, N1 // 4))
    a, b = bf16_operands(f1, f2)
    atten = a.float() @ b.float().transpose(1, 2)  # exact products of the bf16 operands, fp32 sums
    Ro, to, so = R.compute_fine_rt_overlap(atten, score, p1, p2)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert ops.fine_pose_fused_ok(f1.cuda(), f2.cuda())
        Rh, th, sh = ops.fine_pose_from_features(f1.cuda(), f2.cuda(), TEMP, score.cuda(), p1.cuda(), p2.cuda())
    for name, x, y in (("R", Rh, Ro), ("t", th, to), ("score", sh, so)):
        e = (x.cpu() - y).abs().max().item()
        print(f"fused fine pose {B}x{N1}x{N2} {name}: max err {e:.2e}")
        assert e < 1e-4, (name, e)
    if N1 >= 300:
        assert (Rh.cpu() - Rg).abs().max().item() < 5e-3  # and it is the pose the clouds were built with


@torch.no_grad()
@pytest.mark.parametrize("B,N1,N2", [(3, 300, 417), (4, 2048, 2048)])
def test_fused_assignment_vs_streaming_kernels(B, N1, N2):
    """Labels, row weights and soft correspondences against posehead.hip's passes over the stored fp32 matrix."""
    from unopose_amd import ops
    from unopose_amd._lib import call, ptr, stream_ptr

    gen = torch.Generator().manual_seed(7 + N1)
    f1, f2, score, p1, p2, _, _ = constructed_features(B, N1, N2, gen, n_bg=40)
    dev = "cuda"
    a, b = bf16_operands(f1.to(dev), f2.to(dev))
    atten = torch.bmm(a, b.transpose(1, 2), out_dtype=torch.float32).contiguous()
    s1, s2 = score[:, :N1].contiguous().to(dev), score[:, N1:].contiguous().to(dev)
    q = p2.contiguous().to(dev)
    stats, w1, w2 = ops._assign_labels(atten, s1, s2)
    weight = torch.empty(B, N1, device=dev)
    pred = torch.empty(B, N1, 3, device=dev)
    call("unopose_fine_correspondences", ptr(atten), B, N1 + 1, N2 + 1, ptr(s1), ptr(s2), ptr(stats), ptr(w1), ptr(w2), ptr(q),
         ptr(weight), ptr(pred), stream_ptr())
    R_, C_ = N1 + 1, N2 + 1
    ws = torch.empty(B * (R_ + C_) + B * (-(-N1 // 256) - (-N2 // 256)), device=dev)
    w1f, w2f = torch.empty_like(w1), torch.empty_like(w2)
    weightf, predf = torch.empty_like(weight), torch.empty_like(pred)
    call("unopose_fine_assign", ptr(a), ptr(b), B, R_, C_, 256, 1.0 / TEMP, ptr(s1), ptr(s2), ptr(q), ptr(ws), ptr(w1f), ptr(w2f),
         ptr(weightf), ptr(predf), stream_ptr())
    torch.cuda.synchronize()
    assert w1.sum().item() > 0.5 * B * (N1 - 40) and w2.sum().item() > 0.3 * B * N2  # a non-degenerate assignment
    d1, d2 = (w1 != w1f).sum().item(), (w2 != w2f).sum().item()
    print(f"label flips: rows {d1}/{w1.numel()}, cols {d2}/{w2.numel()}")
    assert d1 <= 1e-3 * w1.numel() + 1 and d2 <= 1e-3 * w2.numel() + 1  # only exact near-ties may differ
    same = (w1 == w1f)
    rel = ((weightf - weight).abs() / (weight.abs() + 1e-6))[same]
    print(f"row weights: max rel err {rel.max().item():.2e}; pred max abs err {(predf - pred).abs()[same].max().item():.2e}")
    assert rel.max().item() < 1e-3 if d2 == 0 else rel.median().item() < 1e-4
    on = same & (weight > 1e-3)
    assert (predf - pred).abs()[on].max().item() < 1e-3
    # reciprocal sums of both sides against the streaming statistics (shifted by max there, by 1/temp here)
    rmax, irs = stats[:B * R_].view(B, R_), stats[B * R_:2 * B * R_].view(B, R_)
    rs_f = ws[:B * R_].view(B, R_)
    ref = irs * torch.exp(1.0 / TEMP - rmax)  # 1 / sum exp(x - 1/temp)
    e = ((rs_f - ref).abs() / ref)[:, 1:]
    print(f"row reciprocal sums: max rel err {e.max().item():.2e}")
    assert e.max().item() < 1e-4


def test_fine_assign_rejects_bad_arguments():
    from unopose_amd._lib import call, ptr, stream_ptr
    x = torch.zeros(64, device="cuda")
    with pytest.raises(RuntimeError, match="feature width"):
        call("unopose_fine_assign", ptr(x), ptr(x), 1, 2, 2, 128, 10.0, ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x),
             stream_ptr())
    with pytest.raises(RuntimeError, match="shift"):
        call("unopose_fine_assign", ptr(x), ptr(x), 1, 2, 2, 256, 100.0, ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x), ptr(x),
             stream_ptr())
