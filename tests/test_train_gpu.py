"""GPU: the training path (BASELINE configs[3]) against the reference's own training forward +
backward (tests/golden/train_forward.npz: B = 2, 512 dense / 196 coarse points, tamed weights, injected pose noise)."""
import os

import numpy as np
import pytest
import torch

from train_case import GRAD_KEYS, make_train_batch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _model():
    from oracle.unopose_ref import default_cfg, random_state_dict  # weights only
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.train import freeze_backbone

    m = UNOPose(default_model_cfg(fine_npoint=512))
    m.load_state_dict(random_state_dict(default_cfg(), seed=0, tame=0.1), strict=True)
    return freeze_backbone(m.cuda())


def test_training_forward_backward_matches_reference():
    from unopose_amd.losses import process_loss

    z = np.load(os.path.join(GOLD, "train_forward.npz"))
    model = _model().train()
    batch, aug = make_train_batch()
    ep = {k: v.cuda() for k, v in batch.items()}
    ep["aug_pose"] = (aug[0].cuda(), aug[1].cuda())
    out = model(ep)
    info = process_loss(out)
    info["loss"].backward()
    keys = [k[4:] for k in z.files if k.startswith("ep__")]
    assert len(keys) == 24 and all(k in out for k in keys)  # 2 stages x (3 x 3 losses + acc / fg_num / dis)
    for k in keys:
        got, want = out[k].detach().cpu().numpy(), z["ep__" + k]
        if "loss" in k:
            # ill-conditioned local frames (implementation-defined in the reference, test_geom_gpu) perturb a few PE rows
            assert np.allclose(got, want, rtol=5e-3, atol=1e-4), (k, got, want)
    assert abs(float(info["loss"].detach()) - float(z["loss"])) < 2e-3 * float(z["loss"])
    for k in ("coarse_hard_acc", "coarse_hard_fg_num", "coarse_hard_dis"):  # coarse stage: no PE involved -> tight
        assert np.allclose(out[k].detach().cpu().numpy(), z["ep__" + k], rtol=1e-4, atol=1e-4), k
    params = dict(model.named_parameters())
    for k in GRAD_KEYS:
        g = params[k].grad
        assert g is not None, k
        want = float(z["gradnorm__" + k])
        # parameters of the positional-encoding MLP see the implementation-defined frames directly: 10 %; the rest 3 %
        tol = 0.1 if ".PE." in k else 3e-2
        assert abs(float(g.norm()) - want) < tol * want + 1e-7, (k, float(g.norm()), want)
        head = g.flatten()[:16].cpu().numpy()
        # element-wise on the first 16 entries: within 10 % of the largest of them (summation order + the max-over-k
        # selection in the embedding move individual small entries; the norms above are the tight check)
        assert np.allclose(head, z["gradhead__" + k], rtol=5e-2, atol=0.1 * np.abs(z["gradhead__" + k]).max() + 1e-8), k
    assert params["feature_extraction.rgb_net.vit.blocks.0.attn.qkv.weight"].grad is None  # frozen backbone
    bn = model.fine_point_matching.PE.mlp1.layer0.normlayer.bn
    assert np.allclose(bn.running_mean.cpu().numpy(), z["bn_running_mean"], atol=2e-3)
    assert np.allclose(bn.running_var.cpu().numpy(), z["bn_running_var"], rtol=2e-2, atol=1e-4)
    # back to eval: the fused inference path is unaffected by the excursion (no stale mode, caches keyed on versions)
    model.eval()
    with torch.no_grad():
        o = model({k: v.cuda() for k, v in batch.items() if "label" not in k})
    assert torch.isfinite(o["pred_R"]).all() and "fine_atten_loss0" not in o


def test_train_steps_reduce_the_loss_fp32_and_bf16():
    from unopose_amd.train import build_optimizer, train_step

    batch, _ = make_train_batch()
    batch = {k: v.cuda() for k, v in batch.items()}
    for amp in (None, torch.bfloat16):
        torch.manual_seed(0)
        np.random.seed(0)
        model = _model()
        opt, sched = build_optimizer(model, lr=2e-4, total_iters=100, warmup_iters=0)
        w0 = model.fine_point_matching.out_proj.weight.detach().clone()
        losses = [float(train_step(model, batch, opt, sched, amp_dtype=amp)["loss"]) for _ in range(6)]
        assert all(np.isfinite(losses)) and min(losses[3:]) < losses[0], (amp, losses)
        assert not torch.equal(model.fine_point_matching.out_proj.weight, w0)


def test_train_step_at_baseline_config3_shape():
    """BASELINE configs[3]'s point count (4096 query points per pair; 8 GPUs x DDP is `train.wrap_ddp`, covered by the
    world-2 gloo test): one fp32 training step runs, every loss is finite, every trainable parameter that the forward
    reads receives a gradient, the frozen backbone none."""
    from oracle.unopose_ref import default_cfg, random_state_dict
    from unopose_amd.losses import process_loss
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.train import freeze_backbone

    m = UNOPose(default_model_cfg(fine_npoint=4096))
    m.load_state_dict(random_state_dict(default_cfg(), seed=0, tame=0.1), strict=True)
    m = freeze_backbone(m.cuda()).train()
    batch, _ = make_train_batch(B=2, nq=4096, nt=6000, seed=5)
    out = m({k: v.cuda() for k, v in batch.items()})
    info = process_loss(out)
    assert torch.isfinite(info["loss"]) and all(torch.isfinite(v).all() for k, v in info.items())
    info["loss"].backward()
    missing = [k for k, p in m.named_parameters() if p.requires_grad and p.grad is None]
    assert missing == [], missing
    assert all(p.grad is None for p in m.feature_extraction.rgb_net.vit.parameters())


def test_no_library_convolution_or_batchnorm_in_the_training_step():
    """The eval guard (test_pipeline_gpu.py::test_no_library_gemm_on_the_eval_path) extended to model.train(): during a training
    forward + backward no convolution and no batch-norm op of the library (MIOpen) runs on GPU data -- the PE's SharedMLP is
    csrc/conv_train.hip + csrc/bn_train.hip under autograd, the ViT's patch embedding is the own GEMM -- and the profiler sees the own
    kernels of all three passes.  Self-checking: with the switches off the guard fires."""
    from torch.profiler import ProfilerActivity, profile

    from unopose_amd import ops
    from unopose_amd.losses import process_loss

    model = _model().train()
    batch, aug = make_train_batch()

    def step():
        ep = {k: v.cuda() for k, v in batch.items()}
        ep["aug_pose"] = (aug[0].cuda(), aug[1].cuda())
        model.zero_grad(set_to_none=True)
        process_loss(model(ep))["loss"].backward()
        torch.cuda.synchronize()

    def library_ops():
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step()
        names = [e.key for e in prof.key_averages()]
        lib = sorted({n for n in names if any(t in n.lower() for t in ("miopen", "convolution", "conv2d", "batch_norm", "batchnorm", "igemm", "sp3asm"))})
        return lib, names

    step()  # caches
    lib, names = library_ops()
    assert lib == [], lib
    for k in ("conv1x1_f32_kernel", "conv1x1_wgrad_kernel", "bn_relu_maxpool_kernel", "bn_relu_maxpool_bwd_apply_kernel", "bn_relu_apply_kernel",
              # round 6: the geometric embedding forward + backward (no scatter_ of the composite's max backward, no sin / cos passes)
              "geo_embed_table_kernel", "geo_embed_table_bwd_kernel"):
        assert any(k in n for n in names), k
    assert not any(n in ("aten::scatter_", "aten::sin", "aten::cos") for n in names), [n for n in names if n in ("aten::scatter_", "aten::sin", "aten::cos")]
    ops.TRAIN_OWN_CONV, ops.USE_FUSED_BN_RELU = False, False
    try:
        lib, _ = library_ops()
    finally:
        ops.TRAIN_OWN_CONV, ops.USE_FUSED_BN_RELU = True, True
    assert len(lib) >= 2, lib


@pytest.mark.parametrize("amp", [False, True])
@pytest.mark.parametrize("relu", [False, True])
def test_trainable_linear_on_own_gemm_matches_torch_autograd(amp, relu, monkeypatch):
    """`ops.linear` in differentiable mode: forward and input gradient on csrc/gemm_f32.hip (fp32) / csrc/gemm.hip (autocast),
    weight / bias gradients through torch -- against torch.nn.functional.linear's autograd in float64 (fp32 mode: the bf16 x 3
    products give ~2^-17 relative error) or in the same bf16 autocast (bf16 mode: operand rounding only)."""
    from unopose_amd import ops

    monkeypatch.setattr(ops, "TRAIN_OWN_GEMM_MIN_FLOP", 0.0)  # the size threshold would send this small problem to nn.Linear
    torch.manual_seed(0)
    rows, K, N = 3 * 197, 256, 512
    lin = torch.nn.Linear(K, N).cuda()
    x = torch.randn(3, 197, K, device="cuda", requires_grad=True)
    gy = torch.randn(3, 197, N, device="cuda")
    with ops.differentiable(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = ops.linear(x, lin, relu=relu)
    assert y.grad_fn is not None and type(y.grad_fn).__name__ == "_LinearFnBackward"  # the own-GEMM path, not nn.Linear
    y.float().backward(gy)
    got = (y.detach().float(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = lin.weight.grad = lin.bias.grad = None
    bf = lambda t: t.detach().to(torch.bfloat16).double()  # noqa: E731
    if amp:
        # float64 evaluation of what the bf16 path computes: operands rounded to bf16, products and sums exact.  The ReLU mask is the
        # kernel's own (checked against the float64 pre-activation wherever that is not within bf16 rounding of zero): torch's
        # autocast rounds x W^T to bf16 BEFORE the bias add, the fused epilogue after it, so a comparison with torch's own mask
        # would differ in the elements that land on the other side of zero
        pre = bf(x).reshape(-1, K) @ bf(lin.weight).t() + lin.bias.detach().double()
        mask = (got[0].reshape(-1, N) > 0) if relu else torch.ones_like(pre, dtype=torch.bool)
        if relu:
            clear = pre.abs() > 0.05
            assert torch.equal(mask[clear], (pre > 0)[clear]) and clear.float().mean() > 0.9
        g = bf(gy).reshape(-1, N) * mask
        want = ((pre * mask).reshape(3, 197, N), (g @ bf(lin.weight)).reshape(3, 197, K), g.t() @ bf(x).reshape(-1, K), g.sum(0))
        tol = 2e-2
    else:
        xd, wd, bd = x.detach().double().requires_grad_(), lin.weight.detach().double().requires_grad_(), lin.bias.detach().double().requires_grad_()
        r = torch.nn.functional.linear(xd, wd, bd)
        r = torch.relu(r) if relu else r
        r.backward(gy.double())
        want = (r.detach(), xd.grad, wd.grad, bd.grad)
        tol = 3e-5
    for name, g, w in zip(("y", "dx", "dw", "db"), got, want):
        scale = float(w.abs().max())
        assert float((g.double() - w.double()).abs().max()) < tol * scale, (name, float((g.double() - w.double()).abs().max()), scale)


@pytest.mark.parametrize("rows,N,K", [(20000, 256, 256), (16411, 128, 384), (65536, 512, 256), (513, 256, 128)])
def test_own_linear_weight_gradient_matches_float64(rows, N, K, monkeypatch):
    """csrc/conv_train.hip::linear_wgrad_f32_kernel (dW = dY^T X on the fp32 matrix instruction, row slices combined in double) against
    float64: 2e-6 of the result's scale, odd row counts and a last row pair that is half empty included; then through `ops.linear`'s
    backward (the path the training step takes for its large projections)."""
    from unopose_amd import ops
    from unopose_amd._lib import call, lib, ptr, stream_ptr

    g = torch.Generator().manual_seed(rows + N)
    gy = torch.randn(rows, N, generator=g).cuda()
    x = torch.randn(rows, K, generator=g).cuda()
    splits = lib().unopose_linear_wgrad_f32_splits(rows, N, K)
    assert 1 <= splits <= 1024
    ws = torch.empty(splits * N * K, device="cuda")
    dw = torch.empty(N, K, device="cuda")
    call("unopose_linear_wgrad_f32", ptr(gy), ptr(x), rows, N, K, ptr(ws), ptr(dw), stream_ptr())
    want = gy.double().t() @ x.double()
    assert float((dw.double() - want).abs().max()) < 2e-6 * float(want.abs().max())
    if rows >= ops.TRAIN_OWN_WGRAD_MIN_ROWS and N % 256 == 0 and K % 256 == 0:  # (shapes the forward's own GEMM takes)
        monkeypatch.setattr(ops, "TRAIN_OWN_GEMM_MIN_FLOP", 0.0)
        lin = torch.nn.Linear(K, N).cuda()
        xin = x.clone().requires_grad_(True)
        with ops.differentiable():
            y = ops.linear(xin, lin)
        assert type(y.grad_fn).__name__ == "_LinearFnBackward"
        y.backward(gy)
        assert float((lin.weight.grad.double() - want).abs().max()) < 2e-6 * float(want.abs().max())
        ops.TRAIN_OWN_WGRAD = False
        try:
            lin.weight.grad = None
            with ops.differentiable():
                ops.linear(xin, lin).backward(gy)
        finally:
            ops.TRAIN_OWN_WGRAD = True
        assert float((lin.weight.grad.double() - want).abs().max()) < 1e-4 * float(want.abs().max())  # the library's fp32 GEMM (its own summation order)


@pytest.mark.parametrize("B,n1,n2", [(2, 196, 196), (3, 1000, 1300), (2, 4096, 4096), (2, 5, 300)])
def test_fused_saliency_pair_matches_torch_autograd(B, n1, n2):
    """csrc/saliency_train.hip (softmax(inner, 2) @ s2 and softmax(inner^T, 2) @ s1 for inner = atten[:, 1:, 1:], forward + backward)
    against the reference's expression under torch autograd in float64: values and all three gradients to 2e-5 of their scale;
    row 0 / column 0 of the similarity (the background class) receive exactly zero."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(B + n1 + n2)
    atten = (6 * torch.randn(B, n1 + 1, n2 + 1, generator=g)).cuda().requires_grad_()
    s1 = torch.randn(B, n1, 1, generator=g).cuda().requires_grad_()
    s2 = torch.randn(B, n2, 1, generator=g).cuda().requires_grad_()
    g1, g2 = torch.randn(B, n1, 1, generator=g).cuda(), torch.randn(B, n2, 1, generator=g).cuda()
    m1, m2 = ops.saliency_pair(atten, s1, s2)
    assert "SaliencyFn" in type(m1.grad_fn).__name__
    torch.autograd.backward([m1, m2], [g1, g2])
    ad, d1, d2 = (t.detach().double().requires_grad_() for t in (atten, s1, s2))
    inner = ad[:, 1:, 1:]
    r1 = torch.matmul(torch.softmax(inner, dim=2), d2)
    r2 = torch.matmul(torch.softmax(inner.transpose(1, 2), dim=2), d1)
    torch.autograd.backward([r1, r2], [g1.double(), g2.double()])
    for name, got, want in (("m1", m1, r1), ("m2", m2, r2), ("d_atten", atten.grad, ad.grad), ("ds1", s1.grad, d1.grad), ("ds2", s2.grad, d2.grad)):
        sc = float(want.detach().abs().max())
        e = float((got.detach().double() - want.detach()).abs().max())
        assert e < 2e-5 * sc + 1e-12, (name, e, sc)
    assert float(atten.grad[:, 0].abs().max()) == 0.0 and float(atten.grad[:, :, 0].abs().max()) == 0.0
    ops.TRAIN_FUSED_SALIENCY = False
    try:
        t1, t2 = ops.saliency_pair(atten.detach(), s1.detach(), s2.detach())
    finally:
        ops.TRAIN_FUSED_SALIENCY = True
    assert float((t1 - m1.detach()).abs().max()) < 2e-5 * float(t1.abs().max())


def test_fused_infonce_matches_cross_entropy_pair():
    """ops.infonce_two_way (csrc/posehead.hip statistics + one gradient pass) against the two F.cross_entropy calls of
    loss_utils.py:181-187 in float64: values and the gradient w.r.t. the similarity, ragged sizes, labels incl. background."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(3)
    for B, R, C in ((2, 198, 151), (3, 1030, 2051)):
        a = (10 * torch.randn(B, R, C, generator=g)).cuda().requires_grad_()
        l1 = torch.randint(0, C, (B, R - 1), generator=g).cuda()
        l2 = torch.randint(0, R, (B, C - 1), generator=g).cuda()
        l1[:, ::3] = 0
        w = torch.rand(B, generator=g).cuda()
        got = ops.infonce_two_way(a, l1, l2)
        assert type(got.grad_fn).__name__ == "_InfoNCEFnBackward"
        (got * w).sum().backward()
        ga = a.grad.clone()
        ad = a.detach().double().requires_grad_()
        want = 0.5 * (torch.nn.functional.cross_entropy(ad.transpose(1, 2)[:, :, 1:], l1, reduction="none").mean(1)
                      + torch.nn.functional.cross_entropy(ad[:, :, 1:], l2, reduction="none").mean(1))
        (want * w.double()).sum().backward()
        assert torch.allclose(got.double(), want, rtol=2e-6, atol=1e-5), (got, want)
        scale = float(ad.grad.abs().max())
        assert float((ga.double() - ad.grad).abs().max()) < 1e-5 * scale + 1e-9, (float((ga.double() - ad.grad).abs().max()), scale)


@pytest.mark.parametrize("shape", [(3, 32, 50, 64), (2, 128, 33, 256), (1, 6, 7, 32), (4, 64, 1025, 64), (2, 16, 300, 128)])
def test_fused_bn_relu_maxpool_train_matches_torch(shape):
    """csrc/bn_train.hip, the last SharedMLP layer with its pooling: max_s relu(BatchNorm2d(x)) with batch statistics, forward +
    backward from the POOLED gradient, vs the torch modules in float64 (same tolerances as the unpooled op).  One column is
    duplicated many times, as ball-query padding does: ties between equal maxima must not change any gradient."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(sum(shape) + 5)
    x0 = torch.randn(*shape, generator=g) * 1.7 + 0.4
    x0[..., shape[3] // 2:] = x0[..., :1]  # the second half of every neighbourhood = copies of its first neighbour
    x = x0.cuda().requires_grad_(True)
    dy = torch.randn(*shape[:3], generator=g).cuda()
    bn = torch.nn.BatchNorm2d(shape[1]).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(shape[1], generator=g) + 0.5)
        bn.weight[::3] *= -1.0  # negative scales: the winner is the smallest input
        bn.bias.copy_(torch.randn(shape[1], generator=g) * 0.3)
    ref_bn = torch.nn.BatchNorm2d(shape[1]).double().cuda().train()
    ref_bn.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    xd = x.detach().double().requires_grad_(True)
    yr = torch.relu(ref_bn(xd)).max(dim=3)[0]
    yr.backward(dy.double())
    y = ops.bn_relu_maxpool(x, bn)
    assert "BNReLUMaxPoolTrain" in type(y.grad_fn).__name__  # the fused path was taken
    y.backward(dy)
    sc = float(yr.detach().abs().max())
    assert float((y.detach().double() - yr.detach()).abs().max()) < 2e-6 * sc
    # the gradient of tied winners may land on any of the equal columns: compare what is invariant, the sum over each tie class
    def classes(t):
        half = shape[3] // 2
        return torch.cat([t[..., :1] + t[..., half:].sum(-1, keepdim=True), t[..., 1:half]], -1)
    gs = float(xd.grad.abs().max())
    assert float((classes(x.grad.double()) - classes(xd.grad)).abs().max()) < 2e-6 * gs * shape[3] + 1e-9
    assert float((bn.weight.grad.double() - ref_bn.weight.grad).abs().max()) < 1e-5 * float(ref_bn.weight.grad.abs().max())
    assert float((bn.bias.grad.double() - ref_bn.bias.grad).abs().max()) < 1e-5 * float(ref_bn.bias.grad.abs().max())
    assert float((bn.running_mean.double() - ref_bn.running_mean).abs().max()) < 1e-6
    assert float((bn.running_var.double() - ref_bn.running_var).abs().max()) < 1e-6 * float(ref_bn.running_var.abs().max())
    # other neighbourhood sizes / eval mode: the unfused composition
    bn.eval()
    assert torch.equal(ops.bn_relu_maxpool(x.detach(), bn), torch.relu(bn(x.detach())).max(dim=3)[0])


@pytest.mark.parametrize("B,cin,cout,N,S,xgrad", [(2, 6, 32, 37, 64, False), (3, 32, 64, 50, 64, True), (2, 64, 128, 33, 256, True),
                                                    (1, 64, 128, 1, 64, True), (5, 32, 64, 1030, 64, True), (2, 64, 128, 2048, 64, True)])
def test_own_conv1x1_train_matches_torch(B, cin, cout, N, S, xgrad):
    """csrc/conv_train.hip (bias-free 1 x 1 Conv2d on (B, C, N, S): forward, input gradient, weight gradient on the fp32 matrix
    instruction) vs torch.nn.functional.conv2d in float64: every result within 2e-6 of its tensor's scale (fp32 products and
    accumulation over <= 128 channels; the weight gradient sums up to 262 k positions per workgroup in fp32, partials in double)."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(B * 1000 + cin + cout + N)
    x = torch.randn(B, cin, N, S, generator=g).cuda().requires_grad_(xgrad)
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).cuda()
    dy = torch.randn(B, cout, N, S, generator=g).cuda()
    y = ops.conv1x1(x, conv)
    assert "Conv1x1Fn" in type(y.grad_fn).__name__  # the own path was taken
    y.backward(dy)
    xd = x.detach().double().requires_grad_(xgrad)
    wd = conv.weight.detach().double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xd, wd)
    yr.backward(dy.double())
    assert float((y.detach().double() - yr.detach()).abs().max()) < 2e-6 * float(yr.detach().abs().max())
    assert float((conv.weight.grad.double() - wd.grad).abs().max()) < 2e-6 * float(wd.grad.abs().max()), \
        (float((conv.weight.grad.double() - wd.grad).abs().max()), float(wd.grad.abs().max()))
    if xgrad:
        assert float((x.grad.double() - xd.grad).abs().max()) < 2e-6 * float(xd.grad.abs().max())
    else:
        assert x.grad is None
    # slabs that are not a multiple of 64 positions, and the switch: the module itself
    x2 = torch.randn(1, cin, 3, 33, generator=g).cuda()
    assert torch.equal(ops.conv1x1(x2, conv), conv(x2))


@pytest.mark.parametrize("shape", [(3, 32, 50, 64), (2, 128, 33, 256), (1, 6, 7, 4), (4, 64, 1025, 64)])
def test_fused_bn_relu_train_matches_torch(shape):
    """csrc/bn_train.hip (relu(BatchNorm2d(x)) with batch statistics, forward + backward, running statistics) vs the torch modules in
    float64: outputs / input gradient 2e-6 of the tensor's scale, parameter gradients 1e-5, running statistics 1e-6."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 1.7 + 0.4).cuda().requires_grad_(True)
    dy = torch.randn(*shape, generator=g).cuda()
    bn = torch.nn.BatchNorm2d(shape[1]).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(shape[1], generator=g) + 0.5)
        bn.bias.copy_(torch.randn(shape[1], generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(shape[1], generator=g))
        bn.running_var.copy_(torch.rand(shape[1], generator=g) + 0.5)
    ref_bn = torch.nn.BatchNorm2d(shape[1]).double().cuda().train()
    ref_bn.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    xd = x.detach().double().requires_grad_(True)
    yr = torch.relu(ref_bn(xd))
    yr.backward(dy.double())
    y = ops.bn_relu(x, bn)
    assert y.grad_fn is not None and "BNReLUTrain" in type(y.grad_fn).__name__  # the fused path was taken
    y.backward(dy)
    sc = float(yr.detach().abs().max())
    assert float((y.detach().double() - yr.detach()).abs().max()) < 2e-6 * sc
    assert float((x.grad.double() - xd.grad).abs().max()) < 2e-6 * float(xd.grad.abs().max()) + 1e-9
    assert float((bn.weight.grad.double() - ref_bn.weight.grad).abs().max()) < 1e-5 * float(ref_bn.weight.grad.abs().max())
    assert float((bn.bias.grad.double() - ref_bn.bias.grad).abs().max()) < 1e-5 * float(ref_bn.bias.grad.abs().max())
    assert float((bn.running_mean.double() - ref_bn.running_mean).abs().max()) < 1e-6
    assert float((bn.running_var.double() - ref_bn.running_var).abs().max()) < 1e-6 * float(ref_bn.running_var.abs().max())
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    # eval mode and half the switch: the modules themselves
    bn.eval()
    assert torch.equal(ops.bn_relu(x.detach(), bn), torch.relu(bn(x.detach())))


@pytest.mark.parametrize("B,n,reduction", [(2, 197, "max"), (3, 64, "max"), (2, 197, "mean"), (1, 5, "max")])
def test_geo_embedding_under_autograd_matches_the_composite_in_float64(B, n, reduction):
    """Round 6: GeometricStructureEmbedding under autograd on the table kernels (ops._GeoEmbedFn: csrc/embed.hip forward with the recorded
    arg-max, backward = the table-shaped scatter) against torch autograd over the op-by-op composite (transformer.py:303-350) evaluated in
    float64: the output and the four parameter gradients, for max and mean reduction, 197 / 64 / 5 tokens."""
    import copy

    from unopose_amd import ops
    from unopose_amd.model import UNOPose, default_model_cfg

    g = torch.Generator().manual_seed(B * 1000 + n)
    torch.manual_seed(B * 1000 + n + 1)  # (the module's default init draws from the global generator)
    m = UNOPose(default_model_cfg(fine_npoint=1024)).geo_embedding.cuda()
    m.reduction_a = reduction
    pts = torch.randn(B, n, 3, generator=g).cuda()
    pts = pts / pts.norm(dim=2).max()  # radius-normalised, as the model feeds it
    dE = torch.randn(B, n, n, 256, generator=g).cuda()
    with ops.differentiable():
        out = ops.geo_embedding(pts, m)
    assert out.grad_fn is not None and "GeoEmbed" in type(out.grad_fn).__name__
    out.backward(dE)
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    m64 = copy.deepcopy(m).double()
    for p in m64.parameters():
        p.grad = None
    # the composite in float64 (the ops of ops.geo_embedding_torch, transformer.py:303-350; neighbours from the fp32 distances like the kernel)
    p64 = pts.double()
    dist = torch.cdist(p64, p64)
    knn = torch.cdist(pts, pts).topk(k=4, dim=2, largest=False)[1][:, :, 1:]
    knn_pts = torch.gather(p64.unsqueeze(1).expand(B, n, n, 3), 2, knn.unsqueeze(3).expand(B, n, 3, 3))
    rv = (knn_pts - p64.unsqueeze(2)).unsqueeze(2).expand(B, n, n, 3, 3)
    av = (p64.unsqueeze(1) - p64.unsqueeze(2)).unsqueeze(3).expand(B, n, n, 3, 3)
    a_idx = torch.atan2(torch.linalg.norm(torch.cross(rv, av, dim=-1), dim=-1), (rv * av).sum(-1)) * m.factor_a
    div = m64.embedding.div_term

    def sinus(idx):
        om = idx.unsqueeze(-1) * div
        return torch.stack([torch.sin(om), torch.cos(om)], dim=-1).reshape(*idx.shape, -1)

    a_emb = m64.proj_a(sinus(a_idx))
    ref = m64.proj_d(sinus(dist / m.sigma_d)) + (a_emb.max(dim=3)[0] if reduction == "max" else a_emb.mean(dim=3))
    ref.backward(dE.double())
    # forward: the table form is fp32-class (6-point interpolation); off-diagonal entries (the diagonal's d(i,i) is rounding noise of
    # the composite's |x|^2 - 2 x.x + |x|^2, see _offdiag_err in test_model_gpu.py)
    eye = torch.eye(n, dtype=torch.bool, device="cuda")
    e = (out.double() - ref).abs().amax(-1)
    assert e[:, ~eye].max().item() < 2e-4, e[:, ~eye].max().item()
    # gradients: where two angle terms are within the interpolation error of each other the fp32 table form and the float64 composite can pick
    # different maxima for a few (pair, channel) elements, and dE of those lands on other table rows -- an O(|dE|) difference in single
    # entries (any two precisions of this function disagree like that): bounded in the Frobenius norm, and entry-wise at 1 % of the largest
    for k, p in m64.named_parameters():
        scale = p.grad.abs().max().item()
        err = (got[k].double() - p.grad).abs().max().item()
        fro = ((got[k].double() - p.grad).norm() / p.grad.norm()).item()
        assert err < 1e-2 * scale + 1e-6 and fro < 3e-3, (k, err, scale, fro)


def test_zero_nonfinite_grads_one_launch_equals_per_parameter_nan_to_num():
    """train.zero_nonfinite_grads_ on the GPU (unopose_nan_to_num_multi: one launch over a device table of gradient pointers) against torch.nan_to_num
    per parameter (engine_utils.py:14-18): NaN -> 0, +inf -> 1e5, -inf -> -1e5, everything else untouched; odd sizes, an empty and a strided gradient."""
    from unopose_amd import train

    g = torch.Generator().manual_seed(3)
    shapes = [(257, 33), (1,), (4096, 257), (0,), (7, 5, 3), (100000,)]
    m = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(s)) for s in shapes]).cuda()
    want = []
    for p in m:
        gr = torch.randn(p.shape, generator=g).cuda()
        if gr.numel():
            flat = gr.view(-1)
            idx = torch.randint(0, flat.numel(), (max(1, flat.numel() // 50),), generator=g).cuda()
            flat[idx[0::3]] = float("nan")
            flat[idx[1::3]] = float("inf")
            flat[idx[2::3]] = -float("inf")
        p.grad = gr.clone()
        want.append(torch.nan_to_num(gr, nan=0.0, posinf=1e5, neginf=-1e5))
    strided = torch.nn.Parameter(torch.zeros(6, 4, device="cuda"))
    base = torch.full((6, 8), float("nan"), device="cuda")
    strided.grad = base[:, ::2]  # not contiguous: the torch path
    mod = torch.nn.Module()
    mod.ps, mod.extra = m, torch.nn.ParameterList([strided])
    train.zero_nonfinite_grads_(mod)
    for p, w in zip(m, want):
        assert torch.equal(p.grad, w)
    assert bool((strided.grad == 0).all())
    for _ in range(6):  # the staging ring comes round: still right
        m[0].grad = torch.full((257, 33), float("inf"), device="cuda")
        train.zero_nonfinite_grads_(mod)
        assert bool((m[0].grad == 1e5).all())
