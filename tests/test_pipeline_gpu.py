"""unopose_amd.pipeline.PipelinedForward: two forwards in flight return bit-identical poses to one-at-a-time execution
(same kernels on the same inputs; no atomics on the forward path), tickets resolve in any order, the fp32 path falls back
to depth 1, and a configuration with library stream-K GEMMs is refused."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
KEYS = ("init_R", "init_t", "init_pose_score", "pred_R", "pred_t", "pred_pose_score")


@pytest.fixture(scope="module")
def model():
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import trained_like_

    return trained_like_(UNOPose(default_model_cfg())).cuda().eval()


def batches(n, B=3, img=224):
    from unopose_amd.synthetic import make_batch

    out = []
    for i in range(n):
        ep, _, _ = make_batch(B, S=img, seed=50 + i, device="cuda")
        ep["coarse_rand"] = torch.rand(B, 18000, generator=torch.Generator().manual_seed(i)).cuda()
        out.append(ep)
    return out


@torch.no_grad()
@pytest.mark.parametrize("stages", [False, True])
def test_two_in_flight_equals_one_at_a_time(model, stages):
    """stages=False: two whole forwards side by side; stages=True: ViT half and matcher half on two streams."""
    from unopose_amd.pipeline import PipelinedForward

    eps = batches(6)
    seq = PipelinedForward(model, depth=1)
    ref = [{k: seq.submit(dict(ep)).wait()[k].clone() for k in KEYS} for ep in eps]
    pipe = PipelinedForward(model, depth=2, timing=True, stages=stages)
    assert pipe.depth == 2 and len(pipe.streams) == 2
    tickets = [pipe.submit(dict(ep)) for ep in eps]
    for i in (3, 0, 5, 1, 4, 2):  # results may be collected in any order
        out = tickets[i].result()
        for k in KEYS:
            assert torch.equal(out[k], ref[i][k]), (i, k)
    pipe.drain()
    torch.cuda.synchronize()
    assert pipe.stages is stages
    assert len(pipe.history) == 6 and all(a.elapsed_time(b) > 0 for a, b in pipe.history)


@torch.no_grad()
def test_fp32_pipelines_like_bf16_and_library_gemms_are_refused(model):
    """Round 6: the fp32 path (the reference's default precision) runs two forwards in flight as well -- every fp32 GEMM of the eval path is an
    own kernel (`ops.USE_F32X3`; test_no_library_gemm_on_the_eval_path[fp32]) -- with results bit-equal to one at a time; with the
    own fp32 GEMM switched off it falls back to one at a time; a linear the own kernels do not take raises while forwards overlap."""
    from unopose_amd import ops
    from unopose_amd.pipeline import PipelinedForward

    eps = batches(4, B=2)
    seq = PipelinedForward(model, depth=1, autocast_dtype=None)
    ref = [{k: seq.submit(dict(ep)).wait()[k].clone() for k in KEYS} for ep in eps]
    assert ref[0]["pred_R"].dtype == torch.float32
    for stages in (False, True):
        p32 = PipelinedForward(model, depth=2, autocast_dtype=None, stages=stages)
        assert p32.depth == 2 and len(p32.streams) == 2
        tickets = [p32.submit(dict(ep)) for ep in eps]
        for i in (2, 0, 3, 1):
            out = tickets[i].result()
            for k in KEYS:
                assert torch.equal(out[k], ref[i][k]), (stages, i, k)
        p32.close()
    old = ops.USE_F32X3
    ops.USE_F32X3 = False
    try:
        p1 = PipelinedForward(model, depth=2, autocast_dtype=None)
        assert p1.depth == 1 and p1.streams == [None]
    finally:
        ops.USE_F32X3 = old
    lin = torch.nn.Linear(100, 100).cuda()  # N % 256 != 0: not a shape of csrc/gemm_f32.hip
    prev, ops.FORBID_LIBRARY_BF16_GEMM = ops.FORBID_LIBRARY_BF16_GEMM, True
    try:
        with pytest.raises(RuntimeError, match="library GEMM"):
            ops.linear(torch.randn(64, 100, device="cuda"), lin)
    finally:
        ops.FORBID_LIBRARY_BF16_GEMM = prev
    old = ops.HIP_GEMM_ALL
    ops.HIP_GEMM_ALL = False
    try:
        with pytest.raises(RuntimeError, match="HIP_GEMM_ALL"):
            PipelinedForward(model, depth=2)
    finally:
        ops.HIP_GEMM_ALL = old
    with pytest.raises(ValueError):
        PipelinedForward(model, depth=0)


@torch.no_grad()
def test_patch_embed_on_the_hand_written_gemm(model):
    """ops.patch_embed (K = 588 zero-padded to 640, csrc/gemm.hip) against the fp32 convolution it replaces."""
    from unopose_amd import ops

    g = torch.Generator().manual_seed(0)
    x = torch.randn(20, 3, 224, 224, generator=g).cuda()  # 20 x 256 patches = 5120 rows: above the hand-written GEMM's floor
    conv = model.feature_extraction.rgb_net.vit.patch_embed.proj
    ref = torch.nn.functional.conv2d(x, conv.weight, conv.bias, stride=14).flatten(2).transpose(1, 2)
    patches = x.reshape(20, 3, 16, 14, 16, 14).permute(0, 2, 4, 1, 3, 5).reshape(20, 256, 588)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = ops.patch_embed(patches, conv)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape
    e = (out.float() - ref).abs().max().item()
    print(f"patch embed vs fp32 conv: max err {e:.2e} (|ref| max {ref.abs().max().item():.2f})")
    assert e < 2 ** -6 * max(1.0, ref.abs().max().item())


@torch.no_grad()
def test_forward_is_reproducible_run_to_run(model):
    """The same batch ten times, internal side-stream overlaps on: bit-identical outputs.  (Before every bf16 linear moved to
    csrc/gemm.hip, the library's bf16 GEMM kernels running beside the side-stream geometry kernels perturbed the global
    reference frames in about a quarter of such small-batch forwards: scripts/ubench/lrf_dbg.py.)"""
    ep = batches(1)[0]
    model.internal_overlap = True
    outs = []
    for _ in range(10):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            o = model(dict(ep))
        outs.append({k: o[k].clone() for k in KEYS})
    for o in outs[1:]:
        for k in KEYS:
            assert torch.equal(o[k], outs[0][k]), k


@torch.no_grad()
def test_runner_with_pipelined_chunks_writes_the_same_lines(model, tmp_path):
    from unopose_amd.pipeline import PipelinedForward
    from unopose_amd.runner import inference_and_save
    from unopose_amd.synthetic import make_batch

    images = []
    for i in range(2):
        ep, _, _ = make_batch(5, S=224, seed=70 + i, device="cuda")
        img = {k: v[None] for k, v in ep.items()}
        img.update(score=torch.full((1, 5, 1), 0.5 + 0.1 * i, device="cuda"), obj_id=torch.arange(1, 6, dtype=torch.int32).reshape(1, 5, 1),
                   scene_id=torch.IntTensor([48]), img_id=torch.IntTensor([3 + i]))
        images.append(img)

    class Amp(torch.nn.Module):  # the runner calls model(inputs); the autocast region is the caller's
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, ep):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.m(ep)

    torch.manual_seed(5)  # the coarse stage draws its hypotheses inside forward (U:462): same draws for both runs
    a = inference_and_save(Amp(model), images, str(tmp_path / "a.csv"), instance_batch_size=2)
    torch.manual_seed(5)
    b = inference_and_save(Amp(model), images, str(tmp_path / "b.csv"), instance_batch_size=2, pipeline=PipelinedForward(model, depth=2))
    strip = lambda lines: [",".join(l.split(",")[:-1]) for l in lines]  # noqa: E731  (the last field is the wall-clock time)
    assert strip(a) == strip(b) and len(a) == 10


@torch.no_grad()
def test_runner_pipeline_with_reference_cache_equals_sequential(model, tmp_path):
    """ADVICE round 2 (runner.py:103): pipeline= and ref_cache= together.  Missing reference views are encoded through
    `PipelinedForward.encode_reference` -- the pipeline's autocast dtype, ordered after the forwards in flight -- so the lines equal
    those of the chunk-by-chunk loop run under the same autocast with its own cache; the second image reuses the first one's views."""
    from unopose_amd.pipeline import PipelinedForward
    from unopose_amd.runner import ReferenceCache, inference_and_save
    from unopose_amd.synthetic import make_batch

    images = []
    for i in range(2):
        ep, _, _ = make_batch(5, S=224, seed=80 + i, device="cuda")
        if images:  # same reference views as image 0: every lookup of image 1 is a hit
            for k in ("tem1_rgb", "tem1_choose", "tem1_pts"):
                ep[k] = images[0][k][0]
        img = {k: v[None] for k, v in ep.items()}
        img.update(score=torch.full((1, 5, 1), 0.5 + 0.1 * i, device="cuda"), obj_id=torch.arange(1, 6, dtype=torch.int32).reshape(1, 5, 1),
                   scene_id=torch.IntTensor([48]), img_id=torch.IntTensor([3 + i]), ref_keys=[(48, 1, j) for j in range(5)])
        images.append(img)
    torch.manual_seed(5)
    ca = ReferenceCache(model)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = inference_and_save(model, images, str(tmp_path / "a.csv"), instance_batch_size=2, ref_cache=ca)
    torch.manual_seed(5)
    cb = ReferenceCache(model)
    pipe = PipelinedForward(model, depth=2)
    b = inference_and_save(model, images, str(tmp_path / "b.csv"), instance_batch_size=2, ref_cache=cb, pipeline=pipe)
    pipe.close()
    assert (ca.misses, ca.hits) == (cb.misses, cb.hits) == (5, 5)
    strip = lambda lines: [",".join(l.split(",")[:-1]) for l in lines]  # noqa: E731
    assert strip(a) == strip(b) and len(a) == 10
    for k in ca.store:  # the cached features themselves: same precision, same numbers
        for name in ca.store[k]:
            assert torch.equal(ca.store[k][name], cb.store[k][name]), (k, name)


@torch.no_grad()
def test_pipeline_cold_start_on_a_fresh_model():
    """No warm-up: the FIRST forwards of a freshly constructed model go through the pipeline, so its weight caches are built by
    forward 1 on one pipeline stream while forward 2 is already queued on another (the second stream must wait for
    them).  Both pipeline modes; results = the same batches one at a time on a second, identically initialised model."""
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.pipeline import PipelinedForward
    from unopose_amd.synthetic import trained_like_

    eps = batches(4, B=3, img=224)

    def make():  # (the constructor draws the default initialisation from the global generator)
        torch.manual_seed(1234)
        return trained_like_(UNOPose(default_model_cfg()), seed=3).cuda().eval()

    ref_model = make()
    for stages in (False, True):
        ref_model.internal_overlap = stages  # the in-forward side-stream overlaps are on in stage mode only (pipeline.py)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            want = [ref_model(dict(e)) for e in eps]
        fresh = make()
        pf = PipelinedForward(fresh, depth=2, stages=stages)
        tickets = [pf.submit(dict(e)) for e in eps]
        for t, w in zip(tickets, want):
            o = t.result()
            for k in ("pred_R", "pred_t", "init_R", "pred_pose_score"):
                assert torch.equal(o[k], w[k]), (stages, k)
        pf.close()
        assert fresh.internal_overlap is True  # restored


@torch.no_grad()
def test_stage_mode_at_bench_size_equals_one_at_a_time():
    """The benched mode (stage pipeline, B = 32, 518 x 518 crops, bf16): three consecutive batches, bit-identical to running them
    one at a time."""
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.pipeline import PipelinedForward
    from unopose_amd.synthetic import trained_like_

    model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=518)))).cuda().eval()
    eps = batches(3, B=32, img=518)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        want = [{k: model(dict(e))[k].clone() for k in ("pred_R", "pred_t", "pred_pose_score")} for e in eps]
    pf = PipelinedForward(model, depth=2, stages="auto")
    outs = [pf.submit(dict(e)) for e in eps]
    for t, w in zip(outs, want):
        o = t.result()
        for k in w:
            assert torch.equal(o[k], w[k]), k
    pf.close()


@torch.no_grad()  # like every eval entry point (runner, cli, bench, pipeline): with autograd recording, the fp32 fast paths step aside (ops._no_autograd)
@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_no_library_gemm_on_the_eval_path(model, precision):
    """Regression guard for DESIGN.md section 7 (the library's stream-K bf16 GEMMs hang when forwards overlap; its fp32 SGEMM runs
    at a third of csrc/gemm_f32.hip): during an eval forward NO torch matmul-class op may run on GPU data,
    in either precision -- every contraction is a hand-written kernel (csrc/gemm.hip, gemm_f32.hip, bmm_f32.hip, the attention /
    embedding / PE kernels).  Self-checking: with the own GEMMs switched off the guard must fire."""
    import contextlib

    import torch.nn.functional as F

    offenders = []
    patched = []

    def guard(owner, name):
        orig = getattr(owner, name)

        def wrapper(*a, **k):
            ts = [t for t in a if torch.is_tensor(t)]
            if ts and ts[0].is_cuda and ts[0].is_floating_point() and min(t.dim() for t in ts[:2]) >= 1:
                offenders.append((name, [tuple(t.shape) for t in ts[:2]], [str(t.dtype) for t in ts[:2]], torch.is_autocast_enabled()))
            return orig(*a, **k)

        setattr(owner, name, wrapper)
        patched.append((owner, name, orig))

    from unopose_amd import ops

    ep = batches(1, B=2)[0]
    ctx = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if precision == "bf16" else contextlib.nullcontext
    with ctx():
        model(dict(ep))  # builds the per-module weight caches (their one-off einsum / cat run outside the guarded forward)
    for owner, name in ((torch, "matmul"), (torch, "bmm"), (torch, "mm"), (torch, "addmm"), (torch, "baddbmm"), (torch, "einsum"),
                        (torch, "_addmm_activation"), (F, "linear"), (torch.Tensor, "__matmul__"), (torch.Tensor, "matmul")):
        guard(owner, name)
    lin_forward = torch.nn.Linear.forward

    def linear_forward(self, x):
        if x.is_cuda:
            offenders.append(("nn.Linear.forward", tuple(x.shape), str(x.dtype), torch.is_autocast_enabled()))
        return lin_forward(self, x)

    torch.nn.Linear.forward = linear_forward
    try:
        with ctx():
            model(dict(ep))
        clean = list(offenders)
        del offenders[:]
        ops.HIP_GEMM_ALL, ops.USE_F32X3 = False, False  # the guard itself: with the library paths switched back on it must fire
        try:
            with ctx():
                model(dict(ep))
        finally:
            ops.HIP_GEMM_ALL, ops.USE_F32X3 = True, True
        fired = len(offenders)
    finally:
        torch.nn.Linear.forward = lin_forward
        for owner, name, orig in patched:
            setattr(owner, name, orig)
    assert not clean, clean[:8]
    assert fired > 20


@torch.no_grad()
def test_library_bf16_fallback_is_refused_while_forwards_overlap():
    """ADVICE round 2 (ops.py:202): a bf16 linear whose shape csrc/gemm.hip does not take must not slip to a library GEMM
    silently while PipelinedForward has several forwards in flight -- `ops.linear` raises; one at a time it still runs."""
    from unopose_amd import ops

    lin = torch.nn.Linear(100, 100).cuda()
    x = torch.randn(64, 100, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ops.linear(x, lin)  # allowed: nothing else in flight
        assert y.shape == (64, 100)
        prev, ops.FORBID_LIBRARY_BF16_GEMM = ops.FORBID_LIBRARY_BF16_GEMM, True
        try:
            with pytest.raises(RuntimeError, match="library GEMM"):
                ops.linear(x, lin)
            lin2 = torch.nn.Linear(128, 256).cuda()  # a shape the own kernel takes: unaffected
            assert ops.linear(torch.randn(64, 128, device="cuda"), lin2).shape == (64, 256)
        finally:
            ops.FORBID_LIBRARY_BF16_GEMM = prev
