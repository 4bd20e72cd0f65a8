"""GPU regression test of DESIGN.md section 7 (round 3): the local-frame kernels must return bit-identical results while waves of
ANOTHER kernel that issues MFMAs share their CUs.  Before the library was built without packed-fp32 instructions
(`-fno-slp-vectorize -fno-vectorize`) `query_lrf_group` differed in ~3 of 4 launches beside the MFMA-only neighbour of
scripts/ubench/aggressors.hip and the fused PE kernel in every launch beside the token attention."""
import ctypes
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def aggressors(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("agg") / "aggressors.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
                           os.path.join(ROOT, "scripts", "ubench", "aggressors.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


@torch.no_grad()
def test_frame_kernels_are_stable_beside_mfma_neighbours(aggressors):
    from unopose_amd import ops
    from unopose_amd._lib import call, ptr, stream_ptr
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.synthetic import make_batch, trained_like_

    model = trained_like_(UNOPose(default_model_cfg())).cuda().eval()
    ep, _, _ = make_batch(32, S=224, seed=50, device="cuda")
    g = torch.Generator().manual_seed(0)
    pts, tem = ep["pts"], ep["tem1_pts"]
    c = pts.mean(1, keepdim=True)
    pn = ((pts - c) / (pts - c).norm(dim=2).max(1)[0].reshape(-1, 1, 1)).contiguous()
    w = torch.rand(32, 2048, generator=g).cuda()
    src, ref = torch.randn(32, 2048, 3, generator=g).cuda(), torch.randn(32, 2048, 3, generator=g).cuda()
    pe = model.fine_point_matching.PE
    big = torch.zeros(1 << 20, device="cuda")
    yq = torch.randn(64, 197, 1280, generator=g).cuda().bfloat16()
    vt = torch.randn(64, 256, 256, generator=g).cuda().bfloat16()
    Eb = torch.randn(64, 197, 197, 256, generator=g).cuda().bfloat16()
    outa = torch.empty(64, 197, 256, device="cuda", dtype=torch.bfloat16)

    def mfma_chain():
        for _ in range(4):
            assert aggressors.aggressor_launch(2, big.data_ptr(), 2048, 3000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0

    def token_attn():
        for _ in range(12):
            call("unopose_token_attention", ptr(yq), 1280, ctypes.c_void_p(yq.data_ptr() + 768 * 2), 1280, ptr(vt), ctypes.c_void_p(yq.data_ptr() + 256 * 2), 1280,
                 ptr(Eb), 64, 197, 197, 0.125, ptr(outa), stream_ptr())

    victims = {
        "query_lrf_group": lambda: ops.query_lrf_group(pn, 0.1, 64),
        "pe_bf16x3": lambda: ops.pe_group_mlp_max(pn, pe.r2, pe.ns2, pe.mlp2, bf16x3=True),
        "lrf_global": lambda: ops.lrf_global(tem),
        "weighted_procrustes": lambda: torch.cat([t.reshape(32, -1) for t in ops.weighted_procrustes(src, ref, w, 0.001)], 1),
    }
    side = torch.cuda.Stream()
    for name, fn in victims.items():
        want = fn().clone()
        torch.cuda.synchronize()
        for neighbour in (mfma_chain, token_attn):
            for _ in range(8):
                with torch.cuda.stream(side):
                    neighbour()
                out = fn()
                torch.cuda.synchronize()
                assert torch.equal(out, want), (name, neighbour.__name__, int((out != want).sum()))
