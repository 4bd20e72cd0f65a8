"""Per-stage GPU time of UNOPose.forward (HIP events around each stage; B pairs, autocast bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopose_amd import ops  # noqa: E402
from unopose_amd.model import UNOPose, default_model_cfg  # noqa: E402
from unopose_amd.synthetic import make_batch, trained_like_  # noqa: E402


class T:
    def __init__(self):
        self.ev = []

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev.append((name, e))

    def report(self):
        torch.cuda.synchronize()
        out = {}
        for (n0, e0), (n1, e1) in zip(self.ev[:-1], self.ev[1:]):
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out


@torch.no_grad()
def run(model, ep, t):
    m = model
    t.mark("start")
    dense_pm, dense_fm, dense_po, dense_fo, radius, _ = m._features(ep)
    t.mark("features(ViT+FPS5000)")
    pm_lrf = ops.lrf_global(ep["pts"])
    po_lrf = ops.lrf_global(ep["tem1_pts"])
    B = dense_pm.size(0)
    bg = torch.ones(B, 1, 3, device=dense_pm.device)
    sp_m, sp_m_lrf, sf_m, idx_m = m._sample_wlrf(dense_pm, pm_lrf, dense_fm, m.coarse_npoint)
    sp_o, sp_o_lrf, sf_o, idx_o = m._sample_wlrf(dense_po, po_lrf, dense_fo, m.coarse_npoint)
    t.mark("lrf+fps196+gathers")
    geo_m = m.geo_embedding(torch.cat([bg, sp_m_lrf], 1))
    geo_o = m.geo_embedding(torch.cat([bg, sp_o_lrf], 1))
    t.mark("geo_embedding x2")
    c = m.coarse_point_matching
    f1 = torch.cat([c.bg_token.expand(B, -1, -1).to(sf_m.dtype), c.in_proj(sf_m).to(sf_m.dtype)], 1)
    f2 = torch.cat([c.bg_token.expand(B, -1, -1).to(sf_m.dtype), c.in_proj(sf_o).to(sf_m.dtype)], 1)
    for blk in c.transformers:
        f1, f2 = blk(f1, geo_m, f2, geo_o)
    t.mark("coarse transformers")
    ep = c(sp_m, sf_m, geo_m, sp_o, sf_o, geo_o, radius, ep)
    t.mark("coarse total (incl. transformers again + pose head)")
    f = m.fine_point_matching
    p1_ = (dense_pm - ep["init_t"].unsqueeze(1)) @ ep["init_R"]
    pe1, pe2 = f.PE(p1_), f.PE(dense_po)
    t.mark("PE x2")
    g1 = torch.cat([f.bg_token.expand(B, -1, -1).to(pe1.dtype), f.in_proj(dense_fm).to(pe1.dtype) + pe1], 1)
    g2 = torch.cat([f.bg_token.expand(B, -1, -1).to(pe1.dtype), f.in_proj(dense_fo).to(pe1.dtype) + pe2], 1)
    for blk in f.transformers:
        g1, g2 = blk(g1, geo_m, idx_m, g2, geo_o, idx_o)
    t.mark("fine transformers")
    scores = f.score_heads[f.nblock - 1](torch.cat((g1, g2), 1))
    atten = ops.feature_similarity(f.out_proj(g1), f.out_proj(g2), f.cfg.temp)
    t.mark("fine similarity")
    from unopose_amd.model.unopose import _scores
    score = _scores(scores, dense_pm.shape[1])
    ops.fine_pose(atten, score, dense_pm, dense_po)
    t.mark("fine pose head")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda")
    model = trained_like_(UNOPose(default_model_cfg())).to(dev).eval()
    ep, _, _ = make_batch(B, device=dev)
    ep["coarse_rand"] = torch.rand(B, 18000, device=dev)
    tot = {}
    for it in range(4):
        t = T()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            run(model, dict(ep), t)
        r = t.report()
        if it >= 1:
            for k, v in r.items():
                tot[k] = tot.get(k, 0) + v / 3
    for k, v in tot.items():
        print(f"{v:8.2f} ms  {k}")


if __name__ == "__main__":
    main()
