"""One-off diagnostics for the two round-2 parity questions (run on the GPU box)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import unopose_ref as R
from oracle.pointnet2_oracle import ext as oext
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
from test_geom_gpu import _well_conditioned

torch.set_grad_enabled(False)
GOLD = os.path.join(ROOT, "tests", "golden")
def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}

cfg = R.default_cfg()
sd = R.random_state_dict(cfg, seed=0)
m = UNOPose(default_model_cfg()); m.load_state_dict(sd, strict=True); m = m.cuda().eval()
z = load("geo_embedding_n197")
pts = z["points"]
out = ops.geo_embedding(pts.cuda(), m.geo_embedding).cpu()
full = R.geo_embedding(pts, sd, "geo_embedding", cfg.geo_embedding)
e = (out - full).abs().amax(-1)
eye = torch.eye(197, dtype=torch.bool)
e[:, eye] = 0
d_idx, a_idx = R.geo_embedding_indices(pts, cfg.geo_embedding)
dist = torch.sqrt(R.pairwise_distance(pts, pts))
knn = dist.topk(4, dim=2, largest=False)[1][:, :, 1:]
print("geo: max offdiag err", e.max().item(), "count > 1e-4:", int((e > 1e-4).sum()), "of", e.numel())
flat = e.flatten().topk(12)
for v, ix in zip(flat.values.tolist(), flat.indices.tolist()):
    b, r = divmod(ix, 197 * 197); i, j = divmod(r, 197)
    dd = dist[b, i]
    srt = dd.sort()
    print(f"  err {v:.2e} b={b} i={i} j={j} d={dist[b,i,j].item():.5f} a_idx={a_idx[b,i,j].tolist()} knn={knn[b,i].tolist()} "
          f"4 smallest d: {srt.values[:5].tolist()}")
# composite on the GPU for the same points
comp = ops.geo_embedding_torch(pts.cuda(), m.geo_embedding).cpu()
ec = (comp - full).abs().amax(-1); ec[:, eye] = 0
print("geo: torch-GPU composite vs oracle max offdiag", ec.max().item(), "; HIP vs composite", ((out - comp).abs().amax(-1))[:, ~eye].max().item())

# ---- fine matcher intermediates
sdt = R.random_state_dict(cfg, seed=0, tame=0.1)
m2 = UNOPose(default_model_cfg(fine_npoint=1024)); m2.load_state_dict(sdt, strict=True); m2 = m2.cuda().eval()
z = {k: v.cuda() if torch.is_tensor(v) else v for k, v in load("fine_matcher").items()}
g1, g2 = m2.geo_embedding(z["lrf1"]), m2.geo_embedding(z["lrf2"])
fm = m2.fine_point_matching
fm.taps = {}
fm(z["p1"], z["f1"], g1, z["i1"], z["p2"], z["f2"], g2, z["i2"], z["radius"], {"init_R": z["init_R"], "init_t": z["init_t"]})
t = fm.taps
p1_ = ((z["p1"] - z["init_t"].unsqueeze(1)) @ z["init_R"]).cpu()
for name, p, fo, ref in (("f1", p1_, t["f1"], z["f1_out"]), ("f2", z["p2"].cpu(), t["f2"], z["f2_out"])):
    well = torch.ones(1, p.shape[1], dtype=torch.bool)
    for r, ns in ((0.1, 64), (0.2, 256)):
        well &= _well_conditioned(R.query_and_lrf_group(p.contiguous(), r, ns, oext), r)
    err = (fo[:, :64] - ref).abs().amax(-1).cpu()[0]  # token 0 = bg, tokens 1.. = points 0..
    w = torch.cat([torch.ones(1, dtype=torch.bool), well[0, :63]])
    print(name, "err tokens: max well %.2e, max ill %.2e, n_ill %d, median %.2e" % (err[w].max().item(), err[~w].max().item() if (~w).any() else 0, int((~w).sum()), err.median().item()),
          "; well fraction of the cloud %.3f" % well.float().mean().item())
    print("   worst tokens:", err.topk(6).indices.tolist(), [round(v, 5) for v in err.topk(6).values.tolist()], "ill:", (~w).nonzero().flatten().tolist())
print("score err", (t["score"] - z["score"]).abs().max().item())
print("rowmax err", (t["atten"].max(2)[0] - z["atten_rowmax"]).abs().max().item(), "colmax", (t["atten"].max(1)[0] - z["atten_colmax"]).abs().max().item())
