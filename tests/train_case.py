"""Deterministic inputs of the training-path fixtures (shared by tests/golden/make_train_golden.py, which runs the
REFERENCE on them, and the tests)."""
import torch

from unopose_amd.synthetic import congruent_pair

GRAD_KEYS = ["feature_extraction.rgb_net.output_upscaling.weight", "geo_embedding.proj_a.weight",
             "coarse_point_matching.transformers.0.layers.0.attention.attention.proj_p.weight",
             "coarse_point_matching.score_heads.1.weight", "fine_point_matching.PE.mlp2.layer1.conv.weight",
             "fine_point_matching.PE.mlp1.layer0.normlayer.bn.weight", "fine_point_matching.in_proj.bias",
             "fine_point_matching.transformers.2.dense_layer.attention.attention.scale",
             "fine_point_matching.transformers.1.sparse_layer.layers.1.output.squeeze.weight", "fine_point_matching.out_proj.weight"]


def random_block_outputs(g, B=3, n1=60, n2=50, nblock=3):
    """Random per-block matcher outputs + a pose under which about half the points have a partner."""
    p2 = torch.rand(B, n2, 3, generator=g) - 0.5
    Q = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    Q = Q * torch.sign(torch.det(Q)).reshape(-1, 1, 1)
    t = 0.2 * torch.randn(B, 3, generator=g)
    sel = torch.randint(0, n2, (B, n1), generator=g)
    p1 = torch.gather(p2, 1, sel.unsqueeze(2).expand(-1, -1, 3)) @ Q.transpose(1, 2) + t.unsqueeze(1)
    p1 = p1 + 0.25 * torch.randn(B, n1, 3, generator=g) * (torch.rand(B, n1, 1, generator=g) < 0.5)
    return dict(atten=[3 * torch.randn(B, n1 + 1, n2 + 1, generator=g) for _ in range(nblock)],
                score=[torch.rand(B, n1 + n2, generator=g) * 0.98 + 0.01 for _ in range(nblock)],
                sal=[torch.rand(B, n1 + n2, generator=g) * 0.98 + 0.01 for _ in range(nblock)], p1=p1, p2=p2, R=Q, t=t)


def make_train_batch(B=2, nq=512, nt=1200, seed=31):
    """(batch with rotation_label / translation_label, (aug_R, aug_t in radius-normalised units)): congruent pairs whose
    ground truth is the label; the pose noise the coarse stage would draw is fixed here and injected on both sides."""
    g = torch.Generator().manual_seed(seed)
    eps, Rs, ts = [], [], []
    for _ in range(B):
        ep, R, t = congruent_pair(g, nq, nt, 224, 5e-4)
        eps.append(ep)
        Rs.append(R)
        ts.append(t)
    batch = {k: torch.cat([e[k] for e in eps], 0) for k in eps[0]}
    batch["rotation_label"], batch["translation_label"] = torch.stack(Rs), torch.stack(ts)
    tem = batch["tem1_pts"]
    radius = torch.norm(tem - tem.mean(1, keepdim=True), dim=2).max(1)[0]
    ang = torch.tensor([[0.05, -0.03, 0.02], [-0.04, 0.06, 0.01]])[:B]
    c, s = torch.cos(ang), torch.sin(ang)
    one, zero = torch.ones(B), torch.zeros(B)
    m = lambda rows: torch.stack([torch.stack(r, 1) for r in rows], 1)  # noqa: E731
    noise = m([[c[:, 0], -s[:, 0], zero], [s[:, 0], c[:, 0], zero], [zero, zero, one]]) @ \
        m([[one, zero, zero], [zero, c[:, 1], -s[:, 1]], [zero, s[:, 1], c[:, 1]]])
    aug_R = batch["rotation_label"] @ noise
    aug_t = batch["translation_label"] / (radius.reshape(-1, 1) + 1e-6) + torch.tensor([[0.05, -0.02, 0.03], [-0.03, 0.04, 0.02]])[:B]
    return batch, (aug_R, aug_t)
