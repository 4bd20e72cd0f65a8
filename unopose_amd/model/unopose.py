"""UNOPose for MI355X: same constructor / ``forward(end_points)`` / state_dict contract as
core/unopose/model/oneref_grf_predator_pose_estimation_model.py:11-76 (M), with
C = oneref_predator_coarse_point_matching.py, Fi = oneref_predator_fine_point_matching.py,
F = oneref_feature_extraction.py, U = utils/model_utils.py.  `.eval()`: the fused inference path.  `.train()`: the
reference's training branches (per-block overlap / saliency / correspondence losses, C:78-107, Fi:101-117) on
autograd-recorded composites (ops.differentiable), the frozen backbone still on the fused kernels."""

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..losses import aug_pose_noise, overlap_losses
from .config import to_cfg
from .modules import (GeometricStructureEmbedding, GeometricTransformer, PositionalEncoding,
                      SparseToDenseTransformer, ViTEncoderOneRef, _adjacent)


STACKED_FINE = True  # (module attributes: A/Bs set them from the script) the 2B-stacked fine matcher
# Reference-cloud PE on a side stream under the coarse stage.  (Switched off for part of round 3 while the PE / frame kernels were not
# reproducible beside the token attention of another stream; the cause -- packed-fp32 instructions, DESIGN.md section 7 -- is gone.)
PE_UNDER_COARSE = True
GEOM_UNDER_VIT = 1  # 1: FPS-196 / gathers, 2: + frames + embedding
LRF_UNDER_VIT = True  # the two global frames on the side stream as well
_SIDE_STREAMS = {}  # (device index, launch stream) -> its helper stream, for the life of the process
COARSE_SLOT = True  # round 6: the coarse matcher's inputs gathered into one (2B, 1 + n, C) tensor with a slot for the background token (no concatenations); False: round 5's form
TRAIN_PE_UNDER_COARSE = True  # training: both clouds' PE groups (forward and backward) on the side stream underneath the coarse stage


def _scores(scores, n1, halves=False):
    """C:68-76 / Fi:91-99 -- only `score` is consumed at eval time.  `halves`: the head's output over both clouds as one batch of 2B."""
    return ops.overlap_scores(scores, n1, halves=halves)


def _bg_row(mod, dtype):
    """The background token of a matcher as ONE contiguous row of `dtype` (cached per parameter version: no cast launch per forward)."""
    p = mod.bg_token
    key = (p._version, p.data_ptr(), dtype)
    c = getattr(mod, "_bg_cache", None)
    if c is None or c[0] != key:
        with torch.no_grad():
            c = (key, p.detach().reshape(-1).to(dtype).contiguous())
        mod._bg_cache = c
    return c[1]


def _block_outputs(out_proj, score_head, f1, f2, n1, temp):
    """What the training branch keeps per transformer block (C:62-76, Fi:85-99): the similarity of the projected
    features, the overlap scores and the saliency = overlap scores propagated through the row / column softmax."""
    scores = score_head(torch.cat((f1, f2), dim=1))
    atten = ops.feature_similarity(ops.linear(f1, out_proj), ops.linear(f2, out_proj), temp)
    s1, s2 = scores[:, 1:(n1 + 1)], scores[:, (n1 + 2):]
    m1, m2 = ops.saliency_pair(atten, s1, s2)  # softmax(atten[:, 1:, 1:]) @ s2 and its transposed twin (csrc/saliency_train.hip under train())
    score = torch.clamp(torch.sigmoid(torch.cat((s1, s2), dim=1).squeeze(-1)), min=0, max=1)
    saliency = torch.clamp(torch.sigmoid(torch.cat((m1, m2), dim=1).squeeze(-1)), min=0, max=1)
    return atten, score, saliency


class CoarsePointMatchingOneRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.nblock = cfg.nblock
        self.in_proj = nn.Linear(cfg.input_dim, cfg.hidden_dim)
        self.out_proj = nn.Linear(cfg.hidden_dim, cfg.out_dim)
        self.bg_token = nn.Parameter(torch.randn(1, 1, cfg.hidden_dim) * 0.02)
        self.score_heads = nn.ModuleList([nn.Linear(cfg.hidden_dim, 1) for _ in range(self.nblock)])
        self.transformers = nn.ModuleList([GeometricTransformer(cfg.hidden_dim, 4) for _ in range(self.nblock)])
        self.taps = None  # assign a dict to receive (f1, f2, atten, score) of the next forward (tests)

    def forward_train(self, p1, f1, geo1, p2, f2, geo2, radius, end_points):
        """C:46-107, training branch: every block contributes (similarity, score, saliency) to the losses; the fine stage
        is then started from the ground-truth pose perturbed by `aug_pose_noise` (or by end_points["aug_pose"] =
        (R, t) when the caller injects the draw, as the parity tests do)."""
        B, n1 = f1.shape[:2]
        bg = self.bg_token.expand(B, -1, -1)
        f1 = torch.cat([bg, ops.linear(f1, self.in_proj)], dim=1)
        f2 = torch.cat([bg, ops.linear(f2, self.in_proj)], dim=1)
        attens, scores, sals = [], [], []
        for blk, head in zip(self.transformers, self.score_heads):
            f1, f2 = blk(f1, geo1, f2, geo2)
            a, sc, sa = _block_outputs(self.out_proj, head, f1, f2, n1, self.cfg.temp)
            attens.append(a)
            scores.append(sc)
            sals.append(sa)
        gt_R = end_points["rotation_label"]
        gt_t = end_points["translation_label"] / (radius.reshape(-1, 1) + 1e-6)
        init_R, init_t = end_points["aug_pose"] if "aug_pose" in end_points else aug_pose_noise(gt_R, gt_t)
        overlap_losses(end_points, attens, scores, sals, p1, p2, gt_R, gt_t, self.cfg.loss_predator_thres,
                       self.cfg.loss_dis_thres, "coarse_hard")
        end_points["init_R"], end_points["init_t"] = init_R, init_t
        return end_points

    def forward(self, p1, f1, geo1, p2, f2, geo2, radius, end_points, slot=None):
        if self.training:
            return self.forward_train(p1, f1, geo1, p2, f2, geo2, radius, end_points)
        B, n1 = f1.shape[:2]
        if slot is not None and f1.is_cuda and torch.is_autocast_enabled():
            # `slot` (2B, 1 + n, C_in): both clouds' sparse features behind an unused row per pair (UNOPose._forward_from gathers them that way).
            # in_proj runs over all rows as ONE GEMM, the background token is then written into row 0 of every pair: no concatenation, and the
            # halves of f are what the blocks' stacked self layers expect
            f = ops.set_first_rows_(ops.linear(slot, self.in_proj), _bg_row(self, torch.bfloat16))
            f1, f2 = f[:B], f[B:]
        else:
            f1 = ops.linear(f1, self.in_proj)
            f2 = ops.linear(f2, self.in_proj)
            bg = self.bg_token.expand(B, -1, -1).to(f1.dtype)
            f1 = torch.cat([bg, f1], dim=1)
            f2 = torch.cat([bg, f2], dim=1)
        for blk in self.transformers:
            f1, f2 = blk(f1, geo1, f2, geo2)
        f = _adjacent(f1, f2)  # (the blocks' cross layers wrote the halves of one tensor)
        if f is not None and f1.is_cuda:
            score = _scores(ops.score_head(f, self.score_heads[self.nblock - 1]), n1, halves=True)
            o = ops.linear(f, self.out_proj)
            atten = ops.feature_similarity(o[:B], o[B:], self.cfg.temp)
        else:
            scores = ops.score_head(torch.cat((f1, f2), dim=1), self.score_heads[self.nblock - 1])
            atten = ops.feature_similarity(ops.linear(f1, self.out_proj), ops.linear(f2, self.out_proj), self.cfg.temp)
            score = _scores(scores, n1)
        if self.taps is not None:  # test probe: the tensors the reference's eval branch hands to the pose head
            self.taps.update(f1=f1, f2=f2, atten=atten, score=score)
        n1p, n2p = self.cfg.nproposal1, self.cfg.nproposal2
        rand = end_points.get("coarse_rand")
        if rand is None:  # the reference draws inside forward (U:462)
            rand = torch.rand(B, n1p * 3, device=p1.device)
        init_R, init_t, init_score = ops.coarse_pose(atten, score, p1, p2, rand, n1p, n2p)
        end_points["init_pose_score"] = init_score
        end_points["init_R"] = init_R
        end_points["init_t"] = init_t
        return end_points


class FinePointMatchingOneRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.nblock = cfg.nblock
        d = cfg.hidden_dim
        self.in_proj = nn.Linear(cfg.input_dim, d)
        self.out_proj = nn.Linear(d, cfg.out_dim)
        self.dis_proj = nn.Linear(2 * d, 3)  # present in released checkpoints, unused by forward
        self.bg_token = nn.Parameter(torch.randn(1, 1, d) * 0.02)
        assert cfg.use_lrf and cfg.use_xyz and not cfg.get("use_feature", False), "only the configured PE variant"
        self.PE = PositionalEncoding(d, cfg.pe_radius1, cfg.pe_radius2, cfg.get("nsample1", 32), cfg.get("nsample2", 64))
        self.score_heads = nn.ModuleList([nn.Linear(d, 1) for _ in range(self.nblock)])
        self.transformers = nn.ModuleList(
            [SparseToDenseTransformer(d, 4, cfg.focusing_factor) for _ in range(self.nblock)])
        self.taps = None  # assign a dict to receive (f1, f2, atten, score) of the next forward (tests)

    def forward_train(self, p1, f1, geo1, fps_idx1, p2, f2, geo2, fps_idx2, radius, end_points, pe_ahead=None):
        """Fi:58-117, training branch.  The positional encoding is evaluated per cloud, query first (its BatchNorm layers
        use -- and update -- batch statistics per call, exactly as the reference's two PE calls do).  `pe_ahead` = (grouped features of the
        query cloud in the start pose, of the reference cloud, the stream they were computed on): `UNOPose.forward_train` evaluates
        them underneath the coarse stage (they depend on the points and the injected start pose alone); only the projection runs here."""
        B, n1 = p1.shape[:2]
        bg = self.bg_token.expand(B, -1, -1)
        if pe_ahead is None:
            p1_ = (p1 - end_points["init_t"].unsqueeze(1)) @ end_points["init_R"] if "init_R" in end_points else p1
            pe1, pe2 = self.PE(p1_), self.PE(p2)
        else:
            g1, g2, side = pe_ahead
            torch.cuda.current_stream().wait_stream(side)
            pe1, pe2 = self.PE.project(g1), self.PE.project(g2)
        f1 = torch.cat([bg, ops.linear(f1, self.in_proj) + pe1], dim=1)
        f2 = torch.cat([bg, ops.linear(f2, self.in_proj) + pe2], dim=1)
        attens, scores, sals = [], [], []
        for blk, head in zip(self.transformers, self.score_heads):
            f1, f2 = blk(f1, geo1, fps_idx1, f2, geo2, fps_idx2)
            a, sc, sa = _block_outputs(self.out_proj, head, f1, f2, n1, self.cfg.temp)
            attens.append(a)
            scores.append(sc)
            sals.append(sa)
        gt_R = end_points["rotation_label"]
        gt_t = end_points["translation_label"] / (radius.reshape(-1, 1) + 1e-6)
        overlap_losses(end_points, attens, scores, sals, p1, p2, gt_R, gt_t, self.cfg.loss_predator_thres,
                       self.cfg.loss_dis_thres, "fine")
        return end_points

    def forward(self, p1, f1, geo1, fps_idx1, p2, f2, geo2, fps_idx2, radius, end_points, pe2_groups=None):
        if self.training:
            return self.forward_train(p1, f1, geo1, fps_idx1, p2, f2, geo2, fps_idx2, radius, end_points, pe_ahead=pe2_groups)
        B, n1 = p1.shape[:2]
        if "init_R" in end_points and "init_t" in end_points:
            p1_ = ops.rigid_rows(p1, end_points["init_t"], end_points["init_R"])
        else:
            p1_ = p1
        e_all, f_all = _adjacent(geo1, geo2), _adjacent(f1, f2)
        if (STACKED_FINE and p1_.shape == p2.shape and p1.is_cuda and e_all is not None
                and fps_idx1.shape == fps_idx2.shape):
            # both clouds as ONE batch of 2B through PE, in_proj and the three blocks; the background token
            # rides beside the dense features and is put in front only once, at the end
            # pe2_groups: the reference cloud's grouped features (PE before mlp3) when UNOPose.forward already
            # computed them under the coarse stage; mlp3 then runs HERE, on the main stream, as one 2B GEMM
            d = ops.linear(f_all if f_all is not None else torch.cat([f1, f2], 0), self.in_proj)
            if self.PE.split_ok(p1_) and d.dtype == torch.bfloat16 and (pe2_groups is None or isinstance(pe2_groups, tuple)):
                # both scales of both clouds land in ONE split-layout buffer (the reference cloud's half possibly filled earlier, under
                # the coarse stage); mlp3 runs on csrc/gemm_f32.hip with "+ d" in its epilogue
                if pe2_groups is None:
                    buf = self.PE.groups_split(torch.cat([p1_, p2], 0), torch.empty(2 * B, p2.shape[1], 512, dtype=torch.bfloat16, device=p2.device), 0)
                else:
                    buf = self.PE.groups_split(p1_, pe2_groups[1], 0)
                d = self.PE.project_add(buf, d)
            else:
                pe = self.PE(torch.cat([p1_, p2], 0)) if pe2_groups is None else \
                    self.PE.project(torch.cat([self.PE.groups(p1_), pe2_groups], 0))
                d = d + pe.to(d.dtype)
            bg = _bg_row(self, d.dtype).reshape(1, 1, -1).expand(2 * B, -1, -1)  # (one cached row for every pair: the first block's gather reads it with row distance 0)
            idx_all = torch.cat([fps_idx1, fps_idx2], 0)
            for blk in self.transformers:
                d, bg = blk.forward_stacked(d, bg, e_all, idx_all)
            f = torch.cat([bg, d], dim=1)
            f1, f2 = f[:B], f[B:]
            sc = ops.score_head(f, self.score_heads[self.nblock - 1])
            o = ops.linear(f, self.out_proj)
            if self.taps is None and ops.fine_pose_fused_ok(o[:B], o[B:]):  # the similarity is never materialised
                R, t, s = ops.fine_pose_from_features(o[:B], o[B:], self.cfg.temp, _scores(sc, n1, halves=True), p1, p2)
                return self._finish(end_points, R, t, s, radius)
            scores = torch.cat((sc[:B], sc[B:]), dim=1)
            atten = ops.feature_similarity(o[:B], o[B:], self.cfg.temp)
        else:
            if p1_.shape == p2.shape:  # both clouds through the fused PE kernels as one batch of 2B
                pe = self.PE(torch.cat([p1_, p2], 0))
                pe1, pe2 = pe[:B], pe[B:]
            else:
                pe1, pe2 = self.PE(p1_), self.PE(p2)
            f1 = ops.linear(f1, self.in_proj) + pe1.to(f1.dtype)
            f2 = ops.linear(f2, self.in_proj) + pe2.to(f2.dtype)
            bg = self.bg_token.expand(B, -1, -1).to(f1.dtype)
            f1 = torch.cat([bg, f1], dim=1)
            f2 = torch.cat([bg, f2], dim=1)
            for blk in self.transformers:
                f1, f2 = blk(f1, geo1, fps_idx1, f2, geo2, fps_idx2)
            scores = ops.score_head(torch.cat((f1, f2), dim=1), self.score_heads[self.nblock - 1])
            o1, o2 = ops.linear(f1, self.out_proj), ops.linear(f2, self.out_proj)
            if self.taps is None and ops.fine_pose_fused_ok(o1, o2):
                R, t, s = ops.fine_pose_from_features(o1, o2, self.cfg.temp, _scores(scores, n1), p1, p2)
                return self._finish(end_points, R, t, s, radius)
            atten = ops.feature_similarity(o1, o2, self.cfg.temp)
        score = _scores(scores, n1)
        if self.taps is not None:
            self.taps.update(f1=f1, f2=f2, atten=atten, score=score)
        R, t, s = ops.fine_pose(atten, score, p1, p2)
        return self._finish(end_points, R, t, s, radius)

    @staticmethod
    def _finish(end_points, R, t, s, radius):
        end_points["pred_R"] = R
        end_points["pred_t"] = ops.scale_by_radius(t, radius, multiply=True)
        end_points["pred_pose_score"] = s
        return end_points


class UNOPose(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        cfg = to_cfg(cfg)
        self.cfg = cfg
        self.coarse_npoint = cfg.coarse_npoint
        self.fine_npoint = cfg.fine_npoint
        self.use_ref_rad = cfg.get("use_ref_rad", False)
        self.test_coarse_only = cfg.get("test_coarse_only", False)
        self.feature_extraction = ViTEncoderOneRef(cfg.feature_extraction, self.fine_npoint)
        self.geo_embedding = GeometricStructureEmbedding(cfg.geo_embedding)
        self.coarse_point_matching = CoarsePointMatchingOneRef(cfg.coarse_point_matching)
        self.fine_point_matching = FinePointMatchingOneRef(cfg.fine_point_matching)
        self.taps = None  # assign a dict to receive the sampling intermediates of the next forward (tests)
        self.fixed_init = None  # assign (init_R, init_t) to start the fine stage of the next forwards from that coarse pose (tests)
        self._zero_rows = {}  # (C, dtype, device) -> one zero row (the unused slot row of the coarse matcher's stacked input)
        # side-stream overlaps INSIDE one forward (geometry under the ViT, reference-cloud PE under the coarse stage): +3 % when
        # forwards run one at a time; pipeline.PipelinedForward switches them off (another forward fills those gaps better)
        self.internal_overlap = True

    # ---- F:245-298 -------------------------------------------------------------------------------
    def _features(self, end_points):
        rgb, choose = end_points["rgb"], end_points["rgb_choose"]
        assert choose.size(1) == self.fine_npoint
        net = self.feature_extraction.rgb_net
        dense_pm = end_points["pts"]
        if "dense_po" in end_points and "dense_fo" in end_points:  # precomputed reference (F:252-263)
            dense_fm = net.pixel_features(rgb, choose)
            dense_po, dense_fo = end_points["dense_po"].clone(), end_points["dense_fo"].clone()
            radius = ops.cloud_radius(dense_po)
            dense_pm = ops.scale_by_radius(dense_pm, radius)
            dense_po = ops.scale_by_radius(dense_po, radius)
            return dense_pm, dense_fm, dense_po, dense_fo, radius, None
        if "ref_dense_po" in end_points:  # encode_reference() output: same numbers as the full path below
            radius = end_points["ref_radius"]
            dense_pm = ops.scale_by_radius(dense_pm, radius)
            return dense_pm, net.pixel_features(rgb, choose), end_points["ref_dense_po"], end_points["ref_dense_fo"], radius, None
        tem_rgb, tem_choose, tem_pts = end_points["tem1_rgb"], end_points["tem1_choose"], end_points["tem1_pts"]
        radius = ops.cloud_radius(tem_pts)
        dense_pm = ops.scale_by_radius(dense_pm, radius)
        tem_n = ops.scale_by_radius(tem_pts, radius)
        # The serial geometry chain (FPS 5000->2048: 2047 dependent iterations on B CUs) only needs the
        # points, so it runs on a side HIP stream underneath the ViT GEMMs of the main stream.
        main = torch.cuda.current_stream()
        side = self._side_stream(tem_n.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            idx_o = ops.furthest_point_sample(tem_n, self.fine_npoint)
            dense_po = ops.gather_rows(tem_n, idx_o)
            sparse_up = (ops.sparse_upproj_ok(rgb) and idx_o.shape == choose.shape and net.out_dim == 256 and rgb.shape == tem_rgb.shape
                         and tem_choose.dtype == choose.dtype and tem_choose.is_contiguous())
            if sparse_up:  # query | reference pixel indices in ONE tensor: the reference half is written by the gather itself (8-byte rows)
                both_choose = torch.empty(2 * choose.shape[0], choose.shape[1], dtype=choose.dtype, device=choose.device)
                both_choose[:choose.shape[0]].copy_(choose)
                sel_choose = ops.gather_rows(tem_choose.unsqueeze(-1), idx_o, out=both_choose[choose.shape[0]:].unsqueeze(-1)).squeeze(-1)
            else:
                sel_choose = torch.gather(tem_choose, 1, idx_o.long())
            for t in (idx_o, dense_po, sel_choose):
                t.record_stream(main)
            plan = None
            if sparse_up:
                # which cells of the up-projected map the 2 x 2048 chosen pixels read: known before the ViT has run
                sd = rgb.shape[-1] // 14
                npre = net.vit.cls_token.shape[1] + net.vit.reg_token.shape[1]
                both_choose.record_stream(main)
                plan = ops.upproj_plan(both_choose, rgb.shape[-2], rgb.shape[-1], sd, npre, npre + sd * sd)
                for t in plan.values():
                    if torch.is_tensor(t):
                        t.record_stream(main)
            pre = None
            if GEOM_UNDER_VIT and self.internal_overlap:
                # the other latency-bound, feature-independent steps ride along: both global LRFs, the two
                # 2048->196 FPS chains and the gathers of points / frame coordinates (M:28-47)
                pre = {}
                pre["idx_m"] = ops.furthest_point_sample(dense_pm, self.coarse_npoint)
                pre["idx_o"] = ops.furthest_point_sample(dense_po, self.coarse_npoint)
                pre["sparse_pm"] = ops.gather_rows(dense_pm.float(), pre["idx_m"])
                pre["sparse_po"] = ops.gather_rows(dense_po.float(), pre["idx_o"])
                # (the frame kernels used to change results here, beside the previous batch's token attention in a pipelined runner:
                # packed-fp32 instructions next to another kernel's MFMAs, DESIGN.md section 7 -- the library is built without them now)
                if GEOM_UNDER_VIT > 1 or LRF_UNDER_VIT:
                    pre["sparse_pm_lrf"] = ops.gather_rows(ops.lrf_global(end_points["pts"], self.use_ref_rad), pre["idx_m"])
                    pre["sparse_po_lrf"] = ops.gather_rows(ops.lrf_global(tem_pts, self.use_ref_rad), pre["idx_o"])  # NB App-E.1: full 5000-point cloud
                if GEOM_UNDER_VIT > 1:
                    bg_point = torch.ones(dense_pm.size(0), 1, 3, device=dense_pm.device)
                    pre["geo"] = self.geo_embedding(torch.cat([torch.cat([bg_point, pre["sparse_pm_lrf"]], dim=1),
                                                               torch.cat([bg_point, pre["sparse_po_lrf"]], dim=1)], dim=0))
                for t in pre.values():
                    t.record_stream(main)
        # both crops through the ViT as ONE batch of 2B images
        B = rgb.shape[0]
        if plan is not None:
            acts = net.vit((rgb, tem_rgb), taps_side_by_side=True)
            if torch.is_tensor(acts) and acts.shape[1] == plan["tok_stride"]:
                main.wait_stream(side)
                both = ops.sparse_pixel_features(acts, net.output_upscaling, plan)  # query | reference, stacked
                return dense_pm, both[:B], dense_po, both[B:], radius, pre
            z, (H, W), off = net.upproject(acts, rgb.shape[-2], rgb.shape[-1])
        else:
            z, (H, W), off = net.upproject(net.vit((rgb, tem_rgb), taps_side_by_side=True), rgb.shape[-2], rgb.shape[-1])
        # query | reference features land in ONE (2B,N,256) buffer: the fine matcher takes them stacked
        both = torch.empty(2 * B, choose.shape[1], 256, dtype=torch.float32, device=z.device) \
            if sel_choose.shape == choose.shape else None
        dense_fm = ops.bilinear_sample_native(z[:B], choose, H, W, out=None if both is None else both[:B], tok_offset=off)
        main.wait_stream(side)
        # only the FPS-selected reference pixels are ever interpolated (gather commutes with sampling)
        dense_fo = ops.bilinear_sample_native(z[B:], sel_choose, H, W, out=None if both is None else both[B:], tok_offset=off)
        return dense_pm, dense_fm, dense_po, dense_fo, radius, pre

    @torch.no_grad()
    def encode_reference(self, tem1_rgb, tem1_choose, tem1_pts):
        """Everything `forward` derives from the reference view ALONE: the FPS-2048 subset
        of the radius-normalised cloud, its pixel features, the radius, and the LRF-frame coordinates of
        the FULL cloud (M:30).  Feeding the returned dict back through `end_points` skips the reference
        crop's ViT pass, the 5000->2048 FPS and the reference LRF; unlike the reference's own
        `dense_po`/`dense_fo` shortcut (F:252-263, which re-derives the radius from the subset) the
        results equal the uncached forward."""
        radius = ops.cloud_radius(tem1_pts)
        tem_n = ops.scale_by_radius(tem1_pts, radius)
        idx_o = ops.furthest_point_sample(tem_n, self.fine_npoint)
        sel_choose = torch.gather(tem1_choose, 1, idx_o.long())
        net = self.feature_extraction.rgb_net
        return dict(ref_dense_po=ops.gather_rows(tem_n, idx_o), ref_dense_fo=net.pixel_features(tem1_rgb, sel_choose),
                    ref_radius=radius, ref_lrf=ops.lrf_global(tem1_pts, self.use_ref_rad))

    def _side_stream(self, device):
        """The helper stream paired with the CURRENT stream (one per launch stream, so several forwards in flight on
        different streams do not funnel their side work through one queue).  Process-wide, not per model: streams map onto a few
        hardware queues for good, and every model a process builds must reuse them (pipeline._STREAM_POOL)."""
        cur = torch.cuda.current_stream(device)
        key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), cur.cuda_stream)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        return _SIDE_STREAMS[key]

    def _sample_wlrf(self, pts, pts_lrf, feats, npoint):
        """U:156-177 (gathers done in (B,N,C) layout)."""
        idx = ops.furthest_point_sample(pts, npoint)
        return ops.gather_rows(pts.float(), idx), ops.gather_rows(pts_lrf, idx), ops.gather_rows(feats, idx), idx

    # ---- training (M:25-76 with self.training; F:245-298) ----------------------------------------------------------
    def _lowres_train(self, rgb):
        """(B,3,S,S) -> the up-projected low-resolution map (B, 4S/14, 4S/14, 256) with the gradient path of the reference: the backbone's
        taps (under no_grad and on the fused kernels when it is frozen, F:194-198) -> trainable up-projection."""
        net = self.feature_extraction.rgb_net
        frozen = not any(p.requires_grad for p in net.vit.parameters())
        if frozen:
            with torch.no_grad(), ops.differentiable(False):
                taps = net.vit(rgb)
        else:
            taps = net.vit(rgb)
        B, _, H, W = rgb.shape
        side = H // 14
        z = torch.cat([o[:, 5:, :] for o in taps], dim=2).float()
        z = ops.linear(z, net.output_upscaling).reshape(B, side, side, 4, 4, net.out_dim)
        return z.permute(0, 1, 3, 2, 4, 5).reshape(B, 4 * side, 4 * side, net.out_dim)

    def _pixel_features_train(self, rgb, choose):
        """(B,3,S,S), (B,Np) -> (B,Np,256): `_lowres_train` + the bilinear resize evaluated at the chosen pixels (= F.interpolate + gather,
        recorded by autograd)."""
        return ops.bilinear_sample_pixels(self._lowres_train(rgb), choose, rgb.shape[-2], rgb.shape[-1])

    def forward_train(self, end_points):
        tem_pts = end_points["tem1_pts"]
        radius = torch.norm(tem_pts - tem_pts.mean(1, keepdim=True), dim=2).max(1)[0]
        scale = radius.reshape(-1, 1, 1) + 1e-6
        dense_pm, tem_n = end_points["pts"] / scale, tem_pts / scale
        # The index-only geometry -- the serial 6144 -> 4096 FPS of the reference cloud (3.7 ms at configs[3]'s shape), both global frames,
        # both coarse FPS chains -- carries no gradient and needs the points alone: it runs on a side HIP stream underneath the backbone
        # passes and the up-projections of the main stream (as `_features` does in eval); the main stream joins it before the first use.
        main = torch.cuda.current_stream() if dense_pm.is_cuda else None
        side = self._side_stream(dense_pm.device) if main is not None and self.internal_overlap else None
        geo = {}

        def geometry():
            with torch.no_grad():
                geo["idx_o"] = ops.furthest_point_sample(tem_n, self.fine_npoint)
                geo["dense_po"] = ops.gather_rows(tem_n, geo["idx_o"])
                geo["sel_choose"] = torch.gather(end_points["tem1_choose"], 1, geo["idx_o"].long())
                geo["pm_lrf"] = ops.lrf_global(end_points["pts"], self.use_ref_rad)
                geo["po_lrf"] = ops.lrf_global(tem_pts, self.use_ref_rad)  # NB App-E.1: frames of the FULL cloud, gathered with subset indices
                geo["fps_m"] = ops.furthest_point_sample(dense_pm, self.coarse_npoint)
                geo["fps_o"] = ops.furthest_point_sample(geo["dense_po"], self.coarse_npoint)

        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                geometry()
                for t in geo.values():
                    t.record_stream(main)
            low_m = self._lowres_train(end_points["rgb"])
            low_o = self._lowres_train(end_points["tem1_rgb"])
            main.wait_stream(side)
        else:
            geometry()
            low_m = self._lowres_train(end_points["rgb"])
            low_o = self._lowres_train(end_points["tem1_rgb"])
        H, W = end_points["rgb"].shape[-2:]
        dense_po = geo["dense_po"]
        dense_fm = ops.bilinear_sample_pixels(low_m, end_points["rgb_choose"], H, W)
        # features of the FPS-selected reference pixels only (the reference gathers all of them and then selects: same values,
        # and only the selected ones receive gradient there too)
        dense_fo = ops.bilinear_sample_pixels(low_o, geo["sel_choose"], H, W)
        B = dense_pm.size(0)
        bg_point = torch.ones(B, 1, 3, device=dense_pm.device)
        fps_idx_m, fps_idx_o = geo["fps_m"], geo["fps_o"]
        # (`_sample_wlrf`, U:156-177, with its FPS taken from the side stream)
        sparse_pm, sparse_pm_lrf, sparse_fm = ops.gather_rows(dense_pm.float(), fps_idx_m), ops.gather_rows(geo["pm_lrf"], fps_idx_m), ops.gather_rows(dense_fm, fps_idx_m)
        sparse_po, sparse_po_lrf, sparse_fo = ops.gather_rows(dense_po.float(), fps_idx_o), ops.gather_rows(geo["po_lrf"], fps_idx_o), ops.gather_rows(dense_fo, fps_idx_o)
        # The fine matcher's positional encoding up to its max-pool (grouping + SharedMLP: HBM-bound streams over multi-GB activations, own
        # kernels only) needs the points and the fine stage's start pose -- which in training is the ground truth plus noise, not the
        # coarse network's output (C:46-107) -- so both clouds' halves run on the side stream underneath the latency-bound coarse stage,
        # forward and -- autograd replays every op on the stream of its forward -- backward.  Query cloud first: the BatchNorm layers
        # use and update batch statistics per call in the reference's order.
        pe_ahead = None
        if side is not None and TRAIN_PE_UNDER_COARSE:
            if "aug_pose" not in end_points:  # (the draw the coarse stage would make after its blocks: nothing else draws in between)
                end_points["aug_pose"] = aug_pose_noise(end_points["rotation_label"], end_points["translation_label"] / (radius.reshape(-1, 1) + 1e-6))
            init_R, init_t = end_points["aug_pose"]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                PE = self.fine_point_matching.PE
                g1 = PE.groups((dense_pm - init_t.unsqueeze(1)) @ init_R)
                g2 = PE.groups(dense_po)
            g1.record_stream(main)
            g2.record_stream(main)
            pe_ahead = (g1, g2, side)
        geo_m = self.geo_embedding(torch.cat([bg_point, sparse_pm_lrf], dim=1))
        geo_o = self.geo_embedding(torch.cat([bg_point, sparse_po_lrf], dim=1))
        end_points = self.coarse_point_matching(sparse_pm, sparse_fm, geo_m, sparse_po, sparse_fo, geo_o, radius, end_points)
        return self.fine_point_matching(dense_pm, dense_fm, geo_m, fps_idx_m, dense_po, dense_fo, geo_o, fps_idx_o, radius,
                                        end_points, pe2_groups=pe_ahead)

    def forward(self, end_points):
        if self.training:
            with ops.differentiable():
                return self.forward_train(end_points)
        return self.forward_matching(end_points, self.forward_features(end_points))

    def forward_features(self, end_points):
        """First half of the eval forward: the ViT over both crops, the 5000 -> 2048 FPS and the pixel features (a9, a10, a21).
        `pipeline.PipelinedForward` runs the two halves of consecutive batches on different HIP streams."""
        try:
            return self._features(end_points)
        finally:
            ops.clear_split_memo()

    def forward_matching(self, end_points, feats):
        """Second half: coarse + fine matching on what `forward_features` returned."""
        try:
            return self._matching_half(end_points, feats)
        finally:
            ops.clear_split_memo()

    def _matching_half(self, end_points, feats):
        dense_pm, dense_fm, dense_po, dense_fo, radius, pre = feats
        if pre is not None:
            return self._forward_from(pre, end_points, dense_pm, dense_fm, dense_po, dense_fo, radius)
        dense_pm_lrf = ops.lrf_global(end_points["pts"], self.use_ref_rad)
        # NB (App-E.1): LRF of the FULL tem1 cloud (5000 rows) gathered below with indices into the
        # FPS-2048 subset, exactly as the reference does (M:30, U:167-171).
        if "ref_lrf" in end_points:
            dense_po_lrf = end_points["ref_lrf"]
        else:
            dense_po_lrf = ops.lrf_global(end_points["tem1_pts"], self.use_ref_rad) if "tem1_pts" in end_points \
                else ops.lrf_global(end_points["dense_po"], self.use_ref_rad)
        B = dense_pm.size(0)
        bg_point = torch.ones(B, 1, 3, device=dense_pm.device)
        sparse_pm, sparse_pm_lrf, sparse_fm, fps_idx_m = self._sample_wlrf(dense_pm, dense_pm_lrf, dense_fm,
                                                                         self.coarse_npoint)
        sparse_po, sparse_po_lrf, sparse_fo, fps_idx_o = self._sample_wlrf(dense_po, dense_po_lrf, dense_fo,
                                                                         self.coarse_npoint)
        pre = dict(idx_m=fps_idx_m, idx_o=fps_idx_o, sparse_pm=sparse_pm, sparse_po=sparse_po,
                   sparse_pm_lrf=sparse_pm_lrf, sparse_po_lrf=sparse_po_lrf, sparse_fm=sparse_fm, sparse_fo=sparse_fo)
        return self._forward_from(pre, end_points, dense_pm, dense_fm, dense_po, dense_fo, radius)

    def _forward_from(self, pre, end_points, dense_pm, dense_fm, dense_po, dense_fo, radius):
        """Coarse + fine stages given the sparse subsets (computed inline or ahead on the side stream)."""
        B = dense_pm.size(0)
        bg_point = torch.ones(B, 1, 3, device=dense_pm.device)
        fps_idx_m, fps_idx_o = pre["idx_m"], pre["idx_o"]
        if self.taps is not None:  # test probe (the model never reads fps_idx_* from end_points)
            self.taps.update(fps_idx_m=fps_idx_m, fps_idx_o=fps_idx_o, dense_pm=dense_pm, dense_fm=dense_fm,
                             dense_po=dense_po, dense_fo=dense_fo, radius=radius)
        if "sparse_pm_lrf" not in pre:  # frames on the matcher's stream (see `_features`)
            po_lrf = end_points["ref_lrf"] if "ref_lrf" in end_points else ops.lrf_global(
                end_points["tem1_pts"] if "tem1_pts" in end_points else end_points["dense_po"], self.use_ref_rad)
            pre = dict(pre, sparse_pm_lrf=ops.gather_rows(ops.lrf_global(end_points["pts"], self.use_ref_rad), fps_idx_m),
                       sparse_po_lrf=ops.gather_rows(po_lrf, fps_idx_o))
        sparse_pm, sparse_po, sparse_pm_lrf, sparse_po_lrf = (pre[k] for k in ("sparse_pm", "sparse_po", "sparse_pm_lrf",
                                                                                  "sparse_po_lrf"))
        slot = None
        if ("sparse_fm" not in pre and "sparse_fo" not in pre and COARSE_SLOT and dense_fm.is_cuda and torch.is_autocast_enabled() and not ops.is_differentiable()
                and dense_fm.dtype == dense_fo.dtype and dense_fm.shape[2:] == dense_fo.shape[2:] and fps_idx_m.shape == fps_idx_o.shape
                and dense_fm.is_contiguous() and dense_fo.is_contiguous()):
            # both clouds' sparse features gathered into ONE (2B, 1 + n, C) tensor behind an unused (zero) row per pair: the coarse matcher's
            # in_proj then runs as one GEMM and puts its background token into that row (CoarsePointMatchingOneRef.forward, `slot`)
            n, C = fps_idx_m.shape[1], dense_fm.shape[2]
            slot = torch.empty(2 * B, n + 1, C, dtype=dense_fm.dtype, device=dense_fm.device)
            zr = self._zero_row(C, dense_fm.dtype, dense_fm.device).expand(B, C)
            ops.gather_rows(dense_fm, fps_idx_m, alt=zr, prepend=True, out=slot[:B])
            ops.gather_rows(dense_fo, fps_idx_o, alt=zr, prepend=True, out=slot[B:])
            sparse_fm, sparse_fo = slot[:B, 1:], slot[B:, 1:]
        else:
            sparse_fm = pre["sparse_fm"] if "sparse_fm" in pre else ops.gather_rows(dense_fm, fps_idx_m)
            sparse_fo = pre["sparse_fo"] if "sparse_fo" in pre else ops.gather_rows(dense_fo, fps_idx_o)
        if "geo" in pre:
            geo = pre["geo"]
        else:
            geo = self._geo(bg_point, sparse_pm_lrf, sparse_po_lrf)
        return self._matching(end_points, geo, B, sparse_pm, sparse_fm, sparse_po, sparse_fo, fps_idx_m, fps_idx_o,
                              dense_pm, dense_fm, dense_po, dense_fo, radius, slot=slot)

    def _zero_row(self, C, dtype, device):
        key = (C, dtype, str(device))
        z = self._zero_rows.get(key)
        if z is None:
            z = self._zero_rows[key] = torch.zeros(C, dtype=dtype, device=device)
        return z

    def _geo(self, bg_point, sparse_pm_lrf, sparse_po_lrf):
        # both clouds' embeddings from ONE launch into ONE buffer (the RPE self layers then run 2B at once)
        return self.geo_embedding(torch.cat([torch.cat([bg_point, sparse_pm_lrf], dim=1),
                                             torch.cat([bg_point, sparse_po_lrf], dim=1)], dim=0))

    def _matching(self, end_points, geo, B, sparse_pm, sparse_fm, sparse_po, sparse_fo, fps_idx_m, fps_idx_o, dense_pm,
                  dense_fm, dense_po, dense_fo, radius, slot=None):
        geo_m, geo_o = geo[:B], geo[B:]
        # The reference cloud's positional encoding (Fi:77-80) does not depend on the coarse pose: its two fused
        # group/MLP/max launches run on the side stream UNDER the coarse stage, whose 197-token kernels and
        # hypothesis search are latency-bound and leave most CUs idle.  Only kernels without inter-workgroup
        # waits go there; the mlp3 projection is a library GEMM and stays on the main stream (no two library
        # GEMMs are ever co-scheduled -- DESIGN.md section 7).
        pe2 = None
        if (PE_UNDER_COARSE and self.internal_overlap and not self.test_coarse_only and dense_pm.is_cuda and torch.is_autocast_enabled()
                and dense_pm.shape == dense_po.shape):
            main = torch.cuda.current_stream()
            side = self._side_stream(dense_po.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                PE = self.fine_point_matching.PE
                if PE.split_ok(dense_po) and STACKED_FINE:
                    buf = torch.empty(2 * B, dense_po.shape[1], 512, dtype=torch.bfloat16, device=dense_po.device)
                    pe2 = ("split", PE.groups_split(dense_po, buf, B))
                    buf.record_stream(main)
                else:
                    pe2 = PE.groups(dense_po)
                    pe2.record_stream(main)
        end_points = self.coarse_point_matching(sparse_pm, sparse_fm, geo_m, sparse_po, sparse_fo, geo_o, radius,
                                                end_points, slot=slot)
        if pe2 is not None:
            torch.cuda.current_stream().wait_stream(self._side_stream(dense_po.device))
        if self.fixed_init is not None:  # test hook: the fine stage starts from a GIVEN coarse pose (the model's own one goes to `taps`)
            if self.taps is not None:
                self.taps.update(own_init_R=end_points["init_R"], own_init_t=end_points["init_t"])
            end_points["init_R"], end_points["init_t"] = self.fixed_init
        if self.test_coarse_only:
            end_points["pred_R"] = end_points["init_R"]
            end_points["pred_t"] = end_points["init_t"] * (radius.reshape(-1, 1) + 1e-6)
            end_points["pred_pose_score"] = end_points["init_pose_score"]
            return end_points
        return self.fine_point_matching(dense_pm, dense_fm, geo_m, fps_idx_m, dense_po, dense_fo, geo_o, fps_idx_o,
                                        radius, end_points, pe2_groups=pe2)
