"""Sharded inference runner: the caller side of UNOPose.forward.

Reproduces the I/O contract of the reference's test loop
(core/unopose/engine/oneref_inference_utils_v1.py:13-136): per image, instances are chunked by
``instance_batch_size`` (:42-48), the predicted pose is composed with the reference view's pose
(:83-91), scores are multiplied by the detection score (:99), translations go to millimetres (:98) and
one CSV row per instance is emitted (:114-123).  Unlike the reference -- where every rank writes the same
file (:130-135) -- images are sharded over ranks exactly as detectron2's ``InferenceSampler`` does
(core/unopose/utils/my_distributed_sampler.py:263-271) and the rows are GATHERED to rank 0 over
``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests).  There is no other data-path collective:
pairs are independent.
"""
import json
import time

import numpy as np
import torch
import torch.distributed as dist


def shard_range(total_size, world_size, rank):
    """InferenceSampler._get_local_indices: contiguous blocks whose sizes differ by at most one."""
    shard = total_size // world_size
    left = total_size % world_size
    sizes = [shard + int(r < left) for r in range(world_size)]
    begin = sum(sizes[:rank])
    end = min(sum(sizes[:rank + 1]), total_size)
    return range(begin, end)


def compose_pose(pred_R, pred_t, tem1_pose=None):
    """T_query<-obj = [pred_R | pred_t] @ T_ref<-obj  (oneref_inference_utils_v1.py:83-91)."""
    if tem1_pose is None:
        return pred_R, pred_t
    T = torch.zeros_like(tem1_pose)
    T[:, 3, 3] = 1.0
    T[:, :3, :3] = pred_R
    T[:, :3, 3] = pred_t
    T = T @ tem1_pose
    return T[:, :3, :3], T[:, :3, 3]


def csv_line(scene_id, img_id, obj_id, score, R9, t3_mm, image_time):
    """scene_id,im_id,obj_id,score,R (9, space separated),t (3, mm),time  (:114-123)."""
    return ",".join((str(scene_id), str(img_id), str(obj_id), str(score), " ".join(str(v) for v in R9),
                     " ".join(str(v) for v in t3_mm), f"{image_time}\n"))


_INPUT_KEYS = ("pts", "rgb", "rgb_choose", "fps_idx_m", "tem1_rgb", "tem1_choose", "tem1_pts", "fps_idx_o")
_REF_KEYS = ("ref_dense_po", "ref_dense_fo", "ref_radius", "ref_lrf")


class ReferenceCache:
    """Per-reference-view features kept across images: BOP test sets pair many
    query instances with few reference views, and everything `UNOPose.encode_reference` returns depends
    on the reference view alone.  Keys are whatever identifies a view to the caller (e.g.
    ``(ref_scene_id, ref_im_id, obj_id)``); entries live on the model's device.  Eviction is LRU (a hit
    refreshes the entry), happens only AFTER the batch's result has been assembled -- a key the current call
    needs is never dropped under it -- and entries are own copies, so an eviction really frees memory."""

    def __init__(self, model, max_items=256):
        from collections import OrderedDict

        self.model, self.max_items, self.store = model, max_items, OrderedDict()
        self.hits = self.misses = 0

    def lookup(self, keys, tem1_rgb, tem1_choose, tem1_pts, encode=None):
        """`encode`: the encoder to use for missing views instead of `model.encode_reference` under the caller's autocast state --
        `PipelinedForward.encode_reference` when the forwards go through a pipeline (its precision, its stream ordering)."""
        first = {}
        for i, k in enumerate(keys):
            if k not in self.store:
                first.setdefault(k, i)  # one encode per distinct missing view
        if first:
            sel = torch.as_tensor(list(first.values()), device=tem1_pts.device)
            enc = (encode or self.model.encode_reference)(tem1_rgb[sel], tem1_choose[sel], tem1_pts[sel])
            for j, k in enumerate(first):
                self.store[k] = {name: v[j].clone() for name, v in enc.items()}
        for k in keys:
            self.store.move_to_end(k)  # most recently used last
        self.misses += len(first)
        self.hits += len(keys) - len(first)
        out = {name: torch.stack([self.store[k][name] for k in keys]) for name in _REF_KEYS}
        while len(self.store) > self.max_items:  # least recently used first; `out` no longer needs the store
            self.store.popitem(last=False)
        return out


@torch.no_grad()
def run_image(model, data, instance_batch_size=16, device=None, ref_cache=None, pipeline=None):
    """One image = data[key] with a leading image dim of 1 and n_instance instances (:36-100).
    With `ref_cache` (a ReferenceCache) and `data["ref_keys"]` (one hashable per instance) the reference
    side of every pair comes from the cache.  With `pipeline` (a pipeline.PipelinedForward over `model`) the
    instance chunks of the image are submitted back to back -- the matcher of one chunk runs under the ViT of the next --
    and collected in order; the numbers are those of the chunk-by-chunk loop."""
    if device is not None:
        data = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
    n = data["pts"].size(1)
    Rs, ts, scores = [], [], []
    pending = []
    for s in range(0, n, instance_batch_size):
        e = min(n, s + instance_batch_size)
        inputs = {k: data[k][0][s:e].contiguous() for k in _INPUT_KEYS if k in data}
        if ref_cache is not None and "ref_keys" in data:
            inputs.update(ref_cache.lookup(list(data["ref_keys"][s:e]), inputs["tem1_rgb"], inputs["tem1_choose"],
                                           inputs["tem1_pts"], None if pipeline is None else pipeline.encode_reference))
        pending.append((s, e, pipeline.submit(inputs).result if pipeline is not None else (lambda o=model(inputs): o)))
    for s, e, get in pending:
        out = get()
        R, t = compose_pose(out["pred_R"], out["pred_t"],
                            data["tem1_pose"][0][s:e].contiguous() if "tem1_pose" in data else None)
        Rs.append(R)
        ts.append(t)
        scores.append(out["pred_pose_score"])
    Rs = torch.cat(Rs, 0).reshape(-1, 9).float().cpu().numpy()
    ts = torch.cat(ts, 0).float().cpu().numpy() * 1000
    scores = (torch.cat(scores, 0) * data["score"][0, :, 0]).float().cpu().numpy()
    return Rs, ts, scores


def inference_and_save(model, images, save_path, instance_batch_size=16, device=None, sync=None, ref_cache=None,
                       dets=None, pipeline=None):
    """`images`: an indexable of per-image dicts (the reference's test dataset items, batch dim 1).
    Every rank processes its InferenceSampler shard; rows are gathered to rank 0, which writes the CSV
    and the detections JSON in global image order.  Returns the CSV lines on rank 0, None elsewhere.

    Detections JSON (oneref_inference_utils_v1.py:31,112-113,134-135): a deep copy of the dataset's
    detections -- ``{"<scene:06d>_<img:06d>": [detection dict, ...]}``, every field kept, detections the
    provider filtered out included -- with ``pred_R`` (9 floats, row major) and ``pred_t`` (3 floats, mm) added
    to the entries ``inst_ids`` names.  `dets` defaults to ``images.dets`` (the provider's attribute, like
    ``data_loader.dataset.dets``).  `pipeline`: see `run_image`."""
    from copy import deepcopy

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if dets is None:
        dets = getattr(images, "dets", None)
    dets = deepcopy(dets) if dets is not None else {}
    rows, preds = [], []
    for idx in shard_range(len(images), world, rank):
        data = images[idx]
        if sync is not None:
            sync()
        t0 = time.perf_counter()
        Rs, ts, scores = run_image(model, data, instance_batch_size, device, ref_cache, pipeline)
        if sync is not None:
            sync()
        image_time = time.perf_counter() - t0
        if "seg_time" in data:
            image_time += float(data["seg_time"])
        scene_id, img_id = int(data["scene_id"]), int(data["img_id"])
        inst_ids = np.asarray(data["inst_ids"][0]) if "inst_ids" in data else np.arange(len(scores))
        for k in range(len(scores)):
            rows.append((idx, k, csv_line(scene_id, img_id, int(data["obj_id"][0][k]), scores[k], Rs[k], ts[k],
                                          image_time)))
            preds.append((f"{scene_id:06d}_{img_id:06d}", int(inst_ids[k]), Rs[k].tolist(), ts[k].tolist()))
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((rows, preds), gathered, dst=0)
        if rank != 0:
            return None
        rows = [r for part in gathered for r in part[0]]
        preds = [q for part in gathered for q in part[1]]
    for key, inst_i, R9, t3 in preds:
        if isinstance(dets.get(key), list):  # the reference's layout: a list of detection dicts per image
            entry = dets[key][inst_i]
        else:  # no detections handed in (synthetic runs): same nesting, keyed by instance
            entry = dets.setdefault(key, {}).setdefault(inst_i, {})
        entry["pred_R"], entry["pred_t"] = R9, t3
    rows.sort(key=lambda r: (r[0], r[1]))
    lines = [r[2] for r in rows]
    with open(save_path, "w+") as f:
        f.writelines(lines)
    with open(save_path.replace(".csv", ".json"), "w") as f:
        json.dump(dets, f)
    return lines


def broadcast_module_(module, src=0):
    """One flat broadcast of all parameters and buffers from `src` (weights travel once over xGMI)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return module
    tensors = list(module.parameters()) + list(module.buffers())
    if not tensors:
        return module

    def bcast(flat):
        if flat.is_cuda and dist.get_backend() != "nccl":  # host-memory backends (gloo in the tests)
            host = flat.cpu()
            dist.broadcast(host, src)
            return host.to(flat.device)
        dist.broadcast(flat, src)
        return flat

    # one flat buffer per dtype class: floating-point tensors travel as fp32, integer / bool buffers (BatchNorm's int64
    # `num_batches_tracked`) as int64 -- an fp32 round trip would corrupt counters above 2^24
    for is_float, wire in ((True, torch.float32), (False, torch.int64)):
        group = [t for t in tensors if t.is_floating_point() == is_float]
        if not group:
            continue
        flat = bcast(torch.cat([t.detach().reshape(-1).to(wire) for t in group]))
        off = 0
        with torch.no_grad():  # in-place copy on the parameter itself: bumps `_version`, which keys every derived-weight
            for t in group:    # caches in ops/ (a `.data` alias would leave them stale)
                n = t.numel()
                t.copy_(flat[off:off + n].reshape(t.shape).to(t.dtype))
                off += n
    return module
