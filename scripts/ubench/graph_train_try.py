"""Round 6 experiment (not kept): the whole training step -- forward in train mode, process_loss, backward, gradient hygiene, Adam -- captured once
into ONE hipGraph and replayed (static batch, the pose-noise draw injected through `aug_pose`, Adam(capturable=True) with a tensor learning rate).
Capture works (4.4 s, autograd across the side streams included) and trains, but replays at 110.1 ms per step against 108 ms eager: the ~4900
launches of a step are serialised by their data dependencies and cost the same per node from a graph as from the host, so the cure is fewer,
fused kernels in the 197-token layers, not a cheaper launch.  The class below is the experiment's `GraphedTrainStep`."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.synthetic import make_train_batch, trained_like_
from unopose_amd.losses import aug_pose_noise, process_loss
from unopose_amd.train import flat_and_anneal_factor, freeze_backbone, zero_nonfinite_grads_
dev = torch.device("cuda")


class GraphedTrainStep:
    def __init__(self, model, example_batch, optimizer, scheduler=None, warmup=3):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.static = {k: v.clone() for k, v in example_batch.items() if torch.is_tensor(v)}
        B = self.static["pts"].shape[0]
        self.aug_R = torch.eye(3, device=dev).repeat(B, 1, 1)
        self.aug_t = torch.zeros(B, 3, device=dev)
        self._draw(example_batch)
        model.train()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.info = self._body(zero=False)

    def _draw(self, batch):
        with torch.no_grad():
            tem = batch["tem1_pts"]
            radius = torch.norm(tem - tem.mean(1, keepdim=True), dim=2).max(1)[0]
            R, t = aug_pose_noise(batch["rotation_label"], batch["translation_label"] / (radius.reshape(-1, 1) + 1e-6))
            self.aug_R.copy_(R)
            self.aug_t.copy_(t)

    def _body(self, zero=True):
        ep = dict(self.static)
        ep["aug_pose"] = (self.aug_R, self.aug_t)
        info = process_loss(self.model(ep))
        if zero:
            self.optimizer.zero_grad(set_to_none=True)
        info["loss"].backward()
        zero_nonfinite_grads_(self.model)
        self.optimizer.step()
        return {k: v.detach() for k, v in info.items()}

    def __call__(self, batch):
        for k, v in self.static.items():
            v.copy_(batch[k], non_blocking=True)
        self._draw(batch)
        self.graph.replay()
        if self.scheduler is not None:
            self.scheduler.step()
        return self.info


def build_optimizer(model, lr=1e-4, total_iters=188340):
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=torch.tensor(float(lr), device=dev), betas=(0.5, 0.999), eps=1e-6, weight_decay=0.0, capturable=True)
    return opt, torch.optim.lr_scheduler.LambdaLR(opt, lambda it: flat_and_anneal_factor(it, total_iters))


B, npts, img = 8, 4096, 224
torch.manual_seed(0)
model = freeze_backbone(trained_like_(UNOPose(default_model_cfg(fine_npoint=npts, feature_extraction=dict(img_size=img)))).to(dev))
batch = make_train_batch(B, npts, npts + npts // 2, img, seed=300, device=dev)
opt, sched = build_optimizer(model)
t0 = time.perf_counter()
g = GraphedTrainStep(model, batch, opt, sched)
torch.cuda.synchronize(); print("capture took %.1f s" % (time.perf_counter() - t0), flush=True)
for _ in range(2): info = g(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
ls = []
for _ in range(10):
    info = g(batch); ls.append(float(info["loss"]))
torch.cuda.synchronize()
print("graphed: %.2f ms/step" % ((time.perf_counter() - t0) / 10 * 1e3), ls[0], ls[-1])
