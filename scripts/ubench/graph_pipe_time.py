"""ROUND 6 EXPERIMENT, NOT KEPT: `GraphedPipeline` -- depth hipGraphs of the whole forward (own static inputs each) replayed round-robin on depth
streams -- against the eager PipelinedForward at the reference's contract shape (16 instances, 224 x 224).  Motivation: two, three and four eager
forwards in flight all measure 7.05 ms per step there, one graph 8.7 ms (the forward's dependency chain).  Result (bit-equal poses): graphs depth
1 / 2 / 3 = 1813 / 1874 / 1861 pairs/s bf16, 963 / 889 / 913 fp32 -- replays on different streams do not overlap on this runtime -- against 2264 / 1165
for the eager pipeline in bench.py.  The class is kept here with its measurement."""
import collections
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unopose_amd import ops
from unopose_amd.model import UNOPose, default_model_cfg
from unopose_amd.pipeline import PipelinedForward, Ticket, _pool_streams
from unopose_amd.synthetic import make_batch, trained_like_


class GraphedPipeline:
    """`PipelinedForward` for a stream of SAME-SHAPED batches, each forward replayed as ONE hipGraph: `depth` graphs (own static input
    buffers each), replayed round-robin on `depth` streams, so that consecutive forwards overlap on the GPU while the host pays one launch
    per forward.  For small batches -- the reference's own contract, 16 instances at 224 x 224 -- the eager pipeline is bound by the HOST's
    ~1000 launches per forward (two, three or four forwards in flight all measure 7.05 ms per step) and a single graph by the forward's
    dependency chain (8.7 ms); two graphs in flight are bound by neither.  Outputs are cloned after each replay (the graph's own output
    tensors are overwritten by the slot's next replay).  A batch of another shape (an image's last, shorter chunk) runs eagerly on the
    slot's stream.  `submit` / `drain` / `close` as `PipelinedForward`."""

    OUT_KEYS = ("init_R", "init_t", "init_pose_score", "pred_R", "pred_t", "pred_pose_score")

    def __init__(self, model, depth=2, autocast_dtype=torch.bfloat16, warmup=2):
        if depth not in (1, 2, 3, 4):
            raise ValueError("depth must be 1..4")
        self.model, self.autocast_dtype, self.depth, self.warmup = model, autocast_dtype, depth, warmup
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedPipeline needs the model on a HIP device")
        if not (ops.HIP_GEMM_ALL and ops.USE_HIP_GEMM) or (autocast_dtype is None and not ops.USE_F32X3):
            raise RuntimeError("overlapping forwards need the own GEMMs (ops.HIP_GEMM_ALL / ops.USE_F32X3)")
        self.device = dev
        self.streams = _pool_streams(dev, "pipe", depth)
        self._slots = [None] * depth  # (signature, graph, static inputs, static outputs)
        self._pending = collections.deque()
        self._n = 0
        self._saved_overlap = getattr(model, "internal_overlap", None)

    def _run(self, ep):
        with torch.no_grad():
            if self.autocast_dtype is None:
                return self.model(ep)
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                return self.model(ep)

    @staticmethod
    def _signature(ep):
        return tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in ep.items() if torch.is_tensor(v)))

    def _capture(self, i, ep, sig):
        s = self.streams[i]
        self.drain()  # nothing else in flight while the lazy caches are built and the graph is recorded
        static_in = {k: v.clone() for k, v in ep.items() if torch.is_tensor(v)}
        prev_forbid, ops.FORBID_LIBRARY_BF16_GEMM = ops.FORBID_LIBRARY_BF16_GEMM, True
        try:
            with torch.cuda.stream(s):
                for _ in range(self.warmup):
                    self._run(dict(static_in))
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = self._run(dict(static_in))
        finally:
            ops.FORBID_LIBRARY_BF16_GEMM = prev_forbid
        self._slots[i] = (sig, g, static_in, {k: out[k] for k in self.OUT_KEYS if k in out})

    def submit(self, end_points):
        while len(self._pending) >= self.depth:
            self._pending.popleft().wait()
        i = self._n % self.depth
        self._n += 1
        s = self.streams[i]
        sig = self._signature(end_points)
        if self._slots[i] is None:  # a slot keeps the graph of the FIRST shape it sees (the full chunk); other shapes run eagerly
            self._capture(i, end_points, sig)
        cur = torch.cuda.current_stream(self.device)
        s.wait_stream(cur)
        for v in end_points.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(s)
        slot = self._slots[i]
        with torch.cuda.stream(s):
            if slot[0] == sig:
                for k, v in slot[2].items():
                    v.copy_(end_points[k], non_blocking=True)
                slot[1].replay()
                out = dict(end_points)
                out.update({k: v.clone() for k, v in slot[3].items()})
            else:  # another shape (a shorter last chunk): eagerly, on the same stream
                prev_forbid, ops.FORBID_LIBRARY_BF16_GEMM = ops.FORBID_LIBRARY_BF16_GEMM, True
                try:
                    out = self._run(dict(end_points))
                finally:
                    ops.FORBID_LIBRARY_BF16_GEMM = prev_forbid
            done = torch.cuda.Event()
            done.record(s)
        t = Ticket(out, done, s)
        self._pending.append(t)
        return t

    def drain(self):
        while self._pending:
            self._pending.popleft().wait()

    def close(self):
        self.drain()
        if self._saved_overlap is not None:
            self.model.internal_overlap = self._saved_overlap


torch.set_grad_enabled(False)
dev = torch.device("cuda"); B = int(os.environ.get("GP_B", 16)); S = int(os.environ.get("GP_S", 224))
model = trained_like_(UNOPose(default_model_cfg(feature_extraction=dict(img_size=S)))).to(dev).eval()
batches = []
for i in range(3):
    b, _, _ = make_batch(B, 2048, 5000, S, seed=700 + i, device=dev); b["coarse_rand"] = torch.rand(B, 18000, device=dev); batches.append(b)
def rate(pipe, n=40):
    for i in range(4): pipe.submit(dict(batches[i % 3]))
    pipe.drain(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): t = pipe.submit(dict(batches[i % 3]))
    t.result(); pipe.drain(); torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)
for name, amp in (("bf16", torch.bfloat16), ("fp32", None)):
    seq = PipelinedForward(model, depth=1, autocast_dtype=amp)
    ref = [{k: seq.submit(dict(b)).wait()[k].clone() for k in GraphedPipeline.OUT_KEYS} for b in batches]
    seq.close()
    for depth in (1, 2, 3):
        gp = GraphedPipeline(model, depth=depth, autocast_dtype=amp)
        outs = [gp.submit(dict(batches[i % 3])) for i in range(6)]
        ok = all(torch.equal(outs[i].result()[k], ref[i % 3][k]) for i in range(6) for k in GraphedPipeline.OUT_KEYS)
        print(f"{name} graphs depth {depth}: equal to eager one-at-a-time {ok};  {rate(gp):8.1f} pairs/s", flush=True)
        gp.close()
    ep = PipelinedForward(model, depth=2, autocast_dtype=amp)
    print(f"{name} eager pipeline depth 2: {rate(ep):8.1f} pairs/s", flush=True)
    ep.close()
