"""hipGraph replay of UNOPose.forward.

The forward has static shapes and no host synchronisation, so the ~1000 kernel launches of one step
(host enqueue time ~14 ms, more than the GPU time for batches <= 8 pairs) can be captured once into a
HIP graph and replayed with one launch.  Inputs are copied into static buffers; outputs are the static
tensors of the captured run (clone them if they must outlive the next replay).
"""
import torch


class GraphedForward:
    def __init__(self, model, example_inputs, autocast_dtype=torch.bfloat16, warmup=3):
        self.model = model
        self.autocast_dtype = autocast_dtype
        self.static_in = {k: v.clone() for k, v in example_inputs.items()}
        self.keys = tuple(sorted(self.static_in))
        dev = next(iter(self.static_in.values())).device
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(warmup):  # lazy weight caches, hipFuncSetAttribute, allocator warm-up
                self._run()
        torch.cuda.current_stream(dev).wait_stream(stream)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = self._run()

    def _run(self):
        ep = dict(self.static_in)
        if self.autocast_dtype is None:
            return self.model(ep)
        with torch.autocast("cuda", dtype=self.autocast_dtype):
            return self.model(ep)

    def __call__(self, end_points):
        assert tuple(sorted(k for k in end_points if k in self.static_in)) == self.keys, "input keys changed"
        for k in self.keys:
            self.static_in[k].copy_(end_points[k], non_blocking=True)
        self.graph.replay()
        for k in ("init_R", "init_t", "init_pose_score", "pred_R", "pred_t", "pred_pose_score"):
            if k in self.static_out:
                end_points[k] = self.static_out[k]
        return end_points
