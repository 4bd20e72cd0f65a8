"""Two batches in flight on one GPU: the matcher of batch i under the ViT of batch i+1.

`UNOPose.forward` is a throughput-bound half (the ViT: ~26 ms of GEMMs / attention at 32 pairs, 518x518) followed by a
half made of ~450 small dependent kernels (FPS, geometric transformer on 197 tokens, pose heads: ~14 ms during which most
of the chip idles between launches).  Consecutive batches are independent, so a second HIP stream lets the small
kernels of batch i fill the gaps -- and run beside -- the large kernels of batch i+1: +8 % pairs/s at 518x518, +14 % at
224x224 (bench.py, same box A/B), with every result bit-identical to the one-stream order (same kernels, same inputs).

What makes this safe: (1) NO library GEMM is left on the eval path (`ops.own_gemm_ok`, csrc/gemm.hip, gemm_f32.hip, bmm_f32.hip):
hipBLASLt's stream-K kernels spin on partner workgroups -- three forwards in flight hang with them, none do without;
(2) no kernel of libunopose_hip.so contains packed-fp32 instructions (build.py: -fno-slp-vectorize -fno-vectorize): kernels that do
return wrong values for a few elements when waves of another kernel issuing MFMAs share their CU (DESIGN.md section 7, round 3 --
round 2 had attributed this to the library's bf16 GEMMs).  With both in place pipelined and one-at-a-time execution give
bit-identical poses, also at the bench size in stage mode (tests/test_pipeline_gpu.py, tests/test_coresidency_gpu.py).
The fp32 path is run one batch at a time.

The reference has no counterpart (its runner, engine/oneref_inference_utils_v1.py:13-136, calls the model batch by
batch on the default stream); `runner.inference_and_save` uses this class for consecutive detection batches.
"""
import collections

import torch

from . import ops


# HIP streams are a process-wide resource that maps onto a handful of hardware queues (GPU_MAX_HW_QUEUES = 4 by default): every stream
# a process has EVER created keeps its queue assignment, and two streams that share a queue serialise.  A process that builds one pipeline
# after another (bench.py's legs, a runner that is re-created per dataset) must therefore not mint new streams each time -- measured in
# round 6: the bounded training leg at the end of bench.py's default run took 119 ms per step against 108 ms in a process of its own, its
# side stream sharing a queue with the main one.  All pipelines of a process draw from this pool (they are used one at a time).
_STREAM_POOL = {}


WHOLE_INTERNAL_OVERLAP = True  # the model's in-forward side streams (geometry chain under the ViT, reference PE under the coarse stage) stay on when whole
#                                forwards run side by side too: + 2.0 % at 224 x 224 once the process had 16 hardware queues (2788 -> 2844 pairs/s, round 6;
#                                with 4 queues the extra streams serialised against the pipeline's own and it measured slower: off in rounds 4-5)


def _pool_streams(dev, kind, n, priority=0):
    pool = _STREAM_POOL.setdefault((dev.index, kind), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev, priority=priority))
    return pool[:n]


class Ticket:
    """One submitted forward: `result()` hands back the end_points dict, ordered after the forward on the caller's
    current stream (no host synchronisation)."""

    def __init__(self, out, done, stream):
        self._out, self._done, self._stream = out, done, stream

    def result(self):
        cur = torch.cuda.current_stream()
        if self._stream is not None and cur != self._stream:
            cur.wait_event(self._done)
            for v in self._out.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
        return self._out

    def wait(self):
        """Block the host until this forward has finished."""
        if self._done is not None:
            self._done.synchronize()
        return self._out


class PipelinedForward:
    """model: an eval-mode UNOPose on a HIP device.  depth: forwards in flight (1..4 accepted; 2 is the useful value: bench.py
    --inflight 3 / 4 measure the same rate or less).  autocast_dtype: torch.bfloat16 or None (fp32, the reference's default precision:
    pipelined as well since round 6 -- every fp32 GEMM of the eval path is the own fp32-class kernel, `ops.USE_F32X3`; with that switch off
    fp32 runs one batch at a time, whatever `depth` says).  The model's weights must not change while forwards are in flight (the per-module weight
    caches are rebuilt on whichever stream sees the new version first); after an update call `drain()` then `reset()`.
    `close()` restores the model's `internal_overlap` switch."""

    def __init__(self, model, depth=2, autocast_dtype=torch.bfloat16, run_ahead=None, timing=False, stages="auto"):
        if depth not in (1, 2, 3, 4):
            raise ValueError("depth must be 1..4")
        self.model, self.autocast_dtype = model, autocast_dtype
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("PipelinedForward needs the model on a HIP device")
        self.device = dev
        self.depth = depth if (autocast_dtype is not None or ops.USE_F32X3) else 1
        if self.depth > 1 and not (ops.HIP_GEMM_ALL and ops.USE_HIP_GEMM):
            raise RuntimeError("more than one forward in flight needs ops.HIP_GEMM_ALL (library stream-K GEMMs spin on partner "
                               "workgroups and can hang when forwards overlap)")
        self.streams = _pool_streams(dev, "pipe", self.depth) if self.depth > 1 else [None]
        self._saved_overlap = getattr(model, "internal_overlap", None)
        if hasattr(model, "internal_overlap"):  # (decided per submit: see WHOLE_INTERNAL_OVERLAP)
            model.internal_overlap = self.depth == 1 or WHOLE_INTERNAL_OVERLAP
        # Derived-weight caches (bf16 / split weights, packed PE image, attention weight blocks ...) are built lazily by the FIRST
        # forward, on the stream it runs on.  A forward on ANOTHER pipeline stream must not read them before those kernels have run:
        # every pipeline stream waits once for the completion event of the first forward (pipeline.py:116).
        self._warm = None
        self._warm_seen = set()
        self.last_mode = None
        self._n = 0
        self._pending = collections.deque()
        # run_ahead: forwards the host may enqueue beyond `depth` before it has seen one finish.  Default 1 with several forwards in
        # flight; 0 with ONE forward at a time (fp32): there the next forward's first side-stream wait, enqueued a whole forward early,
        # can share a hardware queue with the running forward's stream (HIP multiplexes streams onto GPU_MAX_HW_QUEUES = 4 queues) --
        # measured 87 -> 101 ms per fp32 forward whenever the stream pool happened to be laid out that way (bf16 stage-mode steps earlier
        # in the process; GPU_MAX_HW_QUEUES=8 also cures it), against 1 % for giving up the run-ahead.
        if run_ahead is None:
            run_ahead = 1 if self.depth > 1 else 0
        self._limit = self.depth + max(0, int(run_ahead))  # forwards the host may have enqueued and not yet seen finish
        # stages: every forward is cut in two -- `forward_features` (the ViT) on one stream, `forward_matching` on a second
        # (same priority since round 6) -- so that ViTs never overlap each other and the latency-bound matcher of batch i always runs
        # beside the ViT of batch i + 1, instead of two whole forwards advancing side by side.  Measured (same box, pairs/s):
        # 518 x 518 crops 861 vs 832, 224 x 224 crops 1988 vs 2086 (there the matcher outlasts the ViT) -> "auto" cuts the
        # forward when the ViT sees >= 1024 tokens.  The in-forward side-stream overlaps stay on in this mode (+1.5 %).
        self.stages = stages if self.depth > 1 and hasattr(model, "forward_features") else False
        self._stage_streams = None
        self.timing = bool(timing)
        self.history = []  # timing=True: (start event, end event) of every forward, on the stream it ran on (bench.py)

    def _forward(self, end_points):
        with torch.no_grad():
            if self.autocast_dtype is None:
                return self.model(end_points)
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                return self.model(end_points)

    def submit(self, end_points):
        """Enqueue one forward over `end_points` (tensors ready on the caller's current stream); returns a Ticket."""
        while len(self._pending) >= self._limit:  # bounded run-ahead: the host never queues more than `_limit` forwards
            self._pending.popleft().wait()
        s = self.streams[self._n % len(self.streams)]
        self._n += 1
        use_stages = self.stages
        if use_stages == "auto":  # decided per batch: a later batch of another resolution gets the mode that suits it
            rgb = end_points.get("rgb")
            use_stages = bool(torch.is_tensor(rgb) and (rgb.shape[-1] // 14) * (rgb.shape[-2] // 14) + 5 >= 1024)
        if hasattr(self.model, "internal_overlap") and self.depth > 1:
            self.model.internal_overlap = bool(use_stages) or WHOLE_INTERNAL_OVERLAP
        self.last_mode = "stages" if use_stages else "whole"  # what this submit chose (bench.py reports it)
        if use_stages:
            if self._stage_streams is None:
                # (both at the default priority: the matching stream at high priority -- rounds 4-6 -- measures 0.7-0.9 % slower on two boxes, the
                #  features stream above the matching stream 1 % slower with 1.5 x the latency: scripts/ubench/prio_ab.sh)
                self._stage_streams = [_pool_streams(self.device, "features", 1)[0], _pool_streams(self.device, "matching0", 1)[0]]
            for ss in self._stage_streams:  # the caches the very first forward built (in EITHER mode) are ordered before this stream's first use
                if self._warm is not None and ss.cuda_stream not in self._warm_seen:
                    ss.wait_event(self._warm)
                    self._warm_seen.add(ss.cuda_stream)
            prev_forbid, ops.FORBID_LIBRARY_BF16_GEMM = ops.FORBID_LIBRARY_BF16_GEMM, True
            try:
                t, start, done = self._submit_stages(end_points)
            finally:
                ops.FORBID_LIBRARY_BF16_GEMM = prev_forbid
            if self._warm is None:
                self._warm = done
                self._warm_seen.update(ss.cuda_stream for ss in self._stage_streams)
        elif s is None:
            start = torch.cuda.Event(enable_timing=True) if self.timing else None
            if start is not None:
                start.record()
            out = self._forward(end_points)
            done = torch.cuda.Event(enable_timing=self.timing)
            done.record()
            t = Ticket(out, done, None)
        else:
            cur = torch.cuda.current_stream(self.device)
            s.wait_stream(cur)
            if self._warm is not None and s.cuda_stream not in self._warm_seen:
                s.wait_event(self._warm)  # the caches the first forward built on its stream
                self._warm_seen.add(s.cuda_stream)
            for v in end_points.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(s)
            prev_forbid, ops.FORBID_LIBRARY_BF16_GEMM = ops.FORBID_LIBRARY_BF16_GEMM, True
            try:
                with torch.cuda.stream(s):
                    start = torch.cuda.Event(enable_timing=True) if self.timing else None
                    if start is not None:
                        start.record(s)
                    out = self._forward(end_points)
                    done = torch.cuda.Event(enable_timing=self.timing)
                    done.record(s)
            finally:
                ops.FORBID_LIBRARY_BF16_GEMM = prev_forbid
            t = Ticket(out, done, s)
            if self._warm is None:
                self._warm = done
                self._warm_seen.add(s.cuda_stream)
        self._pending.append(t)
        if self.timing and start is not None:
            self.history.append((start, done))
            del self.history[:-4096]
        return t

    def _submit_stages(self, end_points):
        sv, st = self._stage_streams
        cur = torch.cuda.current_stream(self.device)
        sv.wait_stream(cur)
        for v in end_points.values():
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(sv)
                v.record_stream(st)
        import contextlib

        ac = (lambda: torch.autocast("cuda", dtype=self.autocast_dtype)) if self.autocast_dtype is not None else contextlib.nullcontext
        with torch.no_grad():
            with torch.cuda.stream(sv), ac():
                start = torch.cuda.Event(enable_timing=True) if self.timing else None
                if start is not None:
                    start.record(sv)
                feats = self.model.forward_features(end_points)
                ready = torch.cuda.Event()
                ready.record(sv)
            flat = list(feats[:5]) + (list(feats[5].values()) if isinstance(feats[5], dict) else [])
            for v in flat:
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(st)
            with torch.cuda.stream(st), ac():
                st.wait_event(ready)
                out = self.model.forward_matching(end_points, feats)
                done = torch.cuda.Event(enable_timing=self.timing)
                done.record(st)
        return Ticket(out, done, st), start, done

    def encode_reference(self, tem1_rgb, tem1_choose, tem1_pts):
        """`model.encode_reference` for a `runner.ReferenceCache` used together with this pipeline: same autocast dtype as the
        pipelined forwards (a cached reference must carry the numbers the uncached forward would compute), on the caller's
        current stream, ordered AFTER every forward in flight (GPU-side waits, the host does not block) so that the encoder's
        kernels never share the device with them; later `submit`s wait for the caller's stream as always."""
        cur = torch.cuda.current_stream(self.device)
        for t in self._pending:
            if t._done is not None:
                cur.wait_event(t._done)
        with torch.no_grad():
            if self.autocast_dtype is None:
                return self.model.encode_reference(tem1_rgb, tem1_choose, tem1_pts)
            with torch.autocast("cuda", dtype=self.autocast_dtype):
                return self.model.encode_reference(tem1_rgb, tem1_choose, tem1_pts)

    def drain(self):
        """Host-wait for everything submitted so far."""
        while self._pending:
            self._pending.popleft().wait()

    def reset(self):
        """Forget which streams have seen the weight caches (call after `drain()` when the model's weights changed)."""
        self.drain()
        self._warm = None
        self._warm_seen.clear()

    def close(self):
        """Drain and give the model its `internal_overlap` switch back."""
        self.drain()
        if self._saved_overlap is not None:
            self.model.internal_overlap = self._saved_overlap

