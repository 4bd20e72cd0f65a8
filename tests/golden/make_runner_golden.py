"""Capture (inputs, model outputs) -> CSV lines + detections JSON from the REFERENCE's own test loop
(core/unopose/engine/oneref_inference_utils_v1.py:13-136), run here with a deterministic stand-in model.
Build container only (imports /root/reference).  Stubs: tqdm (progress bar), orjson (-> json), torch.cuda.synchronize
and Tensor.cuda (no GPU here).  Output: tests/golden/runner_capture.json -- the seed of the inputs, the CSV rows
without their wall-clock `time` column, and the detections JSON the reference wrote.

    python tests/golden/make_runner_golden.py
"""
import json
import os
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

from runner_case import StubModel, make_case  # noqa: E402  (shared with tests/test_runner_cpu.py)


def main():
    class _Bar:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def set_description(self, *a):
            pass

        def update(self, *a):
            pass

    sys.modules["tqdm"] = types.SimpleNamespace(tqdm=_Bar)
    sys.modules["orjson"] = types.SimpleNamespace(dumps=lambda o: json.dumps(o).encode())
    torch.cuda.synchronize = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, "/root/reference")
    from core.unopose.engine.oneref_inference_utils_v1 import inference_and_save_oneref_v1

    images, dets = make_case()

    class _Loader(list):
        pass

    loader = _Loader(images)
    loader.dataset = types.SimpleNamespace(dets=dets)
    model = StubModel()
    model.eval = lambda: model
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "result.csv")
        inference_and_save_oneref_v1(model, loader, path, instance_batch_size=4)
        lines = open(path).read().splitlines()
        out_dets = json.load(open(path.replace(".csv", ".json")))
    cap = {"csv_without_time": [l.rsplit(",", 1)[0] for l in lines], "n_time_fields": [len(l.split(",")) for l in lines],
           "dets": out_dets}
    json.dump(cap, open(os.path.join(HERE, "runner_capture.json"), "w"), indent=0)
    print(f"{len(lines)} rows, {len(out_dets)} images; first row: {lines[0][:120]}")


if __name__ == "__main__":
    main()
