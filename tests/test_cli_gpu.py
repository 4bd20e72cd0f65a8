"""GPU: `python -m unopose_amd.cli` end to end on the synthetic BOP folder -- config file + overrides, checkpoint load (strict), provider,
runner, result path of engine.py:36-52 -- writes the lines the runner writes when driven directly with the same model."""
import json
import os

import numpy as np
import pytest
import torch

import bop_synth

pytestmark = pytest.mark.gpu


@torch.no_grad()
def test_cli_writes_the_runner_lines_at_the_reference_result_path(tmp_path):
    from unopose_amd import cli
    from unopose_amd import provider as P
    from unopose_amd.model import UNOPose, default_model_cfg
    from unopose_amd.runner import inference_and_save
    from unopose_amd.synthetic import trained_like_

    root = str(tmp_path / "bop")
    dcfg, det_path = bop_synth.build(root)
    mcfg = default_model_cfg(fine_npoint=256, feature_extraction=dict(img_size=dcfg["img_size"]))
    torch.manual_seed(3)
    model = trained_like_(UNOPose(mcfg))
    ckpt = str(tmp_path / "model_final.pth")
    torch.save({"model": model.state_dict(), "iteration": 7}, ckpt)  # MyCheckpointer's layout
    cfg = dict(model=dict(cfg=dict(mcfg)), dataloader=dict(test=dict(dataset=dict(cfg=dcfg, eval_dataset_name="ycbv", detetion_path=det_path))),
               test=dict(amp=dict(enabled=False), instance_batch_size=2), misc=dict(output_dir=str(tmp_path / "out"), load_from=""), bop_eval=dict(split="test"))
    cfgf = tmp_path / "cfg.json"
    cfgf.write_text(json.dumps(cfg))

    np.random.seed(11)
    torch.manual_seed(5)  # the coarse stage draws inside forward
    assert cli.main(["--config-file", str(cfgf), f"misc.load_from={ckpt}", "misc.exp_name=_t"]) == 0
    path = tmp_path / "out" / "inference_model_final" / "ycbv" / "result_t_ycbv-test.csv"
    assert path.exists() and path.with_suffix(".json").exists()
    got = path.read_text().splitlines()

    ds = P.BOPTestsetOneRef(dcfg, "ycbv", det_path)
    np.random.seed(11)
    images = [P.collate_image(ds[i]) for i in range(len(ds))]
    torch.manual_seed(5)
    want = inference_and_save(model.cuda().eval(), images, str(tmp_path / "direct.csv"), instance_batch_size=2, device="cuda")
    strip = lambda lines: [",".join(l.strip().split(",")[:-1]) for l in lines]  # noqa: E731  (last field = wall-clock time)
    assert len(got) == len(want) >= 3 and strip(got) == strip(want)
