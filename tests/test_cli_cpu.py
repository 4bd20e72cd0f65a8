"""unopose_amd.cli (SURVEY.md 8(b) "CLI / config boundary"): config loading, key=value overrides and the result-path convention of
core/unopose/engine/engine.py:36-52, without touching a GPU (--print-plan); the run itself is tests/test_model_gpu.py's."""
import json
import os
import subprocess
import sys

import pytest

from unopose_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = dict(model=dict(cfg=dict(coarse_npoint=196)), dataloader=dict(test=dict(dataset=dict(eval_dataset_name="ycbv", detetion_path="d.json", cfg=dict(img_size=224)))),
            test=dict(amp=dict(enabled=False), instance_batch_size=16, save_results_only=False), misc=dict(output_dir="output/unopose", load_from=""),
            bop_eval=dict(split="test"))


def test_result_path_convention_and_overrides(tmp_path):
    cfg = cli.apply_overrides(json.loads(json.dumps(BASE)), ["misc.load_from=ckpts/model_final_wo_optim.pth", "test.amp.enabled=True", "misc.exp_name=_abl",
                                                            "test.instance_batch_size=32", "dataloader.test.dataset.eval_dataset_name=lmo"])
    assert cfg["test"]["amp"]["enabled"] is True and cfg["test"]["instance_batch_size"] == 32
    d, p = cli.result_paths(cfg)
    # engine.py:37-52: <output_dir>/inference_<checkpoint stem>/<dataset>/result<exp_name>_<dataset>-<split>.csv
    assert d == "output/unopose/inference_model_final_wo_optim/lmo" and p == d + "/result_abl_lmo-test.csv"
    assert cli.result_paths(cfg, iteration=500)[0] == "output/unopose/inference_iter_500/lmo"
    with pytest.raises(ValueError):
        cli.apply_overrides(cfg, ["no_equals_sign"])
    assert cli.apply_overrides(cfg, ["misc.note=plain text"])["misc"]["note"] == "plain text"  # non-literals stay strings


def test_config_formats(tmp_path):
    (tmp_path / "c.json").write_text(json.dumps(BASE))
    (tmp_path / "c.py").write_text("model = dict(cfg=dict(coarse_npoint=196))\ndataloader = %r\ntest = %r\nmisc = %r\nbop_eval = %r\n"
                                   % (BASE["dataloader"], BASE["test"], BASE["misc"], BASE["bop_eval"]))
    (tmp_path / "d.py").write_text("cfg = %r\n" % (BASE,))
    import yaml

    (tmp_path / "c.yaml").write_text(yaml.safe_dump(BASE))
    for name in ("c.json", "c.py", "d.py", "c.yaml"):
        assert cli.load_config(str(tmp_path / name)) == BASE, name
    with pytest.raises(ValueError):
        cli.load_config(str(tmp_path / "c.toml"))


def test_print_plan_and_missing_checkpoint(tmp_path):
    cfgf = tmp_path / "c.json"
    cfgf.write_text(json.dumps(BASE))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "unopose_amd.cli", "--config-file", str(cfgf), "--num-gpus", "8", "--print-plan", "misc.load_from=/x/ckpt_12.pth",
                        "test.amp.enabled=True"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["save_path"] == "output/unopose/inference_ckpt_12/ycbv/result_ycbv-test.csv" and plan["amp"] is True and plan["num_gpus"] == 8
    # save_unopose.sh:15-18: a checkpoint that does not exist ends the run before anything is built
    r = subprocess.run([sys.executable, "-m", "unopose_amd.cli", "--config-file", str(cfgf), "misc.load_from=/does/not/exist.pth"],
                       capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 1 and "does not exist" in r.stderr
