"""Command-line entry of the test run (SURVEY.md 8(b) "CLI / config boundary"): what ``core/unopose/save_unopose.sh`` ->
``main_unopose.py --config-file CFG --num-gpus N test.save_results_only=True misc.load_from=CKPT [key=value ...]`` ->
``engine.do_save_results`` (core/unopose/engine/engine.py:36-72) does, on this package:

    python -m unopose_amd.cli --config-file CFG --num-gpus N misc.load_from=CKPT [key=value ...]

* CFG: a ``.json`` / ``.yaml`` file, or a ``.py`` file that leaves a mapping named ``cfg`` (or the LazyConfig layout's top-level
  names ``model``, ``dataloader``, ``test``, ``misc``, ``bop_eval``) -- the keys ``do_save_results`` reads: ``model.cfg`` (or ``model``),
  ``dataloader.test.dataset`` (the provider's fields + ``eval_dataset_name`` + ``detetion_path``, spelled as in the reference),
  ``test.amp.enabled``, ``test.instance_batch_size``, ``misc.output_dir``, ``misc.load_from``, ``misc.exp_name``, ``bop_eval.split``;
  ``key=value`` overrides use dotted keys and Python literals, like detectron2's LazyConfig.apply_overrides.
* result path = the reference's: ``<misc.output_dir>/inference_<checkpoint stem>/<dataset>/result<exp_name>_<dataset>-<split>.csv`` plus
  the sibling ``.json`` (runner.inference_and_save).
* ``--num-gpus N`` > 1 starts N ranks (one per GPU, RCCL) before this process touches a GPU; images are sharded by the
  InferenceSampler rule and rows gathered to rank 0.
* two deliberate differences from ``engine.do_save_results``: (1) ``test.amp.enabled=True`` selects **bf16** autocast here (the reference's
  ``torch.cuda.amp.autocast`` is fp16; bf16 is what the MI355X kernels are built and parity-tested for), ``False`` = fp32 as in the
  reference; (2) the reference goes on to call the BOP evaluation (``bop_eval_utils``) after saving -- this entry stops at the CSV
  unless ``--eval`` is given, which scores it with ``unopose_amd.bop_eval`` (VSD + MSSD + MSPD -> AR, HIP depth renderer).
* extras beyond the reference's line: ``--pipeline`` (two forwards in flight), ``--ref-cache`` (reference views encoded once),
  ``--print-plan`` (resolve config and paths, touch no GPU: used by the CPU tests)."""
import argparse
import ast
import json
import os
import os.path as osp
import subprocess
import sys

from .model.config import Cfg


def load_config(path):
    if path.endswith(".json"):
        with open(path) as f:
            return json.load(f)
    if path.endswith((".yaml", ".yml")):
        import yaml

        with open(path) as f:
            return yaml.safe_load(f)
    if path.endswith(".py"):
        scope = {"__file__": path}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), scope)
        if isinstance(scope.get("cfg"), dict):
            return dict(scope["cfg"])
        return {k: scope[k] for k in ("model", "dataloader", "test", "misc", "bop_eval", "train") if k in scope}
    raise ValueError(f"unsupported config file {path} (json / yaml / py)")


def apply_overrides(cfg, overrides):
    """``a.b.c=value`` with Python-literal values (strings that are not literals stay strings)."""
    for item in overrides:
        if "=" not in item:
            raise ValueError(f"override {item!r} is not key=value")
        key, val = item.split("=", 1)
        try:
            val = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            pass
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise ValueError(f"override {key}: {p} is not a mapping")
        node[parts[-1]] = val
    return cfg


def result_paths(cfg, iteration=None):
    """engine.py:37-52 -> (directory, csv path)."""
    cfg = Cfg(cfg)
    dataset_name = cfg.dataloader.test.dataset.eval_dataset_name
    sub = f"inference_iter_{iteration}" if iteration is not None else f"inference_{osp.splitext(osp.basename(cfg.misc.load_from))[0]}"
    out_dir = osp.join(cfg.misc.output_dir, sub, dataset_name)
    name = f"result{cfg.misc.get('exp_name', '')}_{dataset_name}-{cfg.bop_eval.split}.csv"
    return out_dir, osp.join(out_dir, name)


def _launch_ranks(n, argv, poll_s=0.2):
    """N fresh interpreters with the torchrun environment contract; the parent has not initialised HIP.  All children are polled:
    when any one exits non-zero the others are terminated and that code is returned (as torchrun does) -- a rank that dies must not
    leave the rest blocked in a collective until the watchdog fires."""
    import socket
    import time

    with socket.socket() as s:  # a free port (bind-and-close: the window before rank 0 re-binds it is small; MASTER_PORT overrides)
        s.bind(("127.0.0.1", 0))
        port = os.environ.get("MASTER_PORT") or str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, "-m", "unopose_amd.cli"] + argv, env=env))
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return abs(bad[0])
        if all(c == 0 for c in codes):
            return 0
        time.sleep(poll_s)


def load_checkpoint(model, path):
    """A checkpoint as MyCheckpointer writes it ({"model": state_dict, ...}) or a bare state dict; strict."""
    import torch

    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=True)
    return model


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m unopose_amd.cli", description=__doc__.split("\n\n")[0],
                                 epilog="Differences from engine.do_save_results: test.amp.enabled=True means bf16 autocast (reference: fp16); "
                                        "the BOP evaluation is not started after saving (see unopose_amd.bop_eval).")
    ap.add_argument("--config-file", required=True)
    ap.add_argument("--num-gpus", type=int, default=1)
    ap.add_argument("--eval-only", action="store_true", help="accepted for compatibility with main_unopose.py")
    ap.add_argument("--pipeline", action="store_true")
    ap.add_argument("--ref-cache", action="store_true")
    ap.add_argument("--print-plan", action="store_true")
    ap.add_argument("opts", nargs="*", help="key=value overrides")
    args = ap.parse_args(argv)
    cfg = apply_overrides(load_config(args.config_file), args.opts)
    out_dir, save_path = result_paths(cfg)
    c = Cfg(cfg)
    if args.print_plan:
        print(json.dumps(dict(save_path=save_path, dataset=c.dataloader.test.dataset.eval_dataset_name, checkpoint=c.misc.load_from,
                              amp=bool(c.test.amp.enabled), instance_batch_size=c.test.instance_batch_size, num_gpus=args.num_gpus)))
        return 0
    if not osp.exists(c.misc.load_from):  # save_unopose.sh:15-18
        print(f"{c.misc.load_from} does not exist.", file=sys.stderr)
        return 1
    if args.num_gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _launch_ranks(args.num_gpus, list(argv if argv is not None else sys.argv[1:]))

    import torch
    import torch.distributed as dist

    from .model import UNOPose
    from .provider import BOPTestsetOneRef, collate_image
    from .runner import ReferenceCache, inference_and_save

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import datetime

        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=10))
    torch.set_grad_enabled(False)
    model_cfg = cfg["model"].get("cfg", cfg["model"]) if isinstance(cfg.get("model"), dict) else cfg["model"]
    model = UNOPose(model_cfg)
    if world == 1 or int(os.environ.get("RANK", "0")) == 0:
        load_checkpoint(model, c.misc.load_from)
    model = model.to(dev).eval()
    if world > 1:  # rank 0 read the checkpoint; the weights travel once over RCCL / xGMI (north_star: "RCCL broadcast of DINOv2 weights")
        from .runner import broadcast_module_

        broadcast_module_(model, src=0)
    dcfg = dict(cfg["dataloader"]["test"]["dataset"])
    name, det_path = dcfg.pop("eval_dataset_name"), dcfg.pop("detetion_path", None)
    dataset = BOPTestsetOneRef(dcfg.get("cfg", dcfg), name, det_path)

    class Images:  # batch dim 1, like DataLoader(batch_size=1) over the dataset
        dets = getattr(dataset, "dets", None)

        def __len__(self):
            return len(dataset)

        def __getitem__(self, i):
            return collate_image(dataset[i])

    os.makedirs(out_dir, exist_ok=True)
    amp = bool(c.test.amp.enabled)
    pipe = None
    if args.pipeline:
        from .pipeline import PipelinedForward

        pipe = PipelinedForward(model, depth=2 if amp else 1, autocast_dtype=torch.bfloat16 if amp else None)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp and pipe is None):
        lines = inference_and_save(model, Images(), save_path, instance_batch_size=c.test.instance_batch_size, device=dev,
                                   ref_cache=ReferenceCache(model) if args.ref_cache else None, pipeline=pipe)
    if pipe is not None:
        pipe.close()
    if lines is not None:
        print(f"{len(lines)} estimates -> {save_path}")
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
