"""The training step with the own-kernel thresholds of the trainable linears lowered (ops.TRAIN_OWN_GEMM_MIN_FLOP, ops.TRAIN_OWN_WGRAD_MIN_ROWS):
python scripts/ubench/train_thresholds.py <min_flop> <min_rows> [steps]   (one process per measurement; also the target of a rocprofv3 run)"""
import io, json, os, sys
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from unopose_amd import ops
ops.TRAIN_OWN_GEMM_MIN_FLOP = float(sys.argv[1])
ops.TRAIN_OWN_WGRAD_MIN_ROWS = int(sys.argv[2])
steps = sys.argv[3] if len(sys.argv) > 3 else "10"
tag = f"min_flop {sys.argv[1]} min_rows {sys.argv[2]}"
sys.argv = ["bench.py", "--train", "--steps", steps, "--warmup", "3"]
buf = io.StringIO()
with redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(f"{tag}: {d['ms_per_step']:.2f} ms/step  loss {d['loss_first_last']}", flush=True)
