"""The bench's cached-reference leg alone (one process): pairs/s with reference features from ReferenceCache."""
import io, json, os, sys
from contextlib import redirect_stdout
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import unopose_amd.model.modules as mm
if len(sys.argv) > 1:
    mm.SPARSE_SINGLE_CROP = sys.argv[1] == "True"
flag = mm.SPARSE_SINGLE_CROP
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-fp32", "--no-roofline", "--steps", "5", "--warmup", "2"]
buf = io.StringIO()
with redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print("SPARSE_SINGLE_CROP", flag, "ref_cached", round(d["ref_cached"]["value"], 1), "pairs/s", round(d["ref_cached"]["ms_per_step"], 3), "ms; contract_224 fp32 / bf16", round(d["contract_224"]["fp32"]["value"], 1), round(d["contract_224"]["bf16"]["value"], 1))
