"""pipeline.WHOLE_INTERNAL_OVERLAP on / off for the legs that pipeline whole forwards (224 x 224: batch 32, the contract's batch 16 in both precisions), one process
per measurement.  usage: python scripts/ubench/whole_overlap_ab.py [--one True|False]"""
import io, json, os, subprocess, sys
from contextlib import redirect_stdout
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    import bench
    import unopose_amd.pipeline as pl
    pl.WHOLE_INTERNAL_OVERLAP = sys.argv[2] == "True"
    sys.argv = ["bench.py", "--img", "224", "--no-cpu-baseline", "--no-roofline", "--steps", "40", "--warmup", "5"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    c = d["contract_224"]
    print(f"WHOLE_INTERNAL_OVERLAP={pl.WHOLE_INTERNAL_OVERLAP}: 224 bf16 {d['value']:.1f}  fp32 {d['fp32']['value']:.1f}  contract fp32 {c['fp32']['value']:.1f} bf16 {c['bf16']['value']:.1f}  "
          f"ref_cached {d['ref_cached']['value']:.1f} pairs/s", flush=True)
else:
    for rep in range(3):
        for v in ("False", "True"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", v], stderr=subprocess.DEVNULL)
