#!/bin/bash
# Same-box A/B of the bf16 GEMM's shape policy: every GEMM shape of one forward timed with (a) 256-tiles only, (b) small tiles for
# everything, (c) the default policy.   bash scripts/gemm_policy_ab.sh [img]
R=${GRAFT_REPO_ROOT:-/root/repo}
img=${1:-518}
UNOPOSE_GEMM_SMALL_TILES=0 python3 $R/scripts/gemm_shapes.py $img > /tmp/pol_a.json 2>/dev/null
UNOPOSE_GEMM_SMALL_TILES=100000 python3 $R/scripts/gemm_shapes.py $img > /tmp/pol_b.json 2>/dev/null
python3 $R/scripts/gemm_shapes.py $img > /tmp/pol_c.json 2>/dev/null
python3 - <<'PY'
import json
a, b, c = (json.load(open(f"/tmp/pol_{x}.json")) for x in "abc")
print(f"{'shape':44s} {'n':>4s} {'256-tiles':>10s} {'128-tiles':>10s} {'default':>10s}   (us per launch)")
ta = tb = tc = tbest = 0.0
for k in a:
    n = a[k][0]
    print(f"{k:44s} {n:4d} {a[k][1]:10.1f} {b[k][1]:10.1f} {c[k][1]:10.1f}")
    ta += n * a[k][1]; tb += n * b[k][1]; tc += n * c[k][1]; tbest += n * min(a[k][1], b[k][1], c[k][1])
print(f"per forward: 256-tiles {ta/1e3:.2f} ms, 128-tiles {tb/1e3:.2f} ms, default {tc/1e3:.2f} ms, best-of {tbest/1e3:.2f} ms")
PY
