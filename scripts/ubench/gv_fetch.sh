#!/bin/bash
# usage: gv_fetch.sh <variant>...   fabric read traffic (FETCH_SIZE x 2, the gfx950 correction of scripts/pmc_summary.py) per launch of
# the four ViT linears for gemm_var.py builds; one rocprofv3 --pmc pass (kernel trace only) per variant
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/gf; rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/gf -o gf -- python3 $R/scripts/ubench/gv_fetch.py $v > /dev/null 2>&1
  python3 $R/scripts/rocpd_pmc.py $(find /tmp/gf -name '*.db' | head -1) gemm256 --dispatches > /tmp/gf.csv
  python3 - "$v" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open('/tmp/gf.csv'))]
names = (("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768))
M = 64 * 1374
assert len(rows) == 12, len(rows)
out = []
for i, (n, K, N) in enumerate(names):
    rs = rows[3 * i:3 * i + 3]
    f = sum(float(r["Value"]) for r in rs) / 3 * 2 * 1024 / 1e6
    d = sum(float(r["DurationNs"]) for r in rs) / 3 / 1e3
    alg = 2.0 * (M * K + N * K) / 1e6
    out.append(f"{n} fetch {f:6.0f} MB (operands {alg:4.0f}, x{f / alg:4.2f}; total/alg {(f + 2.0 * M * N / 1e6) / (alg + 2.0 * M * N / 1e6):4.2f}) {d:6.1f} us")
print(f"{sys.argv[1]:12s} " + " | ".join(out))
PY
done
