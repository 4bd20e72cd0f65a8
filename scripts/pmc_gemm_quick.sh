#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the four ViT linears only (scripts/pmc_gemm.py), printed per shape: a quick form of collect_profiles.sh's PMC part
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  lc=$(echo $c | tr A-Z a-z)
  rm -rf /tmp/pmcg_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcg_$c -o pmc -- python3 $R/scripts/pmc_gemm.py 32 > /dev/null 2>&1
  python3 $R/scripts/rocpd_pmc.py $(find /tmp/pmcg_$c -name "*.db" | head -1) gemm --dispatches > /tmp/q_$lc.csv
done
python3 - <<'P'
import csv, sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from pmc_gemm import ORDER, REPS
def load(p): return [(float(r["Value"]), float(r["DurationNs"])) for r in csv.DictReader(open(p)) if ", false, false>(" in r["Kernel"]]
f, w = load("/tmp/q_fetch_size.csv"), load("/tmp/q_write_size.csv")
for i, (name, K, N, g) in enumerate(ORDER):
    fk = sum(v for v, _ in f[i * REPS:(i + 1) * REPS]) / REPS; wk = sum(v for v, _ in w[i * REPS:(i + 1) * REPS]) / REPS
    print(f"{name:5s} fetch x2 {2 * fk * 1024 / 1e6:7.1f} MB  write {wk * 1024 / 1e6:7.1f} MB  duration under PMC {sum(d for _, d in f[i * REPS:(i + 1) * REPS]) / REPS / 1e3:7.1f} us")
P
