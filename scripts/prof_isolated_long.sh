R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python3 $R/bench.py --img 518 --steps 38 --warmup 2 --inflight 1 --no-cpu-baseline --no-roofline --no-fp32 --no-extra > /dev/null 2>&1
python3 $R/scripts/rocpd_stats.py $(find /tmp/ks -name "*.db" | head -1) 70 > $R/gpurun_out/r06_kernel_stats_isolated_b32_s518_40fwd.csv
python3 - <<'P'
import csv, os
p = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out/r06_kernel_stats_isolated_b32_s518_40fwd.csv")
rows = list(csv.reader(l for l in open(p) if not l.startswith("#")))[1:]
tot = sum(float(r[2]) for r in rows); own = sum(float(r[2]) for r in rows if "unopose::" in r[0])
print(open(p).readline().strip()); print("listed %.1f ms, own kernels %.2f %%" % (tot / 1e6, 100 * own / tot))
P
