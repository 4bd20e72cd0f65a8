R=${GRAFT_REPO_ROOT:-/root/repo}
for n in gsnopref prod gsnopref prod; do
  lib=$R/unopose_amd/libunopose_hip_$n.so; [ $n = prod ] && lib=$R/unopose_amd/libunopose_hip.so
  UNOPOSE_LIB=$lib python3 $R/scripts/gemm_shapes.py 518 2>/dev/null > /tmp/gs_$n.json
  python3 - $n <<'PY'
import json, sys
d = json.load(open(f"/tmp/gs_{sys.argv[1]}.json"))
small = {k: v for k, v in d.items() if v[1] < 60}
print(sys.argv[1], "small-GEMM time per forward: %.3f ms" % (sum(v[0] * v[1] for v in small.values()) / 1e3), " ".join(f"{v[1]:.1f}" for v in small.values()))
PY
done
