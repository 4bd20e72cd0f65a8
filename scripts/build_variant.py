"""Build a variant of libunopose_hip.so with extra compiler flags next to the product library, for same-box A/Bs:
    python scripts/build_variant.py noslp -fno-slp-vectorize      -> unopose_amd/libunopose_hip_noslp.so
    UNOPOSE_LIB=unopose_amd/libunopose_hip_noslp.so python bench.py ...
Same sources, same per-file flags as unopose_amd/build.py.  An argument `file.hip=path` compiles `path` in place of csrc/file.hip
(e.g. gemm.hip=scripts/ubench/gemm_r03.hip: last round's GEMM inside this round's library)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unopose_amd import build as B
name = sys.argv[1]
subst = dict(a.split("=", 1) for a in sys.argv[2:] if ".hip=" in a)
extra = [a for a in sys.argv[2:] if ".hip=" not in a]
obj_dir = os.path.join(B.CSRC, "_obj_" + name)
os.makedirs(obj_dir, exist_ok=True)
srcs = sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))
def cc(f):
    o = os.path.join(obj_dir, f[:-4] + ".o")
    subprocess.check_call([B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(f, []), *extra, "-I", B.CSRC, "-c",
                           os.path.join(ROOT, subst[f]) if f in subst else os.path.join(B.CSRC, f), "-o", o])
    return o
with ThreadPoolExecutor(6) as ex:
    objs = list(ex.map(cc, srcs))
out = os.path.join(B.HERE, f"libunopose_hip_{name}.so")
subprocess.check_call([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", out])
print(out)
