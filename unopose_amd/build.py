"""Build libunopose_hip.so (gfx950) in-tree with hipcc.

``python -m unopose_amd.build`` or ``unopose_amd.build.build()``.  hipcc
cross-compiles without a GPU; the resulting ``unopose_amd/libunopose_hip.so``
is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
SO = os.path.join(HERE, "libunopose_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: fp32 distance expressions must round exactly as written so
# FPS / ball_query indices are bit-identical to oracle/ (DESIGN.md "fp-contract").
# -fno-slp-vectorize -fno-vectorize: NO packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).  hipcc -O2 / -O3 forms them
# out of scalar fp32 code (97 in query_lrf_group alone); kernels that contain them (precisely: v_pk_mul_f32 with an op_sel modifier,
# isolated by re-assembling the ISA, scripts/ubench/asm_var.py) return wrong values for a few elements per
# launch whenever waves of ANOTHER kernel issuing MFMAs share their CU -- 23 of 30 launches beside a neighbour that does nothing but
# v_mfma, 0 of 30 when built with this flag or at -O1, and never beside VALU / LDS / memory / barrier neighbours
# (scripts/ubench/coresidency_matrix.py, geom_var.py; the loop vectoriser forms a few more, hence both flags).
# Packed fp32 brings no throughput on gfx950, so the
# flags cost nothing (898 vs 892 pairs/s, inside the run-to-run spread).  tests/test_abi.py compiles every source to ISA with
# these flags and fails if a packed-fp32 instruction is left.
# UNOPOSE_EXTRA_HIPCC_FLAGS come LAST (the last -O wins) and are part of the staleness key below: a changed value rebuilds everything.
FLAGS = ["-O3", "-fno-slp-vectorize", "-fno-vectorize", "-std=c++17", "-fPIC", "-ffp-contract=off",
         f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", *os.environ.get("UNOPOSE_EXTRA_HIPCC_FLAGS", "").split()]
# Dense-math kernels: `nnan` lets fmaxf lower to ONE v_max_f32 instead of canonicalise + max (the PE tile
# loop had 288 v_max for 112 logical maxima).  Not applied to the index-producing files (pointnet2, geom,
# posehead), whose NaN behaviour follows the reference's fminf / fmaxf semantics.
EXTRA_FLAGS = {f: ["-fno-honor-nans"] for f in
               ("pe.hip", "embed.hip", "attn.hip", "attn_f32.hip", "vit_attn.hip", "vit_attn_f32s.hip", "linattn.hip", "bn_train.hip", "fused.hip", "gemm.hip", "gemm_small.hip", "fineassign.hip")}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src, *extra))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "unopose_hip.h"))
    jobs = []
    objs = []
    # objects are only as fresh as the flags they were built with: the flag set is recorded beside them
    key = " ".join(FLAGS) + " | " + repr(sorted(EXTRA_FLAGS.items()))
    key_file = os.path.join(OBJ, "flags.txt")
    if not os.path.exists(key_file) or open(key_file).read() != key:
        force = True
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, hdrs + [os.path.abspath(__file__)]):
            jobs.append([hipcc, *FLAGS, *EXTRA_FLAGS.get(f, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr)

    for f in os.listdir(OBJ):  # objects of sources that no longer exist must not travel with the tree
        if f.endswith(".o") and os.path.join(OBJ, f) not in objs:
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    with open(key_file, "w") as fh:
        fh.write(key)
    if jobs or not os.path.exists(SO):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", SO, *objs])
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
