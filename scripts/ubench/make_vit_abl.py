"""Writes scripts/ubench/vit_attn_abl.hip = unopose_amd/csrc/vit_attn.hip with ablation switches (-DABL=n)
and builds vit_abl<n>.so for scripts/ubench/vit_abl.py.  Timing only: ablated results are wrong."""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
s = open(os.path.join(root, "unopose_amd/csrc/vit_attn.hip")).read()


def rep(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new)


rep('#include "common.h"', '#include "common.h"\n#ifndef ABL\n#define ABL 0\n#endif')
rep("s[qb][r] = __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -m_use));",
    "s[qb][r] = (ABL == 1) ? fmaf(s[qb][r], scale_log2e, -m_use) : __builtin_amdgcn_exp2f(fmaf(s[qb][r], scale_log2e, -m_use));")
rep("          o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][s2].v, o[qb][t], 0, 0, 0);",
    "          if (ABL == 2) { o[qb][t][0] += (float)vf[0] + (float)pf[qb][s2].v[0]; }\n"
    "          else o[qb][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb][s2].v, o[qb][t], 0, 0, 0);")
rep("      for (int qb = 0; qb < QB; ++qb) s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb], 0, 0, 0);",
    "      for (int qb = 0; qb < QB; ++qb) {\n        if (ABL == 3) { s[qb][ks] += (float)kf[0] + (float)qf[qb][ks][0]; }\n"
    "        else s[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb], 0, 0, 0);\n      }")
rep("      const bool grow = mx > m_run[qb] + VA_DEFER;", "      const bool grow = (ABL != 4) && mx > m_run[qb] + VA_DEFER;")
rep("    if (__any(moved)) {  // rare after the first tile", "    if (ABL != 5 && ABL != 4 && __any(moved)) {  // rare after the first tile")
rep("      __syncthreads();\n    }\n  }\n  __syncthreads();  // every wave is done", "      if (ABL != 7) __syncthreads();\n    }\n  }\n  __syncthreads();  // every wave is done")
rep("      if (c + 1 < nchunks) {  // the other buffer", "      if (ABL != 6 && c + 1 < nchunks) {  // the other buffer")
# 8: no LDS fragment reads (K fragments and V fragments come from registers set once)
rep("      const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(buf + (kt + col) * VA_LDK + ks * 16 + hb * 8);",
    "      const bf16x8 kf = (ABL == 8) ? qf[0][ks] : *reinterpret_cast<const bf16x8 *>(buf + (kt + col) * VA_LDK + ks * 16 + hb * 8);")
rep("      union { bf16x8 v; s16x4 h4[2]; } vf;\n", "      union { bf16x8 v; s16x4 h4[2]; } vf;\n      if (ABL == 8) return qf[0][t];\n")
open(os.path.join(here, "vit_attn_abl.hip"), "w").write(s)
for n in (0, 1, 2, 3, 4, 5, 6, 7, 8):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared",
           f"-DABL={n}", f"-I{root}/include", f"-I{root}/unopose_amd/csrc", os.path.join(here, "vit_attn_abl.hip"),
           f"{root}/unopose_amd/csrc/abi.hip", "-o", os.path.join(here, f"vit_abl{n}.so")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-2000:])
print("built")
