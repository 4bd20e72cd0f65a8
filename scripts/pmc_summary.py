"""Fold the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (CSV from scripts/rocpd_pmc.py) into the per-kernel
HBM-bytes-per-launch table bench.py reads (profiles/<tag>_pmc_summary.json)."""
import csv
import json
import re
import sys

SHORT = ["token_attn_kernel<true", "vit_attn_kernel", "group_points_lds_kernel", "pe_group_mlp_max_bf16x3_kernel",
         "ball_query_kernel", "geo_embed_kernel", "geo_knn_kernel", "gemm_bf16_kernel<1", "gemm_bf16_kernel<0, false",
         "fine_assign_kernel<0>", "fine_assign_kernel<1>", "fine_assign_kernel<2>"]
# gemm_bf16_kernel<1>: fc1 + GELU (M = 87936, 768 -> 3072); gemm_bf16_kernel<0, false>: MEAN over the qkv / proj / fc2 launches of
# scripts/pmc_kernels.py (three shapes, three launches each)


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        for s in SHORT:
            if s in r["Kernel"]:
                out[s.replace("<true", "<true>")] = (float(r["MeanValue"]), float(r["MeanDurationNs"]))
    return out


def main(fetch_csv, write_csv, out_json):
    f, w = load(fetch_csv), load(write_csv)
    kernels = {}
    for k in f:
        if k in w:
            kernels[k] = dict(fetch_KiB_raw=f[k][0], write_KiB=w[k][0],
                              hbm_bytes_per_launch=(2.0 * f[k][0] + w[k][0]) * 1024.0,
                              mean_duration_us_under_pmc=f[k][1] / 1e3)
    doc = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python3 "
           "scripts/pmc_kernels.py 32` (B=32 pairs, bench shapes; vit_attn at 64 images x 1374 tokens). Counter unit KiB. "
           "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads exactly 1/2 of the bytes of wide (16 B/lane) "
           "coalesced reads -> doubled; WRITE_SIZE uncorrected (matches the known output sizes to <0.5 %).")
    json.dump(dict(_doc=doc, kernels=kernels), open(out_json, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:40s} {v['hbm_bytes_per_launch']/1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main(*sys.argv[1:])
