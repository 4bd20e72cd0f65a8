"""Fold the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (CSV from scripts/rocpd_pmc.py) into the per-kernel
HBM-bytes-per-launch table bench.py reads (profiles/<tag>_pmc_summary.json).
    pmc_summary.py <fetch.csv> <write.csv> <out.json> [<gemm_fetch_dispatches.csv> <gemm_write_dispatches.csv> [<calib_fetch.csv>]]
The optional per-dispatch CSVs (`rocpd_pmc.py <db> gemm --dispatches` over scripts/pmc_gemm.py) give the four ViT linears one
entry each ("gemm_shapes": attributed by launch order); the calibration CSV (scripts/ubench/fetch_calib.py) records what
fraction of a known 1 GiB read FETCH_SIZE reports for LDS-DMA and for global loads."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SHORT = ["token_attn_kernel<true", "vit_attn_kernel", "group_points_lds_kernel", "pe_group_mlp_max_bf16x3_kernel",
         "ball_query_kernel", "geo_embed_kernel", "geo_embed_table_kernel", "geo_knn_kernel", "gemm256_kernel<1, false, false>", "gemm256_kernel<0, false, false>",
         "fine_assign_kernel<0>", "fine_assign_kernel<1>", "fine_assign_kernel<2>"]
# gemm256_kernel<1, false, false>: fc1 + GELU (M = 87936, 768 -> 3072); <0, false, false>: MEAN over the qkv / proj / fc2 launches of
# scripts/pmc_kernels.py (three shapes, three launches each); the per-shape numbers are under "gemm_shapes"


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        for s in SHORT:
            if s in r["Kernel"]:
                out[s.replace("<true", "<true>")] = (float(r["MeanValue"]), float(r["MeanDurationNs"]))
    return out


def load_dispatches(path, prefix):
    """-> [(value, duration_ns)] of the dispatches whose kernel name starts with `prefix`, in launch order."""
    return [(float(r["Value"]), float(r["DurationNs"])) for r in csv.DictReader(open(path)) if prefix in r["Kernel"]]


def gemm_shapes(fetch_csv, write_csv):
    from pmc_gemm import FOLD, ORDER, REPS

    out = {}
    f, w = load_dispatches(fetch_csv, ", false, false>("), load_dispatches(write_csv, ", false, false>(")  # gemm256_kernel<EPI, false, false>: bf16
    assert len(f) == len(w) == REPS * len(ORDER), (len(f), len(w))
    for i, (name, K, N, gelu) in enumerate(ORDER):
        fs, ws = f[i * REPS:(i + 1) * REPS], w[i * REPS:(i + 1) * REPS]
        fk, wk = sum(v for v, _ in fs) / REPS, sum(v for v, _ in ws) / REPS
        M = 64 * 1374
        out[name] = dict(M=M, K=K, N=N, fetch_KiB_raw=fk, write_KiB=wk, hbm_bytes_per_launch=(2.0 * fk + wk) * 1024.0,
                         # (fold: + the fp32 residual stream read and written by proj / fc2; the row partials either side)
                         algorithmic_bytes=2.0 * (M * K + N * K + M * N) + 4.0 * N + (8.0 * M * N + 8.0 * M * 3 if FOLD and N == 768 else 8.0 * M * 3 + 4.0 * N if FOLD else 0.0),
                         form=("EPI 5 (residual epilogue)" if N == 768 else "EPI 6 / 7 (LayerNorm in the epilogue)") if FOLD else "bias / bias + GELU",
                         mean_duration_us_under_pmc=sum(d for _, d in fs) / REPS / 1e3)
    f3 = load_dispatches(fetch_csv, ", false, true>(")  # gemm256_kernel<EPI, false, true>: fp32-class
    w3 = load_dispatches(write_csv, ", false, true>(")
    x3 = {}
    if len(f3) == len(w3) == 2 * REPS:
        for i, name in enumerate(("qkv", "fc1")):
            K, N = (768, 2304) if name == "qkv" else (768, 3072)
            fk = sum(v for v, _ in f3[i * REPS:(i + 1) * REPS]) / REPS
            wk = sum(v for v, _ in w3[i * REPS:(i + 1) * REPS]) / REPS
            M = 64 * 1374
            x3[name] = dict(M=M, K=K, N=N, fetch_KiB_raw=fk, write_KiB=wk, hbm_bytes_per_launch=(2.0 * fk + wk) * 1024.0,
                            algorithmic_bytes=4.0 * (M * K + N * K + M * N) + 4.0 * N)
    return out, x3


def main(fetch_csv, write_csv, out_json, gemm_fetch=None, gemm_write=None, calib=None):
    f, w = load(fetch_csv), load(write_csv)
    kernels = {}
    for k in f:
        if k in w:
            kernels[k] = dict(fetch_KiB_raw=f[k][0], write_KiB=w[k][0],
                              hbm_bytes_per_launch=(2.0 * f[k][0] + w[k][0]) * 1024.0,
                              mean_duration_us_under_pmc=f[k][1] / 1e3)
    doc = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python3 "
           "scripts/pmc_kernels.py 32` (B=32 pairs, bench shapes; vit_attn at 64 images x 1374 tokens) and `python3 scripts/pmc_gemm.py 32` "
           "(gemm_shapes: the four ViT linears, per-dispatch values attributed by launch order). Counter unit KiB. "
           "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads exactly 1/2 of the bytes of wide (16 B/lane) "
           "coalesced reads -- global_load and LDS-DMA alike, re-measured here under `fetch_calibration` -> doubled; "
           "WRITE_SIZE uncorrected (matches the known output sizes to <0.5 %).")
    out = dict(_doc=doc, kernels=kernels)
    if gemm_fetch and gemm_write:
        out["gemm_shapes"], out["gemm_f32x3_shapes"] = gemm_shapes(gemm_fetch, gemm_write)
    if calib:
        c = {}
        for r in csv.DictReader(open(calib)):
            mode = "lds_dma(buffer_load_dwordx4 ... lds)" if "<0>" in r["Kernel"] else "global_load_dwordx4"
            c[mode] = dict(fetch_KiB_raw=float(r["MeanValue"]), bytes_read=float(1 << 30),
                           reported_fraction=float(r["MeanValue"]) * 1024.0 / float(1 << 30), dispatches=int(r["Dispatches"]))
        out["fetch_calibration"] = c
    json.dump(out, open(out_json, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:40s} {v['hbm_bytes_per_launch']/1e6:10.2f} MB/launch")
    for k, v in out.get("gemm_shapes", {}).items():
        print(f"gemm {k:6s} {v['hbm_bytes_per_launch']/1e6:10.2f} MB/launch   algorithmic {v['algorithmic_bytes']/1e6:10.2f} MB")
    for k, v in out.get("fetch_calibration", {}).items():
        print(f"calibration {k}: FETCH_SIZE reports {v['reported_fraction']:.3f} of the bytes read")


if __name__ == "__main__":
    main(*sys.argv[1:])
