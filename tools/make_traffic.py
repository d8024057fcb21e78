#!/usr/bin/env python3
"""Build profiles/traffic_rNN.json from the two condensed PMC passes (tools/summarize_prof.py output).

    python tools/make_traffic.py <fetch_size.json> <write_size.json> <out.json> [round]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in 1024-byte units and on gfx950 FETCH_SIZE
reports half of a wide (16 B/lane) coalesced stream (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Kernel symbols are
folded into bench.py's kernel families (= kernel function names, csrc/prof.h).
"""
import json
import sys

# bench.py's kernel families are kernel function names (csrc/prof.h); families that keep their template arguments first
FAMILIES = ["conv1d_bf16x3_kernel<2, 2, 2, 2, false>", "conv1d_bf16x3_kernel<3, 2, 1, 4, false>", "conv1d_bf16x3_kernel<2, 2, 1, 4, false>", "conv1d_bf16x3_kernel<1, 4, 1, 4, false>",
            "conv1d_bf16x3_kernel<2, 2, 2, 2, true>", "conv1d_bf16x3_kernel<3, 2, 1, 4, true>", "conv1d_bf16x3_kernel<2, 2, 1, 4, true>", "conv1d_bf16x3_kernel<1, 4, 1, 4, true>",
            "conv1d_mfma_kernel<2, 2, 2, 2>", "conv1d_mfma_kernel<3, 2, 1, 4>", "conv1d_mfma_kernel<2, 2, 1, 4>", "conv1d_mfma_kernel<1, 4, 1, 4>",
            "gemm_bf16x3_big_kernel<2>", "gemm_bf16x3_big_kernel<4>", "gemm_bf16x3_v2_kernel", "gemm_bf16x3_kernel", "gemm_tn_kernel",
            "gemv_fx_combine_kernel", "gemv_fx_kernel", "decode_attn_kernel", "decode_attn16_kernel", "flash_attn_planes_kernel", "flash_attn_bf16x3_kernel",
            "flash_attn_f32_kernel", "aa_act_kernel", "ada_rms_planes512_kernel", "rows_norm_kernel", "split_planes_kernel", "sample_greedy_kernel",
            "conv_post_kernel", "cfm_pack_kernel", "embed_step_kernel"]
CATEGORY = [(f, f) for f in FAMILIES]


def fold(path, counter):
    d = json.load(open(path))
    out = {}
    for name, ctrs in d.get("counters", {}).items():
        if counter not in ctrs:
            continue
        cat = next((c for sub, c in CATEGORY if sub in name), None)
        if cat is None:
            continue
        e = out.setdefault(cat, [0, 0.0])
        e[0] += ctrs[counter]["launches"]
        e[1] += ctrs[counter]["total"]
    return d, {k: (v[0], v[1] / v[0]) for k, v in out.items()}


def main(fetch_path, write_path, out_path, rnd):
    fd, fetch = fold(fetch_path, "FETCH_SIZE")
    _, write = fold(write_path, "WRITE_SIZE")
    kernels = {}
    for cat, (n, f) in fetch.items():
        w = write.get(cat, (0, 0.0))[1]
        kernels[cat] = {"launches_in_pmc_pass": n, "fetch_size_avg_per_launch_raw": f, "write_size_avg_per_launch_raw": w,
                        "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    json.dump({"round": rnd, "workload": "bench.py (configs[2], full pipeline) --steps 1 --warmup 0",
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE under-reports 16-B/lane streams 2x)",
               "kernels": kernels}, open(out_path, "w"), indent=1)
    print("wrote", out_path, sorted(kernels))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 1)
