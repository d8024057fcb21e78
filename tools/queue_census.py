#!/usr/bin/env python3
"""Which HSA queue did each kernel family run on?  Reads a rocprofv3 --kernel-trace CSV (argv[1]) and prints, per (queue, stream),
the number of launches of the decode kernels (gemv_fx / decode_attn) and of the acoustic kernels (LDS-DMA GEMM, attention, convs)."""
import collections
import csv
import sys

rows = collections.defaultdict(collections.Counter)
with open(sys.argv[1]) as f:
    r = csv.DictReader(f)
    for row in r:
        n = row["Kernel_Name"]
        fam = "decode" if ("gemv_fx" in n or "decode_attn" in n) else "acoustic" if ("gemm_bf16x3_v2" in n or "flash_attn" in n or "conv1d" in n) else "other"
        rows[(row["Queue_Id"], row.get("Stream_Id", "-"), row.get("Thread_Id", "-"))][fam] += 1
for (q, s, t), c in sorted(rows.items()):
    print(f"queue {q} stream {s} thread {t}: {dict(c)}")
