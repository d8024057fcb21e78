#!/usr/bin/env python3
"""What a decode launch does beside the acoustic stage: from a rocprofv3 --kernel-trace CSV of the PIPELINED bench run, per queue the
median duration of each decode kernel family and the median gap from the end of the previous kernel of the same queue to its start,
split by whether an acoustic kernel (GEMM / convolution / DiT attention) was running on another queue at that moment.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/lane_timeline.py /tmp/kt
"""
import bisect
import csv
import glob
import os
import statistics
import sys

root = sys.argv[1]
paths = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
assert paths, "no kernel trace found"
rows = []
for p in paths:
    with open(p) as f:
        r = csv.DictReader(f)
        for d in r:
            rows.append((int(d["Start_Timestamp"]), int(d["End_Timestamp"]), d.get("Queue_Id", d.get("Stream_Id", "0")), d["Kernel_Name"]))
rows.sort()
print(f"{len(rows)} dispatches, {len(set(r[2] for r in rows))} queues")
ACOUSTIC = ("gemm_bf16x3", "conv1d_", "flash_attn_planes", "aa_act", "ada_rms")
DECODE = ("gemv_fx", "decode_attn", "sample_greedy", "embed_step", "advance_state", "rows_norm")
ac = [(s, e) for s, e, q, n in rows if any(a in n for a in ACOUSTIC)]
ac_starts = [s for s, _ in ac]
# busy intervals of the acoustic queue(s), merged
merged = []
for s, e in ac:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
m_starts = [m[0] for m in merged]


def acoustic_busy(t):
    i = bisect.bisect_right(m_starts, t) - 1
    return i >= 0 and merged[i][1] > t


last_end = {}
stats = {}
for s, e, q, n in rows:
    fam = next((d for d in DECODE if d in n), None)
    prev = last_end.get(q)
    last_end[q] = e
    if fam is None or prev is None:
        continue
    key = (fam, acoustic_busy(s))
    st = stats.setdefault(key, ([], []))
    st[0].append((e - s) / 1e3)
    st[1].append(max(0, s - prev) / 1e3)
print(f"{'family':16s} {'beside acoustic':16s} {'launches':>9s} {'median us':>10s} {'p90 us':>8s} {'median gap us':>14s} {'p90 gap us':>11s}")
for (fam, busy), (dur, gap) in sorted(stats.items()):
    dur.sort(); gap.sort()
    print(f"{fam:16s} {str(busy):16s} {len(dur):9d} {statistics.median(dur):10.1f} {dur[int(0.9 * len(dur))]:8.1f} {statistics.median(gap):14.1f} {gap[int(0.9 * len(gap))]:11.1f}")
