"""GPU: bench.py keeps its output contract (one JSON line on stdout with the driver's keys, the `roofline` and `cpu_baseline`
objects) -- run at a reduced size so it finishes in about two minutes; the weights are the full-size architecture."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "0", "--batch", "4",
           "--codes", "24", "--text-tokens", "40", "--prompt-frames", "120", "--cpu-codes", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly ONE line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 0 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and 0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1
    assert cb["mel_l1_vs_gpu"] <= 1e-3
    # the B = 16 leg of the CPU baseline is derived from the measured stage times of the one-utterance sample (and says so)
    assert set(cb["stage_seconds"]) == {"gpt", "s2mel", "bigvgan"} and cb["batch16_estimate"]["value"] >= cb["value"]
    # The parity rule of the headline mode (bf16 KV cache) and of the exact mode (fp32 cache), 4 utterances x 64 teacher-forced steps on
    # the full-size model: logit noise within the stated bound, every differing argmax a near-tie of the oracle, the free-running decode
    # leaving the oracle's sequence only there.  Codes may differ in the bf16 mode -- then only at such a step.
    par = cb["decode_parity"]
    assert set(par) == {"bf16", "f32"}
    for mode, pr in par.items():
        assert pr["utterances"] == 4 and pr["steps"] == 64
        assert pr["within_bound"] is True and pr["max_abs_logit_diff"] <= pr["logit_bound"], (mode, pr)
        assert pr["codes_match_rate_teacher_forced"] >= 0.97
        assert pr["first_difference_step_free_running"] == pr["first_mismatch_step_teacher_forced"]
    assert par["f32"]["max_abs_logit_diff"] < par["bf16"]["logit_bound"]
    assert cb["codes_match_rate"] == par["bf16"]["codes_match_rate_teacher_forced"]
    if not cb["greedy_codes_equal_vs_gpu"]:
        assert cb["codes_first_difference_step"] is not None and cb["oracle_score_margin_at_that_step"] <= 2 * par["bf16"]["logit_bound"]
    em = d["exact_mode"]
    assert em["gpt_kv"] == "f32" and em["value"] > 0 and em["codes_match_rate"] >= 0.97 and em["max_abs_logit_diff"] <= em["logit_bound"]
