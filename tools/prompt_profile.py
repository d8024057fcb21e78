"""Host-side profile of one prompt encode (bench.py --workload prompt's step): where the wall time of a cache miss goes."""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "index-tts_amd")]
sys.argv = ["bench.py", "--workload", "prompt"]
import argparse, torch
import bench
ap = argparse.Namespace(workload="prompt", gpt_weights="f32", gpt_kv=None, codes=200, text_tokens=40, prompt_seconds=15.0, batch=0)
from indextts_amd import _lib
_lib.load()
step, _, _, _, _, _ = bench.build_prompt_or_infer(ap, 1, 0, torch.device("cuda", 0))
for _ in range(3):
    step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step(); torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    step(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
