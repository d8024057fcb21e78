"""GPU: the multi-rank product path end to end on the HIP pipeline -- two ranks sharing the one GPU of the box (gloo carries
the collectives; the 8-GPU RCCL run is the driver's), `ShardedSynthesizer` with a `BatchPipeline` per rank, an uneven utterance list:
the waveforms rank 0 gets back, in the caller's order, equal its own single-process synthesis bit for bit."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_synthesizer_two_ranks_on_one_gpu_equals_single_process(device):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "DIST_GPU_OK" in r.stdout, r.stdout[-2000:]
