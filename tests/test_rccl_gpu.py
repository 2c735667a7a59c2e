"""RCCL itself on this stack (torch.distributed backend "nccl"), as far as one GPU allows: a one-rank
group initialises on the device, reduces, gathers objects and tensors, and an all-reduce captured
into a hipGraph replays. That is the API path bench.py --gpus N and the engine fall back to when the
direct all-reduce is unavailable (reference: dist.init_process_group("nccl", ...) at
lightvllm/engine/model_runner.py:54 and the captured forward at :523-536). Run in a child process so
the group does not outlive the test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_rccl_group_and_graph_capture(gpu):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29657")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_probe.py")], capture_output=True, text=True,
                          timeout=240, env=env)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = next(l for l in proc.stdout.splitlines() if l.startswith("RESULT "))
    res = json.loads(line[len("RESULT "):])
    assert res == {"backend": "nccl", "eager_all_reduce": True, "all_gather_object": True,
                   "graph_capture_all_reduce": True, "all_gather": True}, res
