"""Writes tests/golden/host_traces.json by running tests/_host_scenarios.py against the
REFERENCE's own host modules (BlockManager, Sequence, Scheduler, SamplingParams), imported from
/root/reference. Run in the build container only; the GPU box has no reference and only reads the
committed JSON. TEST INFRASTRUCTURE: nothing in light_vllm_amd imports this.

    python oracle/gen_golden.py
"""
from __future__ import annotations

import json
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def reference_namespace():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE)
    from lightvllm.engine.block_manager import BlockManager
    from lightvllm.engine.scheduler import Scheduler
    from lightvllm.engine.sequence import Sequence
    from lightvllm.sampling_params import SamplingParams
    return SimpleNamespace(BlockManager=BlockManager, Scheduler=Scheduler, Sequence=Sequence,
                           SamplingParams=SamplingParams)


def main():
    sys.path.insert(0, ROOT)
    from tests._host_scenarios import run_all
    traces = run_all(reference_namespace())
    out = os.path.join(ROOT, "tests", "golden", "host_traces.json")
    with open(out, "w") as f:
        json.dump(traces, f, separators=(",", ":"))
    print(f"wrote {out} ({os.path.getsize(out)} bytes)")


if __name__ == "__main__":
    main()
