"""The multi-GPU code path of bench.py on real RCCL before the first multi-GPU launch does it.

The N > 1 lines of the benchmark go through torch.distributed's nccl backend (= RCCL on ROCm): communicator creation on
the device, `all_gather_into_tensor` on the uint8 view of a slice of the resident table (dist.all_gather_rows), the
P x P all-reduce of the page-pair shards, per-rank float gathers, barrier and destroy.  A one-GPU box cannot hold two
ranks on RCCL, but it can run every one of those calls on a process group of ONE rank: `bench.py --force-dist`.
Each run is a FRESH child process (subprocess; nothing is exec'ed from this process, which holds the GPU).

What stays unexercised: more than one rank on RCCL (ring / tree set-up over xGMI, skew between ranks); the reference
has no counterpart (its fan-out is a thread pool, deprecated_package/embedder.py:191-224).
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(*argv, force):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MME_DIST_BACKEND", "MME_FORCE_DIST")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--headline-only", *argv]
    if force:
        cmd.append("--force-dist")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, f"{' '.join(cmd)} -> {r.returncode}\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("argv", [("--config", "c4", "--crops", "512", "--table-rows", "4096"), ("--config", "c5", "--crops", "4096")],
                         ids=["c4", "c5"])
def test_bench_through_the_nccl_process_group_at_world_size_1(argv):
    plain = _bench(*argv, force=False)
    forced = _bench(*argv, force=True)  # returning at all means barrier + destroy_process_group came back
    assert plain["config"]["parallelism"] == "single GPU" and plain["allgather_ms"] is None
    assert "backend nccl" in forced["config"]["parallelism"], forced["config"]["parallelism"]
    assert forced["allgather_ms"] is not None and forced["allgather_ms"] > 0.0
    assert forced["allgather_host_ms"] is not None and forced["allgather_bytes"] == plain["config"]["crops_per_gpu"] * 768 * 2
    assert len(forced["ms_per_step_by_rank"]) == 1 and len(forced["startup_s_by_rank"]) == 1
    # the gathered table head, the embeddings and (C5) the page matrix and labels: bit for bit what the run without a
    # process group produced
    assert forced["result_digest"] == plain["result_digest"]
    if argv[1] == "c5":
        assert forced["result_digest"]["labels"] and "page_matrix_f64" in forced["result_digest"]
