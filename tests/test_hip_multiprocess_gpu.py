"""
The multi-process code path with real kernels: WORLD_SIZE processes share GPU 0 and are joined by a gloo group
(the all-to-all of device buffers is staged through host memory there, distributed._all_to_all; RCCL needs one GPU
per rank, which the test pool does not have).  Everything else is what an N-GPU run executes: process-group
bookkeeping, facet sharding, send buffers written in place per destination, split sizes of all_to_all_single,
pending handles with two waves in flight, weighted subgrid ownership, the mirror exchange of the backward pass in
both schedules.  Results are compared with the single-process classes inside each worker (tests/_mp_dist_worker.py).
"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_process_group_path_on_one_gpu(world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_mp_dist_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
        assert f"rank {rank}/{world}: ok" in out


def test_one_rank_rccl_rehearsal_of_the_bench_path():
    """bench.py --rccl-dry: the multi-GPU forward path (DistributedForward, whole-wave ownership, one all-to-all per group
    of waves) with a ONE-rank RCCL process group -- every exchange is a real ``all_to_all_single`` through RCCL on its own
    stream with the real send / receive buffers.  Unmeasured on links; what it pins: RCCL initialises on this box, the
    async handles and stream semantics of distributed._all_to_all hold, the results equal the oracle's, and the JSON
    line is the LAST line on stdout although RCCL prints a banner through C stdio."""
    import json

    root = os.path.dirname(HERE)
    for exchange in ("group", "wave"):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = subprocess.run(
            [sys.executable, os.path.join(root, "bench.py"), "--workload", "8k", "--rccl-dry", "--exchange", exchange, "--steps", "1",
             "--warmup", "0", "--no-cpu-baseline", "--no-backward"],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, check=False,
        )
        assert out.returncode == 0, out.stderr[-3000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["rccl_ranks"] == 1 and "REHEARSAL" in line["config"]["parallelism"]
        assert line["parity"]["ok"], line["parity"]
