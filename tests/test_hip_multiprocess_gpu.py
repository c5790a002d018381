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
