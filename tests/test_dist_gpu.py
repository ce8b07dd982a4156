"""The RCCL branch of woft_amd.dist on real hardware (round-4 review: 'the nccl code path has literally never executed'): a
1-rank `nccl` process group on the GPU box -- communicator creation, device tensors, all_gather / all_reduce / barrier through
RCCL -- running exactly the calls bench.py's N > 1 path makes.  (An 8-GPU node differs only in the number of peers.)"""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np, torch
from types import SimpleNamespace
sys.path.insert(0, os.environ["WOFT_ROOT"])
import torch.distributed as dist
from woft_amd import dist as wd
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
res = [(np.eye(3) * (1 + 0.01 * t), SimpleNamespace(lost=bool(t % 2), N_lost=t, global_H_success=not bool(t % 2))) for t in range(5)]
wd.barrier()
tracks = wd.gather_tracks(res)                      # device = "cuda" for nccl: all_gather of CUDA tensors through RCCL
assert tuple(tracks.shape) == (1, 5, 12) and tracks.dtype == torch.float64 and not tracks.is_cuda
for t in range(5):
    assert np.allclose(tracks[0, t, :9].numpy().reshape(3, 3), np.eye(3) * (1 + 0.01 * t))
    assert tracks[0, t, 9] == float(t % 2) and tracks[0, t, 10] == t
assert wd.max_over_ranks(3.25) == 3.25              # all_reduce(MAX) on a CUDA tensor
assert wd.gather_floats([1.5, 2.0, 7.0]).tolist() == [[1.5, 2.0, 7.0]]
info = wd.bind_to_gpu_node(0, 0, 1)
wd.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("NCCL_1RANK_OK", info.get("bound"), info.get("numa_node"))
'''


def test_one_rank_nccl_group_runs_the_gather_path():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               WOFT_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NCCL_1RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
