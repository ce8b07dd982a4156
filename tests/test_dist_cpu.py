"""world_size-2 gloo test of the multi-GPU runner's only collective (gather of the tracks) and of
the max-over-ranks timing reduction -- the N>1 path of bench.py, on CPU."""
import os
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from woft_amd import dist as wd
    r, w, _ = wd.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    results = []
    for t in range(3):
        H = np.eye(3) * (1 + rank) + 0.01 * t
        results.append((H, SimpleNamespace(lost=bool((t + rank) % 2), N_lost=t, global_H_success=not bool((t + rank) % 2))))
    wd.barrier()
    tracks = wd.gather_tracks(results, device="cpu")
    slow = wd.max_over_ranks(0.5 + rank, device="cpu")
    per_rank = wd.gather_floats([10.0 * rank, rank + 0.25], device="cpu")
    assert per_rank.tolist() == [[0.0, 0.25], [10.0, 1.25]]
    info = wd.bind_to_gpu_node(0, rank, world)        # no GPU here: reports why it did not bind, never raises
    assert info["bound"] is False and info["cores"] >= 1
    q.put((rank, tracks.numpy(), slow))
    wd.finalize()                        # last barrier + destroy_process_group (what bench.py's ranks end with)
    assert not torch.distributed.is_initialized()
    wd.finalize()                        # (idempotent: nothing to do without a group)


def test_gather_tracks_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, tracks, slow = q.get(timeout=120)
        got[rank] = (tracks, slow)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(2):
        tracks, slow = got[rank]
        assert tracks.shape == (2, 3, 12)
        assert slow == 1.5
        for src in range(2):
            for t in range(3):
                assert np.allclose(tracks[src, t, :9].reshape(3, 3), np.eye(3) * (1 + src) + 0.01 * t)
                assert tracks[src, t, 9] == float((t + src) % 2) and tracks[src, t, 10] == t
    assert np.array_equal(got[0][0], got[1][0])


def test_single_process_passthrough():
    sys.path.insert(0, str(ROOT))
    from woft_amd import dist as wd
    res = [(np.eye(3), SimpleNamespace(lost=False, N_lost=0, global_H_success=True))]
    t = wd.gather_tracks(res)
    assert tuple(t.shape) == (1, 1, 12) and wd.max_over_ranks(2.0) == 2.0
    assert wd.gather_floats([1.5, 2]).tolist() == [[1.5, 2.0]]
    assert wd._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
