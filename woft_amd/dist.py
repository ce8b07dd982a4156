"""Multi-GPU runner: independent video sequences, one per rank (one process per GPU, RCCL over
xGMI through torch.distributed's "nccl" backend; "gloo" on CPU for tests).  The tracking path has
no exchange step -- frame t depends on frame t-1 of the same sequence only (TRK:85,272-273) -- so
the only collective is the gather of the finished tracks."""
import os

import numpy as np
import torch
import torch.distributed as dist

RECORD = 12          # per frame: 9 x H (row major) + lost, N_lost, global_H_success


def init_distributed(backend=None):
    """Initialise from the torchrun environment; returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # WOFT_SINGLE_DEVICE=1 (testing the N>1 code path on a 1-GPU box): every rank uses device 0, gloo collectives
    single = os.environ.get("WOFT_SINGLE_DEVICE") == "1"
    dev = 0 if single else local
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or os.environ.get("WOFT_DIST_BACKEND") or \
            ("gloo" if single or not torch.cuda.is_available() else "nccl")
        if torch.cuda.is_available():
            torch.cuda.set_device(dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(dev)
    return rank, world, local


def pack_track(results):
    """[(H 3x3, meta)] -> float64 tensor (T, RECORD)."""
    rec = np.zeros((len(results), RECORD), dtype=np.float64)
    for i, (H, meta) in enumerate(results):
        rec[i, :9] = np.asarray(H, dtype=np.float64).reshape(9)
        rec[i, 9] = float(bool(getattr(meta, "lost", False)))
        rec[i, 10] = float(getattr(meta, "N_lost", 0))
        rec[i, 11] = float(bool(getattr(meta, "global_H_success", True)))
    return torch.from_numpy(rec)


def gather_tracks(results, device=None):
    """All ranks contribute their sequence's track; every rank gets the (world, T, RECORD) tensor."""
    rec = pack_track(results)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec[None]
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    rec = rec.to(device)
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.stack(out).cpu()


def max_over_ranks(value, device=None):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
