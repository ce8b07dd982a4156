"""Multi-GPU runner: independent video sequences, one per rank (one process per GPU, RCCL over
xGMI through torch.distributed's "nccl" backend; "gloo" on CPU for tests).  The tracking path has
no exchange step -- frame t depends on frame t-1 of the same sequence only (TRK:85,272-273) -- so
the only collective is the gather of the finished tracks."""
import os

import numpy as np
import torch
import torch.distributed as dist

RECORD = 12          # per frame: 9 x H (row major) + lost, N_lost, global_H_success


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(dev):
    """NUMA node of HIP device `dev` from its PCI address (sysfs), or None when it cannot be told."""
    try:
        pr = torch.cuda.get_device_properties(dev)
        addr = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
            node = int(f.read())
        return node if node >= 0 else None
    except Exception:
        return None


def bind_to_gpu_node(dev, local_rank, n_local):
    """Pin this rank's host threads to cores of the NUMA node its GPU hangs off (every launch of a frame is issued from
    ONE Python thread per rank: ~250 launches + one device->host read per frame -- it must not migrate across sockets or
    share a core with another rank's loop).  The ranks whose GPUs sit on the same node split that node's cores evenly,
    by device order.  -> dict for the bench line; a no-op (with the reason) where sysfs gives no answer.  WOFT_BIND=0: off."""
    info = {"numa_node": None, "cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    if os.environ.get("WOFT_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity") or not torch.cuda.is_available():
        info["bound"] = False
        return info
    node = gpu_numa_node(dev)
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if node is None:
            cpus = allowed                                    # single-node box / container without topology: all cores
        else:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                cpus = [c for c in _cpulist(f.read()) if c in set(allowed)] or allowed
        # ranks sharing this node: the local devices with the same node, in device order
        same = [d for d in range(n_local) if gpu_numa_node(d) == node] if n_local > 1 else [dev]
        k, n = (same.index(dev), len(same)) if dev in same else (local_rank % max(n_local, 1), max(n_local, 1))
        per = max(1, len(cpus) // n)
        mine = cpus[k * per:(k + 1) * per] or cpus
        os.sched_setaffinity(0, mine)
        info.update(numa_node=node, cores=len(mine), first_core=mine[0], bound=True)
    except Exception as ex:                                   # never fatal: binding is a speed matter only
        info.update(bound=False, error=f"{type(ex).__name__}: {ex}")
    return info


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


def device_index():
    """The HIP device this rank uses (init_distributed's choice)."""
    return 0 if os.environ.get("WOFT_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))


def gather_floats(values, device=None):
    """Every rank contributes a fixed-length list of floats; -> (world, n) float64 tensor on every rank."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    if not dist.is_initialized():            # (an initialised group of ONE rank still goes through its collective)
        return t[None]
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu()


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
    if not dist.is_initialized():
        return rec[None]
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    rec = rec.to(device)
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.stack(out).cpu()


def max_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()


def finalize(clean=True):
    """End of a multi-rank job: every rank meets at one last barrier (rank 0 prints the result line after the others have
    finished their timed work; nobody may tear the communicator down under a collective that is still running) and then
    destroys the process group -- without it RCCL warns at interpreter exit at best and hangs in its watchdog at worst.
    clean=False (an exception is propagating on this rank): no barrier, the group is only destroyed."""
    if not dist.is_initialized():
        return
    try:
        if clean:
            dist.barrier()
    finally:
        dist.destroy_process_group()
