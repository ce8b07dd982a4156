"""Where a host frame's way to the device spends its time (1080p BGR, 6.2 MB): host memcpy into pinned staging, the H2D DMA, and
woft_upload_u8's pipelined pieces.  python tools/micro/upload_probe.py"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from woft_amd import _lib  # noqa: E402

lib = _lib.load()
a = (np.random.rand(1080, 1920, 3) * 255).astype(np.uint8)
stage = torch.empty(a.shape, dtype=torch.uint8).pin_memory()
dev = torch.empty(a.shape, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"np.copyto -> pinned            {timeit(lambda: np.copyto(stage.numpy(), a)):.3f} ms")
print(f"H2D from pinned (async + sync) {timeit(lambda: dev.copy_(stage, non_blocking=True)):.3f} ms")
print(f"copyto + H2D                   {timeit(lambda: (np.copyto(stage.numpy(), a), dev.copy_(stage, non_blocking=True))):.3f} ms")
for ch in (1, 2, 4, 8, 16, 32):
    f = lambda: _lib.check(lib.woft_upload_u8(a.ctypes.data, stage.data_ptr(), dev.data_ptr(), a.nbytes, ch, _lib.stream_ptr()), "up")
    print(f"woft_upload_u8, {ch:2d} pieces      {timeit(f):.3f} ms")
assert torch.equal(dev.cpu(), torch.from_numpy(a))
print(f"tensor.cuda() from pageable    {timeit(lambda: torch.from_numpy(a).cuda()):.3f} ms")
