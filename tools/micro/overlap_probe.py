import sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[2]))
from woft_amd import _lib, ops
hf, wf = 135, 240
P = hf * wf
n = ops.tiled_dims(hf, wf)[2]
mk = lambda rows: (torch.randn(ops._round_up(rows, 256), 256, device="cuda") * 0.1)
a, b = mk(P), mk(n)
sa, sb = (torch.zeros(x.shape[0], 512, dtype=torch.bfloat16, device="cuda") for x in (a, b))
ops.split_bf16_lines(a, sa); ops.split_bf16_lines(b, sb)
vol = torch.zeros(P, n, device="cuda")
big = torch.empty(P, n, device="cuda")
lib = _lib.load()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(gemm_abl, do_gemm, do_fill):
    lib.woft_set_tuning(2, gemm_abl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_event(e0); s2.wait_event(e0)
    if do_gemm:
        with torch.cuda.stream(s1):
            ops.corr_gemm_bf16(sa, sb, P, n, 1 / 16.0, vol, 3)
    if do_fill:
        with torch.cuda.stream(s2):
            big.fill_(1.0)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3
for name, args in [("gemm full", (0, True, False)), ("gemm no-store", (1, True, False)), ("fill only", (0, False, True)),
                   ("gemm no-store || fill", (1, True, True)), ("gemm full || fill", (0, True, True))]:
    ts = sorted(run(*args) for _ in range(7))
    print(f"{name:26s} {ts[3]:8.1f} us")
lib.woft_set_tuning(2, 0)
