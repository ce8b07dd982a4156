"""What one node of a replayed hipGraph costs against one eager launch on this stack (ROCm 7.2, torch 2.10): a chain of N dependent
tiny kernels (x += 1 on 64 floats) and of N ~50-us kernels (the update block's launch length), eager vs torch.cuda.CUDAGraph replay.
Backs DESIGN section 7 item 5 (`alt_graph` slower than eager).  Run on the GPU box: python tools/micro/graph_node_cost.py"""
import time

import torch


def chain(x, n):
    for _ in range(n):
        x.add_(1.0)


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    dev = torch.device("cuda:0")
    n = 190                                                      # launches of one tracked frame
    for name, numel in (("tiny kernel (64 floats)", 64), ("~50 us kernel (64 Mi floats)", 64 << 20)):
        x = torch.zeros(numel, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            chain(x, n)                                          # warm-up on the capture stream
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                chain(x, n)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); chain(x, 20); ev1.record(); torch.cuda.synchronize()
        one = ev0.elapsed_time(ev1) / 20 * 1e3
        te = timed(lambda: chain(x, n), 20)
        tg = timed(g.replay, 20)
        print(f"{name}: one kernel alone {one:7.1f} us | {n}-launch chain: eager {te * 1e3:7.3f} ms = {te / n * 1e6:6.2f} us per launch, "
              f"graph replay {tg * 1e3:7.3f} ms = {tg / n * 1e6:6.2f} us per node")


if __name__ == "__main__":
    main()
