#!/usr/bin/env python
"""Rate at which a replayed hipGraph gets launches onto the GPU on this box: chains of N dependent tiny kernels on one
stream, and the same chains on 2 / 4 forked streams of one graph.  With kernels this small the number is the
submission rate of graph nodes (host / command processor), not a GPU-side latency: one MI355X box, ROCm 7.2:
1.5 us per node for a single chain, 2.7 us per node (all branches together) for a forked graph.  The R50-FPN step has
258 nodes in 4.27 ms, i.e. the submission of a replay (~0.7 ms) runs well ahead of its execution."""
import torch


def chain_us(n, streams=1, numel=64, reps=20):
    dev = torch.device("cuda")
    xs = [torch.zeros(numel, device=dev) for _ in range(streams)]
    side = [torch.cuda.Stream() for _ in range(streams - 1)]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for x in xs:
            x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            cur = torch.cuda.current_stream()
            for st in side:
                st.wait_stream(cur)
            for i, x in enumerate(xs):
                st = cur if i == 0 else side[i - 1]
                with torch.cuda.stream(st):
                    for _ in range(n):
                        x.add_(1.0)
            for st in side:
                cur.wait_stream(st)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    for numel in (64, 1 << 20):
        for streams in (1, 2, 4):
            a, b = chain_us(50, streams, numel), chain_us(250, streams, numel)
            print("numel %8d, %d stream(s): %6.2f us per dependent launch (graph of 250 vs 50 per stream), replay of 50: %7.1f us"
                  % (numel, streams, (b - a) / 200.0, a), flush=True)


if __name__ == "__main__":
    main()
