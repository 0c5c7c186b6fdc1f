#!/usr/bin/env python3
"""Do independent chains of small kernels on different HIP streams overlap on this GPU?  Two/three forked chains of N dependent elementwise
kernels are captured, then replayed by the launch tape on 1 lane (serial) and on one lane per chain."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd.engine import LaunchTape

dev = torch.device("cuda:0")


def build(nchain, nk, numel, lanes):
    xs = [torch.randn(numel, device=dev) for _ in range(nchain)]
    streams = [torch.cuda.Stream() for _ in range(nchain)]

    def fn():
        cur = torch.cuda.current_stream()
        outs = []
        for x, s in zip(xs, streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                y = x
                for _ in range(nk):
                    y = torch.sin(y)
                outs.append(y)
        for s in streams:
            cur.wait_stream(s)
        return outs
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        keep = fn()
    return LaunchTape(g, lanes), keep


BLOCK = torch.randn(8192, 8192, device=dev)


def timed(t, n=5):
    """GPU time of one replay with the whole replay enqueued BEFORE the GPU may start it (a ~10 ms matmul runs first), so that the host's
    ~2.6 us per launch does not pace the kernels"""
    t.replay()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            BLOCK @ BLOCK
        e0.record()
        t.replay()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


for numel in (4096, 262144, 4 * 1024 * 1024):
    for nchain in (2, 3):
        t1, k1 = build(nchain, 100, numel, 1)
        tn, kn = build(nchain, 100, numel, nchain + 1)
        a, b = timed(t1), timed(tn)
        print(f"numel {numel:8d} chains {nchain}: 1 lane {a:8.1f} us ({a/(100*nchain):5.2f} us/kernel), {tn.n_lanes} lanes {b:8.1f} us  -> x{a/b:.2f}")
