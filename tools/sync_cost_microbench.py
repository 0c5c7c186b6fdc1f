"""GPU-side cost of cross-stream ordering on this runtime: N tiny kernels on one stream vs the same kernels ping-ponged between two streams
through event record / wait pairs (the fork/join primitive of functional.run_branches and of autograd's backward stream switching)."""
import time
import torch
a = torch.zeros(1024, device="cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
N = 2000


def one_stream():
    with torch.cuda.stream(sA):
        for _ in range(N):
            a.add_(1.0)


def ping_pong():
    for i in range(N // 2):
        with torch.cuda.stream(sA):
            a.add_(1.0)
        sB.wait_stream(sA)
        with torch.cuda.stream(sB):
            a.add_(1.0)
        sA.wait_stream(sB)


def fork_join(width):
    """N/width rounds of: `width` side streams each wait for A, run one kernel, A waits for all of them"""
    side = [torch.cuda.Stream() for _ in range(width)]
    bufs = [torch.zeros(1024, device="cuda") for _ in range(width)]

    def run():
        for _ in range(N // width):
            for s_, b in zip(side, bufs):
                s_.wait_stream(sA)
                with torch.cuda.stream(s_):
                    b.add_(1.0)
            for s_ in side:
                sA.wait_stream(s_)
    return run


for name, fn in (("one stream", one_stream), ("ping-pong between two streams", ping_pong), ("fork/join width 2", fork_join(2)), ("fork/join width 4", fork_join(4))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-34s host %.2f us/kernel, finished %.2f us/kernel" % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
