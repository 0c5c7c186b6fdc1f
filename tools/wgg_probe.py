#!/usr/bin/env python3
"""The grouped weight-gradient launch that closes the encoder backward of autopet128 B = 4 (job list printed by VX_WGG_DBG=1), alone on the GPU: the whole launch and its parts."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import _hip as H

dev = torch.device("cuda")
B = 4
JOBS = [(64, 32, 512), (64, 64, 512), (256, 64, 512), (256, 64, 512), (32, 96, 4096), (96, 32, 4096), (32, 32, 4096), (32, 96, 4096), (96, 32, 4096)] + [(32, 32, 4096)] * 7 + \
       [(128, 32, 4096)] * 2 + [(16, 16, 32768)] * 6
FOLDS = [(128, 1024), (128, 1024), (16, 2048), (16, 2048)]


def run(jobs, folds, name, reps=30):
    keep = []
    ptrs, dims = [], []
    for ci, co, V in jobs:
        x = torch.randn(B, ci, V, device=dev); dy = torch.randn(B, co, V, device=dev)
        dw = torch.zeros(co, ci, device=dev); db = torch.zeros(co, device=dev)
        keep += [x, dy, dw, db]
        ptrs += [x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr()]
        dims += [ci, co, V, B]
    fptrs, fdims = [], []
    for C, rows in folds:
        part = torch.randn(rows, 2 * C, device=dev); dg = torch.zeros(C, device=dev); dbt = torch.zeros(C, device=dev)
        keep += [part, dg, dbt]
        fptrs += [part.data_ptr(), dg.data_ptr(), dbt.data_ptr()]
        fdims += [C, rows]
    P = (ctypes.c_void_p * max(1, len(ptrs)))(*ptrs); D = (ctypes.c_long * max(1, len(dims)))(*dims)
    FP = (ctypes.c_void_p * max(1, len(fptrs)))(*fptrs); FD = (ctypes.c_int * max(1, len(fdims)))(*fdims)
    st = H.stream_ptr()
    fn = lambda: H.call("vx_pw_wgrad_group", ctypes.addressof(P), ctypes.addressof(D), len(jobs), ctypes.addressof(FP), ctypes.addressof(FD), len(folds), st)
    big = torch.empty(64 << 20, device=dev)
    ts = []
    for _ in range(reps):
        big.zero_()                                   # push the operands out of L2 (256 MB: also most of the memory-side cache)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    byt = sum(B * (ci + co) * V * 4 for ci, co, V in jobs)
    print(f"{name:44s}: {ts[len(ts) // 2]:6.1f} us (min {ts[0]:.1f}); unique operand bytes {byt / 1e6:.0f} MB", flush=True)


run(JOBS, FOLDS, "the whole launch")
run(JOBS, [], "without the folds")
run([j for j in JOBS if j[2] != 32768], FOLDS, "without the six 16 -> 16 jobs at 32^3")
run([j for j in JOBS if j[2] == 32768], [], "only the six 16 -> 16 jobs at 32^3")
run([j for j in JOBS if j[2] == 4096], [], "only the 16^3 jobs")
run([j for j in JOBS if j[2] == 512], [], "only the 8^3 jobs")
run([], FOLDS, "only the folds")
