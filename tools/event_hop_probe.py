#!/usr/bin/env python3
"""Cost of a cross-stream dependency on this runtime: a chain of N tiny kernels on ONE stream against the same chain alternating between two streams
(event record + stream wait per hop), both enqueued behind a long kernel so the host's launch rate does not pace them."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from veloxseg_amd import _hip as H
import ctypes
d = torch.device("cuda:0")
H.LIB.load()
lanes = []
for k in range(2):
    out = ctypes.c_void_p()
    H.call("vx_tape_lane_stream", H.stream_ptr(), k, ctypes.addressof(out))
    lanes.append(torch.cuda.ExternalStream(out.value))
x = torch.zeros(1024, device=d)
big = torch.randn(8192, 8192, device=d)
N = 200
hip = ctypes.CDLL("libamdhip64.so")
# stream memory operations: hipStreamWriteValue32 on the source, hipStreamWaitValue32 (>=) on the destination, on 8-byte signal allocations (hipMallocSignalMemory)
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
sig = []
can = ctypes.c_int(0)
hip.hipDeviceGetAttribute(ctypes.byref(can), 10000 + 8 if False else 0, 0)          # (the attribute's number differs between releases: the calls below report failure themselves)
for _ in range(2):
    q = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(q), 8, 0x2)
    sig.append(q if rc == 0 else None)
if sig[0] is not None:
    hip.hipMemset(sig[0], 0, 8); hip.hipMemset(sig[1], 0, 8); hip.hipDeviceSynchronize()
FLAGS = {"torch wait_stream": None, "hipEventDisableTiming": 0x2, "DisableTiming|DisableSystemFence": 0x2 | 0x20000000, "DisableTiming|ReleaseToDevice": 0x2 | 0x40000000, "flag kernels (set + poll)": "flag", "stream memory ops (WriteValue32 + WaitValue32)": "memop"}
def make_events(flags, n):
    out = []
    for _ in range(n):
        e = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
        out.append(e)
    return out
flagbuf = torch.zeros(64, dtype=torch.int32, device=d)
seq = [0]
def hop(src, dst, ev):
    if ev == "memop":
        seq[0] += 1
        k = 0 if src is lanes[0] else 1                     # one signal word per source lane: its values rise in that lane's order
        assert hip.hipStreamWriteValue32(ctypes.c_void_p(src.cuda_stream), sig[k], seq[0], 0) == 0, "hipStreamWriteValue32 failed"
        assert hip.hipStreamWaitValue32(ctypes.c_void_p(dst.cuda_stream), sig[k], seq[0], 0, 0xFFFFFFFF) == 0, "hipStreamWaitValue32 failed"      # flags 0 = hipStreamWaitValueGte
    elif ev == "flag":
        seq[0] += 1
        H.call("vx_tape_flag_set", flagbuf.data_ptr(), seq[0], src.cuda_stream)
        H.call("vx_tape_flag_wait", flagbuf.data_ptr(), seq[0], dst.cuda_stream)
    elif ev is None:
        dst.wait_stream(src)
    else:
        assert hip.hipEventRecord(ev, ctypes.c_void_p(src.cuda_stream)) == 0
        assert hip.hipStreamWaitEvent(ctypes.c_void_p(dst.cuda_stream), ev, 0) == 0
def run(alternate, evs=None):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0, s1 = lanes
    with torch.cuda.stream(s0):
        for _ in range(3):
            big @ big                      # blocker: everything below is queued while it runs
        e0.record()
    cur = s0
    for i in range(N):
        nxt = (s1 if cur is s0 else s0) if alternate else s0
        if nxt is not cur:
            hop(cur, nxt, None if evs is None else evs[i])
        with torch.cuda.stream(nxt):
            x.add_(1.0)
        cur = nxt
    if cur is not s0:
        s0.wait_stream(cur)
    with torch.cuda.stream(s0):
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N
for name, fl in FLAGS.items():
    if fl == "memop" and sig[0] is None:
        print(f"{name:36s}: hipExtMallocWithFlags(hipMallocSignalMemory) failed on this device")
        continue
    evs = None if fl is None else ([fl] * N if isinstance(fl, str) else make_events(fl, N))
    for _ in range(2):
        a, b = run(False), run(True, evs)
    print(f"{name:36s}: same stream {a:.2f} us per kernel; alternating streams {b:.2f} us per kernel; a cross-stream hop costs {b - a:.2f} us")
