"""veloxseg_amd -- MI355X (gfx950) native forward/backward path for the VeloxSeg 3-D segmentation network.

Drop-in surface (mirrors the reference repository):
    veloxseg_amd.model.VeloxSeg.VeloxSeg      <- model/VeloxSeg.py:16   (same ctor kwargs, state_dict keys, outputs)
    veloxseg_amd.utils.loss.Loss              <- utils/loss.py:10
    veloxseg_amd.utils.load_model.load_model  <- utils/load_model.py:3
All arithmetic runs in hand-written HIP kernels behind the C ABI of include/veloxseg_hip.h.
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys


def _configure_hip_runtime():
    """ROCm 7.2 replays instantiated hipGraphs from pre-recorded AQL packets ("graph packet capture").  On gfx950 that path
    intermittently replays this engine's ~1400-node training graph with corrupt results when the device was idle
    (hipDeviceSynchronize) before the launch; the regular per-node dispatch path is correct and measures the same
    (161.7 vs 162.4 patches/s).  The switch is read once, when the HIP runtime initialises, so it has to
    be in the environment before the first HIP call of the process: importing this package first is enough.
    Returns True when the setting is known to be in effect."""
    want = "0"
    cur = _os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")
    if cur is not None:
        return cur == want
    torch = _sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        return False                      # too late for this process; TrainEngine then launches eagerly instead of replaying
    _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = want
    return True


GRAPH_REPLAY_SAFE = _configure_hip_runtime()
