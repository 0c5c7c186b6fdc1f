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
    """hipGraph dispatch mode of the HIP runtime (ROCm 7.2, gfx950).

    Finding (tools/graph_replay_repro.py): an instantiated graph that is ONE chain of ~1400 kernel nodes is replayed
    through a batched AQL-packet path (DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default) that intermittently returns garbage for this
    step when the device was idle (hipDeviceSynchronize) before the launch.  Graphs with more than one branch take the per-node path,
    are correct in every trial, and run their branches concurrently.  TrainEngine therefore (1) always captures multi-branch
    graphs (engine._forked + functional.run_branches), (2) verifies the replays against the eager pass after capture and falls
    back to eager launches on any mismatch.  VELOXSEG_GRAPH_DISPATCH=nodes additionally turns the packet path off for the whole
    process (slower: the per-node path then enqueues every kernel from the host, about 17 us each); it must be set before the
    first HIP call, i.e. before or at `import veloxseg_amd`."""
    mode = _os.environ.get("VELOXSEG_GRAPH_DISPATCH", "default")
    if mode not in ("default", "nodes"):
        raise ValueError("VELOXSEG_GRAPH_DISPATCH must be 'default' or 'nodes'")
    if mode == "nodes" and "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in _os.environ:
        torch = _sys.modules.get("torch")
        if torch is not None and torch.cuda.is_initialized():
            raise RuntimeError("VELOXSEG_GRAPH_DISPATCH=nodes: import veloxseg_amd before the first CUDA/HIP call of the process")
        _os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
    return mode


GRAPH_DISPATCH = _configure_hip_runtime()
