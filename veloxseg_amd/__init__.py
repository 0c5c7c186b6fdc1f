"""veloxseg_amd -- MI355X (gfx950) native forward/backward path for the VeloxSeg 3-D segmentation network.

Drop-in surface (mirrors the reference repository):
    veloxseg_amd.model.VeloxSeg.VeloxSeg      <- model/VeloxSeg.py:16   (same ctor kwargs, state_dict keys, outputs)
    veloxseg_amd.utils.loss.Loss              <- utils/loss.py:10
    veloxseg_amd.utils.load_model.load_model  <- utils/load_model.py:3
All arithmetic runs in hand-written HIP kernels behind the C ABI of include/veloxseg_hip.h.
"""
__version__ = "0.1.0"
