from .VeloxSeg import VeloxSeg  # noqa: F401
