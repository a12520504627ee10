"""binary-recommendation_amd — MI355X-native hot path (NeuMF / BPR / TwoTower training step)
behind the model surface of leotimus/binary-recommendation's src/models and
trainers/{NFC_plain,twoTower}.py.

The directory name carries a hyphen (it mirrors the reference repo's name), so import it as

    import binrec                      # repo-root shim, or
    importlib.import_module("binary-recommendation_amd")

Hand-written HIP kernels live in csrc/ behind the C-ABI of include/binrec.h
(libbinrec_hip.so, built by build.py / __graft_entry__.build()).
"""
from . import _lib  # noqa: F401
from ._lib import BinrecError  # noqa: F401

__all__ = ["BinrecError"]
