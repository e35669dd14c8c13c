"""seq-recommendations_amd -- the MI355X-native hot path of efikarra/seq-recommendations.

RNN next-item training/prediction (ragged session batches -> embedding-row
gather -> SimpleRNN/LSTM/GRU scan -> full or sampled softmax -> masked CE ->
BPTT -> clipnorm + Adagrad) behind the ``fit_model / predict / evaluate`` surface
of the reference's ``model.py``.  Host code is Python on PyTorch-ROCm (device
memory, streams, torch.distributed); all arithmetic is hand-written HIP for
gfx950 in ``csrc/`` behind the C ABI of ``include/seqrec_hip.h``.

The directory name has a hyphen, so import it with
``importlib.import_module("seq-recommendations_amd")``.
"""
from . import _lib
from ._lib import SeqrecError, LIB_PATH


def require_hip():
    """Load the HIP library or raise -- there is no CPU fallback in the product."""
    return _lib.load()


__all__ = ["require_hip", "SeqrecError", "LIB_PATH"]
