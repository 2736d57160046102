"""`import hv_cuda` shim: the reference scripts import the vote op under this top-level name
(eval_joint.py:10, train_joint.py:10).  Re-exports canonicalvoting_amd.hv_cuda."""
from canonicalvoting_amd.hv_cuda import backward, forward  # noqa: F401
