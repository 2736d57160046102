"""`import MinkowskiEngine as ME` shim (eval_joint.py:9, utils/minkunet.py:28): re-exports the
gfx950 facade in canonicalvoting_amd.me so reference-shaped code runs unchanged."""
from canonicalvoting_amd.me import *  # noqa: F401,F403
from canonicalvoting_amd.me import (CoordinateManager, MinkowskiBatchNorm, MinkowskiConvolution,  # noqa: F401
                                    MinkowskiConvolutionTranspose, MinkowskiReLU, SparseTensor, cat, utils)
from . import modules  # noqa: F401
