"""addk — MI355X-native hot path of Auto-Dynamic-DeepLab (directory `auto-dynamic-deeplab_amd/`, imported as
`addk` through the shim addk.py at the repo root because the directory name is not a Python identifier).

    import addk
    from addk.modeling.ADD import ADD, EDM          # same surface as the reference's modeling package
    from addk.modeling.operations import OPS
    from addk.modeling.genotypes import PRIMITIVES
"""
from . import _lib                                   # noqa: F401
from ._lib import AddkError, LIB_PATH, load          # noqa: F401

__version__ = '0.1.0'


def set_precision(mode):
    """'fp32' (default: exact fp32 MFMA, the parity path) or 'bf16x3' (3-term split-bf16 MFMA, ~1.5e-5 relative per
    product) for the dense convolution forward/data-gradient kernels.  Process-wide."""
    _lib.check(load().addk_set_conv_precision({'fp32': 0, 'bf16x3': 1}[mode]), 'set_precision')


def get_precision():
    return ('fp32', 'bf16x3')[load().addk_get_conv_precision()]
