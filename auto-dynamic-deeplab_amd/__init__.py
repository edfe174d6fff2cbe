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
    """Arithmetic of the halo-patch convolutions (wide k x k stride-1 convs, forward and data gradient): 'fp32' = exact fp32
    MFMA; 'bf16x6' = every operand split into three bf16 terms, six product terms on the bf16 matrix pipe with fp32
    accumulation (as accurate as the fp32 MFMA chain, 2.5x its rate); 'bf16x3' = three terms (fast mode, ~5e-7 rms per
    dot product).  Process-wide; set it before plans are built (packed-weight buffers are sized per mode)."""
    _lib.check(load().addk_set_conv_precision({'fp32': 0, 'bf16x3': 1, 'bf16x6': 2, 'tail_x3': 3}[mode]), 'set_precision')


def get_precision():
    return ('fp32', 'bf16x3', 'bf16x6', 'tail_x3')[load().addk_get_conv_precision()]
