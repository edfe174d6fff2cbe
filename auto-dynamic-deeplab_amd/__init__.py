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
    """Arithmetic of the halo-patch convolutions and their weight gradients (the k x k convs and the wide 1x1 heads): 'fp32' = exact fp32
    MFMA; 'f16x3' (default) = every operand split into two fp16 terms (2 x 11 = 22 significand bits) under an exact power-of-two
    scale, three product terms on the fp16 matrix pipe with fp32 accumulation — split error below the rounding noise of an fp32
    accumulation chain; 'bf16x6' = three bf16 terms, six product terms (rounds 2-5's default, same accuracy, twice the matrix
    instructions); 'tail_x3' = f16x3 in the exit heads, bf16x6 elsewhere.  'bf16x3' is accepted as the old name of mode 1.
    Process-wide; set it before plans are built (packed-weight buffers are sized per mode)."""
    _lib.check(load().addk_set_conv_precision({'fp32': 0, 'f16x3': 1, 'bf16x3': 1, 'bf16x6': 2, 'tail_x3': 3}[mode]), 'set_precision')


def get_precision():
    return ('fp32', 'f16x3', 'bf16x6', 'tail_x3')[load().addk_get_conv_precision()]
