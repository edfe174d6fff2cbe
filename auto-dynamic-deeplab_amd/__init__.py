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
