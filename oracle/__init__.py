"""CPU oracle for the ADD hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a plain PyTorch-CPU fp32 restatement of the algorithm of the
reference's dense convolutional hot path (modeling/operations.py,
modeling/ADD.py, modeling/aspp_train.py, modeling/decoder.py,
modeling/baseline_model.py of HankKung/Auto-Dynamic-DeepLab).  Each function or
class cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the shipped package (`auto-dynamic-deeplab_amd/`, imported as `addk`)
never does and fails loudly when its HIP library is missing.

Parity pin: `tests/golden/*.npz` were produced by `tests/golden/make_golden.py`,
which imports the real reference from /root/reference in the build container
and records inputs, explicit weights and outputs.  `tests/test_oracle_golden.py`
checks this restatement against those vectors (CPU, no GPU needed).
"""
from .net import (  # noqa: F401
    PRIMITIVES, OPS, ReLUConvBN, DilConv, SepConv, Identity, Zero,
    FactorizedReduce, DoubleFactorizedReduce, ASPP_train, Decoder, Cell, ADD,
    EDM, Cell_baseline, Baselin_Model, normalized_shannon_entropy,
    confidence_max, global_batch_norm, Evaluator, cross_entropy_mean_exits, class_weights_from_labels,
)
