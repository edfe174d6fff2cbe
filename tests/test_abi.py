"""C-ABI checks that need no GPU: libaddk.so loads, exports every function include/addk.h declares, and the ctypes
mirrors of the argument structs have the C sizes/offsets (compiled from the header with gcc)."""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

import addk
from addk import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'addk.h')


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(addk_\w+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = addk.load()
    names = _declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), 'libaddk.so does not export %s' % n
    assert set(names) == set(L.EXPORTED_SYMBOLS), set(names) ^ set(L.EXPORTED_SYMBOLS)
    assert lib.addk_version() >= 1


def test_struct_layouts_match_header():
    structs = {'addk_src': L.Src, 'addk_conv_args': L.ConvArgs, 'addk_conv_dgrad_args': L.ConvDgradArgs,
               'addk_conv_wgrad_args': L.ConvWgradArgs, 'addk_dw_args': L.DwArgs, 'addk_sep_args': L.SepArgs, 'addk_sep_bwd_args': L.SepBwdArgs, 'addk_ce_upsample_args': L.CeUpsampleArgs, 'addk_dw_bwd_args': L.DwBwdArgs,
               'addk_bn_finalize_args': L.BnFinalizeArgs, 'addk_bn_bwd_args': L.BnBwdArgs,
               'addk_affine_sum_args': L.AffineSumArgs, 'addk_affine_sum_bwd_args': L.AffineSumBwdArgs,
               'addk_resize_args': L.ResizeArgs, 'addk_edm_args': L.EdmArgs, 'addk_resize_bwd_args': L.ResizeBwdArgs,
               'addk_dw_wreduce_item': L.DwWreduceItem, 'addk_bn_apply_item': L.BnApplyItem,
               'addk_slab_reduce_item': L.SlabReduceItem, 'addk_bn_coeffs_item': L.BnCoeffsItem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "addk.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ['return 0;}']
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, 'abi.c')
        open(c, 'w').write('\n'.join(lines))
        exe = os.path.join(d, 'abi')
        subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = dict(l.split() for l in out.strip().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got['%s.%s' % (cname, f)]) == getattr(cls, f).offset, '%s.%s' % (cname, f)


def test_constants_match():
    src = open(HEADER).read()
    assert int(re.search(r'#define ADDK_MAX_SRC (\d+)', src).group(1)) == L.MAX_SRC
    assert int(re.search(r'#define ADDK_MAX_SLAB (\d+)', src).group(1)) == L.MAX_SLAB
    assert int(re.search(r'#define ADDK_MAX_TERMS (\d+)', src).group(1)) == L.MAX_TERMS


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', str(tmp_path / 'nope.so'))
    import pytest
    with pytest.raises(addk.AddkError):
        L.load()
