"""CPU: libiamx.so loads and exports exactly the C ABI include/iamx.h declares."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, 'include', 'iamx.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(iamx_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_table_agree():
    from imageanalysis_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_library_exports_every_symbol():
    from imageanalysis_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(h, name), name
    L = _lib.lib()
    assert L.iamx_version() >= 100
    assert L.iamx_arch() == b'gfx950'
    assert L.iamx_desc_padded_rows(1) == 128 and L.iamx_desc_padded_rows(128) == 128
    assert L.iamx_desc_padded_rows(129) == 256 and L.iamx_desc_padded_rows(0) == 0
    assert L.iamx_knn2_wg_per_pair(4096) == 16 and L.iamx_knn2_wg_per_pair(1) == 1


def test_argument_checks_do_not_need_a_gpu():
    from imageanalysis_amd import _lib
    L = _lib.lib()
    rc = L.iamx_knn2_l2_u8(None, None, 4, None, None, 4, None, None, None)
    assert rc == -1 and b'null pointer' in L.iamx_last_error()
    rc = L.iamx_ba_residual(None, 1, None, 1, None, None, None, 0, None, None, None)
    assert rc == -1
    assert L.iamx_comm_allgather(None, None, None, 16, None) == -1
    assert L.iamx_knn2sym_sweep(*([None] * 9), 1, 1, 2, None, None, None, None) == -1


def test_product_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from imageanalysis_amd import _lib
    with pytest.raises(_lib.IamxError):
        _lib.require_gpu()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under imageanalysis_amd/ may import or load it."""
    import re
    pkg = os.path.join(REPO, 'imageanalysis_amd')
    pat = re.compile(r'^\s*(from|import)\s+oracle\b|oracle[/.]cpu_ref|liboracle|oracle/_ref', re.M)
    for root, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.sh')):
                text = open(os.path.join(root, f)).read()
                assert not pat.search(text), os.path.join(root, f)
