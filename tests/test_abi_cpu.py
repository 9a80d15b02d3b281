"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/dam_hip.h declares
(no compute calls without a GPU); the ctypes table matches the header's parameter counts."""
import os
import re

import pytest


def header_functions(root):
    txt = open(os.path.join(root, 'include', 'dam_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|int64_t|const char\*)\s+(dam_\w+)\s*\(([^;]*?)\)\s*;', txt, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ('', 'void') else len(args.split(','))
    return out


def test_library_exports_header(dam_lib):
    from conftest import ROOT
    from deep_audio_mixer_amd import _lib
    decl = header_functions(ROOT)
    assert len(decl) >= 5
    for name, nargs in decl.items():
        assert hasattr(dam_lib, name), 'libdam_hip.so does not export %s' % name
        assert name in _lib.SIGNATURES, 'no ctypes signature for %s' % name
        assert len(_lib.SIGNATURES[name][1]) == nargs, name
    assert set(_lib.SIGNATURES) == set(decl)
    assert dam_lib.dam_arch() == b'gfx950'
    header = open(os.path.join(ROOT, 'include', 'dam_hip.h')).read()
    assert dam_lib.dam_abi_version() == _lib.EXPECTED_ABI == int(re.search(r'#define DAM_ABI_VERSION (\d+)', header).group(1))
    assert dam_lib.dam_stft_twiddle_count(2048) == 2048


def test_product_fails_loudly_without_gpu():
    import torch
    from deep_audio_mixer_amd import features
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='GPU only'):
        features.stft_logmag(torch.zeros(1, 4096), hop=1024)


def test_stale_library_is_refused(dam_lib, monkeypatch):
    """A prebuilt libdam_hip.so of another ABI version must not be bound (arguments would be misread as pointers)."""
    from deep_audio_mixer_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'EXPECTED_ABI', _lib.EXPECTED_ABI + 1)
    with pytest.raises(RuntimeError, match='stale'):
        _lib.lib()
