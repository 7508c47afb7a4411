"""The C-ABI shared library loads and exports every symbol include/im360_kernels.h declares (no compute)."""
import ctypes
import os
import re

from imagine360_amd import kernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "im360_kernels.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(im360_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "imagine360_amd", "libim360_kernels.so"))
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_python_binding_covers_header():
    assert set(declared_symbols()) == set(kernels.exported_symbols())
    assert kernels.lib().im360_abi_version() == 1
    assert kernels.lib().im360_last_error() is not None
