"""The C-ABI library loads and exports every symbol include/curla_hip.h declares
(no compute: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "curla_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(curla_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_header_symbols():
    import __graft_entry__ as ge
    ge.build()
    from curla_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in curla_hip.h but not exported"
    # the ctypes table binds exactly the declared entry points
    assert sorted(_lib.SIGNATURES) == names
    assert b"gfx950" in _lib.load().curla_version()


def test_header_is_plain_c():
    """The boundary must compile as C (no torch / C++ types in the signatures)."""
    import subprocess
    src = '#include "curla_hip.h"\nint main(void){return curla_version()==0;}\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-"], input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


def test_missing_library_fails_loudly(monkeypatch):
    from curla_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcurla_hip.so")
    with pytest.raises(_lib.CurlaHipError):
        _lib.load()
    with pytest.raises(_lib.CurlaHipError):
        _lib.call("curla_mean", None, 1, None, None)
