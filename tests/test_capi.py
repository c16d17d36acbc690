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


def test_inline_asm_vector_ops_clear_of_mfma_hazards(tmp_path):
    """common.h's packed Winograd transform is inline asm, invisible to the compiler's hazard recogniser; the
    generated ISA must keep every such VALU write clear of the MFMA RAW / SrcC-WAR windows."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = tmp_path / "conv.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S",
                           "--cuda-device-only", "-o", str(asm), os.path.join(root, "curla_amd/csrc/conv.hip")],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(root, "tools/check_asm_hazards.py"), str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "inline-asm VALU ops checked, 0 hazard(s)" in r.stdout and not r.stdout.startswith("0 inline")
