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


def test_options_api_refuses_unknown_names_and_values():
    """curla_set_option / curla_get_option (curla_amd/csrc/options.h): host-side, no GPU needed.  Every option has its
    default after loading (no CURLA_<NAME> variable is set in the test environment), unknown names and values are
    refused, ``_lib.option`` scopes a value to a block."""
    from curla_amd import _lib
    assert set(_lib.OPTIONS) == {"conv1_u8", "conv1_f32", "s1_fwd", "bwd_split", "gemm_tile", "linear_bwd", "gemm_mfma", "s1_wgrad",
                                 "wgrad1_u8"}
    defaults = {"conv1_u8": "auto", "conv1_f32": "rw", "s1_fwd": "auto", "bwd_split": "auto", "gemm_tile": "auto",
                "linear_bwd": "pair", "gemm_mfma": "auto", "s1_wgrad": "auto", "wgrad1_u8": "auto"}
    for k in _lib.OPTIONS:
        if "CURLA_" + k.upper() not in os.environ:
            assert _lib.get_option(k) == defaults[k]
    with _lib.option("conv1_u8", "rw"):
        assert _lib.get_option("conv1_u8") == "rw"
        with _lib.option("gemm_tile", "64x32"):  # (alias of 6432)
            assert _lib.get_option("gemm_tile") == "6432"
    assert _lib.get_option("conv1_u8") == defaults["conv1_u8"] or "CURLA_CONV1_U8" in os.environ
    with pytest.raises(_lib.CurlaHipError):
        _lib.set_option("conv1_u8", "fastest")
    with pytest.raises(_lib.CurlaHipError):
        _lib.set_option("no_such_option", "1")
    with pytest.raises(_lib.CurlaHipError):
        _lib.get_option("no_such_option")
