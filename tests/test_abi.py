"""CPU: libppo_amd.so builds for gfx950, loads, and exports every symbol that
include/*.h declares; the ctypes table covers the same set.  No compute calls."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            text = open(os.path.join(inc, f)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(ppo_[a-z0-9_]+)\s*\(", text))
    return names


def test_header_symbols_exported(hip_lib):
    names = declared_symbols()
    assert "ppo_gae_scan_f32" in names and len(names) >= 4
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/ but not exported by libppo_amd.so"


def test_ctypes_table_matches_header(hip_lib):
    from ppo_amd import _lib
    assert set(_lib.SIGNATURES) == declared_symbols()


def test_version_and_error_string(hip_lib):
    assert hip_lib.ppo_version() >= 1
    assert isinstance(hip_lib.ppo_last_error(), (bytes, type(None)))


def test_argument_validation_needs_no_gpu(hip_lib):
    # invalid shapes are rejected before any HIP call
    from ppo_amd import _lib
    rc = hip_lib.ppo_gae_scan_f32(None, None, None, None, 0, None, None, 4, 8, 2, 0.9, 0.9, 0.9, 0, None)
    assert rc == -1 and b"bad shape" in hip_lib.ppo_last_error()
    rc = hip_lib.ppo_gae_scan_f32(None, None, None, None, 0, None, None, 4, 8, 8, 0.9, 0.9, 0.9, 0, None)
    assert rc == -1 and b"null" in hip_lib.ppo_last_error()
    assert _lib.PPO_TERM_U8 == 1


def test_code_object_is_gfx950():
    from ppo_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_oracle_builds():
    import oracle
    assert os.path.exists(oracle.build())
    assert isinstance(oracle.lib(), ctypes.CDLL)
