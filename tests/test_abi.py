"""The C-ABI library loads (no GPU needed) and exports exactly what include/fdmi.h declares."""
import os
import re

from conftest import REPO
from foldingdiff_amd import _binding


def _declared():
    src = open(os.path.join(REPO, "include", "fdmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fd_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    declared = _declared()
    assert declared, "no declarations parsed"
    assert sorted(_binding.exported_symbols()) == declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in fdmi.h but not exported by libfdmi.so"


def test_abi_version_and_error_string(lib):
    assert lib.fd_abi_version() == _binding.ABI_VERSION
    assert lib.fd_device_count() >= 0
    # argument errors are reported through codes + fd_last_error, never by crashing
    rc = lib.fd_create(None, 0, None)
    assert rc == -1 and b"null" in lib.fd_last_error()
    rc = lib.fd_profile_every(None, 1)
    assert rc == -1
