"""CPU-only checks of the C-ABI boundary: libmivae.so loads, exports exactly the entry points that
include/mivae.h declares, the ctypes table agrees with the header's arity, and argument validation
that needs no GPU (error codes + mivae_last_error) behaves.  No kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mivae.h")


def header_decls():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(mivae_\w+)\s*\(([^)]*)\)\s*;", text):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return decls


@pytest.fixture(scope="module")
def pkg():
    import nonauto_lm_amd
    return nonauto_lm_amd


def test_header_declares_entry_points():
    d = header_decls()
    assert len(d) >= 25 and "mivae_nice_stack_fwd" in d and "mivae_elbo_bwd" in d


def test_library_exports_every_declared_symbol(pkg):
    assert os.path.exists(pkg._lib.LIB_PATH), "run python nonauto-lm_amd/build.py"
    handle = ctypes.CDLL(pkg._lib.LIB_PATH)
    for name in header_decls():
        assert hasattr(handle, name), f"{name} declared in include/mivae.h but not exported"


def test_ctypes_table_matches_header(pkg):
    d = header_decls()
    assert set(pkg._lib.SIGNATURES) == set(d)
    for name, args in pkg._lib.SIGNATURES.items():
        assert len(args) == d[name], name


def test_argument_validation_without_gpu(pkg):
    lib = pkg._lib.lib()
    assert lib.mivae_version() >= 100
    assert lib.mivae_tokens_mask(None, None, -1, None) == -1
    assert b"n < 0" in lib.mivae_last_error()
    assert lib.mivae_tokens_mask(None, None, 0, None) == 0          # empty input: no launch
    assert lib.mivae_nice_affine_fwd(None, None, None, None, None, None, 2, 1, 3, 7, 0, 0, None) == -1
    assert b"even" in lib.mivae_last_error()
    assert lib.mivae_gauss_logprob_fwd(None, None, None, 2, 1, 3, 4, 0.0, -1.0, 0, None) == -1
    assert lib.mivae_reparam_logq_fwd(None, None, None, None, None, None, 2, 1, 3, 4, 7, None) == -1
    assert b"dtype" in lib.mivae_last_error()
    assert lib.mivae_nice_pack_bytes(32, 64, 1) > 0 and lib.mivae_nice_pack_bytes(0, 64, 1) == 0
    assert lib.mivae_nice_stack_npad(33) == 64 and lib.mivae_nice_stack_npad(64) == 64


def test_product_path_refuses_cpu_tensors(pkg):
    import torch
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.ops.tokens_mask(torch.zeros(2, 3, dtype=torch.int64))


def test_product_never_imports_oracle():
    pkgdir = os.path.join(ROOT, "nonauto-lm_amd")
    for fn in os.listdir(pkgdir):
        if fn.endswith(".py"):
            src = open(os.path.join(pkgdir, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn
