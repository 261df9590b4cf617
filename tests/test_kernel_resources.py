"""Scratch (register spill) budget of the hot kernels, read from the metadata of the built objects (no GPU, no
recompilation): the wide-latent kernels sit at the 256-register line of two waves per SIMD, and a harmless-looking source
change can tip hipcc into spilling inside their task loops (it did once: C5 forward 1.32 -> 1.67 ms).  The numbers below
are the budgets of the committed sources; a kernel that exceeds its budget fails here, on the CPU, before any GPU run."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "nonauto-lm_amd", "lib")
LLVM = "/opt/rocm/lib/llvm/bin"

# (object, regex on the demangled name, max scratch bytes per lane)
BUDGET = [
    ("nice_big.o", r"k_big_chain<4,", 0),
    ("nice_big.o", r"k_big_chain<8, 2,", 0),           # backward chain, h = 256
    ("nice_big.o", r"k_big_chain<8, 0,", 32),           # forward, h = 256: 24-32 bytes (C5 forward 1.37 ms with them)
    ("nice_big.o", r"k_big_chain<8, 1,", 96),           # inverse (sampling direction), h = 256: 56-72 bytes
    ("nice_big.o", r"k_big_nets<4,", 0),
    ("nice_big.o", r"k_big_nets<8, 1, false>", 0),
    ("nice_big.o", r"k_big_nets<8, 1, true>", 64),      # saving route, h = 256: 44 bytes outside the task loop
    ("nice_big.o", r"k_big_dw<4, 1, 0>", 0),
    ("nice_big.o", r"k_big_dw<8,", 0),
    ("nice_big.o", r"k_big_dw2<", 0),
    ("nice_small.o", r"k_stack_bwd_par<1,", 0),         # GELU (the reference default)
    ("nice_small.o", r"k_stack_bwd_par<-1,", 64),       # activation chosen at run time: 20 bytes
    ("nice_stack.o", r"k_stack_fwd<mivae::MmaBF16", 0),
    ("nll.o", r"k_lce_(fwd_s|dh|dwp)<", 0),
]


def _kernels(obj):
    """[(demangled name, scratch bytes per lane)] of the gfx950 code object inside a host object file."""
    with tempfile.TemporaryDirectory() as tmp:
        fb, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "co.elf")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fb}", obj, os.path.join(tmp, "copy.o")])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    names = re.findall(r"^\s*\.name:\s+(\S+)\s*$", notes, re.M)
    scratch = [int(v) for v in re.findall(r"^\s*\.private_segment_fixed_size:\s+(\d+)\s*$", notes, re.M)]
    kernel_names = [n for n in names if n.startswith("_Z") or n.startswith("k_")]
    assert len(kernel_names) == len(scratch), (len(kernel_names), len(scratch))
    dem = subprocess.run(["c++filt"], input="\n".join(kernel_names), capture_output=True, text=True, check=True).stdout.split("\n")
    return [(d.replace("(anonymous namespace)::", ""), s) for d, s in zip(dem, scratch)]


@pytest.mark.skipif(not (os.path.exists(f"{LLVM}/llvm-readelf") and os.path.exists(f"{LLVM}/clang-offload-bundler")
                         and shutil.which("c++filt")), reason="ROCm LLVM tools not present")
def test_hot_kernels_stay_within_their_scratch_budget():
    cache = {}
    seen = 0
    for obj, pat, limit in BUDGET:
        path = os.path.join(LIBDIR, obj)
        if not os.path.exists(path):
            pytest.skip(f"{obj} not built (python nonauto-lm_amd/build.py)")
        if obj not in cache:
            cache[obj] = _kernels(path)
        hits = [(n, s) for n, s in cache[obj] if re.search(pat, n)]
        assert hits, f"no kernel matches {pat} in {obj}"
        for n, s in hits:
            seen += 1
            assert s <= limit, f"{n[:100]}: {s} bytes of scratch per lane (budget {limit})"
    assert seen >= 20
