"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every
symbol include/bma.h declares, validates arguments without launching, and the
product refuses to run on CPU tensors (no fallback)."""

import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "bma.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bma_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bimodalattack_amd import native
    syms = declared_symbols()
    assert len(syms) >= 9
    raw = ctypes.CDLL(native.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/bma.h but not exported"
    assert sorted(native.PROTOTYPES) == syms, "native.PROTOTYPES and include/bma.h disagree"
    assert native.lib.bma_version() == native.ABI_VERSION
    assert native.strerror(0) == "ok" and "align" in native.strerror(-3)


def test_prototype_table_matches_the_header_argument_for_argument():
    """native.PROTOTYPES against include/bma.h: the same number of parameters per entry point, pointers where the header has
    pointers, 64-bit integers where it has int64_t / size_t, floats where it has float (a table one argument short is a
    segmentation fault on the first call, not an exception)."""
    from bimodalattack_amd import native
    text = open(os.path.join(REPO, "include", "bma.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = dict(re.findall(r"\b(bma_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    assert set(decls) == set(native.PROTOTYPES)
    for name, params in decls.items():
        params = " ".join(params.split())
        args = [] if params in ("", "void") else [a.strip() for a in params.split(",")]
        _, table = native.PROTOTYPES[name]
        assert len(args) == len(table), (name, len(args), len(table))
        for a, t in zip(args, table):
            if "*" in a:
                assert t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or t.__name__.startswith("LP_"), (name, a, t)
            elif re.match(r"(const )?(int64_t|size_t)\b", a):
                assert t in (ctypes.c_int64, ctypes.c_size_t, ctypes.c_uint64), (name, a, t)
            elif re.match(r"(const )?float\b", a):
                assert t is ctypes.c_float, (name, a, t)
            elif re.match(r"(const )?double\b", a):
                assert t is ctypes.c_double, (name, a, t)
            elif re.match(r"(const )?(int|unsigned|uint32_t|int32_t)\b", a):
                assert t in (ctypes.c_int, ctypes.c_uint, ctypes.c_uint32, ctypes.c_int32), (name, a, t)


def test_argument_validation_launches_nothing():
    """Every entry point rejects bad arguments before touching the device."""
    from bimodalattack_amd.native import BmaSegment, lib
    assert lib.bma_linf_step(None, None, None, -1, 0.1, 0.1, None, None) == -1
    assert lib.bma_linf_step(None, None, None, 0, 0.1, 0.1, None, None) == 0
    assert lib.bma_linf_step(2, 4, 4, 8, 0.1, 0.1, 4, None) == -3          # misaligned pointer
    assert lib.bma_ce_target(None, 0, 0, None, 1, 0, 10, 0, None, None, None, None, 1.0, None) == -1
    assert lib.bma_ce_target(16, 40, 10, 16, 4, 4, 10, 7, 16, 16, None, None, 1.0, None) == -2   # dtype
    assert lib.bma_ce_target(None, 0, 10, None, 0, 4, 10, 0, None, None, None, None, 1.0, None) == 0
    assert lib.bma_ce_target_ws_bytes(512, 20) == 3 * 512 * 20 * 4
    assert lib.bma_mask_topk(16, 100, 1, 100, 0, None, 101, 16, None, None) == -1   # k > V
    assert lib.bma_mask_topk(16, 5000, 1, 5000, 0, None, 4096, 16, None, None) == -5  # k > 2048
    assert lib.bma_mask_topk(16, 100, 0, 100, 0, None, 5, 16, None, None) == 0
    assert lib.bma_mask_topk_ws_bytes(19, 32064, 256) == 19 * 8 * 256 * 8 and lib.bma_mask_topk_ws_bytes(19, 4096, 256) == 0
    assert lib.bma_mask_topk(16, 9000, 1, 9000, 0, None, 5, 16, 4, None) == -3       # misaligned workspace
    assert lib.bma_rand_positions(16, 0, 65, 1, 16, None) == 0              # any suffix length; empty batch launches nothing
    assert lib.bma_rand_positions(16, 4, 8, 9, 16, None) == -1
    assert lib.bma_sample_scatter(16, 16, 16, 16, 0, 8, 1, 4, 16, None) == 0
    segs = (BmaSegment * 1)(BmaSegment(16, 4, 0))
    assert lib.bma_splice(segs, 1, None, 0, None, 2, 0, 6, 1, 1.0, 16, None) == -3   # D*2 not a multiple of 16
    assert lib.bma_splice(segs, 9, None, 0, None, 2, 0, 8, 1, 1.0, 16, None) == -1
    assert lib.bma_splice(segs, 1, None, 0, None, 0, 0, 8, 1, 1.0, None, None) == 0
    bad = (BmaSegment * 1)(BmaSegment(16, 4, 7))
    assert lib.bma_splice(bad, 1, None, 0, None, 2, 0, 8, 1, 1.0, 16, None) == -1
    # the collective: validation only -- no communicator, no device
    assert lib.bma_allgather_f32(None, -1, None, 0, 1, None, None) == -1
    assert lib.bma_allgather_f32(None, 0, None, 1, 2, None, None) == 0                 # nothing to gather
    assert lib.bma_allgather_f32(16, 4, 16, 2, 2, None, None) == -1                    # rank outside the world
    assert lib.bma_allgather_f32(16, 4, 16, 0, 2, None, None) == -1                    # a world of two needs a communicator
    assert lib.bma_allgather_f32(18, 4, 16, 0, 1, None, None) == -3                    # misaligned
    # round 3 entry points
    assert lib.bma_splice_rows(segs, 1, None, 0, None, 2, 0, 8, 1, 1.0, 16, 0, None, None) == 0        # no rows: nothing to do
    assert lib.bma_splice_rows(segs, 1, None, 0, None, 2, 0, 8, 1, 1.0, None, 5, 16, None) == -1       # no slot map
    assert lib.bma_splice_rows(segs, 1, None, 0, None, 0, 0, 8, 1, 1.0, 16, 5, 16, None) == -1         # an empty block has no rows
    assert lib.bma_add_rmsnorm(None, 16, None, 0.0, 16, 1e-5, 4, 4096, 1, 0, 16, 16, None) == -1
    assert lib.bma_add_rmsnorm(16, 16, None, 0.0, 16, 1e-5, 4, 4100, 1, 0, 16, 16, None) == -3         # row bytes not 16-byte multiples
    assert lib.bma_add_rmsnorm(16, 16, None, 0.0, 16, 1e-5, 0, 4096, 1, 0, 16, 16, None) == 0
    assert lib.bma_add_rmsnorm(16, 16, None, 0.0, 16, 1e-5, 4, 16384, 1, 0, 16, 16, None) == -5        # row longer than 16 KiB
    assert lib.bma_add_rmsnorm_bwd(16, 16, None, None, 1e-5, 4, 4096, 1, 0, 16, None) == -1
    assert lib.bma_rope2(16, 8, 8, 8, 16, 8, 8, 8, 4, 16, 8, 8, 8, 16, 8, 8, 8, 2, 0, 5, 64, 16, 16, 1, 1.0, 1, None) == 0   # B == 0
    assert lib.bma_rope2(16, 8, 8, 8, 16, 8, 8, 8, 4, 16, 8, 8, 8, 16, 8, 8, 8, 2, 1, 5, 64, 16, 16, 1, 0.5, 1, None) == -1  # sin_sign
    assert lib.bma_rope2(16, 8, 8, 8, 16, 8, 8, 8, 4, 16, 8, 4, 8, 16, 8, 8, 8, 2, 1, 5, 64, 16, 16, 1, 1.0, 1, None) == -3  # stride


def test_no_cpu_fallback():
    from bimodalattack_amd import ops
    x = torch.zeros(8)
    with pytest.raises(RuntimeError, match="AMD GPU only"):
        ops.linf_step(x, x, x, 0.1, 0.1)
    with pytest.raises(RuntimeError, match="AMD GPU only"):
        ops.ce_target(torch.zeros(1, 2, 8), torch.zeros(2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="AMD GPU only"):
        ops.mask_topk(torch.zeros(2, 8), None, 2)
    with pytest.raises(RuntimeError, match="AMD GPU only"):
        ops.splice([("shared", torch.zeros(2, 8))], 2)


def test_missing_library_fails_loudly(tmp_path):
    import subprocess, sys
    env = dict(os.environ, BMA_LIB=str(tmp_path / "nope.so"), PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-c", "import bimodalattack_amd.native"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "HIP library not found" in r.stderr and "no CPU or PyTorch fallback" in r.stderr


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, "bimodalattack_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "from oracle" not in src and "import oracle" not in src, f
