"""CPU: the C-ABI shared library loads and exports every symbol include/txe.h declares (no compute calls without a
GPU), and the ctypes prototype table mirrors the header."""
import ctypes
import os
import re

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    text = open(os.path.join(REPO, "include", "txe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|size_t|float|unsigned)\s+(txe_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",") if a.strip() and a.strip() != "void"]
        decls[m.group(2)] = (m.group(1), args)
    return decls


def test_library_builds_loads_and_exports_header_symbols():
    import __graft_entry__ as ge
    lib_path = ge.build()
    lib = ctypes.CDLL(lib_path)
    decls = _header_decls()
    assert len(decls) >= 26
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/txe.h but not exported by libtxe.so"


def test_ctypes_table_mirrors_header():
    from taxoexpan_amd import _lib
    decls = _header_decls()
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    cmap = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
            "unsigned long long": ctypes.c_ulonglong, "unsigned": ctypes.c_uint, "double": ctypes.c_double}
    for name, (ret, args) in decls.items():
        res, argtypes = _lib.SIGNATURES[name]
        assert res is cmap[ret], name
        assert len(args) == len(argtypes), (name, len(args), len(argtypes))
        for a, t in zip(args, argtypes):
            if "*" in a:
                assert t is ctypes.c_void_p, (name, a)
            else:
                ctype = a.rsplit(" ", 1)[0].replace("const ", "").strip()
                assert t is cmap[ctype], (name, a, t)


def test_tail_chain_size_matches_the_header():
    """TXE_TAIL_CHAIN_BYTES (include/txe.h) = the buffer ops._TailChain hands to the C entry points"""
    import re
    from taxoexpan_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "txe.h")).read()
    assert int(re.search(r"#define TXE_TAIL_CHAIN_BYTES (\d+)", hdr).group(1)) == _lib.TAIL_CHAIN_BYTES
    lib = _lib.load()
    assert lib.txe_gat_tail_flush(None, None) == 0                    # no chain: nothing to launch


def test_argument_validation_needs_no_gpu():
    """error paths return codes before anything is launched"""
    from taxoexpan_amd import _lib
    lib = _lib.load()
    assert lib.txe_gat_aggregate_fwd(None, None, 5, None, 0, None, None, 0, 4, 8, 0.2, 0.0, 0, 0, 1.0, None, 0, None, None, 0, None, 0.0, None,
                                     0, None) == -1
    assert lib.txe_readout_fwd(None, 3, None, 0, None, None, 8, None, None, None) == -1
    assert lib.txe_gat_dense_ws_bytes(100, 250, 50, 4, 500, 3) > 0
    assert lib.txe_gat_padded_k(250, 50) == 320 and lib.txe_gat_padded_f(4, 500) == 2048
    assert lib.txe_gat_aggregate_table_supported(4, 500, 2048, 3, 2080) == 1 and lib.txe_gat_aggregate_table_supported(5, 500, 2560, 3, 0) == 0
    assert lib.txe_gat_aggregate_table_fwd(None, None, 5, None, 2048, None, None, None, 3, 4, 500, 0.2, 0, 1.0, None, 0, None, 0, None, 0, None) == -1


def test_host_rng_restatement_matches_library():
    """taxoexpan_amd/rng.py == the hash the kernels inline (evaluated on the host by the library)"""
    from taxoexpan_amd import _lib, rng
    lib = _lib.load()
    for seed in (0, 1, 123456789, 2 ** 61 + 12345):
        idx = np.array([0, 1, 2, 63, 64, 1000, 2 ** 31, 2 ** 40 + 17], dtype=np.uint64)
        want = np.array([lib.txe_dropout_uniform_host(seed, int(i)) for i in idx], dtype=np.float32)
        got = rng.uniform01(seed, idx)
        assert np.array_equal(got, want)
    m = rng.keep_mask(42, (1000, 37), 0.3)
    assert abs(m.mean() - 0.7) < 0.01
    # bit-mask form used for feature dropout: words from the library (host evaluation) == rng.keep_mask_bits
    for seed, rows, cols, p in ((7, 5, 70, 0.1), (2 ** 40 + 3, 3, 32, 0.3), (11, 4, 17, 0.5)):
        wpr = (cols + 31) // 32
        bits = rng.keep_mask_bits(seed, rows, cols, p)
        for r in range(rows):
            for w in range(wpr):
                word = lib.txe_dropout_mask_word_host(seed, r * wpr + w, p)
                for b in range(32):
                    c = w * 32 + b
                    if c < cols:
                        assert bits[r, c] == ((word >> b) & 1), (seed, r, c)
    assert abs(rng.keep_mask_bits(5, 500, 300, 0.1).mean() - 0.9) < 0.005


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from taxoexpan_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.call("txe_gcn_norm", None, 0, None, None)
    except _lib.TxeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected TxeError")


def test_ops_refuse_host_tensors():
    import pytest
    import torch
    from taxoexpan_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.bilinear_project(torch.zeros(4, 3), torch.zeros(1, 3, 2))


def test_lds_direct_copies_are_the_only_users_of_m0():
    """csrc/txe_gemm_split.hip issues its global -> LDS copies from inline asm (`s_mov_b32 m0, <lds address>` + `global_load_lds_dwordx4`):
    M0 is a reserved register that cannot be named in a clobber list, so the source cannot tell the compiler about it.  What CAN be
    checked is the generated gfx950 ISA: in that file every mention of m0 is such a move, immediately followed by the copy that reads
    it -- no compiler-generated M0 user (readlane/movrel, s_sendmsg, GWS, LDS-direct ds_* with M0) exists for the asm to corrupt."""
    import re
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(REPO, "taxoexpan_amd", "csrc", "txe_gemm_split.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "split.s")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", src, "-o", out],
                           capture_output=True, text=True, cwd=os.path.dirname(src))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln.strip() for ln in open(out) if ln.strip() and not ln.strip().startswith((";", ".", "//"))]
    uses = [i for i, ln in enumerate(lines) if re.search(r"\bm0\b", ln.split(";")[0])]
    assert len(uses) >= 20                                   # the copies are there
    for i in uses:
        assert re.match(r"s_mov_b32 m0, s\d+", lines[i]), lines[i]
        assert lines[i + 1].startswith("global_load_lds_dwordx4"), (lines[i], lines[i + 1])
