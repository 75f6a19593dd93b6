"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol include/pygemma_hip.h
declares (no compute without a GPU), the host logic mirrors the reference's (SampleIter split, casts, error
behaviour), and the product never reaches for the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pygemma_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pgx?_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from pygemma_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.load()
    decl = _declared_symbols()
    assert len(decl) >= 20
    for sym in decl:
        assert hasattr(L, sym), f"{sym} declared in include/pygemma_hip.h but not exported"
    assert set(_lib.SYMBOLS) <= set(decl)
    assert b"gfx950" in L.pg_version()
    # and nothing of ours is exported that the header does not declare (VERDICT r1 hygiene: pgx_* test hooks are declared now)
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith(("pg_", "pgx_"))}
    assert exported <= set(decl), sorted(exported - set(decl))


def test_no_gpu_means_loud_failure_not_fallback():
    from pygemma_amd import _lib, lmm
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_lib.PgError):
        _lib.Context(0)
    rng = np.random.default_rng(0)
    with pytest.raises(_lib.PgError):
        lmm.pygemma(rng.standard_normal((20, 1)), rng.standard_normal((20, 4)), np.ones((20, 1)), np.ones(20), eigen=False)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pygemma_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import oracle|from oracle)", src, flags=re.M), f
                code = re.sub(r"//.*|#.*", "", src)   # comments may cite the oracle; code may not load it
                assert "liboracle" not in code and "oracle/" not in code, f


def test_sampleiter_matches_reference_split():
    """lmm/lmm.py:427-434: cols_per_proc = ceil(p/nproc), blocks [r*cols, min((r+1)*cols, p))."""
    from pygemma_amd.lmm import SampleIter
    for p, nproc in [(10, 3), (12226, 8), (100000, 8), (7, 7), (5, 1), (9, 4)]:
        blocks = list(SampleIter(p, nproc))
        cols = int(np.ceil(p / nproc))
        exp = [(r * cols, min((r + 1) * cols, p)) for r in range(nproc) if r * cols < p]
        assert blocks == exp
        assert blocks[0][0] == 0 and blocks[-1][1] == p
        assert all(b0[1] == b1[0] for b0, b1 in zip(blocks, blocks[1:]))


def test_de_mode_raises_like_a_broken_reference_path():
    from pygemma_amd import lmm
    with pytest.raises(NotImplementedError):
        lmm.pygemma(np.zeros((4, 1)), np.zeros((4, 2)), np.ones((4, 1)), np.eye(4), de=True)


def test_drop_in_import_path():
    from pygemma import lmm
    import inspect
    sig = inspect.signature(lmm.pygemma)
    ref = ["Y", "X", "W", "K", "Z", "snps", "verbose", "disable_checks", "de", "grid", "eigen", "nproc"]      # lmm/lmm.py:87
    names = list(sig.parameters)
    assert names[:len(ref)] == ref              # positional order and names of the reference, so every reference call site works
    # extensions come after the reference's parameters, keyword-style, default off: restartable runs, the LRT
    # columns (N2), host-supplied eigenpairs (BASELINE config 5), streaming counters
    assert names[len(ref):] == ["checkpoint", "lrt", "eigenpairs", "stats"]
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect._empty}
    assert d == {"Z": None, "snps": None, "verbose": 0, "disable_checks": True, "de": False, "grid": False, "eigen": True, "nproc": 1,
                 "checkpoint": None, "lrt": False, "eigenpairs": None, "stats": None}
    for name in ("precompute_mat", "calc_lambda_restricted", "calc_beta_vg_ve_restricted_overload", "newton",
                 "likelihood_restricted_lambda_overload", "likelihood_derivative1_restricted_lambda_overload",
                 "likelihood_derivative2_restricted_lambda_overload"):                                   # tests/test_pygemma.py:256-294
        assert callable(getattr(lmm, name))


def test_packed_bed_roundtrip(tmp_path):
    """bed.py: write_bed -> PackedBed.open -> to_float reproduces the dosages (both allele conventions, NaN for missing)."""
    import numpy as np
    from pygemma_amd.bed import PackedBed, write_bed
    rng = np.random.default_rng(5)
    n, p = 37, 19
    G = rng.integers(0, 3, size=(n, p)).astype(np.float64)
    G[rng.random((n, p)) < 0.1] = np.nan
    prefix = str(tmp_path / "t")
    packed = write_bed(prefix, G)
    assert packed.shape == (p, (n + 3) // 4)
    b = PackedBed.open(prefix)
    assert b.shape == (n, p) and b.snps[3] == "rs3"
    np.testing.assert_array_equal(b.to_float(impute=False), G.astype(np.float32))
    b1 = PackedBed.open(prefix + ".bed", count_A1=True)
    np.testing.assert_array_equal(b1.to_float(impute=False), (2 - G).astype(np.float32))
    with open(prefix + ".bed", "r+b") as f:
        f.write(b"\x00")
    import pytest
    with pytest.raises(ValueError):
        PackedBed.open(prefix)
