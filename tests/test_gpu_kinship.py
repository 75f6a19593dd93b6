"""N3: relatedness matrix on the device — column standardisation + lower-triangle syrk — against the
reference's calculate_genetic_relatedness_matrix (experiments/animal_gwas/run_gwas.py:46-56) evaluated in float64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ref_grm(X):
    X = X.astype(np.float64)
    sd = np.std(X, axis=0)
    sd[sd == 0] = 1
    Z = (X - np.mean(X, axis=0)) / sd
    return Z @ Z.T / X.shape[1]


@pytest.mark.parametrize("n,p", [(257, 1000), (640, 333), (1500, 4100)])
def test_kinship_matches_float64_reference(n, p):
    from pygemma_amd import lmm
    rng = np.random.default_rng(n)
    G = rng.binomial(2, rng.uniform(0.05, 0.5, p), size=(n, p)).astype(np.float32)
    G[:, 3] = 1.0                                   # monomorphic column: sd == 0 -> 1, contributes zeros
    K = lmm.kinship(G)
    ref = ref_grm(G)
    assert K.dtype == np.float32 and K.shape == (n, n)
    assert (K.view(np.uint32) == K.T.copy().view(np.uint32)).all()          # mirrored, bit-symmetric
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))
    assert np.abs(K - ref).max() <= 1e-6 * scale.max() * np.sqrt(p) / 8      # fp32 accumulation over p terms
    assert np.abs(K - ref).max() / np.abs(ref).max() <= 1e-6 + 6e-8 * np.sqrt(p)   # random-walk growth of the fp32 accumulation


def test_kinship_without_standardisation_and_feeds_eigensolver():
    from pygemma_amd import lmm, synth
    rng = np.random.default_rng(5)
    n, p = 300, 900
    Z = synth.genotypes(rng, n, p)                  # already standardised
    K0 = lmm.kinship(Z, standardize=False)
    ref = Z.astype(np.float64) @ Z.astype(np.float64).T / p
    assert np.abs(K0 - ref).max() / np.abs(ref).max() <= 2e-6
    X = synth.genotypes(rng, n, 50)
    W = np.ones((n, 1), np.float32)
    y = (0.4 * X[:, :1] + rng.standard_normal((n, 1))).astype(np.float32)
    a = lmm.pygemma(y, X, W, K0)
    b = lmm.pygemma(y, X, W, ref.astype(np.float32))
    np.testing.assert_allclose(a["beta"].to_numpy(), b["beta"].to_numpy(), rtol=2e-3, atol=1e-5)
    np.testing.assert_allclose(a["p_wald"].to_numpy(), b["p_wald"].to_numpy(), rtol=2e-2)
