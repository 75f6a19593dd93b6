"""GPU parity of the rotation GEMM (pg_rotate_dev): bit-exact vs the oracle's k-ordered f32 fma chain,
and within f32-GEMM tolerance of float64 U.T @ X (the reference's sgemm is in the same error class)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pygemma_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,p", [(64, 32), (257, 130), (300, 128), (1000, 516), (1940, 48)])
def test_rotate_bit_exact_vs_oracle(n, p, ctx):
    from oracle import oracle as O
    from pygemma_amd import ops
    rng = np.random.default_rng(n * 7 + p)
    # asymmetric U (not orthogonal on purpose: a transposed or row/col-swapped kernel cannot pass)
    U = rng.standard_normal((n, n)).astype(np.float32)
    X = rng.standard_normal((n, p)).astype(np.float32)
    got = ops.rotate(U, X, ctx=ctx)
    ref = O.rotate(U, X)
    assert got.shape == ref.shape
    assert (got.view(np.uint32) == ref.view(np.uint32)).all(), np.abs(got - ref).max()
    exact = (U.astype(np.float64).T @ X.astype(np.float64)).T
    scale = np.sqrt(n)
    assert np.abs(got[:, :n] - exact).max() <= 2e-6 * scale * 8
    assert (got[:, n:] == 0).all()
