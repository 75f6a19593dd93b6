"""`from pygemma import lmm; lmm.pygemma(Y, X, W, K, snps=...)` — same call as the reference (lmm/lmm.py:87)."""
from pygemma_amd.lmm import SampleIter, pygemma  # noqa: F401
