"""`from pygemma import lmm; lmm.pygemma(Y, X, W, K, snps=...)` — same call as the reference (lmm/lmm.py:87), and the
model-level functions its tests call directly (`lmm.precompute_mat`, `lmm.calc_lambda_restricted`, `lmm.newton`, ...)."""
from pygemma_amd.lmm import *  # noqa: F401,F403
from pygemma_amd.lmm import SampleIter, pygemma  # noqa: F401
