"""Drop-in import path of the reference (`from pygemma import lmm`, README.md:85-88 of rlangefe/pygemma),
backed by the MI355X-native engine in pygemma_amd."""
from . import lmm  # noqa: F401
