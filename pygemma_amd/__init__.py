"""pygemma_amd — MI355X-native engine for pyGEMMA's per-SNP LMM association hot path."""
__version__ = "0.2.0"
