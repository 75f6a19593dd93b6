"""PLINK .bed genotypes kept packed (2 bits per call) all the way to the GPU.

The reference's callers read .bed files with pysnptools into float arrays, impute missing calls with the column mean and
hand the (n, p) float32 matrix to lmm.pygemma (experiments/benchmarks/benchmarks.py:233-244).  `PackedBed` is accepted by
`pygemma_amd.lmm.pygemma` in place of that matrix: the packed SNP records are uploaded as they are (n/4 bytes per SNP
instead of 4n) and decoded, mean-imputed and rotated on the device (pg_rotate_bed_dev).
"""
import os

import numpy as np

__all__ = ["PackedBed", "write_bed"]

_MAGIC = bytes([0x6C, 0x1B, 0x01])   # PLINK 1 .bed, SNP-major


class PackedBed:
    """A SNP-major PLINK .bed image: `data` is a (p, ceil(n/4)) uint8 array (memory-mapped when opened from a file).
    count_A1=False gives the number of A2 alleles per call, like pysnptools.Bed(..., count_A1=False)."""

    def __init__(self, data, n, count_A1=False, snps=None):
        data = np.asarray(data)
        if data.dtype != np.uint8 or data.ndim != 2 or data.shape[1] != (int(n) + 3) // 4:
            raise ValueError("data must be a (p, ceil(n/4)) uint8 array")
        self.data, self.n, self.p = data, int(n), int(data.shape[0])
        self.count_A1 = bool(count_A1)
        self.snps = snps

    @property
    def shape(self):
        return (self.n, self.p)

    @classmethod
    def open(cls, path, count_A1=False):
        """path: the .bed file or its prefix; n and p come from the line counts of .fam and .bim."""
        prefix = path[:-4] if path.endswith(".bed") else path
        with open(prefix + ".fam") as f:
            n = sum(1 for line in f if line.strip())
        snps = []
        with open(prefix + ".bim") as f:
            for line in f:
                t = line.split()
                if t:
                    snps.append(t[1] if len(t) > 1 else t[0])
        p, bpr = len(snps), (n + 3) // 4
        with open(prefix + ".bed", "rb") as f:
            if f.read(3) != _MAGIC:
                raise ValueError(f"{prefix}.bed is not a SNP-major PLINK 1 .bed file")
        if os.path.getsize(prefix + ".bed") != 3 + p * bpr:
            raise ValueError(f"{prefix}.bed: size does not match {p} SNPs x {n} samples")
        data = np.memmap(prefix + ".bed", dtype=np.uint8, mode="r", offset=3, shape=(p, bpr))
        return cls(data, n, count_A1=count_A1, snps=snps)

    def to_float(self, impute=True):
        """Host decode to the (n, p) float32 matrix the reference's callers build (NaN or column-mean for missing calls)."""
        shifts = np.arange(4, dtype=np.uint8) * 2
        codes = ((self.data[:, :, None] >> shifts[None, None, :]) & 3).reshape(self.p, -1)[:, :self.n]
        lut = np.array([2.0, np.nan, 1.0, 0.0] if self.count_A1 else [0.0, np.nan, 1.0, 2.0], np.float64)
        X = lut[codes].T
        if impute:
            mu = np.nanmean(X, axis=0)
            X = np.where(np.isnan(X), mu[None, :], X)
        return np.ascontiguousarray(X, np.float32)


def write_bed(prefix, dosage_A2, snps=None):
    """Write (n, p) A2 dosages (0/1/2, NaN = missing) as prefix.bed/.bim/.fam.  For tests and examples."""
    G = np.asarray(dosage_A2, np.float64)
    n, p = G.shape
    code = np.full(G.shape, 1, np.uint8)              # 01 missing
    code[G == 0] = 0; code[G == 1] = 2; code[G == 2] = 3
    pad = (-n) % 4
    c = np.concatenate([code.T, np.zeros((p, pad), np.uint8)], axis=1).reshape(p, -1, 4)
    packed = (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)
    with open(prefix + ".bed", "wb") as f:
        f.write(_MAGIC)
        f.write(packed.tobytes())
    snps = snps or [f"rs{j}" for j in range(p)]
    with open(prefix + ".bim", "w") as f:
        for j, s in enumerate(snps):
            f.write(f"1\t{s}\t0\t{j + 1}\tA\tG\n")
    with open(prefix + ".fam", "w") as f:
        for i in range(n):
            f.write(f"f{i} i{i} 0 0 0 -9\n")
    return packed
