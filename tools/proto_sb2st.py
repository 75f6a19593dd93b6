#!/usr/bin/env python3
"""NumPy prototype of the two-stage tridiagonalisation behind pg_syevd_dev (csrc/syevd2.hip) — the SAME block and task
structure as the device code, small enough to read; the GPU test hooks (pgx_sy2sb_dev, pgx_sb2st_dev, pgx_bt2_dev) are checked
against it and against LAPACK-level invariants.  CPU only, test/dev tooling (nothing in the product imports it).

    stage 1  dense -> band (half-width b):  panel QR by CholeskyQR2 + Householder reconstruction (LU of E - Q D),
             two-sided block update  A22 <- A22 - V W' - W V'
    stage 2  band -> tridiagonal: bulge chasing, one Householder reflector of length <= b per (sweep, step)
    back     U = Q1 (Q2 Z): stage-2 reflectors applied in (b + g - 1) x g parallelogram blocks (compact WY), then stage 1's

usage: proto_sb2st.py [n] [b] [g]
"""
import sys

import numpy as np


# ------------------------------------------------------------------------------------------------ stage 1
def cholqr2_reconstruct(P):
    """P (m x b, m >= b, full column rank) -> V (m x b unit lower trapezoidal), T (b x b upper), Rs = D R (upper) with
    (I - V T V')' P = [Rs; 0].  CholeskyQR2 for Q, then the LU-based reconstruction of the Householder form of Q D."""
    m, b = P.shape
    G = P.T @ P
    R1 = np.linalg.cholesky(G).T
    Q1 = P @ np.linalg.inv(R1)
    G2 = Q1.T @ Q1
    ortho_err = np.abs(G2 - np.eye(b)).max()          # device: flag the panel if this exceeds 0.05 (fallback to one-stage)
    R2 = np.linalg.cholesky(G2).T
    R = R2 @ R1
    Q = Q1 @ np.linalg.inv(R2)
    # LU without pivoting of  M = E - Q D, D chosen on the fly (d_j = -sign of the eliminated diagonal entry of Q)
    Qt = Q.copy()                                      # columns are eliminated in place: below the diagonal -> L, on/above -> q~
    Dg = np.ones(b)
    L = np.zeros((m, b))
    U = np.zeros((b, b))
    for j in range(b):
        Dg[j] = -1.0 if Qt[j, j] >= 0 else 1.0
        piv = 1.0 - Dg[j] * Qt[j, j]                   # = 1 + |q~_jj| >= 1
        U[:j, j] = -Dg[j] * Qt[:j, j]
        U[j, j] = piv
        L[j, j] = 1.0
        L[j + 1:, j] = -Dg[j] * Qt[j + 1:, j] / piv
        # eliminate row j from the later columns (rows > j)
        Qt[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], Qt[j, j + 1:])
    V = L
    T = U @ np.linalg.inv(V[:b, :].T)                  # T = U Y1^-T
    Rs = Dg[:, None] * R
    return V, T, Rs, ortho_err


def house_qr_small(P):
    """plain Householder QR of an m x b block with m <= b (last panel): V (m x b), T (b x b), Rs (m x b upper trapezoidal)"""
    m, b = P.shape
    A = P.copy()
    V = np.zeros((m, b)); tau = np.zeros(b)
    for j in range(min(m - 1, b)):
        x = A[j:, j]
        v, t, beta = house(x)
        V[j:, j] = v; tau[j] = t
        A[j:, j:] -= t * np.outer(v, v @ A[j:, j:])
        A[j + 1:, j] = 0.0
    T = np.zeros((b, b))
    for j in range(b):
        T[j, j] = tau[j]
        if j:
            T[:j, j] = -tau[j] * T[:j, :j] @ (V[:, :j].T @ V[:, j])
    return V, T, A


def house(x):
    """dlarfg: (v with v[0] = 1, tau, beta) such that (I - tau v v') x = beta e1"""
    alpha = x[0]
    xn2 = float(x[1:] @ x[1:])
    v = x.copy(); v[0] = 1.0
    if xn2 == 0.0:
        v[1:] = 0.0
        return v, 0.0, alpha
    beta = -np.copysign(np.sqrt(alpha * alpha + xn2), alpha)
    tau = (beta - alpha) / beta
    v[1:] = x[1:] / (alpha - beta)
    return v, tau, beta


def sy2sb(A, b):
    """A (n x n symmetric) -> band B (full symmetric storage, half-width b) and the panels' (j, V, T)."""
    A = A.copy()
    n = A.shape[0]
    panels = []
    worst = 0.0
    j = 0
    while n - j - b >= 2:
        m = n - j - b
        P = A[j + b:, j:j + b]
        if m > b:
            V, T, Rs, oe = cholqr2_reconstruct(P)
            worst = max(worst, oe)
            Rfull = np.zeros((m, b)); Rfull[:b] = np.triu(Rs)
        else:
            V, T, Rfull = house_qr_small(P)
        A[j + b:, j:j + b] = Rfull
        A[j:j + b, j + b:] = Rfull.T
        A22 = A[j + b:, j + b:]
        Y = A22 @ V @ T
        W = Y - 0.5 * V @ (T.T @ (V.T @ Y))
        A22 -= V @ W.T + W @ V.T
        panels.append((j, V, T))
        j += b
    return A, panels, worst


def apply_q1(panels, Z, b):
    """Z <- Q1 Z, Q1 = H_panel0 H_panel1 ...: last panel first"""
    for j, V, T in reversed(panels):
        rows = slice(j + b, None)
        Z[rows] -= V @ (T @ (V.T @ Z[rows]))
    return Z


# ------------------------------------------------------------------------------------------------ stage 2
def sb2st(B, b):
    """Bulge chasing on the band matrix B (full symmetric storage for readability), in the device kernel's task structure: step
    (s, k) owns ONE row block R = [r0, r0 + L), r0 = s + 1 + k b, and does, in this order,
        (1) k >= 1: right-apply the previous step's reflector to E = A[R, r0 - b : r0]   (creates the bulge)
        (2) reflector from the first column of E (k = 0: from column s), which becomes (beta, 0, ..., 0)'
        (3) left-apply it to the other columns of E          (4) two-sided on D = A[R, R]
    so a step reads and writes rows R only; step (s + 1, k) needs (s, k + 1) complete (one shared row), nothing else.
    Returns d, e and the reflectors: VV[s, r] = component on row r of sweep s's reflector (row r belongs to step
    (r - s - 1) // b), TAU[s, k]."""
    A = B.copy()
    n = A.shape[0]
    nsteps = (n + b - 1) // b + 1
    VV = np.zeros((n, n)); TAU = np.zeros((n, nsteps))
    for s in range(n - 2):
        vp, tp = None, 0.0
        k = 0
        while True:
            r0 = s + 1 + k * b
            if r0 >= n:
                break
            L = min(b, n - r0)
            R = slice(r0, r0 + L)
            if k >= 1:
                E = A[R, r0 - b:r0]                       # L x b
                E -= tp * np.outer(E @ vp, vp)            # (1)
                x = E[:, 0].copy()
            else:
                x = A[R, s].copy()
            if L >= 2:
                v, t, beta = house(x)
            else:
                v, t, beta = np.ones(1), 0.0, x[0]
            VV[s, R] = v; TAU[s, k] = t
            if k >= 1:
                E[:, 0] = 0.0; E[0, 0] = beta             # (2)
                E[:, 1:] -= t * np.outer(v, v @ E[:, 1:])  # (3)
                A[r0 - b:r0, R] = E.T
            else:
                A[R, s] = 0.0; A[r0, s] = beta
                A[s, R] = A[R, s]
            D = A[R, R]
            p = t * (D @ v)                               # (4)  D <- D - v q' - q v',  q = p - (tau/2)(p'v) v
            q = p - 0.5 * t * (p @ v) * v
            D -= np.outer(v, q) + np.outer(q, v)
            vp, tp = np.concatenate([v, np.zeros(b - L)]), t
            if L < b:
                break
            k += 1
    d = np.diag(A).copy(); e = np.diag(A, -1).copy()
    off = A - np.diag(d) - np.diag(e, -1) - np.diag(e, 1)
    return d, e, VV, TAU, np.abs(off).max()


def apply_q2_reference(VV, TAU, Z, b):
    """Z <- Q2 Z one reflector at a time (Q2 = H^(0) H^(1) ...: last sweep first)"""
    n = Z.shape[0]
    for s in range(n - 3, -1, -1):
        r0 = s + 1; k = 0
        while r0 < n:
            L = min(b, n - r0)
            v = VV[s, r0:r0 + L]; t = TAU[s, k]
            if t != 0.0:
                Z[r0:r0 + L] -= t * np.outer(v, v @ Z[r0:r0 + L])
            r0 += L; k += 1
    return Z


def bt2_blocks(VV, TAU, b, g):
    """The device's blocking: groups of g sweeps [s0, s0 + g), per step k one parallelogram block V_k of (b + g - 1) rows x g
    columns (column i = sweep s0 + i, shifted down by i) with its compact-WY factor T_k.  Yields (s0, [(row0, V_k, T_k), ...])."""
    n = VV.shape[0]
    nsw = n - 2
    for s0 in range(0, nsw, g):
        gg = min(g, nsw - s0)
        blocks = []
        k = 0
        while True:
            row0 = s0 + 1 + k * b
            if row0 >= n:
                break
            rows = min(b + gg - 1, n - row0)
            V = np.zeros((rows, gg)); tau = np.zeros(gg)
            for i in range(gg):
                s = s0 + i
                a = s + 1 + k * b
                if a >= n:
                    continue
                L = min(b, n - a)
                V[a - row0:a - row0 + L, i] = VV[s, a:a + L]
                tau[i] = TAU[s, k]
            T = np.zeros((gg, gg))
            for i in range(gg):                        # forward columnwise larft; tau = 0 columns contribute nothing
                T[i, i] = tau[i]
                if i:
                    T[:i, i] = -tau[i] * T[:i, :i] @ (V[:, :i].T @ V[:, i])
            blocks.append((row0, V, T))
            k += 1
        yield s0, blocks


def apply_q2_blocked(VV, TAU, Z, b, g):
    """Z <- Q2 Z with the blocked order: groups from the last to the first, inside a group the blocks k = 0, 1, ... downwards"""
    groups = list(bt2_blocks(VV, TAU, b, g))
    for s0, blocks in reversed(groups):
        for row0, V, T in blocks:
            rows = slice(row0, row0 + V.shape[0])
            Z[rows] -= V @ (T @ (V.T @ Z[rows]))
    return Z


# ------------------------------------------------------------------------------------------------ driver
def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 203
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    g = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rng = np.random.default_rng(1)
    G = rng.standard_normal((n, 2 * n))
    A = G @ G.T / (2 * n)
    lam_ref = np.linalg.eigvalsh(A)
    B, panels, worst = sy2sb(A, b)
    band_off = max((np.abs(np.diag(B, k)).max() for k in range(b + 1, n)), default=0.0)
    print(f"stage 1: panels {len(panels)}, worst CholQR pass-1 orthogonality error {worst:.2e}, outside band {band_off:.2e}, "
          f"eig diff {np.abs(np.linalg.eigvalsh(B) - lam_ref).max():.2e}")
    Q1 = apply_q1(panels, np.eye(n), b)
    print(f"         |Q1'Q1 - I| {np.abs(Q1.T @ Q1 - np.eye(n)).max():.2e}   |Q1 B Q1' - A| {np.abs(Q1 @ B @ Q1.T - A).max():.2e}")
    d, e, VV, TAU, off = sb2st(B, b)
    Tm = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    print(f"stage 2: off-tridiagonal residue {off:.2e}, eig diff {np.abs(np.linalg.eigvalsh(Tm) - lam_ref).max():.2e}")
    Q2a = apply_q2_reference(VV, TAU, np.eye(n), b)
    Q2b = apply_q2_blocked(VV, TAU, np.eye(n), b, g)
    print(f"         |Q2 T Q2' - B| {np.abs(Q2a @ Tm @ Q2a.T - B).max():.2e}   blocked vs one-by-one {np.abs(Q2a - Q2b).max():.2e}")
    lam, Z = np.linalg.eigh(Tm)
    U = apply_q1(panels, apply_q2_blocked(VV, TAU, Z.copy(), b, g), b)
    print(f"full   : |U'U - I| {np.abs(U.T @ U - np.eye(n)).max():.2e}   |A U - U lam| / |A| {np.abs(A @ U - U * lam).max() / np.abs(A).max():.2e}")


if __name__ == "__main__":
    main()
