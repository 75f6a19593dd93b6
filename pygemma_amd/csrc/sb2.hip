// sb2.hip — H1 (lmm/lmm.py:151-162 / :196-207: the reference calls scipy.linalg.eigh): the TWO-STAGE tridiagonalisation of the
// fp64 eigensolver, written for gfx950.  The one-stage reduction of syevd.hip reads the whole trailing matrix once per column
// (HBM-bound, ~3 launches per column); here almost all the work is fp64-MFMA GEMM:
//
//   stage 1  dense -> band of half-width b = 64.  Per panel of 64 columns: tall-skinny QR by CholeskyQR2 (two Gram GEMMs + two
//            64 x 64 Cholesky factorisations, the orthogonality of the first pass checked on the second Gram), its Householder
//            form (V unit lower trapezoidal, T) rebuilt from Q by an LU factorisation of E - Q D without pivoting (Ballard et al.,
//            "Reconstructing Householder vectors from TSQR"), then the two-sided block update A22 -= V W' + W V' as ONE GEMM of
//            depth 2b with operands [V W] and [W V] (the second read from the first with its k index XOR-ed by 64).
//            A panel that CholeskyQR2 cannot factor to working accuracy (rank-deficient K) raises a device flag; the caller then
//            solves with the one-stage path.
//   stage 2  band -> tridiagonal by bulge chasing in ONE persistent kernel.  A step (sweep s, block K) owns the 64 rows
//            r0 = s + 1 + 64 K .. of the compact band and (1) applies the previous step's reflector from the right, (2) forms its own
//            reflector from the first column, (3) applies it from the left and (4) on both sides of the diagonal block — all in LDS.
//            bc_stationary_kernel: workgroup K does step (s, K) of every sweep and KEEPS its rows in LDS; a row and a reflector per
//            sweep travel between neighbours through mailboxes of self-validating 16-byte chunks.  bc_kernel (matrices with more
//            row blocks than resident workgroups): rows through memory, a progress word per sweep (write-through stores, drained,
//            then the flag: the hand-off of the MI355X guide's Guideline 16).
//   back     U = Q1 (Q2 Z).  Q2: the length-64 reflectors of 64 consecutive sweeps at one block index form a 127 x 64 parallelogram
//            with a compact-WY factor; 2 x 2 super-blocks (two groups x two block indices, 255 rows) on anti-diagonal wavefronts touch
//            disjoint rows 256 apart; a workgroup applies the four blocks to a 256 x 64 slab of Z held in registers.  Q1: blocks of
//            256 reflectors as in the one-stage solver.
#include "sb2.hpp"
#include "dgemm.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace pg {

constexpr int B = SB_B;          // 64
constexpr int P65 = B + 1;       // LDS pitch of a 64 x 64 matrix
constexpr int MAT = B * P65;     // doubles per LDS matrix
#ifndef PG_BT1_BLOCK
#define PG_BT1_BLOCK 256
#endif
constexpr int BT1_BLOCK = PG_BT1_BLOCK;   // reflectors per block of the stage-1 back-transformation

// small matrices in w.sm (b x b each)
enum { SM_G1 = 0, SM_R1, SM_R1INV, SM_G2, SM_XM, SM_M1, SM_M2, SM_RPROD, SM_COUNT };

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_d(double x, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double wave_sum64(double v)
{
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// ---- 64 x 64 building blocks on LDS matrices [64][65], 256 threads -----------------------------------------------------------
// upper Cholesky factor in place: A = R'R (upper triangle of A in, R out, strict lower part zeroed).  Returns false (to every
// thread) when a pivot is not positive / not finite; the factorisation then continues on a substitute pivot so that nothing
// downstream sees a NaN it did not bring itself.
__device__ bool chol_upper64(double *A, int tid)
{
    bool ok = true;
    for (int k = 0; k < B; k++) {
        const double akk = A[k * P65 + k];
        const bool bad = !(akk > 0.0) || !(akk < 1.0e300);
        const double d = bad ? 1.0 : sqrt(akk);
        ok = ok && !bad;
        __syncthreads();
        if (tid < B) {
            const int j = tid;
            if (j == k) A[k * P65 + k] = d;
            else if (j > k) A[k * P65 + j] /= d;
            else A[k * P65 + j] = 0.0;     // (row k, column j < k): strict lower part
        }
        __syncthreads();
        const int i = tid >> 2, j0 = (tid & 3) * 16;
        if (i > k) {
            const double aki = A[k * P65 + i];
#pragma unroll 4
            for (int j = j0; j < j0 + 16; j++)
                if (j >= i) A[i * P65 + j] -= aki * A[k * P65 + j];
        }
        __syncthreads();
    }
    return ok;
}

// X U = Bm for X (64 x 64), U upper triangular: row i of X by forward substitution along j, four lanes per row splitting the
// inner product.  ucoef(k, j) = U[k][j] for k < j, udiag(j) = U[j][j], rhs(i, j) = Bm[i][j].  All 256 threads call.
template <class UC, class UD, class RH>
__device__ void solve_right_upper64(double *X, UC ucoef, UD udiag, RH rhs, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int i = 16 * wave + (lane >> 2), part = lane & 3;
    for (int j = 0; j < B; j++) {
        double s = 0.0;
        for (int k = part; k < j; k += 4) s = fma(X[i * P65 + k], ucoef(k, j), s);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (part == 0) X[i * P65 + j] = (rhs(i, j) - s) / udiag(j);
        wave_sync_lds();
    }
    __syncthreads();
}

// C = A Bm (64 x 64 x 64) from LDS operands given by accessors; out(i, j, value) consumes the result
template <class FA, class FB, class OUT>
__device__ void mm64(FA a, FB b, OUT out, int tid)
{
    const int i = tid >> 2, j0 = (tid & 3) * 16;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = 0.0;
    for (int k = 0; k < B; k++) {
        const double aik = a(i, k);
#pragma unroll
        for (int q = 0; q < 16; q++) acc[q] = fma(aik, b(k, j0 + q), acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 16; q++) out(i, j0 + q, acc[q]);
}

// C = A Bm (64 x 64 x 64, operands in LDS, row-major, pitch P65) on the fp64 MFMA: wavefront w computes rows 16 w .. 16 w + 15 (four
// 16 x 16 tiles); acc[tj][e] = C[16 w + (lane >> 4) + 4 e][16 tj + (lane & 15)].  (The scalar-FMA version mm64 spent 28 us of a
// lone workgroup's time per product — LDS latency, 17 reads per 16 FMAs — where this one needs 5 reads per 4 MFMAs.)
__device__ __forceinline__ void mm64_mfma(const double *A, const double *Bm, doublex4 (&acc)[4], int tid)
{
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[tj][e] = 0.0;
#pragma unroll 4
    for (int ks = 0; ks < B / 4; ks++) {
        const double a = A[(16 * wave + r16) * P65 + 4 * ks + k4];
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
            acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bm[(4 * ks + k4) * P65 + 16 * tj + r16], acc[tj], 0, 0, 0);
    }
}

__device__ __forceinline__ void load64(double *dst, const double *src, long long ld, int tid)
{
    const int i = tid >> 2, j0 = (tid & 3) * 16;
#pragma unroll
    for (int q = 0; q < 16; q++) dst[i * P65 + j0 + q] = src[(long long)i * ld + j0 + q];
}

// row i of X by one lane: X U = rhs with U upper triangular (coefficients wave-uniform: LDS broadcast reads); a lane reads only
// what it wrote itself, so the 64 rows of one wavefront need no synchronisation at all.  Called by ONE wavefront (lane = row).
template <class UC, class UD, class RH>
__device__ __forceinline__ void solve_rows_wave(double *X, UC ucoef, UD urcp, RH rhs, int lane)
{
    for (int j = 0; j < B; j++) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = 0;
        for (; k + 4 <= j; k += 4) {          // four independent chains: eight LDS reads in flight
            const double x0 = X[lane * P65 + k], x1 = X[lane * P65 + k + 1], x2 = X[lane * P65 + k + 2], x3 = X[lane * P65 + k + 3];
            const double u0 = ucoef(k, j), u1 = ucoef(k + 1, j), u2 = ucoef(k + 2, j), u3 = ucoef(k + 3, j);
            s0 = fma(x0, u0, s0); s1 = fma(x1, u1, s1); s2 = fma(x2, u2, s2); s3 = fma(x3, u3, s3);
        }
        for (; k < j; k++) s0 = fma(X[lane * P65 + k], ucoef(k, j), s0);
        X[lane * P65 + j] = (rhs(lane, j) - ((s0 + s1) + (s2 + s3))) * urcp(j);
    }
}

// X C = R for X in place (X holds R on entry; LDS, pitch P65), C upper triangular with coef(k, j) = C[k][j] (k < j) and
// rcp(j) = 1 / C[j][j], by ONE wavefront, in column blocks of 16: the part of the right-hand side that earlier blocks determine
// is an fp64-MFMA product (A operand = the finished columns of X, B operand = C through the accessor), the 16 x 16 triangle is
// solved with lane = row, the row's 16 unknowns in registers and the coefficients broadcast from LDS — 480 dependent terms per
// lane instead of the 2016 of solve_rows_wave (43 us for the two solves of recon_kernel, its largest part).
template <class UC, class UD>
__device__ __forceinline__ void solve_rows_blocked(double *X, UC coef, UD rcp, int lane)
{
    const int r16 = lane & 15, k4 = lane >> 4;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        if (b > 0) {
            doublex4 acc[4];
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[ti][e] = X[(16 * ti + k4 + 4 * e) * P65 + 16 * b + r16];
#pragma unroll
            for (int ks = 0; ks < 4 * b; ks++) {
                const double cb = coef(4 * ks + k4, 16 * b + r16);
#pragma unroll
                for (int ti = 0; ti < 4; ti++)
                    acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[(16 * ti + r16) * P65 + 4 * ks + k4], cb, acc[ti], 0, 0, 0);
            }
            wave_sync_lds();
#pragma unroll
            for (int ti = 0; ti < 4; ti++)
#pragma unroll
                for (int e = 0; e < 4; e++) X[(16 * ti + k4 + 4 * e) * P65 + 16 * b + r16] = acc[ti][e];
            wave_sync_lds();
        }
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = X[lane * P65 + 16 * b + j];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < j; k++) {
                if (k & 1) s1 = fma(x[k], coef(16 * b + k, 16 * b + j), s1);
                else s0 = fma(x[k], coef(16 * b + k, 16 * b + j), s0);
            }
            x[j] = (x[j] - (s0 + s1)) * rcp(16 * b + j);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) X[lane * P65 + 16 * b + j] = x[j];
        wave_sync_lds();
    }
}

// ---- pass 1 of CholeskyQR2: G -> R1 (upper) and R1^-1, one wavefront, everything in registers -------------------------------
// Lane j keeps column j of the matrix (64 doubles).  Right-looking Cholesky: at step k the pivot and the row R[k][i] reach every lane
// through v_readlane (compile-time lane numbers: all loops are unrolled), no LDS, no barriers.  Then X R = I with lane i = row i of X,
// the coefficient R[k][j] read from lane j's register k.  (The LDS version with run-time loops spent its time in LDS latency: 82 us;
// the first, with workgroup barriers: 112 us.)
__global__ __launch_bounds__(64) void chol_inv_kernel(const double *G, double *R, double *Rinv, int *fail)
{
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);     // runs beside the trailing update's MFMA waves (look-ahead): a latency chain, served first
    double a[B], rc[B];
#pragma unroll
    for (int i = 0; i < B; i++) a[i] = G[i * B + lane];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < B; k++) {
        const double akk = readlane_d(a[k], k);
        const bool bad = !(akk > 0.0) || !(akk < 1.0e300);
        ok = ok && !bad;
        const double d = bad ? 1.0 : sqrt(akk), rcp = 1.0 / d;
        rc[k] = rcp;
        const double rk = (lane > k) ? a[k] * rcp : ((lane == k) ? d : 0.0);     // R[k][lane]
        a[k] = rk;
#pragma unroll
        for (int i = k + 1; i < B; i++) a[i] = fma(-readlane_d(rk, i), rk, a[i]);   // A[i][lane] -= R[k][i] R[k][lane]  (used for lane >= i)
    }
    if (!ok && lane == 0) atomicOr(fail, 1);
    // X R = I through LDS (blocked: solve_rows_blocked; the register version — 2016 dependent v_readlane + FMA pairs — took as long
    // as the factorisation itself)
    extern __shared__ double lds[];
    double *Rm = lds, *Xl = lds + MAT, *rcs = lds + 2 * MAT;
#pragma unroll
    for (int i = 0; i < B; i++) {
        const double r = (lane >= i) ? a[i] : 0.0;
        Rm[i * P65 + lane] = r;
        R[i * B + lane] = r;
        Xl[i * P65 + lane] = (i == lane) ? 1.0 : 0.0;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < B; k++) rcs[k] = rc[k];
    }
    wave_sync_lds();
    solve_rows_blocked(Xl, [&](int k, int j) { return Rm[k * P65 + j]; }, [&](int j) { return rcs[j]; }, lane);
#pragma unroll
    for (int i = 0; i < B; i++) Rinv[i * B + lane] = (lane >= i) ? Xl[i * P65 + lane] : 0.0;
}

// ---- pass 1, blocked (r4): the same R1 and R1^-1 in ~13 us instead of 47 ----------------------------------------------------------
// chol_inv_kernel above is 2 016 dependent (two v_readlane + FMA) steps in one wavefront, fully unrolled (~50 KB of code: beside
// another kernel its instruction fetches alone made it 2 - 3 x slower).  Here the factorisation walks four row panels of 16 rows: ONE
// wavefront holds the panel's 16 x 64 entries with lane = COLUMN and runs the 16 elimination steps of the panel on all columns at once
// (120 readlane + FMA pairs per panel: the multipliers R[k][i] are wave-uniform), which yields the panel's rows of R — diagonal block and
// everything right of it — without a separate triangular solve; the trailing blocks are updated as 16 x 16 x 16 MFMA products by all
// four wavefronts.  1 / sqrt(pivot) from v_rsq_f64 + two Newton steps (the pivot's square root and reciprocal in ~100 cycles instead
// of ~300).  R^-1: the four 16 x 16 diagonal inverses by one wavefront each (lane = row, coefficients by readlane), then block column K
// by wavefront K as MFMA products: X_IK = -X_II sum_{I < M <= K} R_IM X_MK.
__device__ __forceinline__ double rsqrt_newton(double a)
{
    double r = __builtin_amdgcn_rsq(a);
    // two Newton steps r <- r (1.5 - 0.5 a r^2): quadratic from the instruction's ~2^-26 to below 2^-52
    const double h = 0.5 * a;
    r = fma(r, fma(-h * r, r, 0.5), r);
    r = fma(r, fma(-h * r, r, 0.5), r);
    return r;
}
// acc (16 x 16, MFMA C layout: row (lane >> 4) + 4 e, column lane & 15) += A B with A(i, k) and B(k, j) given by accessors, K = 16
template <class FA, class FB>
__device__ __forceinline__ void tile16_mma(doublex4 &acc, FA a, FB b, int lane)
{
    const int r16 = lane & 15, k4 = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(r16, 4 * ks + k4), b(4 * ks + k4, r16), acc, 0, 0, 0);
}
__global__ __launch_bounds__(256) void chol_inv16_kernel(const double *G, double *R, double *Rinv, int *fail)
{
    extern __shared__ double lds[];      // (dynamic: 67 KB — a static allocation above 64 KB is not honoured at launch)
    double *A = lds;                     // the matrix; its upper triangle becomes R
    double *X = lds + MAT;               // R^-1
    double *rinv_d = lds + 2 * MAT;      // 1 / R[j][j]
    __shared__ int bad_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    __builtin_amdgcn_s_setprio(3);
    {
        const int i = tid >> 2, j0 = (tid & 3) * 16;
#pragma unroll
        for (int q = 0; q < 16; q++) { A[i * P65 + j0 + q] = G[i * B + j0 + q]; X[i * P65 + j0 + q] = 0.0; }
    }
    if (tid == 0) bad_sh = 0;
    __syncthreads();
    for (int J = 0; J < 4; J++) {
        const int o = 16 * J;
        if (wave == 0) {
            // rows o .. o + 15, all columns >= o: lane = column
            double a[16];
#pragma unroll
            for (int r = 0; r < 16; r++) a[r] = A[(o + r) * P65 + lane];
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const double akk = readlane_d(a[k], o + k);      // (o + k is wave-uniform: v_readlane with a scalar lane select)
                const bool bad = !(akk > 0.0) || !(akk < 1.0e300);
                ok = ok && !bad;
                const double inv = bad ? 1.0 : rsqrt_newton(akk), d = bad ? 1.0 : akk * inv;
                if (lane == o + k) rinv_d[o + k] = inv;
                const double rk = (lane > o + k) ? a[k] * inv : ((lane == o + k) ? d : 0.0);      // R[o + k][lane]
                a[k] = rk;
#pragma unroll
                for (int i = k + 1; i < 16; i++) a[i] = fma(-readlane_d(rk, o + i), rk, a[i]);      // row o + i -= R[o+k][o+i] R[o+k][:]
            }
            if (!ok && lane == 0) bad_sh = 1;
#pragma unroll
            for (int r = 0; r < 16; r++) A[(o + r) * P65 + lane] = (lane >= o + r) ? a[r] : 0.0;
        }
        __syncthreads();
        // trailing blocks (I, K), J < I <= K <= 3:  A[I][K] -= R[J][I]' R[J][K]
        {
            int t = 0;
            for (int I = J + 1; I < 4; I++)
                for (int K = I; K < 4; K++, t++) {
                    if ((t & 3) != wave) continue;
                    doublex4 acc;
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = A[(16 * I + k4 + 4 * e) * P65 + 16 * K + r16];
                    tile16_mma(acc, [&](int i, int k) { return -A[(o + k) * P65 + 16 * I + i]; }, [&](int k, int j) { return A[(o + k) * P65 + 16 * K + j]; }, lane);
#pragma unroll
                    for (int e = 0; e < 4; e++) A[(16 * I + k4 + 4 * e) * P65 + 16 * K + r16] = acc[e];
                }
        }
        __syncthreads();
    }
    // ---- R^-1.  Diagonal blocks: wavefront J inverts R_JJ, lane = row i of X_JJ, x[j] = (delta_ij - sum_{k < j} x[k] R[k][j]) / R[j][j]
    {
        const int o = 16 * wave;
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            double s_ = (lane == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < j; k++) s_ = fma(-x[k], A[(o + k) * P65 + o + j], s_);      // (uniform address: an LDS broadcast)
            x[j] = (lane <= j && lane < 16) ? s_ * rinv_d[o + j] : 0.0;
        }
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 16; j++) X[(o + lane) * P65 + o + j] = x[j];
        }
    }
    __syncthreads();
    // block column K (wavefront K), rows I = K - 1 .. 0
    if (wave >= 1) {
        const int K = wave;
        for (int I = K - 1; I >= 0; I--) {
            doublex4 sacc;
#pragma unroll
            for (int e = 0; e < 4; e++) sacc[e] = 0.0;
            for (int Mb = I + 1; Mb <= K; Mb++)
                tile16_mma(sacc, [&](int i, int k) { return A[(16 * I + i) * P65 + 16 * Mb + k]; }, [&](int k, int j) { return X[(16 * Mb + k) * P65 + 16 * K + j]; }, lane);
            // S through LDS (X[I][K] is not in use yet), then X_IK = -X_II S
#pragma unroll
            for (int e = 0; e < 4; e++) X[(16 * I + k4 + 4 * e) * P65 + 16 * K + r16] = sacc[e];
            wave_sync_lds();
            doublex4 xa;
#pragma unroll
            for (int e = 0; e < 4; e++) xa[e] = 0.0;
            tile16_mma(xa, [&](int i, int k) { return -X[(16 * I + i) * P65 + 16 * I + k]; }, [&](int k, int j) { return X[(16 * I + k) * P65 + 16 * K + j]; }, lane);
            wave_sync_lds();
#pragma unroll
            for (int e = 0; e < 4; e++) X[(16 * I + k4 + 4 * e) * P65 + 16 * K + r16] = xa[e];
            wave_sync_lds();
        }
    }
    __syncthreads();
    if (bad_sh != 0 && tid == 0) atomicOr(fail, 1);
    {
        const int i = tid >> 2, j0 = (tid & 3) * 16;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = j0 + q;
            R[i * B + c] = (c >= i) ? A[i * P65 + c] : 0.0;
            Rinv[i * B + c] = (c >= i) ? X[i * P65 + c] : 0.0;
        }
    }
}

// ---- pass 2 + Householder reconstruction ------------------------------------------------------------------------------------
// in : G2 = Q1'Q1, R1, Q1top = first 64 rows of Q1 (ld 64)
// out: Rs = D R2 R1 into the band block Aband (ld lda; only the upper triangle is written), V's top block (unit lower) into VW (ld 2b)
//      and Vst (ld ldv), T (64 x 64), Xm = -R2^-1 D U^-1 (so that the rows of V below the top block are Q1 Xm)
// R2: when |G2 - I| < 1e-8 (every panel of a well-conditioned K) its first-order form I + triu(G2 - I, 1) + diag(G2 - I)/2 and
// R2^-1 = 2I - R2 are exact to working precision (the neglected terms are of second order, < 1e-16); otherwise the Cholesky route.
__global__ __launch_bounds__(256) void recon_kernel(const double *G2, const double *R1g, const double *Q1top, double *Aband, long long lda,
                                                    double *VW, double *Vst, long long ldv, double *Tout, double *Xm, double *Rprod, int *fail)
{
    // three LDS matrices (100 KB: the kernel has to find room on a CU beside the workgroups of the trailing update it overlaps):
    // M0: G2 -> R2 -> T;   M1: R2^-1 -> Xm (in place);   M2: R1 -> top block of Q -> its LU
    extern __shared__ double lds[];
    __builtin_amdgcn_s_setprio(3);     // runs beside the trailing update's MFMA waves (look-ahead): a latency chain, served first
    double *M0 = lds, *M1 = lds + MAT, *M2 = lds + 2 * MAT;
    __shared__ double Dg[B], piv[B], prc[B];
    __shared__ double red[4];
    const int tid = threadIdx.x, i = tid >> 2, j0 = (tid & 3) * 16, lane = tid & 63, wave = tid >> 6;
#ifdef PG_RECON_TIME
    long long tk[8]; int nt = 0;
#define RT_MARK() do { __syncthreads(); tk[nt++] = wall_clock64(); } while (0)
    RT_MARK();
#else
#define RT_MARK() do { } while (0)
#endif
    // orthogonality of pass 1 = |G2 - I|_max: CholeskyQR2 reaches working accuracy when this is well below 1
    double err = 0.0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const double g = G2[i * B + j0 + q];
        M0[i * P65 + j0 + q] = g;
        const double dlt = fabs(g - ((i == j0 + q) ? 1.0 : 0.0));
        err = (dlt > err || !(dlt == dlt)) ? dlt : err;      // a NaN sticks
    }
    for (int s = 1; s < 64; s <<= 1) { const double o = __shfl_xor(err, s, 64); err = (o > err || !(o == o)) ? o : err; }
    if (lane == 0) red[wave] = err;
    load64(M2, R1g, B, tid);                                                      // M2 = R1
    __syncthreads();
    double e_all = red[0];
    for (int w = 1; w < 4; w++) e_all = (red[w] > e_all || !(red[w] == red[w])) ? red[w] : e_all;
    if (tid == 0 && !(e_all <= 0.05)) atomicOr(fail, 1);
    if (e_all < 1.0e-8) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = j0 + q;
            const double g = M0[i * P65 + c];
            const double r2 = (c > i) ? g : ((c == i) ? 1.0 + 0.5 * (g - 1.0) : 0.0);
            M0[i * P65 + c] = r2;                                                 // M0 = R2
            M1[i * P65 + c] = (c == i) ? 2.0 - r2 : -r2;                          // M1 = R2^-1 = 2I - R2
        }
        __syncthreads();
    } else {
        const bool ok = chol_upper64(M0, tid);                                    // M0 = R2
        if (!ok && tid == 0) atomicOr(fail, 1);
        solve_right_upper64(M1, [&](int k, int j) { return M0[k * P65 + j]; }, [&](int j) { return M0[j * P65 + j]; },
                            [&](int a, int b) { return a == b ? 1.0 : 0.0; }, tid);   // M1 = R2^-1
    }
    RT_MARK();
    {
        doublex4 acc[4];
        mm64_mfma(M0, M2, acc, tid);                                              // R2 R1 (scaled by D on the way out, by whoever reads it)
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int e = 0; e < 4; e++) Rprod[(16 * wave + (lane >> 4) + 4 * e) * B + 16 * tj + (lane & 15)] = acc[tj][e];
    }
    __syncthreads();                                                              // R1 is done with: M2 takes the top block of Q
    load64(M2, Q1top, B, tid);                                                    // Q1's top block through LDS
    __syncthreads();
    {
        doublex4 acc[4];
        mm64_mfma(M2, M1, acc, tid);
        __syncthreads();                                                          // every read of Q1's top block is done
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int e = 0; e < 4; e++) M2[(16 * wave + (lane >> 4) + 4 * e) * P65 + 16 * tj + (lane & 15)] = acc[tj][e];   // M2 = top block of Q = Q1 R2^-1
    }
    __syncthreads();
    RT_MARK();
    // LU of E - Q D without pivoting on the top block: ONE wavefront, the matrix in registers (lane j = column j), right-looking with
    // compile-time lane numbers (v_readlane) — no LDS traffic, no barriers inside the 64 steps.  After step k row k holds the
    // eliminated entries W[k][j] = q~_j[k] of Q for j >= k (U[k][j] = -D_j W[k][j], U[k][k] = piv_k = 1 + |W[k][k]|), column k the
    // multipliers L[r][k], r > k (= V's top block).  (Crout order on two wavefronts with three workgroup barriers per step: 222 us
    // for the whole kernel; the first version, rank-one updates by the whole workgroup: 320.)
    if (wave == 0) {
        // lane = ROW r, m[c] = entry (r, c).  Step k: lane k holds row k (W[k][c], c >= k); every lane below multiplies its entry of
        // column k by -D_k / pivot (its multiplier L[r][k], left in m[k]) and subtracts L[r][k] W[k][c] from the rest of its row —
        // W[k][c] by v_readlane from lane k, one FMA per entry, lanes at or above k neutralised by a zero multiplier.
        double m[B];
#pragma unroll
        for (int c = 0; c < B; c++) m[c] = M2[lane * P65 + c];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const double wkk = readlane_d(m[k], k);
            // 1 / pivot by v_rcp_f64 + two Newton steps (the pivot is 1 + |w_kk| in [1, 2]): the IEEE division is ~300 dependent cycles, once per
            // step of a chain of 64 (r4: 7 of this kernel's 71 us)
            const double dk = (wkk >= 0.0) ? -1.0 : 1.0, pv = 1.0 - dk * wkk;
            double rc = __builtin_amdgcn_rcp(pv);
            rc = fma(rc, fma(-pv, rc, 1.0), rc);
            rc = fma(rc, fma(-pv, rc, 1.0), rc);
            const double sc = -dk * rc;
            if (lane == 0) { Dg[k] = dk; piv[k] = pv; prc[k] = rc; }
            const double l = (lane > k) ? m[k] * sc : 0.0;       // L[lane][k]
#pragma unroll
            for (int c = k + 1; c < B; c++) m[c] = fma(-l, readlane_d(m[c], k), m[c]);
            if (lane > k) m[k] = l;
        }
#pragma unroll
        for (int c = 0; c < B; c++) M2[lane * P65 + c] = m[c];
    }
    __syncthreads();
    RT_MARK();
    // T = U Y1^-T: T Y1' = U, Y1' unit upper triangular with Y1'[k][j] = L[j][k]  (wave 0 -> M0);  Xm U = -R2^-1 D  (wave 1, in place on M1)
    if (wave == 0) {
        for (int c = 0; c < B; c++) M0[lane * P65 + c] = (lane == c) ? piv[c] : ((lane < c) ? -Dg[c] * M2[lane * P65 + c] : 0.0);    // U
        wave_sync_lds();
        solve_rows_blocked(M0, [&](int k, int j) { return M2[j * P65 + k]; }, [&](int) { return 1.0; }, lane);
    } else if (wave == 1) {
        for (int c = 0; c < B; c++) M1[lane * P65 + c] = -M1[lane * P65 + c] * Dg[c];
        wave_sync_lds();
        solve_rows_blocked(M1, [&](int k, int j) { return -Dg[j] * M2[k * P65 + j]; }, [&](int j) { return prc[j]; }, lane);
    }
    __syncthreads();
    RT_MARK();
#ifdef PG_RECON_TIME
    if (tid == 0) printf("recon ticks(10ns): load+R2 %lld | 2 products %lld | LU %lld | solves %lld\n", tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3]);
#endif
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int c = j0 + q;
        Tout[i * B + c] = M0[i * P65 + c];
        Xm[i * B + c] = M1[i * P65 + c];
        const double v = (i == c) ? 1.0 : ((i > c) ? M2[i * P65 + c] : 0.0);
        VW[(long long)i * (2 * B) + c] = v;
        Vst[(long long)i * ldv + c] = v;
        if (c >= i) Aband[(long long)i * lda + c] = Dg[i] * Rprod[i * B + c];     // Rs = D (R2 R1): written by this same thread above
    }
}

// ---- last panel: m <= 64 rows, plain Householder QR in LDS ------------------------------------------------------------------
// P: the m x 64 block (ld lda), overwritten by R (upper trapezoidal; below the diagonal zeros).  V -> VW / Vst, T -> Tout.
__global__ __launch_bounds__(256) void small_qr_kernel(int m, double *P, long long lda, double *VW, double *Vst, long long ldv, double *Tout)
{
    extern __shared__ double lds[];
    double *A = lds, *V = lds + MAT, *Gm = lds + 2 * MAT, *Tm = lds + 3 * MAT;
    __shared__ double vcol[B], wrow[B], tau[B], sc[2];
    const int tid = threadIdx.x, i = tid >> 2, j0 = (tid & 3) * 16;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        A[i * P65 + j0 + q] = (i < m) ? P[(long long)i * lda + j0 + q] : 0.0;
        V[i * P65 + j0 + q] = 0.0;
        Tm[i * P65 + j0 + q] = 0.0;
    }
    if (tid < B) tau[tid] = 0.0;
    __syncthreads();
    const int nref = (m - 1 < B) ? m - 1 : B;
    for (int j = 0; j < nref; j++) {
        if (tid < 64) {                                   // wave 0: reflector of column j, rows j .. m-1
            const double x = (tid >= j && tid < m) ? A[tid * P65 + j] : 0.0;
            const double alpha = __shfl(x, j, 64);
            const double xn2 = wave_sum64((tid > j) ? x * x : 0.0);
            double beta = alpha, t = 0.0, scal = 0.0;
            if (xn2 != 0.0) { beta = -copysign(sqrt(alpha * alpha + xn2), alpha); t = (beta - alpha) / beta; scal = 1.0 / (alpha - beta); }
            const double v = (tid == j) ? 1.0 : ((tid > j && tid < m) ? x * scal : 0.0);
            vcol[tid] = v;
            V[tid * P65 + j] = v;
            if (tid == 0) { tau[j] = t; sc[0] = t; sc[1] = beta; }
        }
        __syncthreads();
        if (tid < B) {                                    // w_c = v' A[:, c] for c > j
            double s = 0.0;
            if (tid > j) for (int r = j; r < m; r++) s = fma(vcol[r], A[r * P65 + tid], s);
            wrow[tid] = s;
        }
        __syncthreads();
        const double t = sc[0];
#pragma unroll 4
        for (int c = j0; c < j0 + 16; c++) {
            if (c > j && i >= j) A[i * P65 + c] -= t * vcol[i] * wrow[c];
            else if (c == j && i >= j) A[i * P65 + c] = (i == j) ? sc[1] : 0.0;
        }
        __syncthreads();
    }
    // Gram of V and the dlarft recurrence (forward, columnwise)
    mm64([&](int a, int k) { return V[k * P65 + a]; }, [&](int k, int b) { return V[k * P65 + b]; },
         [&](int a, int b, double v) { Gm[a * P65 + b] = v; }, tid);
    __syncthreads();
    for (int j = 0; j < B; j++) {
        const double tj = tau[j];
        double v = 0.0;
        if (tid < j) {
            for (int l = tid; l < j; l++) v = fma(Tm[tid * P65 + l], Gm[l * P65 + j], v);
            v *= -tj;
        }
        __syncthreads();
        if (tid < j) Tm[tid * P65 + j] = v;
        if (tid == j) Tm[j * P65 + j] = tj;
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int c = j0 + q;
        Tout[i * B + c] = Tm[i * P65 + c];
        if (i < m) {
            P[(long long)i * lda + c] = A[i * P65 + c];
            VW[(long long)i * (2 * B) + c] = V[i * P65 + c];
            Vst[(long long)i * ldv + c] = V[i * P65 + c];
        }
    }
}

// [V W] (m x 128 row-major) -> its transpose (128 x ldt row-major, columns c0 ..): both operands of the rank-128 update k-major, so
// that dgemm's loads walk along m with 16-byte accesses (with k contiguous per row the update ran at 0.30 MFMA-busy)
__global__ __launch_bounds__(256) void transpose_vw_kernel(long long m, const double *VW, double *VWt, long long ldt)
{
    __shared__ double tile[64][65];
    const long long r0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = (r0 + r < m) ? VW[(r0 + r) * (2 * B) + c0 + tx] : 0.0;
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (r0 + tx < m) VWt[(long long)(c0 + c) * ldt + r0 + tx] = tile[tx][c];
}

// rows [r_begin, m) of the panel's V (VW columns 0..63) into the reflector store
__global__ void copy_v_kernel(long long m, long long r_begin, const double *VW, double *Vst, long long ldv)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = r_begin + idx / B, c = idx % B;
    if (r < m) Vst[r * ldv + c] = VW[r * (2 * B) + c];
}

// ---- fused panel kernels (r4): the chain of thin products between the two big GEMMs of a panel ------------------------------------
// Until r4 every thin product of a panel (P R1^-1, Q1'Q1, Q1 Xm, X T, V'Y, T'M1, Y + V M2) was its own dgemm launch of 10 - 17 us —
// a 128-row tile kernel walking K = 64 with one barrier per 8 k — plus split-K reductions, a copy and a transposition: 12 launches
// and ~150 us per panel beside the two big GEMMs.  Here each step works on blocks of 64 rows, one 256-thread workgroup per block, the
// 64 x 64 operands in LDS, products on the fp64 MFMA (wavefront w: rows / Gram columns 16 w ..), and consecutive steps that need no
// grid-wide sum in between are one kernel.  Sums over all row blocks (the Grams) leave one 64 x 64 partial per block; a second kernel
// adds them in block order (deterministic).

// C = At' Bm (64 x 64 x 64, operands in LDS row-major with pitch P65): wavefront w computes rows 16 w .. 16 w + 15 of C
__device__ __forceinline__ void mm64t_mfma(const double *At, const double *Bm, doublex4 (&acc)[4], int tid)
{
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
#pragma unroll 4
    for (int ks = 0; ks < B / 4; ks++) {
        const double a = At[(4 * ks + k4) * P65 + 16 * wave + r16];
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
            acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bm[(4 * ks + k4) * P65 + 16 * tj + r16], acc[tj], 0, 0, 0);
    }
}
// C += A Bm with the accumulators as they come (mm64_mfma zeroes them)
__device__ __forceinline__ void mm64_mfma_acc(const double *A, const double *Bm, doublex4 (&acc)[4], int tid)
{
    const int lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
#pragma unroll 4
    for (int ks = 0; ks < B / 4; ks++) {
        const double a = A[(16 * wave + r16) * P65 + 4 * ks + k4];
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
            acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bm[(4 * ks + k4) * P65 + 16 * tj + r16], acc[tj], 0, 0, 0);
    }
}
// 64 rows x 64 columns of a row-major matrix (row stride ld) -> LDS [64][P65]; rows >= rows_valid are zeros
__device__ __forceinline__ void load64_rows(double *dst, const double *src, long long ld, int rows_valid, int tid)
{
    const int i = tid >> 2, j0 = (tid & 3) * 16;
    if (i < rows_valid) {          // (8-byte loads: the panel's row stride is n, odd for an odd test size)
#pragma unroll
        for (int q = 0; q < 16; q++) dst[i * P65 + j0 + q] = src[(long long)i * ld + j0 + q];
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) dst[i * P65 + j0 + q] = 0.0;
    }
}
// LDS [64 rows r][P65] -> the transposed image: column c of the block becomes 64 consecutive doubles of row (c0 + c) of dstT
__device__ __forceinline__ void store64_transposed(const double *Z, double *dstT, long long ldt, long long r0, int rows_valid, int tid)
{
    const int c = tid >> 2, q0 = (tid & 3) * 16;
#pragma unroll
    for (int q = 0; q < 16; q++)
        if (q0 + q < rows_valid) dstT[(long long)c * ldt + r0 + q0 + q] = Z[(q0 + q) * P65 + c];
}

// Gram of a tall 64-column matrix by blocks of 64 rows: MUL = false: partial(b) = P_b' P_b;  MUL = true: Q_b = P_b Rinv (written to Q),
// partial(b) = Q_b' Q_b.  P: m x 64 with row stride ldp; Q: m x 64, row stride 64.
template <bool MUL>
__global__ __launch_bounds__(256) void panel_gram_kernel(long long m, const double *P, long long ldp, const double *Rinv, double *Q, double *partials)
{
    extern __shared__ double lds[];
    double *Xs = lds, *Rs = lds + MAT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    const long long r0 = (long long)blockIdx.x * B;
    const int rows = (int)((m - r0 < B) ? m - r0 : B);
    load64_rows(Xs, P + r0 * ldp, ldp, rows, tid);
    if (MUL) load64(Rs, Rinv, B, tid);
    __syncthreads();
    doublex4 acc[4];
    if (MUL) {
        mm64_mfma(Xs, Rs, acc, tid);
        __syncthreads();                                   // every read of P_b is done: Xs takes Q_b
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = 16 * wave + k4 + 4 * e, c = 16 * tj + r16;
                Xs[i * P65 + c] = acc[tj][e];              // rows past the end are exact zeros (zero rows of P_b)
                if (i < rows) Q[(r0 + i) * B + c] = acc[tj][e];
            }
        __syncthreads();
    }
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[tj][e] = 0.0;
    mm64t_mfma(Xs, Xs, acc, tid);
    double *out = partials + (size_t)blockIdx.x * B * B;
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) out[(16 * wave + k4 + 4 * e) * B + 16 * tj + r16] = acc[tj][e];
}

// out (64 x 64) = sum over the blocks' partials, in block order
// (128 workgroups: 32 entries x 8 interleaved classes of blocks each, so that a thread has at most ~20 loads, all in flight at once; a
// thread per entry walking all ~156 blocks was a latency chain of 8.8 us per sum, three sums per panel)
__global__ __launch_bounds__(256) void sum_partials_kernel(int nparts, const double *partials, double *out)
{
    __shared__ double red[8][33];
    const int o = threadIdx.x & 31, zs = threadIdx.x >> 5, idx = blockIdx.x * 32 + o;
    double s = 0.0;
#pragma unroll 8
    for (int z = zs; z < nparts; z += 8) s += partials[(size_t)z * B * B + idx];
    red[zs][o] = s;
    __syncthreads();
    if (zs == 0) {
        double t = red[0][o];
#pragma unroll
        for (int q = 1; q < 8; q++) t += red[q][o];
        out[idx] = t;
    }
}

// rows of V below the top block: V_b = Q_b Xm -> [V W] (ld 2b), the reflector store (ld ldv) and the transposed copy [V W]' (rows 0..63);
// block 0 is the top block the reconstruction kernel wrote: it is only transposed
__global__ __launch_bounds__(256) void panel_v_kernel(long long m, const double *Q, const double *Xm, double *VW, double *Vst, long long ldv, double *VWt, long long ldt)
{
    extern __shared__ double lds[];
    double *Qs = lds, *Xs = lds + MAT, *Vs = lds + 2 * MAT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    const long long r0 = (long long)blockIdx.x * B;
    const int rows = (int)((m - r0 < B) ? m - r0 : B);
    if (blockIdx.x == 0) {
        load64_rows(Vs, VW, 2 * B, rows, tid);
        __syncthreads();
        store64_transposed(Vs, VWt, ldt, 0, rows, tid);
        return;
    }
    load64_rows(Qs, Q + r0 * B, B, rows, tid);
    load64(Xs, Xm, B, tid);
    __syncthreads();
    doublex4 acc[4];
    mm64_mfma(Qs, Xs, acc, tid);
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 16 * wave + k4 + 4 * e, c = 16 * tj + r16;
            Vs[i * P65 + c] = acc[tj][e];
            if (i < rows) { VW[(r0 + i) * (2 * B) + c] = acc[tj][e]; Vst[(r0 + i) * ldv + c] = acc[tj][e]; }
        }
    __syncthreads();
    store64_transposed(Vs, VWt, ldt, r0, rows, tid);
}

// X_b = sum of the split-K slabs of X = A22 V (rows in the aligned coordinates of that product: pad rows in front);
// Y_b = X_b T -> W columns of [V W];  partial(b) = V_b' Y_b
__global__ __launch_bounds__(256) void panel_xy_kernel(long long m, long long pad, int slices, const double *slabs, long long slab_stride, const double *T,
                                                       double *VW, double *partials)
{
    extern __shared__ double lds[];
    double *Xs = lds, *Ts = lds + MAT, *Vs = lds + 2 * MAT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    const long long r0 = (long long)blockIdx.x * B;
    const int rows = (int)((m - r0 < B) ? m - r0 : B);
    {
        const int i = tid >> 2, j0 = (tid & 3) * 16;
        double x[16];
#pragma unroll
        for (int q = 0; q < 16; q++) x[q] = 0.0;
        if (i < rows)
            for (int z = 0; z < slices; z++) {
                const double *src = slabs + (size_t)z * slab_stride + (pad + r0 + i) * B + j0;
#pragma unroll
                for (int q = 0; q < 16; q += 2) { const double2 v = *reinterpret_cast<const double2 *>(src + q); x[q] += v.x; x[q + 1] += v.y; }
            }
#pragma unroll
        for (int q = 0; q < 16; q++) Xs[i * P65 + j0 + q] = x[q];
    }
    load64(Ts, T, B, tid);
    load64_rows(Vs, VW + r0 * (2 * B), 2 * B, rows, tid);
    __syncthreads();
    doublex4 acc[4];
    mm64_mfma(Xs, Ts, acc, tid);
    __syncthreads();                                       // every read of X_b is done: Xs takes Y_b
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 16 * wave + k4 + 4 * e, c = 16 * tj + r16;
            Xs[i * P65 + c] = acc[tj][e];
            if (i < rows) VW[(r0 + i) * (2 * B) + B + c] = acc[tj][e];
        }
    __syncthreads();
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[tj][e] = 0.0;
    mm64t_mfma(Vs, Xs, acc, tid);
    double *out = partials + (size_t)blockIdx.x * B * B;
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) out[(16 * wave + k4 + 4 * e) * B + 16 * tj + r16] = acc[tj][e];
}

// M2 = -1/2 T' M1 (every workgroup for itself: 64 MFMAs per wavefront);  W_b = Y_b + V_b M2 -> [V W] and rows 64..127 of [V W]'
__global__ __launch_bounds__(256) void panel_w_kernel(long long m, const double *T, const double *M1, double *VW, double *VWt, long long ldt)
{
    extern __shared__ double lds[];
    double *Ts = lds, *Ms = lds + MAT, *Vs = lds + 2 * MAT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    const long long r0 = (long long)blockIdx.x * B;
    const int rows = (int)((m - r0 < B) ? m - r0 : B);
    load64(Ts, T, B, tid);
    load64(Ms, M1, B, tid);
    load64_rows(Vs, VW + r0 * (2 * B), 2 * B, rows, tid);
    __syncthreads();
    doublex4 acc[4];
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[tj][e] = 0.0;
    mm64t_mfma(Ts, Ms, acc, tid);                          // T' M1
    __syncthreads();                                       // every read of M1 (and of T) is done: Ms takes M2
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) Ms[(16 * wave + k4 + 4 * e) * P65 + 16 * tj + r16] = -0.5 * acc[tj][e];
    // the accumulators start from Y_b (in the MFMA's own layout: 16 lanes = 128 consecutive bytes of a row)
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 16 * wave + k4 + 4 * e;
            acc[tj][e] = (i < rows) ? VW[(r0 + i) * (2 * B) + B + 16 * tj + r16] : 0.0;
        }
    __syncthreads();
    mm64_mfma_acc(Vs, Ms, acc, tid);                       // Y_b + V_b M2
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 16 * wave + k4 + 4 * e, c = 16 * tj + r16;
            Ts[i * P65 + c] = acc[tj][e];                  // T is done with (read before the barrier above)
            if (i < rows) VW[(r0 + i) * (2 * B) + B + c] = acc[tj][e];
        }
    __syncthreads();
    store64_transposed(Ts, VWt + (size_t)B * ldt, ldt, r0, rows, tid);
}

// The two-sided update restricted to the NEXT panel's 64 columns (look-ahead): C (m x 64, row stride ldc: rows = the whole trailing matrix,
// columns = its first 64) -= V W_top' + W V_top', with [V W] the current panel's block vectors (m x 128, row stride 2b) and _top their first
// 64 rows.  One workgroup per 64 rows; 128 MFMAs per wavefront.  The full update then leaves these columns alone.
__global__ __launch_bounds__(256) void panel_next_update_kernel(long long m, const double *VW, double *C, long long ldc)
{
    extern __shared__ double lds[];
    double *Vb = lds, *Wb = lds + MAT, *WtT = lds + 2 * MAT, *VtT = lds + 3 * MAT;      // V_b, W_b (negated) | W_top', V_top' (k-major)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    const long long r0 = (long long)blockIdx.x * B;
    const int rows = (int)((m - r0 < B) ? m - r0 : B);
    {
        const int i = tid >> 2, j0 = (tid & 3) * 16;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const double v = (i < rows) ? VW[(r0 + i) * (2 * B) + j0 + q] : 0.0, w_ = (i < rows) ? VW[(r0 + i) * (2 * B) + B + j0 + q] : 0.0;
            Vb[i * P65 + j0 + q] = -v;
            Wb[i * P65 + j0 + q] = -w_;
            // top block, transposed: WtT[k][c] = W[c][k], VtT[k][c] = V[c][k]   (i = c here; the panel has more than 64 rows)
            WtT[(j0 + q) * P65 + i] = VW[(long long)i * (2 * B) + B + j0 + q];
            VtT[(j0 + q) * P65 + i] = VW[(long long)i * (2 * B) + j0 + q];
        }
    }
    doublex4 acc[4];
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 16 * wave + k4 + 4 * e;
            acc[tj][e] = (i < rows) ? C[(r0 + i) * ldc + 16 * tj + r16] : 0.0;
        }
    __syncthreads();
    mm64_mfma_acc(Vb, WtT, acc, tid);
    mm64_mfma_acc(Wb, VtT, acc, tid);
#pragma unroll
    for (int tj = 0; tj < 4; tj++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int i = 16 * wave + k4 + 4 * e;
            if (i < rows) C[(r0 + i) * ldc + 16 * tj + r16] = acc[tj][e];
        }
}

// =================================================================================================================================
// doubles of the stage-1 panel work space: two transposed copies [V W]' (128 x ldt each) + one 64 x 64 Gram partial per block of 64 rows
static size_t sb2_panel_work(int n)
{
    const size_t ldt = ((size_t)n + 128 + 127) / 128 * 128;
    return 2 * 128 * ldt + ((size_t)n / B + 2) * B * B;
}
size_t sb2_bytes(int n)
{
    const size_t nn = (size_t)n * n;
    const size_t ng = ((size_t)n + SB_G - 1) / SB_G + 1, kmax = ((size_t)n + B - 1) / B + 1;
    return 8 * (2 * nn + (size_t)n * 4 * B + (size_t)n * SB_LD + (size_t)n * (kmax + 1) + 2 * ng * kmax * 128 * SB_G + (kmax + 1) * SB_G * n +
                (ng + 1) * B * B + 2 * (size_t)BT1_BLOCK * BT1_BLOCK + 2 * (size_t)BT1_BLOCK * n + 64 * B * B + (kmax + 1) * SB_MAIL_LD + sb2_panel_work(n)) +
           4 * ((size_t)n + 64) + 4096;
}

thread_local DevArena *g_arena = nullptr;
int dev_alloc(void **p, size_t bytes)
{
    if (bytes == 0) bytes = 8;
    if (g_arena && g_arena->base) {
        const size_t at = (g_arena->off + 255) & ~(size_t)255;
        if (at + bytes <= g_arena->cap) { *p = g_arena->base + at; g_arena->off = at + bytes; return PG_OK; }
    }
    const hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) { set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); *p = nullptr; return PG_ENOMEM; }
    return PG_OK;
}
void dev_free(void *p)
{
    if (!p) return;
    if (g_arena && g_arena->base && (char *)p >= g_arena->base && (char *)p < g_arena->base + g_arena->cap) return;
    (void)hipFree(p);
}
static int alloc_d2(double **p, size_t count) { return dev_alloc(reinterpret_cast<void **>(p), count * sizeof(double)); }

int sb2_alloc(int n, Sb2Work &w)
{
    w = Sb2Work{};
    w.n = n;
    w.npan = 0;
    for (int j = 0; n - j - B >= 2; j += B) w.npan++;
    w.nk = (n + B - 1) / B + 1;
    w.ng = (std::max(n - 2, 1) + SB_G - 1) / SB_G;
    w.kmax = (n + B - 1) / B + 1;
    const size_t nn = (size_t)n * n;
    int rc = PG_OK;
    struct { double **p; size_t cnt; } req[] = {
        {&w.Vst, nn}, {&w.Tst, (size_t)(w.npan + 1) * B * B}, {&w.VW, ((size_t)n + 384) * 2 * B}, {&w.Qb, ((size_t)n + 384) * B}, {&w.sm, (size_t)16 * B * B},
        {&w.S, ((size_t)n + 2) * SB_LD}, {&w.VV, nn}, {&w.TAU, (size_t)n * w.nk}, {&w.Vp, (size_t)w.ng * w.kmax * 128 * SB_G},
        {&w.Vtp, (size_t)w.ng * w.kmax * 128 * SB_G}, {&w.Wws, ((size_t)w.kmax + 1) * SB_G * n}, {&w.G, (size_t)BT1_BLOCK * BT1_BLOCK},
        {&w.T, (size_t)BT1_BLOCK * BT1_BLOCK}, {&w.W, (size_t)BT1_BLOCK * n}, {&w.W2, (size_t)BT1_BLOCK * n}, {&w.mail, ((size_t)w.kmax + 1) * SB_MAIL_LD},
        {&w.Pw, sb2_panel_work(n)}};
    for (auto &r : req) { if (!rc) rc = alloc_d2(r.p, r.cnt); }
    if (!rc) rc = dev_alloc(reinterpret_cast<void **>(&w.prog), ((size_t)n + 16) * sizeof(int));
    if (!rc) rc = dev_alloc(reinterpret_cast<void **>(&w.fail), 4 * sizeof(int));
    if (rc) sb2_free(w);
    return rc;
}

void sb2_free(Sb2Work &w)
{
    for (double *p : {w.Vst, w.Tst, w.VW, w.Qb, w.sm, w.S, w.VV, w.TAU, w.Vp, w.Vtp, w.Wws, w.G, w.T, w.W, w.W2, w.mail, w.Pw}) dev_free(p);
    dev_free(w.prog);
    dev_free(w.fail);
    w = Sb2Work{};
}

// ---- stage 1 ------------------------------------------------------------------------------------------------------------------
int sy2sb_device(pg_ctx *ctx, int n, double *A, Sb2Work &w)
{
    hipStream_t st = ctx->stream;
    constexpr int LDS4 = 4 * MAT * 8, CHOL_LDS = (2 * MAT + B) * 8, CHOL16_LDS = (2 * MAT + B) * 8;
    // the dynamic-LDS limit is a per-DEVICE property of the function: set on every call (a few microseconds, once per solve), never
    // cached per process — a second device in the same process would launch these kernels without it
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&recon_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&small_qr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&chol_inv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CHOL_LDS));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&chol_inv16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, CHOL16_LDS));
    PG_HIP(hipMemsetAsync(w.Vst, 0, (size_t)n * n * 8, st));
    PG_HIP(hipMemsetAsync(w.Tst, 0, (size_t)(w.npan + 1) * B * B * 8, st));
    PG_HIP(hipMemsetAsync(w.fail, 0, 4 * sizeof(int), st));
    // The two big GEMMs of a panel work on 128 x 128 tiles aligned to multiples of 128 in A's own coordinates (only then do the
    // fully updated diagonal tiles of one panel's rank-2b update coincide with the next panel's): every other panel their origin lies
    // 64 rows above the trailing matrix, and [V W] is read with 64 zero rows in front (kept zero for the whole reduction).
    PG_HIP(hipMemsetAsync(w.VW, 0, (size_t)128 * 2 * B * 8, st));
    double *const VW0 = w.VW + (size_t)128 * 2 * B, *const Qb0 = w.Qb + (size_t)128 * B;
    // transposed copy of [V W] for the rank-128 update (the back-transformation's work space is free during stage 1): 128 rows of
    // ldt doubles, the panel's rows from column 128 on, the 128 columns in front kept zero
    const long long ldt = ((long long)n + 128 + 127) / 128 * 128;
    // two copies, used by alternating panels (with look-ahead the next panel's V is written while the update still reads this one's),
    // then the Gram partials of the fused panel kernels: one 64 x 64 block per 64 rows
    double *const VWt_base = w.Pw;
    auto VWt_of = [&](int pan_) { return VWt_base + (size_t)(pan_ & 1) * 128 * ldt; };
    double *const parts = VWt_base + (size_t)2 * 128 * ldt;
    PG_HIP(hipMemsetAsync(VWt_base, 0, (size_t)2 * 128 * ldt * 8, st));
    bool chol16 = true;      // the blocked first-pass Cholesky (PG_SB2_CHOL16=0: the one-wavefront version)
    if (const char *e_ = getenv("PG_SB2_CHOL16")) chol16 = atoi(e_) != 0;
    bool fused = n >= 2 * B + 2;
    if (const char *e_ = getenv("PG_SB2_FUSED")) fused = fused && atoi(e_) != 0;      // A/B timing and tests
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_gram_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MAT * 8));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_gram_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * MAT * 8));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_v_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * MAT * 8));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_xy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * MAT * 8));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * MAT * 8));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&panel_next_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * MAT * 8));
    double *sm = w.sm;
    auto SM = [&](int k) { return sm + (size_t)k * B * B; };
    const long long ld = n;
    // factorisation of the panel at column j (everything that needs only the panel's own columns): on the stream of context c
    auto factor = [&](pg_ctx *c, int j, int pan) -> int {
        hipStream_t cs = c->stream;
        const long long m = n - j - B;
        double *P = A + (size_t)(j + B) * ld + j;                 // m x 64 panel below the diagonal block
        double *Vs = w.Vst + (size_t)(j + B) * ld + j;
        double *Tp = w.Tst + (size_t)pan * B * B;
        int rc = PG_OK;
        if (m > B && fused) {
            // CholeskyQR2 with the fused panel kernels: Gram partials per block of 64 rows + their sum; R1, R1^-1; Q1 = P R1^-1 and the
            // partials of G2 = Q1'Q1 in one kernel; reconstruction; V = Q1 Xm straight into [V W], the reflector store and [V W]'
            const unsigned nb = (unsigned)((m + B - 1) / B);
            panel_gram_kernel<false><<<nb, 256, 2 * MAT * 8, cs>>>(m, P, ld, nullptr, nullptr, parts);
            sum_partials_kernel<<<B * B / 32, 256, 0, cs>>>((int)nb, parts, SM(SM_G1));
            if (chol16) chol_inv16_kernel<<<1, 256, CHOL16_LDS, cs>>>(SM(SM_G1), SM(SM_R1), SM(SM_R1INV), w.fail);
            else chol_inv_kernel<<<1, 64, CHOL_LDS, cs>>>(SM(SM_G1), SM(SM_R1), SM(SM_R1INV), w.fail);
            panel_gram_kernel<true><<<nb, 256, 2 * MAT * 8, cs>>>(m, P, ld, SM(SM_R1INV), Qb0, parts);
            sum_partials_kernel<<<B * B / 32, 256, 0, cs>>>((int)nb, parts, SM(SM_G2));
            recon_kernel<<<1, 256, 3 * MAT * 8, cs>>>(SM(SM_G2), SM(SM_R1), Qb0, P, ld, VW0, Vs, ld, Tp, SM(SM_XM), SM(SM_RPROD), w.fail);
            panel_v_kernel<<<nb, 256, 3 * MAT * 8, cs>>>(m, Qb0, SM(SM_XM), VW0, Vs, ld, VWt_of(pan) + 128, ldt);
        } else if (m > B) {
            // CholeskyQR2: G1 = P'P, R1; Q1 = P R1^-1; G2 = Q1'Q1, R2; (Q = Q1 R2^-1 only through its top block and Xm)
            rc = dgemm(c, true, B, B, m, 1.0, P, ld, P, ld, 0.0, SM(SM_G1), B);
            if (rc) return rc;
            chol_inv_kernel<<<1, 64, CHOL_LDS, cs>>>(SM(SM_G1), SM(SM_R1), SM(SM_R1INV), w.fail);
            rc = dgemm(c, false, m, B, B, 1.0, P, ld, SM(SM_R1INV), B, 0.0, Qb0, B);
            if (!rc) rc = dgemm(c, true, B, B, m, 1.0, Qb0, B, Qb0, B, 0.0, SM(SM_G2), B);
            if (rc) return rc;
            recon_kernel<<<1, 256, 3 * MAT * 8, cs>>>(SM(SM_G2), SM(SM_R1), Qb0, P, ld, VW0, Vs, ld, Tp, SM(SM_XM), SM(SM_RPROD), w.fail);
            // rows 64.. of V = Q1[64:, :] Xm  -> VW[:, 0:64]
            rc = dgemm(c, false, m - B, B, B, 1.0, Qb0 + (size_t)B * B, B, SM(SM_XM), B, 0.0, VW0 + (size_t)B * 2 * B, 2 * B);
            if (rc) return rc;
            copy_v_kernel<<<(unsigned)(((m - B) * B + 255) / 256), 256, 0, cs>>>(m, B, VW0, Vs, ld);
        } else {
            small_qr_kernel<<<1, 256, LDS4, cs>>>((int)m, P, ld, VW0, Vs, ld, Tp);
        }
        PG_HIP(hipGetLastError());
        return PG_OK;
    };
    // Look-ahead (opt-in: PG_SB2_LOOKAHEAD=<CUs to reserve>): the rank-128 update of panel j is issued as its first tile column (which
    // holds the next panel) and the rest; the rest runs on a second stream whose CU mask leaves some CUs out, the next panel's
    // factorisation — a chain of single-workgroup kernels and small GEMMs, ~300 us of mostly idle chip — follows the first tile column
    // on the caller's stream.  [V W] is read by the update from its transposed copy, so the factorisation may overwrite V, Q and the
    // small matrices at once.  (First version: the factorisation on a high-priority stream beside an unmasked update.  The two did
    // run side by side, but the chain's single wavefronts shared SIMDs with the update's MFMA waves and slowed down by what they hid:
    // Cholesky 61 -> 145 us, reconstruction 182 -> 205 us, phase 116 ms against 118.7.)
    pg_ctx side = *ctx;
    side.scratch = nullptr; side.scratch_bytes = 0; side.stream = nullptr;
    hipEvent_t e_col = nullptr, e_fac = nullptr;
    bool ahead = false;
    if (const char *e_ = getenv("PG_SB2_LOOKAHEAD")) {
        const int req = atoi(e_);
        bool ok;
        if (req == -2) {       // -2: a second stream of the LOWEST priority: the update's workgroups yield freed slots to the panel kernels
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            ok = hipStreamCreateWithPriority(&side.stream, hipStreamNonBlocking, least) == hipSuccess;
        } else if (req < 0) ok = hipStreamCreateWithFlags(&side.stream, hipStreamNonBlocking) == hipSuccess;       // -1: a plain second stream
        else {
            const int reserve = std::max(1, std::min(req, ctx->num_cu / 2));
            uint32_t mask[16];
            const int words = (ctx->num_cu + 31) / 32;
            for (int q = 0; q < 16; q++) mask[q] = 0;
            for (int c = reserve; c < ctx->num_cu; c++) mask[c / 32] |= 1u << (c % 32);
            ok = words <= 16 && hipExtStreamCreateWithCUMask(&side.stream, (uint32_t)words, mask) == hipSuccess;
        }
        ahead = req != 0 && ok && hipEventCreateWithFlags(&e_col, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e_fac, hipEventDisableTiming) == hipSuccess;
        if (!ahead) (void)hipGetLastError();
    }
    auto finish = [&](int rc) {
        if (side.stream) { (void)hipStreamSynchronize(side.stream); (void)hipStreamDestroy(side.stream); }
        if (e_col) (void)hipEventDestroy(e_col);
        if (e_fac) (void)hipEventDestroy(e_fac);
        if (side.scratch) (void)hipFree(side.scratch);
        return rc;
    };
    int pan = 0;
    bool factored = false;          // the current panel's factorisation has been issued (by the previous iteration's look-ahead)
    for (int j = 0; n - j - B >= 2; j += B, pan++) {
        const long long m = n - j - B;
        double *Tp = w.Tst + (size_t)pan * B * B;
        int rc = PG_OK;
        if (!factored) rc = factor(ctx, j, pan);
        else if (hipStreamWaitEvent(st, e_fac, 0) != hipSuccess) rc = PG_EHIP;
        if (rc) return finish(rc);
        // two-sided update of A22 = A[j+64:, j+64:] (m x m):  Y = A22 V T,  W = Y - 1/2 V (T' (V' Y)),  A22 -= V W' + W V'
        const long long off = j + B, off_al = off & ~(long long)127, pad = off - off_al;      // pad = 0 or 64
        double *A22al = A + (size_t)off_al * ld + off_al;
        double *V = VW0, *Wc = VW0 + B;                           // columns 0..63 / 64..127 of VW (ld 128)
        double *const VWt = VWt_of(pan);
        DgemmDesc::Partials xp;
        {   // X = A22 V: A22 symmetric with its lower triangle (+ diagonal tiles) valid
            DgemmDesc d;
            d.symA = true; d.M = m + pad; d.N = B; d.K = m + pad; d.alpha = 1.0; d.beta = 0.0;
            d.A = A22al; d.lda = ld; d.B = VW0 - (size_t)pad * 2 * B; d.ldb = 2 * B; d.C = Qb0 - (size_t)pad * B; d.ldc = B;
            if (fused && m > B) d.partials = &xp;            // the slices of a split K are summed by the kernel that consumes X
            rc = dgemm_ex(ctx, d);
        }
        if (fused && m > B) {
            if (rc) return finish(rc);
            const unsigned nb = (unsigned)((m + B - 1) / B);
            const double *slabs = (xp.slices > 1) ? xp.ws : Qb0 - (size_t)pad * B;
            panel_xy_kernel<<<nb, 256, 3 * MAT * 8, st>>>(m, pad, xp.slices, slabs, (m + pad) * B, Tp, VW0, parts);      // Y = X T, partials of V'Y
            sum_partials_kernel<<<B * B / 32, 256, 0, st>>>((int)nb, parts, SM(SM_M1));
            panel_w_kernel<<<nb, 256, 3 * MAT * 8, st>>>(m, Tp, SM(SM_M1), VW0, VWt + 128, ldt);                            // W = Y - 1/2 V T'(V'Y)
            if (hipGetLastError() != hipSuccess) return finish(PG_EHIP);
        } else {
            if (!rc) rc = dgemm(ctx, false, m, B, B, 1.0, Qb0, B, Tp, B, 0.0, Wc, 2 * B);               // Y  = X T           -> W columns
            if (!rc) rc = dgemm(ctx, true, B, B, m, 1.0, V, 2 * B, Wc, 2 * B, 0.0, SM(SM_M1), B);       // M1 = V' Y
            if (!rc) rc = dgemm(ctx, true, B, B, B, -0.5, Tp, B, SM(SM_M1), B, 0.0, SM(SM_M2), B);      // M2 = -1/2 T' M1
            if (!rc) rc = dgemm(ctx, false, m, B, B, 1.0, V, 2 * B, SM(SM_M2), B, 1.0, Wc, 2 * B);      // W  = Y + V M2
            if (rc) return finish(rc);
            transpose_vw_kernel<<<dim3((unsigned)((m + 63) / 64), 2), 256, 0, st>>>(m, VW0, VWt + 128, ldt);
            if (hipGetLastError() != hipSuccess) return finish(PG_EHIP);
        }
        // A22 -= [V W] [W V]' on the lower triangle: first tile column, then (beside the next panel's factorisation) the rest
        const long long mm = m + pad;
        const bool next = n - (j + B) - B >= 2;
        DgemmDesc d;
        d.transA = true; d.kxorB = B; d.K = 2 * B; d.alpha = -1.0; d.beta = 1.0; d.lower_only = true; d.lda = ldt; d.ldb = ldt; d.ldc = ld;
        const bool split = ahead && next && mm > 128 && m > B;
        factored = false;
        if (split && fused) {
            // Look-ahead (r4).  The next panel is the trailing matrix's first 64 columns: they are updated first, by a thin kernel of its
            // own (6 us; the first 128-column tile column as a GEMM launch took 54: ~60 tiles on 256 CUs), the next factorisation follows
            // on this stream; everything else of the update goes to the second stream: the other half of tile column 0 (aligned columns
            // 64..127; only when the trailing matrix starts on a tile boundary — otherwise that tile column is the 64 zero columns in front
            // and the next panel, nothing more) as a 64-wide strip, and the lower triangle of the tiles from (1, 1) on.
            if (hipEventRecord(e_col, st) != hipSuccess) return finish(PG_EHIP);
            if (hipStreamWaitEvent(side.stream, e_col, 0) != hipSuccess) return finish(PG_EHIP);
            if (pad == 0) {
                DgemmDesc s_ = d;
                s_.lower_only = false; s_.M = mm; s_.N = B;
                s_.A = VWt + 128; s_.B = VWt + 128 + B; s_.C = A22al + B;
                rc = dgemm_ex(&side, s_);
                if (rc) return finish(rc);
            }
            d.M = mm - 128; d.N = mm - 128;
            d.A = VWt + 128 - pad + 128; d.B = d.A; d.C = A22al + (size_t)128 * ld + 128;
            rc = dgemm_ex(&side, d);
            if (rc) return finish(rc);
            if (hipEventRecord(e_fac, side.stream) != hipSuccess) return finish(PG_EHIP);
            panel_next_update_kernel<<<(unsigned)((m + B - 1) / B), 256, 4 * MAT * 8, st>>>(m, VW0, A + (size_t)off * ld + off, ld);
            if (hipGetLastError() != hipSuccess) return finish(PG_EHIP);
            rc = factor(ctx, j + B, pan + 1);
            if (rc) return finish(rc);
            factored = true;
            continue;
        }
        if (split) {
            // the part of the update beyond the first tile column: on the masked stream, as soon as [V W]' is there
            if (hipEventRecord(e_col, st) != hipSuccess) return finish(PG_EHIP);
            if (hipStreamWaitEvent(side.stream, e_col, 0) != hipSuccess) return finish(PG_EHIP);
            d.M = mm - 128; d.N = mm - 128;
            d.A = VWt + 128 - pad + 128; d.B = d.A; d.C = A22al + (size_t)128 * ld + 128;
            rc = dgemm_ex(&side, d);
            if (rc) return finish(rc);
            if (hipEventRecord(e_fac, side.stream) != hipSuccess) return finish(PG_EHIP);
        }
        d.M = mm; d.N = split ? 128 : mm;
        d.A = VWt + 128 - pad; d.B = d.A; d.C = A22al;
        rc = dgemm_ex(ctx, d);
        if (rc) return finish(rc);
        if (split) {
            // the next panel's factorisation behind the first tile column, beside the rest
            rc = factor(ctx, j + B, pan + 1);
            if (rc) return finish(rc);
            factored = true;
        }
    }
    if (factored && hipStreamWaitEvent(st, e_fac, 0) != hipSuccess) return finish(PG_EHIP);
    return finish(PG_OK);
}

// ---- band storage ---------------------------------------------------------------------------------------------------------------
// Row i of S holds the 128 columns j in (i - 128, i] of row i, column j in slot j & 127 (a circular window: the slot does not
// move when the window does): the band (i - j <= b) from A, zeros in the room for the bulge.  A whole row is 1 KB, 16-byte aligned
// for every i, so one wavefront moves one row with one 16-byte access per lane.
__global__ void band_extract_kernel(int n, const double *A, double *S)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)(n + 2) * SB_LD) return;
    const int i = (int)(idx / SB_LD), p = (int)(idx % SB_LD);
    const int j = i - ((i - p) & (SB_LD - 1));
    S[idx] = (i < n && j >= 0 && i - j <= B) ? A[(size_t)i * n + j] : 0.0;
}
__global__ void band_de_kernel(int n, const double *S, double *d, double *e)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = S[(size_t)i * SB_LD + (i & (SB_LD - 1))];
    if (i + 1 < n) e[i] = S[(size_t)(i + 1) * SB_LD + (i & (SB_LD - 1))];
}

// ---- stage 2: bulge chasing -------------------------------------------------------------------------------------------------------
// agent-scope relaxed accesses: global_load/store ... sc1 (write-through stores; loads served by L2, never by this CU's L1)
__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BC_SC1 = 16;     // aux bits of the raw buffer accesses: sc1
__device__ __forceinline__ void unpack2(u32x4 v, double &a, double &b) { a = __hiloint2double((int)v.y, (int)v.x); b = __hiloint2double((int)v.w, (int)v.z); }
__device__ __forceinline__ u32x4 pack2(double a, double b)
{
    u32x4 v;
    v.x = (unsigned)__double2loint(a); v.y = (unsigned)__double2hiint(a); v.z = (unsigned)__double2loint(b); v.w = (unsigned)__double2hiint(b);
    return v;
}
constexpr int BC_DONE = INT_MAX / 2;
#ifndef PG_BC_EARLY_SEND
#define PG_BC_EARLY_SEND 1
#endif
#ifndef PG_BC_POLL_SLEEP
#define PG_BC_POLL_SLEEP 1      // s_sleep units (64 clocks) between two polls of a mailbox
#endif
#ifndef PG_BC_RAW_BARRIER
#define PG_BC_RAW_BARRIER 1
#endif
// debugging aid (pgx_sb2_set_debug): a host-mapped int array the bulge-chasing kernel leaves its position in (sweep, step, phase)
static int *g_bc_debug = nullptr;
void sb2_set_debug(int *p) { g_bc_debug = p; }
#ifndef PG_BC_TIME
#define BC_T(ix) do { } while (0)
#else
#define BC_T(ix) do { if (tid == 0) { const long long t_ = wall_clock64(); tacc[ix] += t_ - tlast; tlast = t_; } } while (0)
#endif
#ifndef PG_BC_DEBUG
#define BC_HB(ph) do { } while (0)
#else
#define BC_HB(ph) do { if (dbg && tid == 0) { __hip_atomic_store(&dbg[0], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); \
    __hip_atomic_store(&dbg[1], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __hip_atomic_store(&dbg[2], ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } } while (0)
#endif

// prog[s] = number of completed steps of sweep s (BC_DONE when the sweep has ended); ctl[1] = abort flag.
constexpr int WP = SB_LD + 1;       // LDS pitch of one band row (129 doubles: rows and columns both walk conflict-free)
#ifndef PG_BC_WAVES
#define PG_BC_WAVES 16              // wavefronts per workgroup: several per SIMD hide each other's LDS latency (busy time per step at n = 10 000: 4 waves 8.4 us, 8: 5.2, 16: 4.9)
#endif
constexpr int NW = PG_BC_WAVES, RW = B / NW;     // RW: rows (or columns) of a 64 x 64 block per wavefront
constexpr int BC_LDS_BYTES = (B * WP + 4 * B + 2 * NW * B + 4) * 8;    // band rows | v, v', w, q | partial sums | tau, beta, tau', abort

// sum over the 64 lanes, returned to every lane: DPP inside each row of 16 lanes (no LDS round trips), the four row sums
// combined through scalar registers
__device__ __forceinline__ double dpp_mov(double x, const int ctrl_sel)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    switch (ctrl_sel) {     // dpp_ctrl must be an immediate
        case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
        case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;   // quad_perm [2,3,0,1]
        case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break; // row_half_mirror
        default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break; // row_mirror
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double x)
{
    x += dpp_mov(x, 0);
    x += dpp_mov(x, 1);
    x += dpp_mov(x, 2);
    x += dpp_mov(x, 3);          // every lane: the sum of its row of 16
    return (readlane_d(x, 0) + readlane_d(x, 16)) + (readlane_d(x, 32) + readlane_d(x, 48));
}

__global__ __launch_bounds__(64 * NW) void bc_kernel(int n, double *S, double *VV, double *TAU, int nk, int *prog, int *ctl, int *fail, int *dbg)
{
    // 72 KB of LDS: dynamic, with the launch attribute raised — a STATIC allocation above 64 KB compiles but is not honoured at launch
    // (accesses beyond 64 KB read 0 and drop writes; first GPU run of this kernel).
    // Wn holds the step's 64 band rows exactly as they lie in memory (row r0 + rl, slot = column & 127): loading and storing are plain
    // row copies, one 16-byte access per lane.  With eb = (r0 - 64) & 127 the row reads, from slot eb on and cyclically,
    //     E(rl, c) = Wn[rl][(eb + c) & 127], c < 64        D(rl, c) = Wn[rl][(eb + 64 + c) & 127], c <= rl   (lower triangle only)
    extern __shared__ double bc_lds[];
    double *Wn = bc_lds;
    double *vcur = Wn + B * WP, *vprev = vcur + B, *wv = vprev + B, *qv = wv + B;
    double (*part)[B] = reinterpret_cast<double (*)[B]>(qv + B);
    double (*part2)[B] = part + NW;
    double *sc = reinterpret_cast<double *>(part2 + NW);
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;     // (lane, wq): one of 64 columns/rows x one NW-th of the other index
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(S, 0, (int)((size_t)(n + 2) * SB_LD * 8), 0x00020000);   // wave-uniform
    // Sweeps are dealt round-robin: workgroup w takes sweeps w, w + G, w + 2G, ... in order, so the sweep a workgroup waits for
    // always belongs to its left neighbour, which is resident (the grid never exceeds one workgroup per CU of an otherwise idle
    // stream).  (A dynamic queue — lane 0 fetching the next sweep with an atomic, broadcast through LDS — was the first version:
    // the compiler rotated that `if (tid == 0)` region into the loop latch and sent the other 63 lanes of wave 0 into the next
    // iteration's barriers ahead of lane 0; the kernel re-ran sweep 0 for ever.  A scalar loop counter cannot diverge.)
#ifdef PG_BC_TIME
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#endif
    for (int s = blockIdx.x; s <= n - 3; s += gridDim.x) {
        double taup = 0.0;
        for (int k = 0;; k++) {
            const int r0 = s + 1 + k * B;
            if (r0 >= n) break;
            const int L = (n - r0 < B) ? n - r0 : B;
            const int eb = (r0 - B) & (SB_LD - 1), db = r0 & (SB_LD - 1);
            const int ecol = (eb + lane) & (SB_LD - 1), dcol = (db + lane) & (SB_LD - 1);     // this lane's column of E / of D
#define EIX(rl_, c_) ((rl_) * WP + ((eb + (c_)) & (SB_LD - 1)))
#define DIX(rl_, c_) ((rl_) * WP + ((db + (c_)) & (SB_LD - 1)))
            BC_HB(1);
            if (s > 0) {
                // step (s, k) reads rows r0 .. r0 + 63: the last of them is the first row of step (s - 1, k + 1)
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(&prog[s - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 2) {
                        if (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        if (++spins > (1 << 20)) {            // never expected: the sweep ahead is always running; do not hang the GPU
                            __hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (atomicOr(&fail[1], 1) == 0) { fail[2] = s; fail[3] = k; }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
            }
            BC_HB(2); BC_T(0);
            // ---- load: wave wq copies rows RW wq .. RW wq + RW - 1 (all accesses in flight), rows past the end are zero
            {
                u32x4 rv[RW];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {
                    const int rl = RW * wq + rr;
                    if (rl < L) rv[rr] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(((r0 + rl) * SB_LD + 2 * lane) * 8), 0, BC_SC1);
                    else rv[rr] = u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {
                    const int rl = RW * wq + rr;
                    double v0, v1;
                    unpack2(rv[rr], v0, v1);
                    Wn[rl * WP + 2 * lane] = v0;
                    Wn[rl * WP + 2 * lane + 1] = v1;
                }
            }
            double x0 = 0.0;
            __syncthreads();
            BC_HB(3); BC_T(1);
            // ---- (1) right-apply the previous reflector to E: u = E vprev;  E -= taup u vprev'   (lane = row, wq = RW columns)
            if (k >= 1) {
                double s_ = 0.0;
#pragma unroll
                for (int c = RW * wq; c < RW * wq + RW; c++) s_ = fma(Wn[EIX(lane, c)], vprev[c], s_);
                part[wq][lane] = s_;
                __syncthreads();
                double u = 0.0;
#pragma unroll
                for (int w_ = 0; w_ < NW; w_++) u += part[w_][lane];
                const double tu = taup * u;
#pragma unroll
                for (int c = RW * wq; c < RW * wq + RW; c++) Wn[EIX(lane, c)] = fma(-tu, vprev[c], Wn[EIX(lane, c)]);
                if (wq == 0) x0 = Wn[EIX(lane, 0)];       // first column, updated by this thread itself
            } else if (wq == 0) x0 = Wn[EIX(lane, B - 1)];   // k = 0: column s, the last column of E
            // ---- (2) reflector (wave 0)
            if (wq == 0) {
                const double alpha = readlane_d(x0, 0);
                const double xn2 = wave_sum_dpp((lane >= 1) ? x0 * x0 : 0.0);
                double beta = alpha, tau = 0.0, scal = 0.0;
                if (xn2 != 0.0) { beta = -copysign(sqrt(alpha * alpha + xn2), alpha); tau = (beta - alpha) / beta; scal = 1.0 / (alpha - beta); }
                const double v = (lane == 0) ? 1.0 : x0 * scal;     // rows >= L carry x0 = 0
                vcur[lane] = v;
                if (lane == 0) { sc[0] = tau; sc[1] = beta; TAU[(size_t)s * nk + k] = tau; }
                if (lane < L) VV[(size_t)s * n + r0 + lane] = v;
                Wn[EIX(lane, (k >= 1) ? 0 : B - 1)] = (lane == 0) ? beta : 0.0;
            }
            __syncthreads();
            BC_HB(4); BC_T(5);
            const double tau = sc[0];
            // ---- (3a), (4a) column products with the new reflector: w = E'v, p = D v   (lane = column, wq = RW rows;
            // D symmetric with its lower triangle stored)
            {
                double sw = 0.0, sp = 0.0;
#pragma unroll
                for (int r = RW * wq; r < RW * wq + RW; r++) {
                    const double vr = vcur[r];
                    sp = fma(Wn[(r >= lane) ? r * WP + dcol : DIX(lane, r)], vr, sp);
                    if (k >= 1) sw = fma(Wn[r * WP + ecol], vr, sw);
                }
                part[wq][lane] = sw;
                part2[wq][lane] = sp;
            }
            __syncthreads();
            BC_T(6);
            if (wq == 0) {
                double w_ = 0.0;
#pragma unroll
                for (int q = 0; q < NW; q++) w_ += part[q][lane];
                wv[lane] = w_;
            }
            if (wq == 1) {
                double p = 0.0;
#pragma unroll
                for (int q = 0; q < NW; q++) p += part2[q][lane];
                p *= tau;
                const double pv = wave_sum_dpp(p * vcur[lane]);
                qv[lane] = p - 0.5 * tau * pv * vcur[lane];
            }
            __syncthreads();
            BC_HB(5); BC_T(7);
            // ---- (3b), (4b) rank updates in place (lane = column, wq = RW rows): this lane's w, q, v once, the rows' v, q broadcast
            {
                const double wc = wv[lane], qc = qv[lane], vc = vcur[lane];
#pragma unroll
                for (int r = RW * wq; r < RW * wq + RW; r++) {
                    const double vr = vcur[r], qr = qv[r];
                    if (k >= 1 && lane >= 1) Wn[r * WP + ecol] = fma(-tau * vr, wc, Wn[r * WP + ecol]);
                    if (lane <= r) Wn[r * WP + dcol] -= vr * qc + qr * vc;
                }
            }
            if (tid < B) vprev[tid] = vcur[tid];
            taup = tau;
            __syncthreads();
            BC_T(2);
            // ---- store: rows back as they came (write-through 16-byte stores)
#pragma unroll
            for (int rr = 0; rr < RW; rr++) {
                const int rl = RW * wq + rr;
                if (rl >= L) break;
                __builtin_amdgcn_raw_buffer_store_b128(pack2(Wn[rl * WP + 2 * lane], Wn[rl * WP + 2 * lane + 1]), rsrc,
                                                       (unsigned)(((r0 + rl) * SB_LD + 2 * lane) * 8), 0, BC_SC1);
            }
            // every storing wave drains its stores, the workgroup meets, then one lane publishes the step
            BC_HB(6); BC_T(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            BC_HB(7); BC_T(4);
            if (tid == 0) __hip_atomic_store(&prog[s], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#undef EIX
#undef DIX
        }
        if (tid == 0) __hip_atomic_store(&prog[s], BC_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef PG_BC_TIME
    if (dbg && tid == 0 && blockIdx.x == 1)      // 100 MHz ticks spent by workgroup 1, per phase
        for (int q = 0; q < 8; q++) __hip_atomic_store(&dbg[4 + q], (int)(tacc[q] / 100), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// ---- stage 2, band rows stationary in LDS ------------------------------------------------------------------------------------------
// The kernel above carries every step's 64 rows through memory (load, store, flag: about half of a step's 11 us).  When the
// matrix has no more row blocks than the chip has CUs, the rows can stay where they are worked on: workgroup K keeps the 64 rows
// s + 1 + 64 K .. of sweep s in a ring in LDS (row i in ring slot i & 63) and does step (s, K) of every sweep.  Between two sweeps
// its window slides down by one row: the top row leaves for workgroup K - 1 (workgroup 0: it is final, to memory), the new bottom
// row arrives from workgroup K + 1, and the reflector of step (s, K) goes down to workgroup K + 1, which applies it from the
// right.  These three messages (a mailbox per workgroup in memory, write-through 16-byte stores) are all the memory traffic of a
// step.  One mailbox slot per direction is enough: the dependences of the sweeps themselves order
// each send after the receiver has read the previous one.
// Message format: every double travels as one 16-byte chunk {low word, tag, high word, tag} with tag = sweep + 1, so each aligned
// 8-byte half validates itself (8-byte accesses are single-copy atomic) and the receiver polls the payload directly: no drain,
// no separate flag, no second round trip (the first version — payload, s_waitcnt, sequence flag, then the receiver's two dependent
// loads — spent 2.1 us per hop; the sweep period is two hops plus the work between them).
// (r4, measured and rejected: four polls of a mailbox in flight, a quarter of a round trip apart, to sample the memory more often than once
// per round trip — stage 2 59.4 against 51.9 ms: the extra reads of the line the neighbour is about to write delay that write.)
//   mailbox of workgroup K (MB_LD doubles = 4 KB): chunks [0, 128) the row going up | chunks [128, 192) the reflector going down,
//   chunk 192 its tau
constexpr int MB_LD = SB_MAIL_LD, MB_V = 128, MB_TAU = 192;
__device__ __forceinline__ u32x4 mb_pack(double x, unsigned tag) { return u32x4{(unsigned)__double2loint(x), tag, (unsigned)__double2hiint(x), tag}; }
__device__ __forceinline__ double mb_value(u32x4 c) { return __hiloint2double((int)c.z, (int)c.x); }
// after an expired wait: raise the abort flag for every workgroup and leave a note
__device__ __forceinline__ void bc_give_up(int *ctl, int *fail, int s, int K)
{
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_store(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (atomicOr(&fail[1], 1) == 0) { fail[2] = s; fail[3] = K; }
    }
}
#define BC_POLL_GUARD(spins_, ctl_, bad_)                                                                                         \
    if ((++(spins_) & 63) == 0 &&                                                                                                 \
        ((spins_) > (1 << 20) || __hip_atomic_load(&(ctl_)[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { (bad_) = true; break; }

#ifdef PG_BCS_TIME      // tools/bcs_time.py: s_memtime at the phase boundaries of workgroup 40, sweeps 2000 .. 5999, summed per phase
__device__ long long g_bcs_time[12];
#ifndef PG_BCS_WAVE
#define PG_BCS_WAVE 0
#endif
#define BCS_T(ix) do { if (K == 40 && tid == 64 * PG_BCS_WAVE && s >= 2000 && s < 6000) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); if (!(s == 2000 && (ix) == 0)) g_bcs_time[ix] += t_ - bcs_last; bcs_last = t_; } } while (0)
#else
#define BCS_T(ix) do { } while (0)
#endif
// The barriers of a step order LDS accesses only: the mailbox stores (write-through, ~1 000 cycles to their acknowledgement) and the
// reflector's copy to memory must NOT be waited for — __syncthreads() puts s_waitcnt vmcnt(0) in front of the barrier, and wavefront 0,
// which sends, is the one every barrier of the step waits for.
#if PG_BC_RAW_BARRIER
#define BC_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#else
#define BC_BARRIER() __syncthreads()
#endif
template <int NWT>      // wavefronts per workgroup: 16 (one workgroup per CU) or 8 (two per CU: matrices of up to 2 x 64 x CUs rows)
__global__ __launch_bounds__(64 * NWT) void bc_stationary_kernel(int n, double *S, double *VV, double *TAU, int nk, double *mail, int nwg, int *ctl, int *fail, int test_fault)
{
    constexpr int NW = NWT, RW = B / NWT;
    if (test_fault && blockIdx.x == 1) return;      // tests only (PG_BC_TEST_FAULT=1): a workgroup that never shows up — its neighbours' waits must expire
    extern __shared__ double bc_lds[];
    double *Wn = bc_lds;
    double *vcur = Wn + B * WP, *vprev = vcur + B, *wv = vprev + B, *qv = wv + B;
    double (*part)[B] = reinterpret_cast<double (*)[B]>(qv + B);
    double (*part2)[B] = part + NW;
    double *sc = reinterpret_cast<double *>(part2 + NW);     // [0] tau, [1] beta of this step, [2] tau of the reflector from above
    int *abort_sh = reinterpret_cast<int *>(sc + 3);
    const int tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;
    const int K = blockIdx.x;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(S, 0, (int)((size_t)(n + 2) * SB_LD * 8), 0x00020000);
    const auto rmail = __builtin_amdgcn_make_buffer_rsrc(mail, 0, (int)((size_t)nwg * MB_LD * 8), 0x00020000);
    // the window of sweep 0
#pragma unroll
    for (int rr = 0; rr < RW; rr++) {
        const int i = 1 + K * B + RW * wq + rr;
        double v0 = 0.0, v1 = 0.0;
        if (i < n) unpack2(__builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)((i * SB_LD + 2 * lane) * 8), 0, 0), v0, v1);
        Wn[(i & (B - 1)) * WP + 2 * lane] = v0;
        Wn[(i & (B - 1)) * WP + 2 * lane + 1] = v1;
    }
    if (tid < B) vprev[tid] = 0.0;
    if (tid == 0) { sc[2] = 0.0; *abort_sh = 0; }
    __syncthreads();
    int s = 0;
#ifdef PG_BCS_TIME
    long long bcs_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (; s <= n - 3; s++) {
        const int r0 = s + 1 + K * B;
        if (r0 >= n) break;
        BCS_T(0);      // loop overhead + final barrier of the previous step
        const int L = (n - r0 < B) ? n - r0 : B;
        const int eb = (r0 - B) & (SB_LD - 1), db = r0 & (SB_LD - 1), rb = r0 & (B - 1);
        const int ecol = (eb + lane) & (SB_LD - 1), dcol = (db + lane) & (SB_LD - 1);
#define ROW(rl_) ((((rl_) + rb) & (B - 1)) * WP)
#define EIX(rl_, c_) (ROW(rl_) + ((eb + (c_)) & (SB_LD - 1)))
#define DIX(rl_, c_) (ROW(rl_) + ((db + (c_)) & (SB_LD - 1)))
        // ---- receive: the new bottom row (row r0 + 63 was the top row of step (s - 1, K + 1)) and the reflector of step (s, K - 1)
        if (wq == 1 && s >= 1) {
            double v0 = 0.0, v1 = 0.0;
            if (r0 + B - 1 < n) {
                const unsigned want = (unsigned)s, off = (unsigned)(((K + 1) * MB_LD * 8) + 32 * lane);
                u32x4 c0, c1;
                int spins = 0;
                bool bad = false;
                for (;;) {
                    c0 = __builtin_amdgcn_raw_buffer_load_b128(rmail, off, 0, BC_SC1);
                    c1 = __builtin_amdgcn_raw_buffer_load_b128(rmail, off + 16, 0, BC_SC1);
                    asm volatile("" ::: "memory");
                    if (__all(c0.y == want && c0.w == want && c1.y == want && c1.w == want)) break;
                    BC_POLL_GUARD(spins, ctl, bad)
                    __builtin_amdgcn_s_sleep(PG_BC_POLL_SLEEP);
                }
                if (bad) { bc_give_up(ctl, fail, s, K); *abort_sh = 1; }
                v0 = mb_value(c0); v1 = mb_value(c1);
            }
            Wn[ROW(B - 1) + 2 * lane] = v0;
            Wn[ROW(B - 1) + 2 * lane + 1] = v1;
        }
        if (wq == 0 && K >= 1) {
            const unsigned want = (unsigned)(s + 1), base = (unsigned)((K - 1) * MB_LD * 8);
            u32x4 c0, c1;
            int spins = 0;
            bool bad = false;
            for (;;) {
                c0 = __builtin_amdgcn_raw_buffer_load_b128(rmail, base + 16 * (MB_V + lane), 0, BC_SC1);
                c1 = __builtin_amdgcn_raw_buffer_load_b128(rmail, base + 16 * MB_TAU, 0, BC_SC1);
                asm volatile("" ::: "memory");
                if (__all(c0.y == want && c0.w == want && c1.y == want && c1.w == want)) break;
                BC_POLL_GUARD(spins, ctl, bad)
                __builtin_amdgcn_s_sleep(PG_BC_POLL_SLEEP);
            }
            if (bad) { bc_give_up(ctl, fail, s, K); *abort_sh = 1; }
            vprev[lane] = mb_value(c0);
            if (lane == 0) sc[2] = mb_value(c1);
        }
        BC_BARRIER();
        BCS_T(1);      // waiting for the two messages (+ barrier)
        if (*abort_sh != 0) return;
        const double taup = sc[2];
        double x0 = 0.0;
        // ---- (1) right-apply the reflector from above to E: E <- E - tau' (E v') v'^T.  NW lanes per row: thread -> row wq + NW (lane / NW), the RW
        // columns lane % NW + NW cc; a row's dot product with v' is a DPP sum over its group of NW lanes, so nothing of this phase goes
        // through LDS and there is no barrier inside it — one after it (wavefront 0 reads column 0 for the reflector).
        // History of this phase, which is on the sweeps' critical cycle twice: lane = row, wq = RW columns, every thread adding the NW partials of
        // its row from LDS behind a barrier 2 430 cycles; the partials added once per row (a second barrier) 2 130; whole rows per wavefront
        // with 64-lane reductions (readlane) slower than either (stage 2 58.4 against 55.5 ms).
        if (K >= 1) {
            // rows wq, wq + NW, ..: with the pitch of 129 doubles the lane groups of a 32-lane LDS pass sit 32 banks apart; columns lane % NW + NW cc:
            // the NW lanes of a group read consecutive doubles
            const int rl = wq + NW * (lane / NW), c0 = lane % NW;
            double er[RW], vp[RW], t = 0.0;
#pragma unroll
            for (int cc = 0; cc < RW; cc++) { er[cc] = Wn[EIX(rl, c0 + NW * cc)]; vp[cc] = vprev[c0 + NW * cc]; }
#pragma unroll
            for (int cc = 0; cc < RW; cc++) t = fma(er[cc], vp[cc], t);
            t += dpp_mov(t, 0);
            t += dpp_mov(t, 1);
            if (NW >= 8) t += dpp_mov(t, 2);
            if (NW >= 16) t += dpp_mov(t, 3);
            const double tu = taup * t;
#pragma unroll
            for (int cc = 0; cc < RW; cc++) Wn[EIX(rl, c0 + NW * cc)] = fma(-tu, vp[cc], er[cc]);
            BC_BARRIER();
            if (wq == 0) x0 = Wn[EIX(lane, 0)];
        } else if (wq == 0) x0 = Wn[EIX(lane, B - 1)];
        BCS_T(2);      // right-apply (dot, barrier, update)
        // ---- (2) reflector (wave 0)
        if (wq == 0) {
            const double alpha = readlane_d(x0, 0);
            const double xn2 = wave_sum_dpp((lane >= 1) ? x0 * x0 : 0.0);
            double beta = alpha, tau = 0.0, scal = 0.0;
            if (xn2 != 0.0) {
                const double nn = alpha * alpha + xn2;
                if (nn > 1.0e-280 && nn < 1.0e280) {
                    // the square root and the two divisions of the textbook formulas are ~800 cycles of dependent IEEE expansions, and this
                    // wavefront's path from the messages' arrival to the reflector's departure is on the sweeps' critical cycle twice:
                    // v_rsq / v_rcp + Newton steps instead (to the last bit or two; what matters is tau v'v = 2 to rounding, as before)
                    const double r = rsqrt_newton(nn);
                    double nrm = nn * r;
                    nrm = fma(0.5 * r, fma(-nrm, nrm, nn), nrm);                  // sqrt(nn)
                    beta = -copysign(nrm, alpha);
                    tau = fma(fabs(alpha), r, 1.0);                               // (beta - alpha) / beta = 1 + |alpha| / nrm
                    const double den = fabs(alpha) + nrm;                         // alpha - beta = sign(alpha) (|alpha| + nrm)
                    double ri = __builtin_amdgcn_rcp(den);
                    ri = fma(ri, fma(-den, ri, 1.0), ri);
                    ri = fma(ri, fma(-den, ri, 1.0), ri);
                    scal = copysign(ri, alpha);
                } else { beta = -copysign(sqrt(nn), alpha); tau = (beta - alpha) / beta; scal = 1.0 / (alpha - beta); }
            }
            const double v = (lane == 0) ? 1.0 : x0 * scal;
            // the reflector leaves for workgroup K + 1 straight from this wavefront's registers, BEFORE the barrier (r4; r3: from the last
            // wavefront behind it): the time from the messages' arrival to this store is on the sweeps' critical cycle twice
            if (PG_BC_EARLY_SEND && r0 + B < n) {
                const unsigned tag = (unsigned)(s + 1);
                __builtin_amdgcn_raw_buffer_store_b128(mb_pack(v, tag), rmail, (unsigned)(K * MB_LD * 8 + 16 * (MB_V + lane)), 0, BC_SC1);
                if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(mb_pack(tau, tag), rmail, (unsigned)(K * MB_LD * 8 + 16 * MB_TAU), 0, BC_SC1);
            }
            vcur[lane] = v;
            if (lane == 0) { sc[0] = tau; sc[1] = beta; TAU[(size_t)s * nk + K] = tau; }
            if (lane < L) VV[(size_t)s * n + r0 + lane] = v;
            Wn[EIX(lane, (K >= 1) ? 0 : B - 1)] = (lane == 0) ? beta : 0.0;
        }
        BC_BARRIER();
        BCS_T(3);      // reflector + send + barrier
        const double tau = sc[0];
        if (!PG_BC_EARLY_SEND && wq == NW - 1 && r0 + B < n) {
            const unsigned tag = (unsigned)(s + 1);
            __builtin_amdgcn_raw_buffer_store_b128(mb_pack(vcur[lane], tag), rmail, (unsigned)(K * MB_LD * 8 + 16 * (MB_V + lane)), 0, BC_SC1);
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(mb_pack(tau, tag), rmail, (unsigned)(K * MB_LD * 8 + 16 * MB_TAU), 0, BC_SC1);
        }
        // ---- (3a), (4a) w = E'v, p = D v   (lane = column, wq = RW rows).  (r4: the right-apply's arrangement transposed — NW lanes per column,
        // DPP sums, no partials — brings nothing here: 43.4 against 42.7 ms; the symmetric D is read from its lower triangle in two patterns.)
        {
            double sw = 0.0, sp = 0.0;
#pragma unroll
            for (int r = RW * wq; r < RW * wq + RW; r++) {
                const double vr = vcur[r];
                sp = fma(Wn[(r >= lane) ? ROW(r) + dcol : DIX(lane, r)], vr, sp);
                if (K >= 1) sw = fma(Wn[ROW(r) + ecol], vr, sw);
            }
            part[wq][lane] = sw;
            part2[wq][lane] = sp;
        }
        BC_BARRIER();
        BCS_T(4);      // w = E'v, p = D v partial sums + barrier
        if (wq == 0) {
            double w_ = 0.0;
#pragma unroll
            for (int q = 0; q < NW; q++) w_ += part[q][lane];
            wv[lane] = w_;
        }
        if (wq == 1) {
            double p = 0.0;
#pragma unroll
            for (int q = 0; q < NW; q++) p += part2[q][lane];
            p *= tau;
            const double pv = wave_sum_dpp(p * vcur[lane]);
            qv[lane] = p - 0.5 * tau * pv * vcur[lane];
        }
        BC_BARRIER();
        BCS_T(5);      // the two reductions + barrier
        // ---- (3b), (4b) rank updates in place (lane = column, wq = RW rows)
        {
            const double wc = wv[lane], qc = qv[lane], vc = vcur[lane];
#pragma unroll
            for (int r = RW * wq; r < RW * wq + RW; r++) {
                const double vr = vcur[r], qr = qv[r];
                if (K >= 1 && lane >= 1) Wn[ROW(r) + ecol] = fma(-tau * vr, wc, Wn[ROW(r) + ecol]);
                if (lane <= r) Wn[ROW(r) + dcol] -= vr * qc + qr * vc;
                // the top row is finished as soon as wavefront 0 has updated it (its first row): it leaves for workgroup K - 1 (or for
                // memory) BEFORE the wavefront's other rows — the row's arrival starts the neighbour's next step (r4: it left after them)
                if (r == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wavefront's own LDS writes of row 0
                    const double a_ = Wn[ROW(0) + 2 * lane], b_ = Wn[ROW(0) + 2 * lane + 1];
                    if (K == 0) __builtin_amdgcn_raw_buffer_store_b128(pack2(a_, b_), rsrc, (unsigned)((r0 * SB_LD + 2 * lane) * 8), 0, 0);
                    else {
                        const unsigned tag = (unsigned)(s + 1), off = (unsigned)(K * MB_LD * 8 + 32 * lane);
                        __builtin_amdgcn_raw_buffer_store_b128(mb_pack(a_, tag), rmail, off, 0, BC_SC1);
                        __builtin_amdgcn_raw_buffer_store_b128(mb_pack(b_, tag), rmail, off + 16, 0, BC_SC1);
                    }
                }
            }
        }
        BCS_T(6);      // rank updates + top row out
        BC_BARRIER();
#undef ROW
#undef EIX
#undef DIX
    }
    // the rows still in the window go back to memory
#pragma unroll
    for (int rr = 0; rr < RW; rr++) {
        const int i = s + 1 + K * B + RW * wq + rr;
        if (i < n)
            __builtin_amdgcn_raw_buffer_store_b128(pack2(Wn[(i & (B - 1)) * WP + 2 * lane], Wn[(i & (B - 1)) * WP + 2 * lane + 1]), rsrc,
                                                   (unsigned)((i * SB_LD + 2 * lane) * 8), 0, 0);
    }
}

int sb2st_device(pg_ctx *ctx, int n, const double *A, double *d, double *e, Sb2Work &w, bool allow_stationary, bool *used_stationary)
{
    if (used_stationary) *used_stationary = false;
    hipStream_t st = ctx->stream;
    band_extract_kernel<<<(unsigned)(((size_t)(n + 2) * SB_LD + 255) / 256), 256, 0, st>>>(n, A, w.S);
    PG_HIP(hipMemsetAsync(w.prog, 0, ((size_t)n + 16) * sizeof(int), st));
    PG_HIP(hipMemsetAsync(w.VV, 0, (size_t)n * n * 8, st));
    PG_HIP(hipMemsetAsync(w.TAU, 0, (size_t)n * w.nk * 8, st));
    if (n >= 3) {
        constexpr int BC_LDS = BC_LDS_BYTES, BC_LDS8 = (B * WP + 4 * B + 2 * 8 * B + 4) * 8;
        // per device, on every call (see sy2sb_device)
        PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS));
        PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bc_stationary_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS));
        PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bc_stationary_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS8));
        // one workgroup per block of 64 rows, all of them resident at once (they wait for each other): one per CU with 16 wavefronts,
        // two per CU with 8 (76 KB of LDS each; a step's busy time 5.2 instead of 4.9 us); larger matrices take the kernel that carries
        // the rows through memory.  Should the grid not become resident after all, the bounded waits raise the flag and the caller
        // falls back.
        const int nblk = (n - 1 + B - 1) / B;
        int per_cu = (nblk <= ctx->num_cu) ? 1 : 2;
        if (const char *e_ = getenv("PG_BC_PER_CU")) per_cu = std::max(1, std::min(atoi(e_), 2));      // A/B and tests
        // (with every CU holding two workgroups the sweep period doubles: 512 row blocks 0.47 s against 0.40 s for the memory kernel at 513,
        // while 469 blocks, n = 30 000, take 0.21 s — the second workgroup per CU is used up to 15/8 of the CUs)
        bool stationary = nblk <= std::min(per_cu == 1 ? ctx->num_cu : ctx->num_cu * 15 / 8, w.kmax);
        // every workgroup waits for its neighbours, so the whole grid has to be resident: ask the runtime how many workgroups of this
        // kernel (its registers, its LDS) one CU takes instead of assuming it from the CU count alone
        if (stationary) {
            int occ = 0;
            const hipError_t oe = (per_cu == 1)
                ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bc_stationary_kernel<NW>, 64 * NW, (size_t)BC_LDS)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bc_stationary_kernel<8>, 64 * 8, (size_t)BC_LDS8);
            if (oe != hipSuccess) { (void)hipGetLastError(); occ = 0; }
            stationary = occ >= per_cu && (long long)nblk <= (long long)std::min(occ, per_cu) * ctx->num_cu;
        }
        if (const char *e_ = getenv("PG_BC_STATIONARY")) stationary = stationary && atoi(e_) != 0;
        stationary = stationary && allow_stationary;
        if (used_stationary) *used_stationary = stationary;
        if (stationary) {
            PG_HIP(hipMemsetAsync(w.mail, 0, (size_t)nblk * MB_LD * 8, st));
            const int test_fault = (getenv("PG_BC_TEST_FAULT") && atoi(getenv("PG_BC_TEST_FAULT")) != 0) ? 1 : 0;
            const int exp_waves = getenv("PG_BC_STAT_WAVES") ? atoi(getenv("PG_BC_STAT_WAVES")) : 0;      // A/B: wavefronts per workgroup of the one-per-CU form
            if (per_cu == 1 && exp_waves == 8) bc_stationary_kernel<8><<<nblk, 64 * 8, BC_LDS8, st>>>(n, w.S, w.VV, w.TAU, w.nk, w.mail, nblk, w.prog + n, w.fail, test_fault);
            else if (per_cu == 1 && exp_waves == 4) {
                constexpr int BC_LDS4 = (B * WP + 4 * B + 2 * 4 * B + 4) * 8;
                PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bc_stationary_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, BC_LDS4));
                bc_stationary_kernel<4><<<nblk, 64 * 4, BC_LDS4, st>>>(n, w.S, w.VV, w.TAU, w.nk, w.mail, nblk, w.prog + n, w.fail, test_fault);
            } else
            if (per_cu == 1) bc_stationary_kernel<NW><<<nblk, 64 * NW, BC_LDS, st>>>(n, w.S, w.VV, w.TAU, w.nk, w.mail, nblk, w.prog + n, w.fail, test_fault);
            else bc_stationary_kernel<8><<<nblk, 64 * 8, BC_LDS8, st>>>(n, w.S, w.VV, w.TAU, w.nk, w.mail, nblk, w.prog + n, w.fail, test_fault);
        } else {
            // a sweep trails the one ahead by two blocks: n / 128 sweeps are in flight at most; workgroups beyond that would only poll
            int nwg = n / (2 * B) + 4;
            nwg = std::max(1, std::min(nwg, std::min(ctx->num_cu, 256)));
            if (const char *e_ = getenv("PG_BC_NWG")) nwg = std::max(1, std::min(atoi(e_), 256));     // A/B and debugging
            bc_kernel<<<nwg, 64 * NW, BC_LDS, st>>>(n, w.S, w.VV, w.TAU, w.nk, w.prog, w.prog + n, w.fail, g_bc_debug);
        }
    }
    band_de_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, w.S, d, e);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// ---- back-transformation with the stage-2 reflectors ----------------------------------------------------------------------------
// Block (G, k): sweeps s0 = G g .. s0 + g - 1 at block index k act on rows row0 = s0 + 1 + k b .. row0 + b + g - 2.  Column i of V
// (sweep s0 + i) occupies rows i .. i + b - 1 of the block.  One workgroup builds V (128 x 64, zero padded), its compact-WY factor
// T (forward, columnwise: H_{s0} H_{s0+1} ... = I - V T V') and V T.
__global__ __launch_bounds__(256) void bt2_prep_kernel(int n, int nk, int ng, const double *VV, const double *TAU, double *Vp, double *Vtp)
{
    extern __shared__ double lds[];
    constexpr int VR = 128;
    double *V = lds;                    // [128][65]
    double *Gm = lds + VR * P65;        // [64][65]
    double *Tm = Gm + MAT;              // [64][65]
    __shared__ double tau[SB_G];
    const int G = blockIdx.x, k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    const int s0 = G * SB_G, row0 = s0 + 1 + k * B;
    if (row0 >= n) return;
    const size_t blk = ((size_t)k * ng + G) * VR * SB_G;
    // V[r][i] = VV[s0 + i][row0 + r] for i <= r < i + 64, sweep and row in range
    for (int idx = tid; idx < VR * SB_G; idx += 256) {
        const int r = idx & 127, i = idx >> 7;       // consecutive threads walk along a sweep's row of VV
        const int s = s0 + i;
        double v = 0.0;
        if (s <= n - 3 && r >= i && r < i + B && row0 + r < n) v = VV[(size_t)s * n + row0 + r];
        V[r * P65 + i] = v;
    }
    if (tid < SB_G) {
        const int s = s0 + tid;
        tau[tid] = (s <= n - 3 && s + 1 + k * B < n) ? TAU[(size_t)s * nk + k] : 0.0;
    }
    for (int idx = tid; idx < SB_G * SB_G; idx += 256) Tm[(idx >> 6) * P65 + (idx & 63)] = ((idx >> 6) == (idx & 63)) ? 1.0 : 0.0;
    __syncthreads();
    {   // Gram G = V'V on the MFMA (K = 128 rows): wavefront w its sweeps 16 w .. 16 w + 15
        doublex4 acc[4];
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[tj][e] = 0.0;
#pragma unroll 4
        for (int ks = 0; ks < VR / 4; ks++) {
            const double a = V[(4 * ks + k4) * P65 + 16 * wave + r16];
#pragma unroll
            for (int tj = 0; tj < 4; tj++) acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, V[(4 * ks + k4) * P65 + 16 * tj + r16], acc[tj], 0, 0, 0);
        }
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int e = 0; e < 4; e++) Gm[(16 * wave + k4 + 4 * e) * P65 + 16 * tj + r16] = acc[tj][e];
    }
    __syncthreads();
    // T (forward, columnwise: T[i][j] = -tau_j sum_{l=i}^{j-1} T[i][l] G[l][j], T[j][j] = tau_j) is the inverse of the upper triangular
    // matrix with diagonal 1 / tau_j and G above it: X M = I by the blocked row solve, rcp(j) = tau_j (a reflector with tau = 0 gets
    // its zero row and column).  The 64-step recurrence with two workgroup barriers per step was most of this kernel's 11 ms.
    if (wave == 0) solve_rows_blocked(Tm, [&](int kk, int j) { return Gm[kk * P65 + j]; }, [&](int j) { return tau[j]; }, lane);
    __syncthreads();
    // V and V T out (row-major 128 x 64); V T on the MFMA: wavefront w the rows 32 w .. 32 w + 31
    for (int idx = tid; idx < VR * SB_G; idx += 256) Vp[blk + idx] = V[(idx >> 6) * P65 + (idx & 63)];
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
        doublex4 acc[4];
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[tj][e] = 0.0;
#pragma unroll 4
        for (int ks = 0; ks < SB_G / 4; ks++) {
            const double a = V[(32 * wave + 16 * rt + r16) * P65 + 4 * ks + k4];
#pragma unroll
            for (int tj = 0; tj < 4; tj++) acc[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Tm[(4 * ks + k4) * P65 + 16 * tj + r16], acc[tj], 0, 0, 0);
        }
#pragma unroll
        for (int tj = 0; tj < 4; tj++)
#pragma unroll
            for (int e = 0; e < 4; e++) Vtp[blk + (size_t)(32 * wave + 16 * rt + k4 + 4 * e) * SB_G + 16 * tj + r16] = acc[tj][e];
    }
}

// One reflector block applied to one slab of 64 columns of Z in ONE pass:  Zs <- Zs - (V T) (V' Zs).
// The slab (127 rows x 64 columns) is read from HBM once, into REGISTERS, in the accumulator layout of the fp64 MFMA (a 16 x 16 tile:
// lane l, component e = row (l >> 4) + 4 e, column l & 15) — which is also the layout of its B operand for the four k-steps of a
// 16-row tile, so W = V' Zs needs no LDS copy of the slab, and the same registers are the C operand of the second product.
// W (64 x 64) goes through LDS; V and V T (128 x 64 each, L2-resident: every slab of the block reads the same two) are staged in
// chunks of 16 k-rows like dgemm's operands.  Four waves, one column tile of 16 each (all rows): the zero pattern of the
// parallelogram (V has 64 non-zeros per column of 127) is then the same for every wave and known at compile time — 46 of the 64
// tile products remain.  76 KB of LDS: two workgroups per CU overlap each other's slab traffic.  History at n = 10 000: two
// batched GEMMs per wavefront (40 MB of HBM traffic per block) 166 ms; fused with the slab in LDS (one workgroup per CU, nothing
// overlapped) 142 ms; slab in registers, waves as 2 x 2, all 64 tile products 102 ms; this version: see DESIGN.
constexpr int BT2_NS = 64;                 // columns of Z per workgroup
constexpr int BT2_ZP = BT2_NS + 16;        // LDS pitch of W: consecutive k-rows on disjoint bank halves (as in dgemm.hpp)
constexpr int BT2_AP = 128 + 16;           // pitch of a staged operand chunk [CK][128]
constexpr int BT2_CK = 16;                 // k-rows per staged operand chunk
constexpr int BT2_LDS_BYTES = (64 * BT2_ZP + 2 * BT2_CK * BT2_AP) * 8;
struct Bt2Args {
    int n, nb;
    long long row_first, h_last;
    const double *Vp, *Vtp;       // block of batch element 0
    long long blk_stride;         // elements between consecutive batch elements
    double *Z;
    int vec;                      // rows of Z are 16-byte aligned (n even)
};
__global__ __launch_bounds__(256, 2) void bt2_apply_kernel(Bt2Args ar)
{
    extern __shared__ double lds[];
    double *Ws = lds;                               // [64][ZP]    Ws[sweep][c]
    double *As = Ws + 64 * BT2_ZP;                  // [2][CK][AP] operand chunk, k-major
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int z = blockIdx.y, n = ar.n;
    const long long row0 = ar.row_first + (long long)z * (B + SB_G);
    const int h = (z == ar.nb - 1) ? (int)ar.h_last : B + SB_G - 1;
    const int c0 = blockIdx.x * BT2_NS, ncols = (n - c0 < BT2_NS) ? n - c0 : BT2_NS;
    const double *V = ar.Vp + (long long)z * ar.blk_stride, *Vt = ar.Vtp + (long long)z * ar.blk_stride;
    double *Zg = ar.Z + (size_t)row0 * n + c0;
    constexpr int CK = BT2_CK, NA = 128 / CK, NB2 = SB_G / CK;
    // ---- slab in.  Wavefront w owns column tile w (columns 16 w .. 16 w + 15) of the slab, all 128 rows, from here to the end:
    // B operand of the first product, C operand and result of the second.  Full, 16-byte-aligned slabs come in whole rows (512
    // contiguous bytes, 16 per lane) through the operand staging space, 32 rows at a time, and are picked up from LDS in the
    // accumulator layout; lane-wise 8-byte loads in that layout were the slow part of the kernel (dgemm's epilogue had the same
    // disease).  Edge slabs (last columns, odd n) keep the element-wise path.
    doublex4 zr[8];
    const int rsub = lane >> 4, csub = lane & 15;
    const int colw = wave * 16 + csub;              // this lane's column of the slab
    const bool vec = ar.vec && ncols == BT2_NS;
    constexpr int SP2 = BT2_NS + 8;                 // pitch of a staged slab row
    if (vec) {
        double2 cv[4][4];           // all 16 loads of the thread in flight at once: the memory latency is paid once per slab, not per chunk
#pragma unroll
        for (int ch = 0; ch < 4; ch++)
#pragma unroll
            for (int ps = 0; ps < 4; ps++) {
                const int row = 32 * ch + 8 * ps + (tid >> 5);
                cv[ch][ps] = (row < h) ? *reinterpret_cast<const double2 *>(Zg + (size_t)row * n + 2 * (tid & 31)) : make_double2(0.0, 0.0);
            }
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
#pragma unroll
            for (int ps = 0; ps < 4; ps++) *reinterpret_cast<double2 *>(As + (8 * ps + (tid >> 5)) * SP2 + 2 * (tid & 31)) = cv[ch][ps];
            __syncthreads();
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int e = 0; e < 4; e++) zr[2 * ch + tt][e] = As[(16 * tt + rsub + 4 * e) * SP2 + colw];
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = t * 16 + rsub + 4 * e;
                zr[t][e] = (row < h && colw < ncols) ? Zg[(size_t)row * n + colw] : 0.0;
            }
    }
    // ---- phase A: W (64 sweeps x 64 columns) = V' Zs, K = 128 rows; operand chunk = CK rows of V (contiguous doubles).
    // Sweep tile I of V' is non-zero on the rows 16 I .. 16 I + 78 only (column i of V: rows i .. i + 63): row tiles I .. I + 4 of
    // the eight — 20 tile products of 32; the others would add exact zeros.
    doublex4 accA[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int e = 0; e < 4; e++) accA[i][e] = 0.0;
    double2 stg[CK / 8];
    auto gloadA = [&](int kt) {      // chunk kt: rows CK kt ..: CK * 64 contiguous doubles; thread -> double2 number tid (+ 256 per pass)
#pragma unroll
        for (int ps = 0; ps < CK / 8; ps++) stg[ps] = *reinterpret_cast<const double2 *>(V + (size_t)kt * CK * SB_G + 2 * (ps * 256 + tid));
    };
    auto lstoreA = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < CK / 8; ps++) {
            const int e0 = 2 * (ps * 256 + tid);           // element of the chunk: row e0 / 64, sweep e0 % 64
            double *d = As + (buf * CK + (e0 >> 6)) * BT2_AP + (e0 & 63);
            d[0] = stg[ps].x; d[1] = stg[ps].y;
        }
    };
    static_assert(BT2_CK == 16, "the zero pattern below is per row tile of 16");
    gloadA(0);
    lstoreA(0);
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < NA; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < NA) gloadA(kt + 1);
#pragma unroll
        for (int kk = 0; kk < CK; kk += 4) {
            const int kr = kk + rsub;
#pragma unroll
            for (int I = 0; I < 4; I++) {
                if (kt < I || kt > I + 4) continue;
                accA[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(As[(buf * CK + kr) * BT2_AP + I * 16 + csub], zr[kt][kk / 4], accA[I], 0, 0, 0);
            }
        }
        if (kt + 1 < NA) lstoreA(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int I = 0; I < 4; I++)
#pragma unroll
        for (int e = 0; e < 4; e++) Ws[(I * 16 + rsub + 4 * e) * BT2_ZP + colw] = accA[I][e];
    // ---- phase B: Zs (128 x 64) -= (V T) W, K = 64 sweeps, accumulated onto the slab registers themselves (V T enters negated);
    // operand chunk = CK sweeps of all 128 rows of V T (row-major [row][sweep]).  Row tile R of V T is non-zero from sweep 16 R - 63
    // on, i.e. on the sweep tiles max(0, R - 4) .. 3: 26 tile products of 32.
    double rt[CK / 8][4];
    auto gloadB = [&](int kt) {      // thread -> row tid / 2, 4 consecutive sweeps (+ 8 per pass)
#pragma unroll
        for (int ps = 0; ps < CK / 8; ps++) {
            const double *src = Vt + (size_t)(tid >> 1) * SB_G + kt * CK + ps * 8 + (tid & 1) * 4;
            const double2 a = *reinterpret_cast<const double2 *>(src), b2 = *reinterpret_cast<const double2 *>(src + 2);
            rt[ps][0] = a.x; rt[ps][1] = a.y; rt[ps][2] = b2.x; rt[ps][3] = b2.y;
        }
    };
    auto lstoreB = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < CK / 8; ps++)
#pragma unroll
            for (int q = 0; q < 4; q++) As[(buf * CK + ps * 8 + (tid & 1) * 4 + q) * BT2_AP + (tid >> 1)] = rt[ps][q];
    };
    gloadB(0);
    lstoreB(0);
    __syncthreads();          // also: W complete in LDS
#pragma unroll
    for (int kt = 0; kt < NB2; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < NB2) gloadB(kt + 1);
#pragma unroll
        for (int kk = 0; kk < CK; kk += 4) {
            const int kr = kk + rsub;
            const double b2 = Ws[(kt * CK + kr) * BT2_ZP + colw];
#pragma unroll
            for (int R = 0; R < 8; R++) {
                if (kt < R - 4) continue;
                zr[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(-As[(buf * CK + kr) * BT2_AP + R * 16 + csub], b2, zr[R], 0, 0, 0);
            }
        }
        if (kt + 1 < NB2) lstoreB(buf ^ 1);
        __syncthreads();
    }
    // ---- slab out
    if (vec) {
        // chunk ch = rows 32 ch ..: into the staging space, then everybody stores whole rows
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int e = 0; e < 4; e++) As[(16 * tt + rsub + 4 * e) * SP2 + colw] = zr[2 * ch + tt][e];
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 4; ps++) {
                const int row = 32 * ch + 8 * ps + (tid >> 5);
                if (row < h)
                    *reinterpret_cast<double2 *>(Zg + (size_t)row * n + 2 * (tid & 31)) = *reinterpret_cast<const double2 *>(As + (8 * ps + (tid >> 5)) * SP2 + 2 * (tid & 31));
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int R = 0; R < 8; R++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = R * 16 + rsub + 4 * e;
                if (row < h && colw < ncols) Zg[(size_t)row * n + colw] = zr[R][e];
            }
    }
}

// ---- four reflector blocks per pass ------------------------------------------------------------------------------------------------
// bt2_apply_kernel moves a slab of Z through HBM once per block: 157 x (read + write) of Z at n = 10 000, 250 GB, and that — not
// the MFMA work — sets its time.  Blocks (G, k), (G, k + 1), (G - 1, k), (G - 1, k + 1) lie on 255 consecutive rows (each block 127 rows,
// starting 64 apart) and may be applied in exactly that order, so a workgroup that keeps 256 rows x 64 columns in registers applies all
// four on ONE trip of the slab: half the traffic, a quarter of the launches' slab latencies.  The 2 x 2 super-blocks (u, ks) — groups
// glast - 2u and glast - 2u - 1, block indices 2 ks and 2 ks + 1 — depend on each other exactly like the blocks do, so they run on
// anti-diagonals u + ks = ts, their slabs 256 rows apart.  Same per-wave layout as above with 16 row tiles instead of 8.
struct Bt2SArgs {
    int n, ng, ts, u_first;
    const double *Vp, *Vtp;       // block (G, k) at ((k ng + G) 128 x 64)
    double *Z;
    int vec;
};
#ifdef PG_BT2_TIME       // tools/bt2_time.py: s_memtime of workgroup (slab 10, first super-block) at the phase boundaries; what is read back is the LAST launch's
                         // (one block per trip, the chip nearly empty: a workgroup alone on its CU)
__device__ long long g_bt2_time[16];
#define BT2_T(ix) do { if (blockIdx.x == 10 && blockIdx.y == 0 && threadIdx.x == 0) g_bt2_time[ix] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define BT2_T(ix) do { } while (0)
#endif
__global__ __launch_bounds__(256, 2) void bt2_apply4_kernel(Bt2SArgs ar)
{
    extern __shared__ double lds[];
    double *Ws = lds;                               // [64][ZP]
    double *As = Ws + 64 * BT2_ZP;                  // [2][CK][AP]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = ar.n, u = ar.u_first + blockIdx.y, ks = ar.ts - u;
    const int Ghi = ar.ng - 1 - 2 * u, Glo = Ghi - 1, k0 = 2 * ks;
    const long long a = (long long)Ghi * SB_G + 1 + (long long)k0 * B, slab0 = a - B;      // row of slab tile 0
    auto exists = [&](int G, int k) { return G >= 0 && (long long)G * SB_G + 1 + (long long)k * B < n; };
    const bool e0 = exists(Ghi, k0), e1 = exists(Ghi, k0 + 1), e2 = exists(Glo, k0), e3 = exists(Glo, k0 + 1);
    if (!(e0 || e1 || e2 || e3)) return;
    // rows of the slab any of the four touches (slab rows; tiles 0..7 | 4..11 | 4..11 | 8..15)
    const int rlo = e2 ? 0 : ((e0 || e3) ? 64 : 128), rhi = e1 ? 256 : ((e0 || e3) ? 192 : 128);
    const int c0 = blockIdx.x * BT2_NS, ncols = (n - c0 < BT2_NS) ? n - c0 : BT2_NS;
    double *Zg = ar.Z + c0;                         // + grow * n
    constexpr int CK = BT2_CK, NA = 128 / CK, NB2 = SB_G / CK;
    constexpr size_t BLK = (size_t)128 * SB_G;
    doublex4 zr[16];
    const int rsub = lane >> 4, csub = lane & 15;
    const int colw = wave * 16 + csub;
    const bool vec = ar.vec && ncols == BT2_NS;
    constexpr int SP2 = BT2_NS + 8;
    auto row_ok = [&](int r) { const long long g = slab0 + r; return r >= rlo && r < rhi && g >= 0 && g < n; };
    BT2_T(0);
    if (vec) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            double2 cv[4][4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++)
#pragma unroll
                for (int ps = 0; ps < 4; ps++) {
                    const int row = 128 * half + 32 * ch + 8 * ps + (tid >> 5);
                    cv[ch][ps] = row_ok(row) ? *reinterpret_cast<const double2 *>(Zg + (size_t)(slab0 + row) * n + 2 * (tid & 31)) : make_double2(0.0, 0.0);
                }
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
#pragma unroll
                for (int ps = 0; ps < 4; ps++) *reinterpret_cast<double2 *>(As + (8 * ps + (tid >> 5)) * SP2 + 2 * (tid & 31)) = cv[ch][ps];
                __syncthreads();
#pragma unroll
                for (int tt = 0; tt < 2; tt++)
#pragma unroll
                    for (int e = 0; e < 4; e++) zr[8 * half + 2 * ch + tt][e] = As[(16 * tt + rsub + 4 * e) * SP2 + colw];
                __syncthreads();
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; t++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = t * 16 + rsub + 4 * e;
                zr[t][e] = (row_ok(row) && colw < ncols) ? Zg[(size_t)(slab0 + row) * n + colw] : 0.0;
            }
    }
    // one block on the row tiles T0 .. T0 + 7 of the slab registers (the body of bt2_apply_kernel).  Operand chunks are fetched TWO
    // steps ahead (a step is 8 - 16 MFMAs, 0.2 - 0.4 us: less than an L2 round trip), the first two chunks of the second product
    // during the last two steps of the first.
    // The first two chunks of the NEXT block are asked for during the last two steps of the current one (the staging registers of the first
    // product are idle by then): a block no longer starts with an exposed round trip to L2.
    double2 stg[2][CK / 8];
    bool prefetched = false;
    auto apply = [&](auto T0C, const double *V, const double *Vt, const double *Vnext) {
        constexpr int T0 = decltype(T0C)::value;
        doublex4 accA[4];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) accA[i][e] = 0.0;
        double rt[2][CK / 8][4];
        auto gloadAp = [&](const double *Vp_, int kt, int set) {
#pragma unroll
            for (int ps = 0; ps < CK / 8; ps++) stg[set][ps] = *reinterpret_cast<const double2 *>(Vp_ + (size_t)kt * CK * SB_G + 2 * (ps * 256 + tid));
        };
        auto gloadA = [&](int kt, int set) { gloadAp(V, kt, set); };
        auto lstoreA = [&](int buf, int set) {
#pragma unroll
            for (int ps = 0; ps < CK / 8; ps++) {
                const int e0_ = 2 * (ps * 256 + tid);
                double *d = As + (buf * CK + (e0_ >> 6)) * BT2_AP + (e0_ & 63);
                d[0] = stg[set][ps].x; d[1] = stg[set][ps].y;
            }
        };
        auto gloadB = [&](int kt, int set) {
#pragma unroll
            for (int ps = 0; ps < CK / 8; ps++) {
                const double *src = Vt + (size_t)(tid >> 1) * SB_G + kt * CK + ps * 8 + (tid & 1) * 4;
                const double2 x = *reinterpret_cast<const double2 *>(src), y = *reinterpret_cast<const double2 *>(src + 2);
                rt[set][ps][0] = x.x; rt[set][ps][1] = x.y; rt[set][ps][2] = y.x; rt[set][ps][3] = y.y;
            }
        };
        auto lstoreB = [&](int buf, int set) {
#pragma unroll
            for (int ps = 0; ps < CK / 8; ps++)
#pragma unroll
                for (int q = 0; q < 4; q++) As[(buf * CK + ps * 8 + (tid & 1) * 4 + q) * BT2_AP + (tid >> 1)] = rt[set][ps][q];
        };
        if (!prefetched) { gloadA(0, 0); gloadA(1, 1); }
        lstoreA(0, 0);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < NA; kt++) {
            const int buf = kt & 1;
            if (kt + 2 < NA) gloadA(kt + 2, kt & 1);            // set kt & 1 went to LDS one step ago
            else gloadB(kt + 2 - NA, (kt + 2 - NA) & 1);
#pragma unroll
            for (int kk = 0; kk < CK; kk += 4) {
                const int kr = kk + rsub;
#pragma unroll
                for (int I = 0; I < 4; I++) {
                    if (kt < I || kt > I + 4) continue;
                    accA[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(As[(buf * CK + kr) * BT2_AP + I * 16 + csub], zr[T0 + kt][kk / 4], accA[I], 0, 0, 0);
                }
            }
            if (kt + 1 < NA) lstoreA(buf ^ 1, (kt + 1) & 1);
            else lstoreB(0, 0);        // the first chunk of the second product takes the buffer the step before last released: no barrier of its own
            __syncthreads();
        }
        // W = V'Z stays in the accumulators: the C layout of v_mfma_f64_16x16x4 (lane (r, c): rows r + 4 e of column c) IS the B layout of the next
        // product's k-step 4 e (row 4 e + r of column c), so accA[kt][kk / 4] is the operand — no trip through LDS, no barrier (r3-r4: W was
        // written to LDS, 41 KB, and read back once per k-step)
        static_assert(BT2_CK == 16, "accA[kt] = the 16 reflectors of chunk kt");
        if (T0 == 4 && V == ar.Vp + ((size_t)k0 * (size_t)ar.ng + Ghi) * BLK) BT2_T(8);      // first block: end of the first product
#pragma unroll
        for (int kt = 0; kt < NB2; kt++) {
            const int buf = kt & 1;
            if (kt + 2 < NB2) gloadB(kt + 2, kt & 1);
            else if (Vnext) gloadAp(Vnext, kt + 2 - NB2, kt + 2 - NB2);
#pragma unroll
            for (int kk = 0; kk < CK; kk += 4) {
                const int kr = kk + rsub;
                const double b2 = accA[kt][kk / 4];
#pragma unroll
                for (int R = 0; R < 8; R++) {
                    if (kt < R - 4) continue;
                    zr[T0 + R] = __builtin_amdgcn_mfma_f64_16x16x4f64(-As[(buf * CK + kr) * BT2_AP + R * 16 + csub], b2, zr[T0 + R], 0, 0, 0);
                }
            }
            if (kt + 1 < NB2) lstoreB(buf ^ 1, (kt + 1) & 1);
            __syncthreads();
        }
        prefetched = Vnext != nullptr;
    };
    const size_t ngs = (size_t)ar.ng;
    BT2_T(1);
    {
        const double *Vb[4] = {ar.Vp + ((size_t)k0 * ngs + Ghi) * BLK, ar.Vp + ((size_t)(k0 + 1) * ngs + Ghi) * BLK, ar.Vp + ((size_t)k0 * ngs + Glo) * BLK,
                               ar.Vp + ((size_t)(k0 + 1) * ngs + Glo) * BLK};
        const double *Tb[4] = {ar.Vtp + ((size_t)k0 * ngs + Ghi) * BLK, ar.Vtp + ((size_t)(k0 + 1) * ngs + Ghi) * BLK, ar.Vtp + ((size_t)k0 * ngs + Glo) * BLK,
                               ar.Vtp + ((size_t)(k0 + 1) * ngs + Glo) * BLK};
        const bool ex[4] = {e0, e1, e2, e3};
        auto next_of = [&](int q) -> const double * { for (int r = q + 1; r < 4; r++) if (ex[r]) return Vb[r]; return nullptr; };
        if (e0) apply(std::integral_constant<int, 4>{}, Vb[0], Tb[0], next_of(0));
        BT2_T(2);
        if (e1) apply(std::integral_constant<int, 8>{}, Vb[1], Tb[1], next_of(1));
        BT2_T(3);
        if (e2) apply(std::integral_constant<int, 0>{}, Vb[2], Tb[2], next_of(2));
        BT2_T(4);
        if (e3) apply(std::integral_constant<int, 4>{}, Vb[3], Tb[3], nullptr);
    }
    BT2_T(6);
    // ---- slab out
    if (vec) {
#pragma unroll
        for (int ch = 0; ch < 8; ch++) {
#pragma unroll
            for (int tt = 0; tt < 2; tt++)
#pragma unroll
                for (int e = 0; e < 4; e++) As[(16 * tt + rsub + 4 * e) * SP2 + colw] = zr[2 * ch + tt][e];
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 4; ps++) {
                const int row = 32 * ch + 8 * ps + (tid >> 5);
                if (row_ok(row))
                    *reinterpret_cast<double2 *>(Zg + (size_t)(slab0 + row) * n + 2 * (tid & 31)) = *reinterpret_cast<const double2 *>(As + (8 * ps + (tid >> 5)) * SP2 + 2 * (tid & 31));
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int R = 0; R < 16; R++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = R * 16 + rsub + 4 * e;
                if (row_ok(row) && colw < ncols) Zg[(size_t)(slab0 + row) * n + colw] = zr[R][e];
            }
    }
    BT2_T(7);
}

// the blocks (V, V T) of every (group, block index): needs only the reflectors, so the caller may run it on a second stream beside the
// divide & conquer (st = nullptr: the context's stream)
int bt2_prep_device(pg_ctx *ctx, int n, Sb2Work &w, hipStream_t st)
{
    if (n < 3) return PG_OK;
    if (!st) st = ctx->stream;
    constexpr int VR = 128;
    const size_t lds = (size_t)(VR * P65 + 2 * MAT) * 8;
    // per device, on every call (see sy2sb_device)
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bt2_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bt2_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BT2_LDS_BYTES));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bt2_apply4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BT2_LDS_BYTES));
    bt2_prep_kernel<<<dim3(w.ng, w.kmax), 256, lds, st>>>(n, w.nk, w.ng, w.VV, w.TAU, w.Vp, w.Vtp);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

int bt2_device(pg_ctx *ctx, int n, double *Z, Sb2Work &w, bool prepared)
{
    if (n < 3) return PG_OK;
    hipStream_t st = ctx->stream;
    constexpr int VR = 128, HGT = B + SB_G - 1;      // rows of a block
    const int ng = w.ng;
    if (!prepared) { const int rc = bt2_prep_device(ctx, n, w, st); if (rc) return rc; }
    const size_t blk = (size_t)VR * SB_G;
    const int glast = ng - 1;
    const int nslab = (n + BT2_NS - 1) / BT2_NS;
    bool four = true;       // four blocks per trip of a slab (bt2_apply4_kernel); PG_BT2_FOUR=0: one block per trip
    if (const char *e_ = getenv("PG_BT2_FOUR")) four = atoi(e_) != 0;
    if (four) {
        const int nu = (ng + 1) / 2, kcount = (n - 1 + B - 1) / B, nks = (kcount + 1) / 2;
        const int lds_pad = getenv("PG_BT2_LDS_PAD") ? atoi(getenv("PG_BT2_LDS_PAD")) : 0;      // A/B: > 6 KB leaves room for ONE workgroup per CU only
        if (lds_pad > 0) PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bt2_apply4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, BT2_LDS_BYTES + lds_pad));
        for (int ts = 0; ts <= (nu - 1) + (nks - 1); ts++) {
            const int u_lo = std::max(0, ts - (nks - 1)), u_hi = std::min(ts, nu - 1);
            Bt2SArgs ar;
            ar.n = n; ar.ng = ng; ar.ts = ts; ar.u_first = u_lo; ar.Vp = w.Vp; ar.Vtp = w.Vtp; ar.Z = Z;
            ar.vec = ((n & 1) == 0 && (uintptr_t)Z % 16 == 0) ? 1 : 0;
            if (const char *e_ = getenv("PG_BT2_VEC")) ar.vec = ar.vec && atoi(e_) != 0;      // A/B: 0 = the slab straight between memory and the MFMA layout
            bt2_apply4_kernel<<<dim3(nslab, u_hi - u_lo + 1), 256, BT2_LDS_BYTES + lds_pad, st>>>(ar);
        }
        PG_HIP(hipGetLastError());
        return PG_OK;
    }
    // wavefront t: blocks (G, k) with (glast - G) + k = t; their rows start (g + b) apart
    for (int t = 0;; t++) {
        const int kmin = std::max(0, t - glast);
        const long long base = (long long)SB_G * (glast - t) + 1;          // row0(k) = base + k (g + b)
        // largest k with G = glast - t + k <= glast (k <= t) and row0 < n
        long long kmx = t;
        if (base + kmx * (SB_G + B) >= n) kmx = (n - 1 - base) / (SB_G + B);   // floor; base may be negative only when kmin > 0
        if (n - 1 - base < 0) kmx = -1;
        if (kmx < kmin) break;     // wavefronts are non-empty up to the last one (t <= glast: k = 0 exists; beyond: once empty, always empty)
        const int nb = (int)(kmx - kmin + 1);
        const long long row_first = base + (long long)kmin * (SB_G + B), row_last = base + kmx * (SB_G + B);
        const int G0 = glast - t + kmin;
        Bt2Args ar;
        ar.n = n; ar.nb = nb; ar.row_first = row_first; ar.h_last = std::min<long long>(HGT, n - row_last);
        ar.Vp = w.Vp + ((size_t)kmin * ng + G0) * blk; ar.Vtp = w.Vtp + ((size_t)kmin * ng + G0) * blk;
        ar.blk_stride = (long long)(ng + 1) * blk; ar.Z = Z; ar.vec = ((n & 1) == 0 && (uintptr_t)Z % 16 == 0) ? 1 : 0;
        bt2_apply_kernel<<<dim3(nslab, nb), 256, BT2_LDS_BYTES, st>>>(ar);
    }
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// ---- back-transformation with the stage-1 reflectors ----------------------------------------------------------------------------
__global__ void bt1_diag_t_kernel(int m, int pan0, const double *Tst, double *T)
{
    // diagonal 64 x 64 blocks of the aggregated T (ld BT1_BLOCK) = the panels' own factors
    const int sub = blockIdx.x, tid = threadIdx.x;
    if (sub * B >= m) return;
    const double *src = Tst + (size_t)(pan0 + sub) * B * B;
    for (int idx = tid; idx < B * B; idx += blockDim.x) {
        const int i = idx / B, j = idx % B;
        T[(size_t)(sub * B + i) * BT1_BLOCK + sub * B + j] = src[idx];
    }
}

int bt1_device(pg_ctx *ctx, int n, double *Z, Sb2Work &w)
{
    hipStream_t s = ctx->stream;
    const int nref = w.npan * B;                      // reflector columns 0 .. nref-1; column i has its unit at row i + 64
    if (nref == 0) return PG_OK;
    constexpr int BB = BT1_BLOCK;
    const int nblk = (nref + BB - 1) / BB;
    for (int b = nblk - 1; b >= 0; b--) {
        const int i0 = b * BB, m = std::min(BB, nref - i0), nsub = (m + B - 1) / B;
        const long long rows = n - i0 - B;
        if (rows <= 0) continue;
        const double *V = w.Vst + (size_t)(i0 + B) * n + i0;      // rows i0 + 64 .., columns i0 .. i0 + m - 1
        double *Zr = Z + (size_t)(i0 + B) * n;
        int rc = dgemm(ctx, true, m, m, rows, 1.0, V, n, V, n, 0.0, w.G, BB);
        if (rc) return rc;
        PG_HIP(hipMemsetAsync(w.T, 0, (size_t)BB * BB * sizeof(double), s));
        bt1_diag_t_kernel<<<nsub, 256, 0, s>>>(m, i0 / B, w.Tst, w.T);
        PG_HIP(hipGetLastError());
        for (int k = 1; k < nsub; k++) {     // block column k of T: T(0:r, r:r+w) = -T(0:r,0:r) G(0:r, r:r+w) T(r:r+w, r:r+w); W is free here
            const int r = k * B, wdt = std::min(B, m - r);
            rc = dgemm(ctx, false, r, wdt, r, 1.0, w.T, BB, w.G + r, BB, 0.0, w.W, B);
            if (!rc) rc = dgemm(ctx, false, r, wdt, wdt, -1.0, w.W, B, w.T + (size_t)r * BB + r, BB, 0.0, w.T + r, BB);
            if (rc) return rc;
        }
        rc = dgemm(ctx, true, m, n, rows, 1.0, V, n, Zr, n, 0.0, w.W, n);          // W  = V' Z
        if (!rc) rc = dgemm(ctx, false, m, n, m, 1.0, w.T, BB, w.W, n, 0.0, w.W2, n);   // W2 = T W
        if (!rc) rc = dgemm(ctx, false, rows, n, m, -1.0, V, n, w.W2, n, 1.0, Zr, n);   // Z -= V W2
        if (rc) return rc;
    }
    return PG_OK;
}

}  // namespace pg

#ifdef PG_BCS_TIME
extern "C" int pgx_bcs_time(long long *out12, int reset)
{
    if (reset) { long long z[12] = {0}; return hipMemcpyToSymbol(HIP_SYMBOL(pg::g_bcs_time), z, sizeof(z)) == hipSuccess ? 0 : -1; }
    return hipMemcpyFromSymbol(out12, HIP_SYMBOL(pg::g_bcs_time), sizeof(long long) * 12) == hipSuccess ? 0 : -1;
}
#endif

#ifdef PG_BT2_TIME
extern "C" int pgx_bt2_time(long long *out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(pg::g_bt2_time), sizeof(long long) * 16) == hipSuccess ? 0 : -1;
}
#endif
