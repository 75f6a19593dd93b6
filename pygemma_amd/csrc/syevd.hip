// syevd.hip — H1: eigendecomposition of the relatedness matrix K (lmm/lmm.py:151-162 / :196-207; the
// reference calls scipy.linalg.eigh = LAPACK ssyevr in float32).  Built from scratch for gfx950, in fp64:
//
//   1. blocked Householder tridiagonalisation  K = Q T Q'   (panel of NB reflectors: one HBM-bound symv per
//      column + small panel GEMVs, then a rank-2NB trailing update on fp64 MFMA)
//   2. divide and conquer on T (rank-one merges: deflation on the host — O(n) sequential — secular
//      equation, Loewner-corrected vectors and the eigenvector GEMMs on the device)
//   3. back-transformation  U = Q Z  with blocked compact-WY reflectors (fp64 MFMA GEMMs)
//
// Delivered like the reference: ascending eigenvalues clamped at 0 (lmm.py:157) in float32, U with
// eigenvector j in column j in float32 — plus the fp64 pair for the invariant checks.  Parity is judged by
// invariants (residual, orthogonality, agreement with host LAPACK in fp64), not by agreement with the
// reference's float32 LAPACK (SURVEY 8c Tier B).
#include "common.hpp"
#include "dgemm.hpp"
#include "sb2.hpp"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <thread>
#include <vector>

namespace pg {

constexpr int NB = 64;  // reflectors per panel
#ifndef PG_SYMV_OCC
#define PG_SYMV_OCC 2     // workgroups of the symmetric symv per CU the register allocation is held to (A/B knob)
#endif
static int alloc_d(double **p, size_t count);

// ---------------------------------------------------------------------------------------------
// A (N x N fp64, full symmetric) from the lower triangle of K (n x n float32); N = n or n + 1: the extra row/column is zero
__global__ void sym_from_lower_kernel(long long n, long long N, const float *K, double *A)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * N) return;
    long long i = idx / N, j = idx % N;
    A[idx] = (i < n && j < n) ? (double)(i >= j ? K[i * n + j] : K[j * n + i]) : 0.0;
}

__device__ __forceinline__ double block_sum(double v, double *sh)
{
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; w++) t += sh[w];
    return t;
}

// sum of cnt partials, the same additions on every wavefront of every workgroup (lane-strided, then xor butterfly)
__device__ __forceinline__ double det_sum(const double *p, int cnt)
{
    double s = 0.0;
    for (int b = threadIdx.x & 63; b < cnt; b += 64) s += p[b];
    for (int st = 1; st < 64; st <<= 1) s += __shfl_xor(s, st, 64);
    return s;
}
// Householder scalars of column i from alpha = a[i+1] and |a[i+2:]|^2 (dlarfg)
__device__ __forceinline__ void larfg_scalars(double alpha, double xnorm2, double &beta, double &tau, double &scal)
{
    if (xnorm2 == 0.0) { beta = alpha; tau = 0.0; scal = 0.0; }
    else {
        beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
        tau = (beta - alpha) / beta;
        scal = 1.0 / (alpha - beta);
    }
}
// element r of the reflector v_i (v[i+1] = 1, zeros above), from the un-scaled column a
__device__ __forceinline__ double vval(const double *acol, double scal, int i, int r)
{
    return (r == i + 1) ? 1.0 : ((r > i + 1) ? acol[r] * scal : 0.0);
}

// Per column i of the panel (ci = i - i0 columns already in it) three launches:
//   col_kernel      : finish W[ci-1] (needs the global w.v of the previous column), form column i of A with the
//                     pending panel updates  a = A[:,i] - V W[i,:]' - W V[i,:]'  and partial sums of |a[i+2:]|^2
//   (no launch of its own: every consumer derives beta, tau and the scale of v from the norm partials, v_r = a_r * scal)
//   symv_dots_kernel: y = A[i+1:, i+1:] v (HBM-bound, 16-byte loads, one wavefront per row) and, in extra
//                     workgroups, the 2*ci panel dot products W[c,:].v, V[c,:].v
//   w_update_kernel : w = tau (y - V t1 - W t2), partial sums of w.v; stores v into the panel, Vall and vcur
// P = [V ; W ; V] stacked (3*NB x n), panel column c of V at P[c*n + r], of W at P[(NB+c)*n + r]
#ifndef PG_COL_CG
#define PG_COL_CG 8       // (4: 465 ms of tridiagonalisation at n = 10 000, 8: 460, 16: 463)
#endif
__global__ __launch_bounds__(64 * PG_COL_CG) void col_kernel(int n, int i, int ci, int nblk_prev, const double *A, double *P, const double *vprev,
                                                  const double *wtmp, const double *tauvec, const double *wvpart, double *acol, double *normpart)
{
    // 64 rows per workgroup, the panel columns split CG ways over the waves (short latency chains)
    constexpr int CG = PG_COL_CG;
    __shared__ double part[CG][64];
    __shared__ double sh[CG];
    const int rr = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int r = i + blockIdx.x * 64 + rr;
    double *V = P, *W = P + (size_t)NB * n;
    double alpha = 0.0, wi = 0.0;
    if (ci > 0) {
        const double dot = det_sum(wvpart, nblk_prev);
        alpha = -0.5 * tauvec[i - 1] * dot;
        wi = wtmp[i] + alpha * vprev[i];              // W[ci-1][i]
    }
    double acc = 0.0;
    if (r < n) {
#pragma unroll 4
        for (int c = cg; c + 1 < ci; c += CG) acc -= V[(size_t)c * n + r] * W[(size_t)c * n + i] + W[(size_t)c * n + r] * V[(size_t)c * n + i];
    }
    part[cg][rr] = acc;
    __syncthreads();
    double sq = 0.0;
    if (cg == 0 && r < n) {
        double ps = part[0][rr];
#pragma unroll
        for (int g = 1; g < CG; g++) ps += part[g][rr];
        acc = A[(size_t)r * n + i] + ps;
        if (ci > 0) {
            const int c = ci - 1;
            const double wr = wtmp[r] + alpha * vprev[r];
            W[(size_t)c * n + r] = wr;                 // rows <= i-1 of this column stay zero (panel memset)
            acc -= V[(size_t)c * n + r] * wi + wr * V[(size_t)c * n + i];
        }
        acol[r] = acc;
        if (r >= i + 2) sq = acc * acc;
    }
    sq = block_sum(sq, sh);
    if (threadIdx.x == 0) normpart[blockIdx.x] = sq;
}

__global__ __launch_bounds__(256) void symv_dots_kernel(int n, int i, int ci, int nsymv, const double *A, const double *P, const double *acol,
                                                        const double *normpart, int nblk_col, double *hh, double *dvec, double *evec, double *tauvec,
                                                        double *y, double *t)
{
    __shared__ double sh[4];
    double beta, tau, scal;
    larfg_scalars(acol[i + 1], det_sum(normpart, nblk_col), beta, tau, scal);
    if (blockIdx.x == 0 && threadIdx.x == 0) { hh[0] = beta; hh[1] = tau; hh[2] = scal; dvec[i] = acol[i]; evec[i] = beta; tauvec[i] = tau; }
    if ((int)blockIdx.x < nsymv) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int r = i + 1 + blockIdx.x * 4 + wave;
        if (r >= n) return;
        const double *row = A + (size_t)r * n;
        double acc = 0.0;
        for (int c = i + 1 + lane; c < n; c += 64) acc = fma(row[c], vval(acol, scal, i, c), acc);
        for (int s = 1; s < 64; s <<= 1) acc += __shfl_xor(acc, s, 64);
        if (lane == 0) y[r] = acc;
    } else {
        const int b = blockIdx.x - nsymv;   // t[b] = W[b,:].v (b < ci) ; t[ci + b] = V[b,:].v
        const double *src = (b < ci) ? P + (size_t)(NB + b) * n : P + (size_t)(b - ci) * n;
        double acc = 0.0;
        for (int r = i + 1 + threadIdx.x; r < n; r += blockDim.x) acc = fma(src[r], vval(acol, scal, i, r), acc);
        acc = block_sum(acc, sh);
        if (threadIdx.x == 0) t[b] = acc;
    }
}

// ---- symmetric product from the LOWER triangle only (n even): half the HBM bytes of the full-row kernel above.
// Trailing rows are cut into 64-row blocks I (from s = i+1), columns into 128-column blocks J from the even origin
// cb = s & ~1 (16-byte aligned double2 loads; the extra column i, if any, meets v[i] = 0).  One wavefront per tile
// (I, J), J <= (64 I + 63 + s - cb) / 128: lane l owns columns c, c+1 = cb + 128 J + 2l and uses every loaded element
// twice,  rowacc[r] += A[r][c] v[c] (c <= r)  and  colacc[c] += A[r][c] v[r] (c < r).  The 64 row sums leave through
// a reduce-scatter into rowpart[J][r], the column sums into colpart[I][c]; w_update_kernel adds the partials in a
// fixed order (deterministic, no atomics).  Workgroups with blockIdx.x >= nbr compute the panel dot products.
template <int NVP>
__device__ __forceinline__ double reduce_scatter_t(double (&v)[NVP], int lane)
{
#pragma unroll
    for (int st = 1; st < NVP; st <<= 1) {
        const bool b = (lane & st) != 0;
#pragma unroll
        for (int m = 0; m < NVP / (2 * st); m++) {
            const double keep = b ? v[2 * m + 1] : v[2 * m];
            const double send = b ? v[2 * m] : v[2 * m + 1];
            v[m] = keep + __shfl_xor(send, st, 64);
        }
    }
    return v[0];
}

// 32 rows of a tile: row sums (reduce-scattered: lane l and l+32 both end with the sum of row rbase + l%32) and the
// running column sums.  Two calls per 64-row tile keep the row accumulators at 32 doubles per lane.
template <bool INTERIOR>
__device__ __forceinline__ double symv_half(int n, int rbase, int c0, bool cok, const double *Ac, const double *vrow, double vc0, double vc1,
                                            int lane, double &col0, double &col1)
{
    double rowacc[32];
#pragma unroll
    for (int bt = 0; bt < 4; bt++) {
        double2 a[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = rbase + bt * 8 + q;
            a[q] = (INTERIOR || (cok && r < n)) ? *reinterpret_cast<const double2 *>(Ac + (size_t)r * n) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int r = rbase + bt * 8 + q;
            const double vr = vrow[bt * 8 + q];           // LDS broadcast (zero for rows past the end)
            double ax = a[q].x, ay = a[q].y;
            if (!INTERIOR) { if (c0 > r) ax = 0.0; if (c0 + 1 > r) ay = 0.0; }
            rowacc[bt * 8 + q] = fma(ax, vc0, ay * vc1);
            if (!INTERIOR) { if (c0 >= r) ax = 0.0; if (c0 + 1 >= r) ay = 0.0; }
            col0 = fma(ax, vr, col0);
            col1 = fma(ay, vr, col1);
        }
    }
    double tot = reduce_scatter_t<32>(rowacc, lane);
    tot += __shfl_xor(tot, 32, 64);
    return tot;
}

__global__ __launch_bounds__(256, PG_SYMV_OCC) void symv_sym_kernel(int n, int i, int ci, int nbr, const double *A, const double *P, const double *acol,
                                                       const double *normpart, int nblk_col, double *hh, double *dvec, double *evec, double *tauvec,
                                                       double *rowpart, double *colpart, double *t)
{
    __shared__ double sh[4];
    __shared__ double vrows[64];
    const int s = i + 1, cb = s & ~1, delta = s - cb;
    double beta, tau, scal;
    larfg_scalars(acol[i + 1], det_sum(normpart, nblk_col), beta, tau, scal);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        hh[0] = beta; hh[1] = tau; hh[2] = scal; dvec[i] = acol[i]; evec[i] = beta; tauvec[i] = tau;
    }
    if ((int)blockIdx.x < nbr) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int I = blockIdx.x, J = blockIdx.y * 4 + wave;
        const int r0 = s + 64 * I;
        if (threadIdx.x < 64) vrows[threadIdx.x] = (r0 + threadIdx.x < n) ? vval(acol, scal, i, r0 + threadIdx.x) : 0.0;
        __syncthreads();
        if (J > (64 * I + 63 + delta) / 128) return;
        const int c0 = cb + 128 * J + 2 * lane;
        const bool cok = c0 < n;                       // n even and c0 even: c0 + 1 < n as well
        const double vc0 = cok ? vval(acol, scal, i, c0) : 0.0, vc1 = cok ? vval(acol, scal, i, c0 + 1) : 0.0;
        const double *Ac = A + (cok ? c0 : 0);
        const bool interior = (r0 + 64 <= n) && (cb + 128 * J + 127 < r0);   // every column left of every row, all in range
        double col0 = 0.0, col1 = 0.0;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
            const double tot = interior ? symv_half<true>(n, r0 + 32 * hf, c0, cok, Ac, vrows + 32 * hf, vc0, vc1, lane, col0, col1)
                                        : symv_half<false>(n, r0 + 32 * hf, c0, cok, Ac, vrows + 32 * hf, vc0, vc1, lane, col0, col1);
            const int r = r0 + 32 * hf + lane;
            if (lane < 32 && r < n) rowpart[(size_t)J * n + r] = tot;
        }
        if (cok) { colpart[(size_t)I * n + c0] = col0; colpart[(size_t)I * n + c0 + 1] = col1; }
    } else if (blockIdx.y == 0) {
        const int b = blockIdx.x - nbr;   // t[b] = W[b,:].v (b < ci) ; t[ci + b] = V[b,:].v
        const double *src = (b < ci) ? P + (size_t)(NB + b) * n : P + (size_t)(b - ci) * n;
        double acc = 0.0;
        for (int r = i + 1 + threadIdx.x; r < n; r += blockDim.x) acc = fma(src[r], vval(acol, scal, i, r), acc);
        acc = block_sum(acc, sh);
        if (threadIdx.x == 0) t[b] = acc;
    }
}

#ifndef PG_WUPD_CG
#define PG_WUPD_CG 16      // waves per 64-row block of the w update, each taking every CG-th panel column / partial sum (4 -> 16: 471 -> 457 ms of tridiagonalisation at n = 10 000; 148 workgroups of 4 waves left the loads un-hidden)
#endif
template <bool SYM, int CG = PG_WUPD_CG>
__global__ __launch_bounds__(64 * CG) void w_update_kernel(int n, int i, int ci, int nbr, double *P, const double *acol, const double *hh, const double *y,
                                                       const double *rowpart, const double *colpart, const double *t,
                                                       double *Vall, double *vcur, double *wtmp, double *partial)
{
    __shared__ double part[CG][64];
    __shared__ double sh[CG];
    const int rr = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int r = i + 1 + blockIdx.x * 64 + rr;
    const double *V = P, *W = P + (size_t)NB * n;
    const double tau = hh[1], scal = hh[2];
    double acc = 0.0;
    if (r < n) {
#pragma unroll 4
        for (int c = cg; c < ci; c += CG) acc -= V[(size_t)c * n + r] * t[c] + W[(size_t)c * n + r] * t[ci + c];
        if (SYM) {
            // (A v)_r from the symmetric partials, split over the 4 waves, each in a fixed order
            const int s = i + 1, cb = s & ~1, delta = s - cb;
            const int Ir = (r - s) >> 6, nJ = (64 * Ir + 63 + delta) / 128 + 1;
            double ya = 0.0;
#pragma unroll 8
            for (int J = cg; J < nJ; J += CG) ya += rowpart[(size_t)J * n + r];
            const int Jc = (r - cb) >> 7;
            int Imin = (128 * Jc - delta) / 64;
            if (Jc == 0) Imin = 0;
#pragma unroll 8
            for (int I = Imin + cg; I < nbr; I += CG) ya += colpart[(size_t)I * n + r];
            acc += ya;
        }
    }
    part[cg][rr] = acc;
    __syncthreads();
    double wv = 0.0;
    if (cg == 0 && r < n) {
        double ps = part[0][rr];
#pragma unroll
        for (int g = 1; g < CG; g++) ps += part[g][rr];
        acc = (SYM ? 0.0 : y[r]) + ps;
        acc *= tau;
        wtmp[r] = acc;
        const double v = vval(acol, scal, i, r);       // the reflector, stored here for everything downstream
        P[(size_t)ci * n + r] = v; P[(size_t)(2 * NB + ci) * n + r] = v; vcur[r] = v;
        Vall[(size_t)r * n + i] = v;
        wv = acc * v;
    }
    wv = block_sum(wv, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = wv;
}

// W[ci,:] = w - (tau/2)(w.v) v   (last column of a panel; the others are finished by the next col_kernel)
__global__ __launch_bounds__(256) void w_final_kernel(int n, int i, int ci, int nblk, double *P, const double *v, const double *wtmp,
                                                      const double *tauvec, const double *partial)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const double dot = det_sum(partial, nblk);        // before the bounds check: every lane takes part in the shuffles
    const double alpha = -0.5 * tauvec[i] * dot;
    if (r >= n) return;
    P[(size_t)(NB + ci) * n + r] = (r > i) ? wtmp[r] + alpha * v[r] : 0.0;
}

struct SytrdWork {
    double *A = nullptr, *P = nullptr, *Vall = nullptr, *acol = nullptr, *vcur = nullptr, *y = nullptr, *t = nullptr,
           *wtmp = nullptr, *partial = nullptr, *d = nullptr, *e = nullptr, *tau = nullptr, *rowpart = nullptr, *colpart = nullptr;
    // t: [0, 2*NB) panel dot products, [2*NB, 2*NB+3) beta/tau/scale of the current column
    // partial: [0, n/64+2) w.v partial sums, [n/64+2, 2(n/64+2)) column-norm partial sums
};

// Householder tridiagonalisation of the symmetric fp64 matrix A (n x n, full storage, destroyed).
// Outputs (device): d[n], e[n-1], tau[n-1], Vall (n x n row-major, column i = v_i).
static int sytrd_device(pg_ctx *ctx, int n, SytrdWork &w)
{
    hipStream_t s = ctx->stream;
    // The stream is drained once per panel, always (~2 ms per solve at n = 10 000).  This loop queues ~3 launches per column;
    // rocprofv3's counter mode (--pmc) keeps per-dispatch state for every dispatch still in flight and runs off the end of it
    // (SIGSEGV inside librocprofiler-sdk's hsa intercept, reached from the hipLaunchKernel of symv_sym_kernel) once ~10^4 launches
    // are queued without a synchronisation (profiles/r02_pmc_abort_diagnosis.txt).  The bound on the queue depth therefore
    // lives here, for every caller, not in an opt-in knob; PG_SYEVD_PANEL_SYNC=0 turns it off for A/B timing.
    const char *ps_env = getenv("PG_SYEVD_PANEL_SYNC");
    const bool panel_sync = !(ps_env && ps_env[0] == '0' && ps_env[1] == '\0');
    for (int i0 = 0; i0 < n - 1; i0 += NB) {
        const int nbc = std::min(NB, n - 1 - i0);
        if (panel_sync) PG_HIP(hipStreamSynchronize(s));
        PG_HIP(hipMemsetAsync(w.P, 0, (size_t)3 * NB * n * sizeof(double), s));
        int nblk_prev = 0;
        double *normpart = w.partial + (n / 64 + 2);
        for (int ci = 0; ci < nbc; ci++) {
            const int i = i0 + ci;
            const int nblk_col = (n - i + 63) / 64;
            col_kernel<<<nblk_col, 64 * PG_COL_CG, 0, s>>>(n, i, ci, nblk_prev, w.A, w.P, w.vcur, w.wtmp, w.tau, w.partial, w.acol, normpart);
            double *hh = w.t + 2 * NB;
            const int nt = n - i - 1;
            const int nblk = (nt + 63) / 64;
            if ((n & 1) == 0 && w.rowpart) {
                const int nbr = nblk, ywaves = (64 * (nbr - 1) + 64) / 128 + 1;
                symv_sym_kernel<<<dim3(nbr + 2 * ci, (ywaves + 3) / 4), 256, 0, s>>>(n, i, ci, nbr, w.A, w.P, w.acol, normpart, nblk_col, hh, w.d, w.e, w.tau,
                                                                                  w.rowpart, w.colpart, w.t);
                w_update_kernel<true><<<nblk, 64 * PG_WUPD_CG, 0, s>>>(n, i, ci, nbr, w.P, w.acol, hh, w.y, w.rowpart, w.colpart, w.t, w.Vall, w.vcur, w.wtmp, w.partial);
            } else {
                const int nsymv = (nt + 3) / 4;
                symv_dots_kernel<<<nsymv + 2 * ci, 256, 0, s>>>(n, i, ci, nsymv, w.A, w.P, w.acol, normpart, nblk_col, hh, w.d, w.e, w.tau, w.y, w.t);
                w_update_kernel<false><<<nblk, 64 * PG_WUPD_CG, 0, s>>>(n, i, ci, nblk, w.P, w.acol, hh, w.y, w.rowpart, w.colpart, w.t, w.Vall, w.vcur, w.wtmp, w.partial);
            }
            nblk_prev = nblk;
            if (ci == nbc - 1) w_final_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, i, ci, nblk, w.P, w.vcur, w.wtmp, w.tau, w.partial);
        }
        PG_HIP(hipGetLastError());
        const int off = i0 + nbc;
        const long long nt = n - off;
        if (nt > 0) {
            // A22 -= V W' + W V'   as one TN GEMM with K = 2*NB:  [V;W]' [W;V]
            // (lower triangle only when the symmetric symv is in use: nothing reads the upper one any more)
            int rc = dgemm(ctx, true, nt, nt, 2 * NB, -1.0, w.P + off, n, w.P + (size_t)NB * n + off, n, 1.0,
                           w.A + (size_t)off * n + off, n, (n & 1) == 0 && w.rowpart != nullptr);
            if (rc) return rc;
        }
    }
    // last diagonal entry
    PG_HIP(hipMemsetAsync(w.P, 0, (size_t)3 * NB * n * sizeof(double), s));
    col_kernel<<<1, 64 * PG_COL_CG, 0, s>>>(n, n - 1, 0, 0, w.A, w.P, w.vcur, w.wtmp, w.tau, w.partial, w.acol, w.partial + (n / 64 + 2));
    PG_HIP(hipMemcpyAsync(w.d + (n - 1), w.acol + (n - 1), sizeof(double), hipMemcpyDeviceToDevice, s));
    PG_HIP(hipGetLastError());
    return PG_OK;
}


// =============================================================================================
// Divide and conquer for the symmetric tridiagonal eigenproblem  T = Z diag(lam) Z'.
// Cuppen's rank-one splitting with Gu/Eisenstat's stable eigenvector formula (the structure of LAPACK
// dstedc/dlaed0-4, re-derived for a host + GPU split): small leaves by implicit QL on the host; per merge
// the O(n) sort + deflation scan on the host, everything O(n^2)/O(n^3) on the device — one thread per
// secular root, one wavefront per Loewner product, fp64 MFMA GEMM for the eigenvector update.
// =============================================================================================
constexpr int DC_LEAF = 32;

// implicit-shift QL for a small symmetric tridiagonal (EISPACK tql2 lineage); Z (m x m row-major, ldz) in/out
static int host_tql2(int m, double *d, double *e, double *Z, int ldz)
{
    const double eps = 2.220446049250313e-16;
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Z[(size_t)i * ldz + j] = (i == j) ? 1.0 : 0.0;
    if (m == 1) return 0;
    e[m - 1] = 0.0;
    for (int l = 0; l < m; l++) {
        int iter = 0, mm;
        do {
            for (mm = l; mm < m - 1; mm++) {
                double dd = fabs(d[mm]) + fabs(d[mm + 1]);
                if (fabs(e[mm]) <= eps * dd) break;
            }
            if (mm != l) {
                if (iter++ == 300) return -1;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = hypot(g, 1.0);
                g = d[mm] - d[l] + e[l] / (g + copysign(r, g));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = mm - 1; i >= l; i--) {
                    double f = s * e[i], b = c * e[i];
                    r = hypot(f, g);
                    e[i + 1] = r;
                    if (r == 0.0) { d[i + 1] -= p; e[mm] = 0.0; break; }
                    s = f / r; c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * s + 2.0 * c * b;
                    p = s * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    for (int k = 0; k < m; k++) {
                        double *zk = Z + (size_t)k * ldz;
                        f = zk[i + 1];
                        zk[i + 1] = s * zk[i] + c * f;
                        zk[i] = c * zk[i] - s * f;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[mm] = 0.0;
            }
        } while (mm != l);
    }
    return 0;
}

__global__ void scatter_leaves_kernel(int n, int ldS, const double *S, const int *leaf_start, const int *leaf_size, const int *leaf_of_row, double *Q)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * ldS) return;
    int r = (int)(idx / ldS), j = (int)(idx % ldS);
    int lf = leaf_of_row[r];
    if (j < leaf_size[lf]) Q[(size_t)r * n + leaf_start[lf] + j] = S[idx];
}

__global__ void gather_z_kernel(int n, const double *Q, const int *zrow, double *z)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = zrow[i];
    z[i] = (r >= 0) ? Q[(size_t)r * n + i] : 0.0;
}

struct Rot { int a, b; double c, s; };

// One merge of a level, as the batched kernels below see it (the lower levels of the tree hold hundreds of small merges: eight
// launches each made those levels launch-bound — 156 merges of 64 rows: 6.9 ms at n = 10 000)
struct MergeDesc { int s, n1, nm, k, k1, k2, rot_off, nrot; double rho; };

// apply the deflation rotations (in order) to columns of Q, rows [r0, r0+nrow): one thread per row
__device__ __forceinline__ void givens_body(int n, int r0, int nrow, double *Q, const Rot *rots, int nrot, int bx)
{
    int r = r0 + bx * blockDim.x + threadIdx.x;
    if (r >= r0 + nrow) return;
    double *row = Q + (size_t)r * n;
    for (int t = 0; t < nrot; t++) {
        Rot g = rots[t];
        double x = row[g.a], y = row[g.b];
        row[g.a] = g.c * x + g.s * y;
        row[g.b] = g.c * y - g.s * x;
    }
}
__global__ void givens_kernel(int n, int r0, int nrow, double *Q, const Rot *rots, int nrot) { givens_body(n, r0, nrow, Q, rots, nrot, blockIdx.x); }
__global__ void givens_batched_kernel(int n, const MergeDesc *md, double *Q, const Rot *rots)
{
    const MergeDesc m = md[blockIdx.y];
    if (m.nrot > 0) givens_body(n, m.s, m.nm, Q, rots + m.rot_off, m.nrot, blockIdx.x);
}

// Tp[r][jj] = Qin[r0+r][col[jj]] (jj < k: non-deflated, in secular order);  deflated columns go straight to Qout
__device__ __forceinline__ void permute_cols_body(int n, int r0, int nm, int k, const double *Qin, const int *col, double *Tp, double *Qout, long long bx)
{
    long long idx = bx * blockDim.x + threadIdx.x;
    if (idx >= (long long)nm * nm) return;
    int r = (int)(idx / nm), jj = (int)(idx % nm);
    double v = Qin[(size_t)(r0 + r) * n + col[jj]];
    if (jj < k) Tp[(size_t)r * k + jj] = v;
    else Qout[(size_t)(r0 + r) * n + r0 + jj] = v;
}
__global__ void permute_cols_kernel(int n, int r0, int nm, int k, const double *Qin, const int *col, double *Tp, double *Qout)
{
    permute_cols_body(n, r0, nm, k, Qin, col, Tp, Qout, blockIdx.x);
}
// batched: the work space of the merge that starts at row s is the slice s n .. of Tp / Um (nm k <= nm n doubles), s .. of zh
__global__ void permute_cols_batched_kernel(int n, const MergeDesc *md, const double *Qin, const int *col, double *Tp, double *Qout)
{
    const MergeDesc m = md[blockIdx.y];
    permute_cols_body(n, m.s, m.nm, m.k, Qin, col + m.s, Tp + (size_t)m.s * n, Qout, blockIdx.x);
}

// One WAVEFRONT per root of  f(lam) = 1 + rho * sum_i w_i^2 / (dl_i - lam)  (rho > 0, dl strictly increasing);
// the 64 lanes split every sum over i and butterfly-reduce, so control flow is wave-uniform.
// Root j lies in (dl_j, dl_{j+1}) (last: (dl_{k-1}, dl_{k-1} + rho*|w|^2]).  The origin is moved to the
// nearer pole and the iteration runs on the offset tau, so that every difference dl_i - lam_j is obtained
// as (dl_i - dl_origin) - tau without cancellation.  Rational ("middle way") steps, bracket-safeguarded.
__device__ __forceinline__ double wave_sum(double v)
{
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
    return v;
}
__device__ __forceinline__ void secular_body(int k, const double *dl, const double *w, double rho, const int *rp, double *Dm, double *lam_out, int bx)
{
    const int lane = threadIdx.x & 63;
    const int j = bx * 4 + (threadIdx.x >> 6);
    if (j >= k) return;
    const double eps = 1.1102230246251565e-16;
    int org;
    double lo, hi, tau;
    const bool last = (j == k - 1);
    if (last) {
        double sw = 0.0;
        for (int i = lane; i < k; i += 64) sw += w[i] * w[i];
        sw = wave_sum(sw);
        org = k - 1; lo = 0.0; hi = rho * sw; tau = 0.5 * hi;
        if (k == 1) tau = hi;
    } else {
        const double dj = dl[j], gap = dl[j + 1] - dj, half = 0.5 * gap;
        double fm = 0.0;   // f at the midpoint, deltas taken from dl[j]
        for (int i = lane; i < k; i += 64) fm += rho * w[i] * w[i] / ((dl[i] - dj) - half);
        fm = 1.0 + wave_sum(fm);
        if (fm >= 0.0) { org = j; lo = 0.0; hi = half; tau = 0.5 * half; }
        else { org = j + 1; lo = -half; hi = 0.0; tau = -0.5 * half; }
        if (fm == 0.0) { lo = hi = tau = half; }
    }
    const double dorg = dl[org];
    const int jp = last ? k - 1 : j;   // psi: i <= jp, phi: i > jp
    for (int it = 0; it < 400 && lo != hi; it++) {
        double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0, sabs = 0.0;
        for (int i = lane; i < k; i += 64) {
            const double del = (dl[i] - dorg) - tau;
            const double rdel = 1.0 / del, t = rho * w[i] * w[i] * rdel, dt = t * rdel;      // one division per pole instead of two
            if (i <= jp) { psi += t; dpsi += dt; } else { phi += t; dphi += dt; }
            sabs += fabs(t);
        }
        psi = wave_sum(psi); dpsi = wave_sum(dpsi); phi = wave_sum(phi); dphi = wave_sum(dphi); sabs = wave_sum(sabs);
        const double f = 1.0 + psi + phi;
        if (f == 0.0) break;
        if (f < 0.0) lo = tau; else hi = tau;   // f is increasing on the interval
        if (fabs(f) <= 8.0 * eps * (1.0 + sabs + fabs(tau) * (dpsi + dphi))) break;
        if (hi - lo <= 2.0 * eps * fmax(fabs(lo), fabs(hi))) break;
        double tnew;
        if (last) {
            // one pole at dl_{k-1} (delta 0) + constant:  c + S/(0 - t) = 0
            const double A = -tau, S = (dpsi + dphi) * A * A, c = f - (dpsi + dphi) * A;
            tnew = (c > 0.0) ? S / c : 2.0 * tau;
        } else {
            const double A = (dl[j] - dorg) - tau, B = (dl[j + 1] - dorg) - tau;
            const double S = dpsi * A * A, R = dphi * B * B, c = f - dpsi * A - dphi * B;
            const double qa = c, qb = c * (A + B) + S + R, qc = A * B * f;
            double eta;
            if (qa == 0.0) eta = qc / qb;
            else {
                const double disc = sqrt(fabs(qb * qb - 4.0 * qa * qc));
                eta = (qb <= 0.0) ? (qb - disc) / (2.0 * qa) : 2.0 * qc / (qb + disc);
            }
            if (f * eta >= 0.0) eta = -f / (dpsi + dphi);
            tnew = tau + eta;
        }
        if (!(tnew > lo && tnew < hi) || it > 40) {
            // bisection; geometric when the bracket spans orders of magnitude on one side of 0
            if (lo > 0.0 && hi > 4.0 * lo) tnew = sqrt(lo * hi);
            else if (hi < 0.0 && lo < 4.0 * hi) tnew = -sqrt(lo * hi);
            else if (lo == 0.0 && hi > 0.0) tnew = (it > 60) ? hi * 1e-3 : 0.5 * hi;
            else if (hi == 0.0 && lo < 0.0) tnew = (it > 60) ? lo * 1e-3 : 0.5 * lo;
            else tnew = 0.5 * (lo + hi);
        }
        if (tnew == tau) break;
        tau = tnew;
    }
    for (int i = lane; i < k; i += 64) Dm[(size_t)rp[i] * k + j] = (dl[i] - dorg) - tau;   // row of pole i: its slot in the type-grouped order
    if (lane == 0) lam_out[j] = dorg + tau;
}
__global__ __launch_bounds__(256) void secular_kernel(int k, const double *dl, const double *w, double rho, const int *rp, double *Dm, double *lam_out)
{
    secular_body(k, dl, w, rho, rp, Dm, lam_out, blockIdx.x);
}
__global__ __launch_bounds__(256) void secular_batched_kernel(int n, const MergeDesc *md, const double *dl, const double *w, const int *rp, double *Dm, double *lam_out)
{
    const MergeDesc m = md[blockIdx.y];
    secular_body(m.k, dl + m.s, w + m.s, m.rho, rp + m.s, Dm + (size_t)m.s * n, lam_out + m.s, blockIdx.x);
}

// zhat_i = sign(w_i) sqrt| Dm[i][i] * prod_{j != i} Dm[i][j] / (dl_i - dl_j) |   (one wavefront per i)
__device__ __forceinline__ void zhat_body(int k, const double *dl, const double *w, const int *rp, const double *Dm, double *zh, int bx)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = bx * 4 + wave;
    if (i >= k) return;
    const double *row = Dm + (size_t)rp[i] * k;
    const double di = dl[i];
    double prod = 1.0;
    for (int j = lane; j < k; j += 64) prod *= (j == i) ? row[j] : row[j] / (di - dl[j]);
    for (int s = 1; s < 64; s <<= 1) prod *= __shfl_xor(prod, s, 64);
    if (lane == 0) zh[i] = copysign(sqrt(fabs(prod)), w[i]);
}
__global__ __launch_bounds__(256) void zhat_kernel(int k, const double *dl, const double *w, const int *rp, const double *Dm, double *zh) { zhat_body(k, dl, w, rp, Dm, zh, blockIdx.x); }
__global__ __launch_bounds__(256) void zhat_batched_kernel(int n, const MergeDesc *md, const double *dl, const double *w, const int *rp, const double *Dm, double *zh)
{
    const MergeDesc m = md[blockIdx.y];
    zhat_body(m.k, dl + m.s, w + m.s, rp + m.s, Dm + (size_t)m.s * n, zh + m.s, blockIdx.x);
}

// U[:, j] = (zh_i / Dm[i][j])_i, normalised; in place over Dm.  64 columns per workgroup (coalesced across j), the rows
// split over the 4 waves, column norms combined through LDS in a fixed order.
// Two launches, rows cut into chunks of UV_ROWS so that a merge of k = 7 000 poles fills the chip (110 workgroups of the one-kernel
// version left 60 % of the CUs idle, each thread walking 1 750 dependent load + divide steps twice: 18 ms per solve, now ~2):
// (1) the quotients in place and the sums of squares of a chunk's rows per column -> part[chunk][j]; (2) the chunks' sums added in
// chunk order, rows scaled.
constexpr int UV_ROWS = 256;
__device__ __forceinline__ void uvec_norm_body(int k, const double *zh, const int *rp, double *Dm, double *part, int bx, int by)
{
    __shared__ double sh[4][64];
    const int jj = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = bx * 64 + jj;
    const int i0 = by * UV_ROWS, i1 = (i0 + UV_ROWS < k) ? i0 + UV_ROWS : k;
    if (i0 >= k || bx * 64 >= k) return;      // (batched launches are sized for the largest merge of the level; uniform per workgroup)
    double ss = 0.0;
    if (j < k)
        for (int i = i0 + rg; i < i1; i += 4) { const size_t at = (size_t)rp[i] * k + j; const double v = zh[i] / Dm[at]; Dm[at] = v; ss += v * v; }
    sh[rg][jj] = ss;
    __syncthreads();
    if (rg == 0 && j < k) part[(size_t)by * k + j] = ((sh[0][jj] + sh[1][jj]) + sh[2][jj]) + sh[3][jj];
}
__device__ __forceinline__ void uvec_scale_body(int k, const double *zh, const int *rp, double *Dm, const double *part, int nchunk, int bx, int by)
{
    const int jj = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = bx * 64 + jj;
    if (j >= k) return;
    double ss = 0.0;
    for (int c = 0; c < nchunk; c++) ss += part[(size_t)c * k + j];
    const double inv = 1.0 / sqrt(ss);
    const int i0 = by * UV_ROWS, i1 = (i0 + UV_ROWS < k) ? i0 + UV_ROWS : k;
    for (int i = i0 + rg; i < i1; i += 4) { const size_t at = (size_t)rp[i] * k + j; Dm[at] *= inv; }      // the quotients were left there by uvec_norm_kernel
}
__global__ __launch_bounds__(256) void uvec_norm_kernel(int k, const double *zh, const int *rp, double *Dm, double *part) { uvec_norm_body(k, zh, rp, Dm, part, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(256) void uvec_scale_kernel(int k, const double *zh, const int *rp, double *Dm, const double *part, int nchunk)
{
    uvec_scale_body(k, zh, rp, Dm, part, nchunk, blockIdx.x, blockIdx.y);
}
// batched: the column norms' partial sums of the merge at row s live at part + s * pstride (pstride = n / UV_ROWS + 2 >= its chunks)
__global__ __launch_bounds__(256) void uvec_norm_batched_kernel(int n, const MergeDesc *md, const double *zh, const int *rp, double *Dm, double *part, int pstride)
{
    const MergeDesc m = md[blockIdx.z];
    uvec_norm_body(m.k, zh + m.s, rp + m.s, Dm + (size_t)m.s * n, part + (size_t)m.s * pstride, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void uvec_scale_batched_kernel(int n, const MergeDesc *md, const double *zh, const int *rp, double *Dm, const double *part, int pstride)
{
    const MergeDesc m = md[blockIdx.z];
    uvec_scale_body(m.k, zh + m.s, rp + m.s, Dm + (size_t)m.s * n, part + (size_t)m.s * pstride, (m.k + UV_ROWS - 1) / UV_ROWS, blockIdx.x, blockIdx.y);
}

// Q_new = [Q1 0; 0 Q2] U for every merge of a level: blockIdx.y = 2 * merge + (0: upper rows x columns of types 1, 2 | 1: lower rows x
// types 2, 3), blockIdx.x = 64 x 64 tile of that product; operands through LDS, fp64 MFMA (wavefront w: rows 16 w .. of the tile).
// The sizes here are small (merges of up to 1024 rows): what matters is ONE launch per level instead of two per merge.
__global__ __launch_bounds__(256) void merge_gemm_batched_kernel(int n, const MergeDesc *md, const double *Tp, const double *Um, double *Qout)
{
    constexpr int KC = 32, AP = KC + 1, TP = 65;            // 34 KB of static LDS (above 64 KB a static allocation is not honoured)
    __shared__ double As[64 * AP], Bs[KC * TP];
    const MergeDesc m = md[blockIdx.y >> 1];
    const int part = blockIdx.y & 1, k = m.k;
    if (k == 0) return;
    const int n1 = m.n1, n2 = m.nm - m.n1, k12 = m.k1 + m.k2, k23 = k - m.k1;
    const int M = part ? n2 : n1, K = part ? k23 : k12;
    const int tn = (k + 63) / 64, tm = (M + 63) / 64;
    if ((int)blockIdx.x >= tm * tn) return;
    const int ti = blockIdx.x / tn, tj = blockIdx.x % tn;
    const double *A = Tp + (size_t)m.s * n + (part ? (size_t)n1 * k + m.k1 : 0);       // M x K, ld k
    const double *Bm = Um + (size_t)m.s * n + (part ? (size_t)m.k1 * k : 0);           // K x k, ld k
    double *C = Qout + (size_t)(m.s + (part ? n1 : 0)) * n + m.s;                      // M x k, ld n
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, k4 = lane >> 4;
    doublex4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[t][e] = 0.0;
    for (int k0 = 0; k0 < K; k0 += KC) {
        for (int idx = tid; idx < 64 * KC; idx += 256) {
            const int r = idx / KC, c = idx % KC;               // A tile: 64 rows x KC
            As[r * AP + c] = (64 * ti + r < M && k0 + c < K) ? A[(size_t)(64 * ti + r) * k + k0 + c] : 0.0;
            const int rb = idx >> 6, cb = idx & 63;             // B tile: KC rows x 64
            Bs[rb * TP + cb] = (k0 + rb < K && 64 * tj + cb < k) ? Bm[(size_t)(k0 + rb) * k + 64 * tj + cb] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KC / 4; ks++) {
            const double a = As[(16 * wave + r16) * AP + 4 * ks + k4];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[(4 * ks + k4) * TP + 16 * t + r16], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int row = 64 * ti + 16 * wave + k4 + 4 * e, col = 64 * tj + 16 * t + r16;
            if (row < M && col < k) C[(size_t)row * n + col] = acc[t][e];
        }
}

__global__ void copy_block_kernel(int n, int r0, int nm, const double *Qin, double *Qout)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nm * nm) return;
    int r = r0 + (int)(idx / nm), c = r0 + (int)(idx % nm);
    Qout[(size_t)r * n + c] = Qin[(size_t)r * n + c];
}

__global__ void permute_final_kernel(int n, const double *Qin, const int *col, double *Qout)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * n) return;
    int r = (int)(idx / n), j = (int)(idx % n);
    Qout[idx] = Qin[(size_t)r * n + col[j]];
}

struct MergePlan {
    int s, n1, nm, k;
    double rho;
    std::vector<double> dl, w, ddefl;   // non-deflated poles/weights; values of the deflated columns
    std::vector<int> col;               // global source column for each output slot (k non-deflated, then deflated)
    std::vector<int> rowpos;            // slot (0..k-1) of sorted pole i: non-deflated columns are stored grouped by type
    int k1 = 0, k2 = 0;                 // columns living in the upper block only / in both (Givens-mixed); the rest: lower only
    std::vector<Rot> rots;
};

// host: sort + deflation (the dlaed2 scan).  d, z: block-local arrays of length nm (z2 already sign-adjusted).
static void plan_merge(MergePlan &mp, const double *d_in, const double *z_in, double beta)
{
    const int nm = mp.nm;
    const double eps = 1.1102230246251565e-16;
    std::vector<double> d(d_in, d_in + nm), z(nm);
    const double sgn = (beta < 0.0) ? -1.0 : 1.0;
    for (int i = 0; i < nm; i++) z[i] = z_in[i] * ((i >= mp.n1) ? sgn : 1.0) * 0.7071067811865475244;
    mp.rho = 2.0 * fabs(beta);
    // ascending order of the poles: sort (value, index) pairs — contiguous keys; equal values keep their index order, as a stable sort of
    // the indices would (this runs on the host between two launches: ~0.5 ms of idle GPU per top-level merge with the indirect sort)
    std::vector<int> idx(nm);
    {
        std::vector<std::pair<double, int>> key(nm);
        for (int i = 0; i < nm; i++) key[i] = {d[i], i};
        std::sort(key.begin(), key.end());
        for (int i = 0; i < nm; i++) idx[i] = key[i].second;
    }
    double dmax = 0.0, zmax = 0.0;
    for (int i = 0; i < nm; i++) { dmax = std::max(dmax, fabs(d[i])); zmax = std::max(zmax, fabs(z[i])); }
    const double tol = 8.0 * eps * std::max(dmax, zmax);
    mp.k = 0; mp.dl.clear(); mp.w.clear(); mp.ddefl.clear(); mp.col.clear(); mp.rots.clear(); mp.rowpos.clear(); mp.k1 = mp.k2 = 0;
    std::vector<int> defl_cols;
    // column types as in dlaed2: 1 = non-zero in the upper block only, 3 = lower only, 2 = both (after a Givens rotation of
    // columns from the two blocks): the update GEMM then only multiplies the non-zero parts (half the flops of a dense product)
    std::vector<int> typ(nm);
    for (int i = 0; i < nm; i++) typ[i] = (i < mp.n1) ? 1 : 3;
    if (mp.rho * zmax <= tol) {
        for (int jj = 0; jj < nm; jj++) { defl_cols.push_back(idx[jj]); mp.ddefl.push_back(d[idx[jj]]); }
    } else {
        int pj = -1;
        std::vector<int> nd;
        for (int jj = 0; jj < nm; jj++) {
            const int nj = idx[jj];
            if (mp.rho * fabs(z[nj]) <= tol) { defl_cols.push_back(nj); mp.ddefl.push_back(d[nj]); continue; }
            if (pj < 0) { pj = nj; continue; }
            double s_ = z[pj], c_ = z[nj];
            const double tau = sqrt(c_ * c_ + s_ * s_), t = d[nj] - d[pj];      // |z| <= 1 here (unit vectors' components): no need for hypot's range care
            c_ /= tau; s_ = -s_ / tau;
            if (fabs(t * c_ * s_) <= tol) {
                z[nj] = tau; z[pj] = 0.0;
                if (typ[pj] != typ[nj]) typ[nj] = 2;
                mp.rots.push_back({mp.s + pj, mp.s + nj, c_, s_});
                const double tt = d[pj] * c_ * c_ + d[nj] * s_ * s_;
                d[nj] = d[pj] * s_ * s_ + d[nj] * c_ * c_;
                d[pj] = tt;
                defl_cols.push_back(pj); mp.ddefl.push_back(d[pj]);
                pj = nj;
            } else { nd.push_back(pj); pj = nj; }
        }
        if (pj >= 0) nd.push_back(pj);
        mp.k = (int)nd.size();
        for (int q : nd) { mp.dl.push_back(d[q]); mp.w.push_back(z[q]); }
        mp.rowpos.assign(mp.k, 0);
        int slot = 0;
        for (int t = 1; t <= 3; t++)
            for (int i = 0; i < mp.k; i++)
                if (typ[nd[i]] == t) { mp.rowpos[i] = slot++; mp.col.push_back(mp.s + nd[i]); if (t == 1) mp.k1++; else if (t == 2) mp.k2++; }
    }
    for (int q : defl_cols) mp.col.push_back(mp.s + q);
}

struct StedcWork {
    double *Qa = nullptr, *Qb = nullptr, *Tp = nullptr, *Um = nullptr, *z = nullptr, *dnew = nullptr, *dl = nullptr, *w = nullptr, *zh = nullptr, *S = nullptr, *unorm = nullptr;
    int *ibuf = nullptr;   // zrow | col | leaf tables
    Rot *rots = nullptr;
    MergeDesc *mdesc = nullptr;   // the merges of one level (batched kernels)
    bool borrowed = false;        // Qa, Tp, Um belong to the caller (set before stedc_alloc): buffers of the two-stage path that are idle by then
};

// T = tridiag(d, e) (host arrays, length n / n-1) -> ascending eigenvalues (host) and Z (device, n x n row-major,
// eigenvector j in column j).  *Zout points into the work buffers.
static int stedc_device(pg_ctx *ctx, int n, const double *d_in, const double *e_in, std::vector<double> &evals, StedcWork &wk, double **Zout)
{
    hipStream_t st = ctx->stream;
    const bool timing = getenv("PG_SYEVD_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto mark = [&](const char *what, int a, int b) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const double t = now();
        fprintf(stderr, "[stedc n=%d] %-10s %6d %6d %8.2f ms\n", n, what, a, b, (t - t_prev) * 1e3);
        t_prev = t;
    };
    std::vector<double> d(d_in, d_in + n), e(e_in, e_in + (n > 1 ? n - 1 : 0));
    // The deflation tolerances of the merges compare eigenvalue-scale quantities with components of unit vectors (as dlaed2 does):
    // like dstedc, work on T scaled to max-norm ~ 1 — by a power of two, so that the scaling itself rounds nothing.
    double orgnrm = 0.0;
    for (double v : d) orgnrm = std::max(orgnrm, fabs(v));
    for (double v : e) orgnrm = std::max(orgnrm, fabs(v));
    double unscale = 1.0;
    if (orgnrm > 0.0 && std::isfinite(orgnrm)) {
        const int ex = std::ilogb(orgnrm);
        const double sc = std::ldexp(1.0, -ex);
        unscale = std::ldexp(1.0, ex);
        for (double &v : d) v *= sc;
        for (double &v : e) v *= sc;
    }
    // ---- leaves
    const int nleaf = (n + DC_LEAF - 1) / DC_LEAF;
    std::vector<int> bs(nleaf + 1);
    for (int l = 0; l <= nleaf; l++) bs[l] = (int)((long long)n * l / nleaf);
    for (int l = 1; l < nleaf; l++) { const double b = fabs(e[bs[l] - 1]); d[bs[l] - 1] -= b; d[bs[l]] -= b; }
    int maxleaf = 0;
    for (int l = 0; l < nleaf; l++) maxleaf = std::max(maxleaf, bs[l + 1] - bs[l]);
    std::vector<double> S((size_t)n * maxleaf, 0.0);
    std::vector<int> itab(3 * (size_t)n + 16, 0);
    int *leaf_start = itab.data(), *leaf_size = itab.data() + n, *leaf_of_row = itab.data() + 2 * n;
    {   // the leaves are independent small dense problems: a few host threads share them (disjoint slices of d, S and itab)
        std::atomic<int> next(0), failed(0);
        auto work = [&]() {
            for (int l = next.fetch_add(1); l < nleaf; l = next.fetch_add(1)) {
                const int s = bs[l], m = bs[l + 1] - bs[l];
                std::vector<double> dd(d.begin() + s, d.begin() + s + m), ee(m, 0.0), Zl((size_t)m * m);
                for (int i = 0; i + 1 < m; i++) ee[i] = e[s + i];
                if (host_tql2(m, dd.data(), ee.data(), Zl.data(), m) != 0) { failed = 1; return; }
                for (int i = 0; i < m; i++) { d[s + i] = dd[i]; for (int j = 0; j < m; j++) S[(size_t)(s + i) * maxleaf + j] = Zl[(size_t)i * m + j]; }
                leaf_start[l] = s; leaf_size[l] = m;
                for (int i = 0; i < m; i++) leaf_of_row[s + i] = l;
            }
        };
        const int nthr = (int)std::max(1u, std::min(8u, std::min((unsigned)nleaf, std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        for (int t = 1; t < nthr; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (failed) { set_error("stedc: leaf QL did not converge"); return PG_EINVAL; }
    }
    PG_HIP(hipMemsetAsync(wk.Qa, 0, (size_t)n * n * 8, st));
    PG_HIP(hipMemsetAsync(wk.Qb, 0, (size_t)n * n * 8, st));
    PG_HIP(hipMemcpyAsync(wk.S, S.data(), S.size() * 8, hipMemcpyHostToDevice, st));
    PG_HIP(hipMemcpyAsync(wk.ibuf, itab.data(), 3 * (size_t)n * 4, hipMemcpyHostToDevice, st));
    scatter_leaves_kernel<<<(unsigned)(((size_t)n * maxleaf + 255) / 256), 256, 0, st>>>(n, maxleaf, wk.S, wk.ibuf, wk.ibuf + n, wk.ibuf + 2 * n, wk.Qa);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(st));   // host staging buffers go out of scope below
    mark("leaves", nleaf, maxleaf);

    double *Qin = wk.Qa, *Qout = wk.Qb;
    std::vector<int> blocks(bs);   // boundaries of the current level
    // the tables that travel between host and device at every level live in page-locked memory kept with the context: the copies are
    // then asynchronous in fact, not only in name (staged from pageable vectors each of them blocked the host)
    const size_t nn = (size_t)n;
    const size_t hp_need = nn * (8 + 8 + 4 + 4 + 4 + 16) + (nn + 1) * sizeof(Rot) + 1024;
    if (ctx->hpin_bytes < hp_need) {
        if (ctx->hpin) { (void)hipHostFree(ctx->hpin); ctx->hpin = nullptr; ctx->hpin_bytes = 0; }
        PG_HIP(hipHostMalloc(&ctx->hpin, hp_need, hipHostMallocDefault));
        ctx->hpin_bytes = hp_need;
    }
    double *zhost = (double *)ctx->hpin, *dnew = zhost + nn, *dlw = dnew + nn;      // dlw: 2 n
    Rot *allrots = (Rot *)(dlw + 2 * nn);
    int *zrow = (int *)(allrots + nn + 1), *colbuf = zrow + nn, *rpbuf = colbuf + nn;
    std::fill(rpbuf, rpbuf + nn, 0);
    while (blocks.size() > 2) {
        const int nb = (int)blocks.size() - 1;
        std::vector<MergePlan> plans;
        std::fill(zrow, zrow + nn, -1);
        for (int b = 0; b + 1 < nb; b += 2) {
            MergePlan mp; mp.s = blocks[b]; mp.n1 = blocks[b + 1] - blocks[b]; mp.nm = blocks[b + 2] - blocks[b];
            for (int i = 0; i < mp.nm; i++) zrow[mp.s + i] = (i < mp.n1) ? (mp.s + mp.n1 - 1) : (mp.s + mp.n1);
            plans.push_back(std::move(mp));
        }
        PG_HIP(hipMemcpyAsync(wk.ibuf, zrow, (size_t)n * 4, hipMemcpyHostToDevice, st));
        gather_z_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, Qin, wk.ibuf, wk.z);
        PG_HIP(hipMemcpyAsync(zhost, wk.z, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        PG_HIP(hipStreamSynchronize(st));
        mark("  gather z", nb, 0);
        size_t rot_total = 0;
        for (auto &mp : plans) {
            plan_merge(mp, d.data() + mp.s, zhost + mp.s, e[mp.s + mp.n1 - 1]);
            for (int i = 0; i < mp.nm; i++) colbuf[mp.s + i] = mp.col[i];
            for (int i = 0; i < mp.k; i++) rpbuf[mp.s + i] = mp.rowpos[i];
            rot_total += mp.rots.size();
        }
        PG_REQUIRE(rot_total <= nn, "stedc: more deflation rotations than rows");      // (a rotation deflates a row: at most nm - 1 per merge)
        std::fill(dlw, dlw + 2 * nn, 0.0);
        {
            size_t at = 0;
            for (auto &mp : plans) {
                for (auto &r : mp.rots) allrots[at++] = r;
                for (int i = 0; i < mp.k; i++) { dlw[mp.s + i] = mp.dl[i]; dlw[nn + mp.s + i] = mp.w[i]; }
            }
        }
        PG_HIP(hipMemcpyAsync(wk.ibuf + n, colbuf, (size_t)n * 4, hipMemcpyHostToDevice, st));
        PG_HIP(hipMemcpyAsync(wk.ibuf + 2 * n, rpbuf, (size_t)n * 4, hipMemcpyHostToDevice, st));   // (the leaf tables are done with)
        if (rot_total > 0) PG_HIP(hipMemcpyAsync(wk.rots, allrots, rot_total * sizeof(Rot), hipMemcpyHostToDevice, st));
        PG_HIP(hipMemcpyAsync(wk.dl, dlw, (size_t)n * 8, hipMemcpyHostToDevice, st));
        PG_HIP(hipMemcpyAsync(wk.w, dlw + nn, (size_t)n * 8, hipMemcpyHostToDevice, st));
        mark("  plan+copy", nb, 0);
        size_t roff = 0;
        int nm_max = 0, k_max = 0;
        for (auto &mp : plans) { nm_max = std::max(nm_max, mp.nm); k_max = std::max(k_max, mp.k); }
        const int batch_max = getenv("PG_DC_BATCH_MAX") ? atoi(getenv("PG_DC_BATCH_MAX")) : 1024;      // 0: every merge on its own (A/B, tests)
        const bool batched = plans.size() >= 2 && nm_max <= batch_max;
        if (batched) {
            // the small merges of a low level: one launch per kernel for all of them (work space slices by the merge's first row)
            std::vector<MergeDesc> md(plans.size());
            for (size_t q = 0; q < plans.size(); q++) {
                const MergePlan &mp = plans[q];
                md[q] = MergeDesc{mp.s, mp.n1, mp.nm, mp.k, mp.k1, mp.k2, (int)roff, (int)mp.rots.size(), mp.rho};
                roff += mp.rots.size();
            }
            PG_HIP(hipMemcpyAsync(wk.mdesc, md.data(), md.size() * sizeof(MergeDesc), hipMemcpyHostToDevice, st));
            const unsigned nmrg = (unsigned)plans.size();
            const int pstride = n / UV_ROWS + 2;
            if (rot_total > 0) givens_batched_kernel<<<dim3((nm_max + 255) / 256, nmrg), 256, 0, st>>>(n, wk.mdesc, Qin, wk.rots);
            permute_cols_batched_kernel<<<dim3((unsigned)(((size_t)nm_max * nm_max + 255) / 256), nmrg), 256, 0, st>>>(n, wk.mdesc, Qin, wk.ibuf + n, wk.Tp, Qout);
            if (k_max > 0) {
                const int *rp = wk.ibuf + 2 * n;
                const unsigned nchunk = (unsigned)((k_max + UV_ROWS - 1) / UV_ROWS), ktile = (unsigned)((k_max + 63) / 64);
                secular_batched_kernel<<<dim3((k_max + 3) / 4, nmrg), 256, 0, st>>>(n, wk.mdesc, wk.dl, wk.w, rp, wk.Um, wk.dnew);
                zhat_batched_kernel<<<dim3((k_max + 3) / 4, nmrg), 256, 0, st>>>(n, wk.mdesc, wk.dl, wk.w, rp, wk.Um, wk.zh);
                uvec_norm_batched_kernel<<<dim3(ktile, nchunk, nmrg), 256, 0, st>>>(n, wk.mdesc, wk.zh, rp, wk.Um, wk.unorm, pstride);
                uvec_scale_batched_kernel<<<dim3(ktile, nchunk, nmrg), 256, 0, st>>>(n, wk.mdesc, wk.zh, rp, wk.Um, wk.unorm, pstride);
                const unsigned tiles = (unsigned)(((nm_max + 63) / 64) * ktile);
                merge_gemm_batched_kernel<<<dim3(tiles, 2 * nmrg), 256, 0, st>>>(n, wk.mdesc, wk.Tp, wk.Um, Qout);
            }
            PG_HIP(hipGetLastError());
        }
        for (auto &mp : plans) {
            if (batched) break;
            const int nm = mp.nm, k = mp.k, s = mp.s;
            if (!mp.rots.empty()) {
                givens_kernel<<<(nm + 255) / 256, 256, 0, st>>>(n, s, nm, Qin, wk.rots + roff, (int)mp.rots.size());
                roff += mp.rots.size();
            }
            permute_cols_kernel<<<(unsigned)(((size_t)nm * nm + 255) / 256), 256, 0, st>>>(n, s, nm, k, Qin, wk.ibuf + n + s, wk.Tp, Qout);
            if (k > 0) {
                const int *rp = wk.ibuf + 2 * n + s;
                secular_kernel<<<(k + 3) / 4, 256, 0, st>>>(k, wk.dl + s, wk.w + s, mp.rho, rp, wk.Um, wk.dnew + s);
                zhat_kernel<<<(k + 3) / 4, 256, 0, st>>>(k, wk.dl + s, wk.w + s, rp, wk.Um, wk.zh);
                {
                    const int nchunk = (k + UV_ROWS - 1) / UV_ROWS;
                    uvec_norm_kernel<<<dim3((k + 63) / 64, nchunk), 256, 0, st>>>(k, wk.zh, rp, wk.Um, wk.unorm);
                    uvec_scale_kernel<<<dim3((k + 63) / 64, nchunk), 256, 0, st>>>(k, wk.zh, rp, wk.Um, wk.unorm, nchunk);
                }
                // Q_new = [Q1 0; 0 Q2] U: the upper rows only meet the columns of types 1, 2 (slots [0, k1+k2)), the lower rows
                // those of types 2, 3 (slots [k1, k))
                const int n1 = mp.n1, n2 = nm - mp.n1, k12 = mp.k1 + mp.k2, k23 = k - mp.k1;
                double *Ctop = Qout + (size_t)s * n + s, *Cbot = Qout + (size_t)(s + n1) * n + s;
                int rc = PG_OK;
                if (k12 > 0) rc = dgemm(ctx, false, n1, k, k12, 1.0, wk.Tp, k, wk.Um, k, 0.0, Ctop, n);
                else PG_HIP(hipMemset2DAsync(Ctop, (size_t)n * 8, 0, (size_t)k * 8, (size_t)n1, st));
                if (rc) return rc;
                if (k23 > 0) rc = dgemm(ctx, false, n2, k, k23, 1.0, wk.Tp + (size_t)n1 * k + mp.k1, k, wk.Um + (size_t)mp.k1 * k, k, 0.0, Cbot, n);
                else PG_HIP(hipMemset2DAsync(Cbot, (size_t)n * 8, 0, (size_t)k * 8, (size_t)n2, st));
                if (rc) return rc;
            }
            PG_HIP(hipGetLastError());
        }
        std::vector<int> nblocks;
        for (int b = 0; b + 1 < nb; b += 2) nblocks.push_back(blocks[b]);
        if (nb % 2 == 1) {   // odd block carried to the next level unchanged
            const int s = blocks[nb - 1], m = blocks[nb] - s;
            copy_block_kernel<<<(unsigned)(((size_t)m * m + 255) / 256), 256, 0, st>>>(n, s, m, Qin, Qout);
            nblocks.push_back(s);
        }
        nblocks.push_back(n);
        PG_HIP(hipMemcpyAsync(dnew, wk.dnew, (size_t)n * 8, hipMemcpyDeviceToHost, st));
        PG_HIP(hipStreamSynchronize(st));
        for (auto &mp : plans) {
            for (int i = 0; i < mp.k; i++) d[mp.s + i] = dnew[mp.s + i];
            for (int i = mp.k; i < mp.nm; i++) d[mp.s + i] = mp.ddefl[i - mp.k];
        }
        { int ksum = 0; for (auto &mp : plans) ksum += mp.k; mark("level", (int)plans.size(), ksum); }
        blocks.swap(nblocks);
        std::swap(Qin, Qout);
    }
    // ---- final ascending order
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] < d[b]; });
    evals.resize(n);
    for (int i = 0; i < n; i++) evals[i] = d[idx[i]] * unscale;
    PG_HIP(hipMemcpyAsync(wk.ibuf, idx.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    permute_final_kernel<<<(unsigned)(((size_t)n * n + 255) / 256), 256, 0, st>>>(n, Qin, wk.ibuf, Qout);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(st));
    *Zout = Qout;
    return PG_OK;
}

static int stedc_alloc(int n, StedcWork &wk)
{
    int rc = PG_OK;
    double **bufs[] = {&wk.Qa, &wk.Qb, &wk.Tp, &wk.Um, &wk.z, &wk.dnew, &wk.dl, &wk.w, &wk.zh, &wk.S, &wk.unorm};
    size_t sizes[] = {(size_t)n * n, (size_t)n * n, (size_t)n * n, (size_t)n * n, (size_t)n, (size_t)n, (size_t)n, (size_t)n, (size_t)n, (size_t)n * (DC_LEAF + 1),
                      ((size_t)n / UV_ROWS + 2) * n};
    for (int k = 0; k < 11 && !rc; k++)
        if (!*bufs[k]) rc = alloc_d(bufs[k], sizes[k]);          // (borrowed buffers are already set)
    if (!rc) rc = dev_alloc(reinterpret_cast<void **>(&wk.ibuf), (3 * (size_t)n + 16) * 4);
    if (!rc) rc = dev_alloc(reinterpret_cast<void **>(&wk.rots), ((size_t)n + 1) * sizeof(Rot));
    if (!rc) rc = dev_alloc(reinterpret_cast<void **>(&wk.mdesc), ((size_t)n / DC_LEAF + 2) * sizeof(MergeDesc));
    return rc;
}
static void stedc_free(StedcWork &wk)
{
    if (wk.borrowed) wk.Qa = wk.Tp = wk.Um = nullptr;
    for (double *p : {wk.Qa, wk.Qb, wk.Tp, wk.Um, wk.z, wk.dnew, wk.dl, wk.w, wk.zh, wk.S, wk.unorm}) dev_free(p);
    dev_free(wk.ibuf);
    dev_free(wk.rots);
    dev_free(wk.mdesc);
    wk = StedcWork{};
}

static int alloc_d(double **p, size_t count) { return dev_alloc(reinterpret_cast<void **>(p), count * sizeof(double)); }

}  // namespace pg

using namespace pg;

// ---- internal test hooks (not part of include/pygemma_hip.h) -----------------------------------------
extern "C" int pgx_dgemm_dev(pg_ctx *ctx, int transA, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                             const double *B, int64_t ldb, double beta, double *C, int64_t ldc)
{
    PG_REQUIRE(ctx && A && B && C, "pgx_dgemm_dev: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    return dgemm(ctx, transA != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc);
}

// every mode of the fp64 GEMM behind one test hook: flags bit 0 transA (A stored K x M), bit 1 transB (B stored N x K), bit 2 lower_only
// (C symmetric, tiles above the diagonal skipped), bit 3 symA (A symmetric, lower triangle + full 128 x 128 diagonal tiles valid),
// bit 4 no split-K; kxorB: B's k index XOR-ed (multiple of 8)
extern "C" int pgx_dgemm_ex_dev(pg_ctx *ctx, int flags, int kxorB, int64_t M, int64_t N, int64_t K, double alpha, const double *A, int64_t lda,
                                const double *B, int64_t ldb, double beta, double *C, int64_t ldc)
{
    PG_REQUIRE(ctx && A && B && C, "pgx_dgemm_ex_dev: NULL argument");
    PG_HIP(hipSetDevice(ctx->device));
    DgemmDesc d;
    d.transA = (flags & 1) != 0; d.transB = (flags & 2) != 0; d.lower_only = (flags & 4) != 0; d.symA = (flags & 8) != 0;
    d.allow_splitk = (flags & 16) == 0; d.kxorB = kxorB;
    d.M = M; d.N = N; d.K = K; d.alpha = alpha; d.beta = beta; d.A = A; d.lda = lda; d.B = B; d.ldb = ldb; d.C = C; d.ldc = ldc;
    return dgemm_ex(ctx, d);
}

// the time stamps one workgroup of the ring GEMM left (PG_DGEMM_TUNE bit 3): 64 values, see RING_STAMP in dgemm.hpp
namespace pg { long long *g_ring_stamp_buf = nullptr; }
extern "C" int pgx_ring_stamps(long long *out64)
{
    PG_REQUIRE(out64, "pgx_ring_stamps: NULL argument");
    if (!pg::g_ring_stamp_buf) {            // first call: make the buffer; the GEMMs launched from now on (with PG_DGEMM_TUNE bit 3) fill it
        PG_HIP(hipMalloc(reinterpret_cast<void **>(&pg::g_ring_stamp_buf), 64 * sizeof(long long)));
        PG_HIP(hipMemset(pg::g_ring_stamp_buf, 0, 64 * sizeof(long long)));
    }
    PG_HIP(hipDeviceSynchronize());
    PG_HIP(hipMemcpy(out64, pg::g_ring_stamp_buf, 64 * sizeof(long long), hipMemcpyDeviceToHost));
    return PG_OK;
}

// ---- lmm/lmm.py:124-125  K <- Z K Z'  (Z: n x q, K: q x q; float32 or float64 each) on the device ----------------------------------
// Two fp64-MFMA products, T = Z K and R = T Z', and ONE rounding to float32 at the end (the reference rounds once per float32 BLAS
// product when both inputs are float32, or once at lmm.py:127-128 when one is float64: either way the result below is within one
// float32 rounding of the exact product, which is at least as close as the reference's own).
static __global__ void widen_kernel(int64_t rows, int64_t cols, const void *src, int is64, int64_t ld, double *dst)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t r = idx / cols, c = idx % cols;
    dst[idx] = is64 ? static_cast<const double *>(src)[r * ld + c] : (double)static_cast<const float *>(src)[r * ld + c];
}
static __global__ void narrow_kernel(int64_t rows, int64_t cols, const double *src, float *dst, int64_t ld)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    dst[(idx / cols) * ld + idx % cols] = (float)src[idx];
}
extern "C" int pg_zkzt_dev(pg_ctx *ctx, int64_t n, int64_t q, const void *Z, int z_is_f64, int64_t ldz, const void *K, int k_is_f64,
                           int64_t ldk, float *out, int64_t ldo)
{
    PG_REQUIRE(ctx && Z && K && out && n > 0 && q > 0 && ldz >= q && ldk >= q && ldo >= n, "pg_zkzt_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    double *Zd = nullptr, *Kd = nullptr, *T = nullptr, *R = nullptr;
    int rc = alloc_d(&Zd, (size_t)n * q);
    if (!rc) rc = alloc_d(&Kd, (size_t)q * q);
    if (!rc) rc = alloc_d(&T, (size_t)n * q);
    if (!rc) rc = alloc_d(&R, (size_t)n * n);
    hipStream_t st = ctx->stream;
    if (!rc) {
        widen_kernel<<<(unsigned)((n * q + 255) / 256), 256, 0, st>>>(n, q, Z, z_is_f64, ldz, Zd);
        widen_kernel<<<(unsigned)((q * q + 255) / 256), 256, 0, st>>>(q, q, K, k_is_f64, ldk, Kd);
        rc = dgemm(ctx, false, n, q, q, 1.0, Zd, q, Kd, q, 0.0, T, q);             // T = Z K
    }
    if (!rc) {
        DgemmDesc d;                                                              // R = T Z'  (B = Z stored N x K)
        d.transB = true; d.M = n; d.N = n; d.K = q; d.alpha = 1.0; d.beta = 0.0; d.A = T; d.lda = q; d.B = Zd; d.ldb = q; d.C = R; d.ldc = n;
        rc = dgemm_ex(ctx, d);
    }
    if (!rc) {
        narrow_kernel<<<(unsigned)((n * n + 255) / 256), 256, 0, st>>>(n, n, R, out, ldo);
        if (hipGetLastError() != hipSuccess) rc = PG_EHIP;
    }
    (void)hipStreamSynchronize(st);
    dev_free(Zd); dev_free(Kd); dev_free(T); dev_free(R);
    return rc;
}

extern "C" int pgx_sytrd_dev(pg_ctx *ctx, int64_t n64, const float *K, double *d, double *e, double *tau, double *Vall)
{
    PG_REQUIRE(ctx && K && d && e && tau && Vall && n64 >= 2 && n64 <= 65536, "pgx_sytrd_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    const int n = (int)n64;
    SytrdWork w;
    int rc = PG_OK;
    double **bufs[] = {&w.A, &w.P, &w.acol, &w.vcur, &w.y, &w.t, &w.wtmp, &w.partial, &w.rowpart, &w.colpart};
    size_t sizes[] = {(size_t)n * n, (size_t)3 * NB * n, (size_t)n, (size_t)n, (size_t)n, (size_t)2 * NB + 8, (size_t)n, 2 * ((size_t)n / 64 + 2),
                      ((size_t)n / 128 + 2) * n, ((size_t)n / 64 + 2) * n};
    for (int k = 0; k < 10 && !rc; k++) rc = alloc_d(bufs[k], sizes[k]);
    w.Vall = Vall; w.d = d; w.e = e; w.tau = tau;
    if (!rc) {
        sym_from_lower_kernel<<<(unsigned)(((size_t)n * n + 255) / 256), 256, 0, ctx->stream>>>(n, n, K, w.A);
        (void)hipMemsetAsync(w.Vall, 0, (size_t)n * n * 8, ctx->stream);   // rows <= i of reflector i are never written
        rc = sytrd_device(ctx, n, w);
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (int k = 0; k < 10; k++) if (*bufs[k]) (void)hipFree(*bufs[k]);
    return rc;
}

extern "C" int pgx_stedc_dev(pg_ctx *ctx, int64_t n64, const double *d_host, const double *e_host, double *evals_host, double *Z_dev)
{
    PG_REQUIRE(ctx && d_host && e_host && evals_host && Z_dev && n64 >= 1 && n64 <= 65536, "pgx_stedc_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    const int n = (int)n64;
    StedcWork wk;
    int rc = stedc_alloc(n, wk);
    std::vector<double> ev;
    double *Z = nullptr;
    if (!rc) rc = stedc_device(ctx, n, d_host, e_host, ev, wk, &Z);
    if (!rc) {
        for (int i = 0; i < n; i++) evals_host[i] = ev[i];
        hipError_t e2 = hipMemcpyAsync(Z_dev, Z, (size_t)n * n * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e2 != hipSuccess) { set_error("copy failed"); rc = PG_EHIP; }
    }
    (void)hipStreamSynchronize(ctx->stream);
    stedc_free(wk);
    return rc;
}

// compact-WY factor of a block of reflectors (forward, columnwise):  H_0 ... H_{m-1} = I - V T V'
// G = V'V (m x m, ld NB).  T upper triangular (ld NB).  One workgroup.
// Back-transformation block: BB reflectors = BB/NB sub-blocks of NB.  T (BB x BB, upper triangular, compact WY:
// H_0 .. H_{m-1} = I - V T V') is built from the Gram matrix G = V'V: the NB x NB diagonal blocks by the dlarft
// recurrence (one workgroup per sub-block), the off-diagonal block columns by T(0:r, r:r+w) = -T(0:r,0:r) G(0:r, r:r+w) T(r:r+w, r:r+w).
#ifndef PG_BT_BLOCK
#define PG_BT_BLOCK 256
#endif
constexpr int BB = PG_BT_BLOCK;
__global__ __launch_bounds__(64) void larft_kernel(int m, const double *G, const double *tau, double *T)
{
    __shared__ double Ts[NB][NB + 1];
    const int off = blockIdx.x * NB, mb = (m - off < NB) ? m - off : NB;
    G += (size_t)off * BB + off; T += (size_t)off * BB + off; tau += off;
    __shared__ double Gs[NB][NB + 1];          // the sub-block of G: the recurrence below walks its columns, 64 dependent steps
    const int i = threadIdx.x;
    for (int j = 0; j < NB; j++) if (i < NB) Ts[i][j] = 0.0;
    for (int l = 0; l < mb; l++) if (i < mb) Gs[l][i] = G[(size_t)l * BB + i];     // row l, coalesced over i
    __syncthreads();
    for (int j = 0; j < mb; j++) {
        const double tj = tau[j];
        double v = 0.0;
        if (i < j) {
            for (int l = i; l < j; l++) v += Ts[i][l] * Gs[l][j];
            v *= -tj;
        }
        __syncthreads();
        if (i < j) Ts[i][j] = v;
        if (i == j) Ts[j][j] = tj;
        __syncthreads();
    }
    for (int j = 0; j < NB; j++) if (i < NB) T[(size_t)i * BB + j] = Ts[i][j];
}

// leading n x n block of X (row stride N) and the first n eigenvalues -> float32 / float64 outputs
__global__ void finalize_kernel(long long n, long long N, const double *X, const double *ev, float *U32, float *ev32, double *U64)
{
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n * n) {
        const double x = X[(idx / n) * N + idx % n];
        if (U32) U32[idx] = (float)x;
        if (U64) U64[idx] = x;
    }
    if (idx < n && ev32) ev32[idx] = (float)fmax(ev[idx], 0.0);   // lmm.py:157 np.maximum(0.0, eigenVals), then float32
}

// Z <- Q Z with Q = H_0 H_1 ... H_{n-2} (reflectors in Vall columns, scalars tau on the device)
static int backtransform_device(pg_ctx *ctx, int n, const double *Vall, const double *tau, double *Z, double *G, double *T, double *W, double *W2)
{
    // G, T: BB x BB;  W, W2: BB x n.  Blocks of BB = 256 reflectors: every block is one read-modify-write pass over the
    // rows i0.. of Z, so wide blocks cut that HBM traffic (the NB = 64 panels of the tridiagonalisation made 157 passes).
    hipStream_t s = ctx->stream;
    const int nref = n - 1;
    const int nblk = (nref + BB - 1) / BB;
    for (int b = nblk - 1; b >= 0; b--) {
        const int i0 = b * BB, m = std::min(BB, nref - i0), nsub = (m + NB - 1) / NB;
        const double *V = Vall + (size_t)i0 * n + i0;   // rows i0.., columns i0..i0+m-1 (row i0 itself is zero)
        const long long rows = n - i0;
        int rc = dgemm(ctx, true, m, m, rows, 1.0, V, n, V, n, 0.0, G, BB);
        if (rc) return rc;
        PG_HIP(hipMemsetAsync(T, 0, (size_t)BB * BB * sizeof(double), s));
        larft_kernel<<<nsub, 64, 0, s>>>(m, G, tau + i0, T);
        PG_HIP(hipGetLastError());
        for (int k = 1; k < nsub; k++) {     // block column k of T; W is free here and serves as scratch
            const int r = k * NB, wdt = std::min(NB, m - r);
            rc = dgemm(ctx, false, r, wdt, r, 1.0, T, BB, G + r, BB, 0.0, W, NB);                         // X = T(0:r,0:r) G(0:r, r:r+w)
            if (!rc) rc = dgemm(ctx, false, r, wdt, wdt, -1.0, W, NB, T + (size_t)r * BB + r, BB, 0.0, T + r, BB);   // T(0:r, r:r+w) = -X T_kk
            if (rc) return rc;
        }
        rc = dgemm(ctx, true, m, n, rows, 1.0, V, n, Z + (size_t)i0 * n, n, 0.0, W, n);          // W  = V' Z
        if (!rc) rc = dgemm(ctx, false, m, n, m, 1.0, T, BB, W, n, 0.0, W2, n);                  // W2 = T W
        if (!rc) rc = dgemm(ctx, false, rows, n, m, -1.0, V, n, W2, n, 1.0, Z + (size_t)i0 * n, n);  // Z -= V W2
        if (rc) return rc;
    }
    return PG_OK;
}

static int syevd_onestage(pg_ctx *ctx, int n0, const float *K, float *evals, float *U, double *evals64, double *U64)
{
    // Odd n: work on N = n + 1 with a zero extra row/column.  It stays exactly decoupled through the reduction (v and w
    // have a zero there), its diagonal entry of T is then set above every other eigenvalue, so it comes out last with
    // eigenvector e_N and is dropped — and the even-n symmetric symv (half the HBM bytes, 16-byte loads) serves every n.
    const int n = (n0 > 1 && (n0 & 1)) ? n0 + 1 : n0;
    hipStream_t st = ctx->stream;
    // PG_SYEVD_TIMING=1: phase wall times on stderr (each mark synchronises the stream: diagnostic runs only)
    const bool timing = getenv("PG_SYEVD_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const double t = now();
        fprintf(stderr, "[pg_syevd_dev n=%d] %-16s %8.1f ms\n", n0, what, (t - t_prev) * 1e3);
        t_prev = t;
    };
    SytrdWork w;
    StedcWork wk;
    double *G = nullptr, *T = nullptr, *W = nullptr, *W2 = nullptr, *dev_ev = nullptr;
    int rc = PG_OK;
    double **bufs[] = {&w.A, &w.P, &w.Vall, &w.acol, &w.vcur, &w.y, &w.t, &w.wtmp, &w.partial, &w.d, &w.e, &w.tau, &G, &T, &W, &W2, &dev_ev,
                       &w.rowpart, &w.colpart};
    size_t sizes[] = {(size_t)n * n, (size_t)3 * NB * n, (size_t)n * n, (size_t)n, (size_t)n, (size_t)n, (size_t)2 * NB + 8, (size_t)n, 2 * ((size_t)n / 64 + 2),
                      (size_t)n, (size_t)n, (size_t)n, (size_t)BB * BB, (size_t)BB * BB, (size_t)BB * n, (size_t)BB * n, (size_t)n,
                      ((size_t)n / 128 + 2) * n, ((size_t)n / 64 + 2) * n};
    const int nbuf = (int)(sizeof(sizes) / sizeof(sizes[0]));
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(st);
        for (int k = 0; k < nbuf; k++) if (*bufs[k]) { (void)hipFree(*bufs[k]); *bufs[k] = nullptr; }
        stedc_free(wk);
    };
    for (int k = 0; k < nbuf && !rc; k++) rc = alloc_d(bufs[k], sizes[k]);
    if (!rc) rc = stedc_alloc(n, wk);
    if (rc) { cleanup(); return rc; }
    mark("allocate");
    std::vector<double> hd(n), he(n, 0.0), ev;
    double *Z = nullptr;
    if (n == 1) {
        // trivial: T = K
        sym_from_lower_kernel<<<1, 256, 0, st>>>(1, 1, K, w.A);
        if (hipMemcpyAsync(hd.data(), w.A, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = PG_EHIP;
    } else {
        // no early return from here on: every failure falls through to cleanup() (~8 n^2 doubles leaked on this path)
        if (hipMemsetAsync(w.Vall, 0, (size_t)n * n * 8, st) != hipSuccess || hipMemsetAsync(w.tau, 0, (size_t)n * 8, st) != hipSuccess) {
            set_error("pg_syevd_dev: hipMemsetAsync failed: %s", hipGetErrorString(hipGetLastError()));
            rc = PG_EHIP;
        }
        if (!rc) {
            sym_from_lower_kernel<<<(unsigned)(((size_t)n * n + 255) / 256), 256, 0, st>>>(n0, n, K, w.A);
            rc = sytrd_device(ctx, n, w);
        }
        if (!rc) {
            if (hipMemcpyAsync(hd.data(), w.d, (size_t)n * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(he.data(), w.e, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) { set_error("pg_syevd_dev: reading T back failed: %s", hipGetErrorString(hipGetLastError())); rc = PG_EHIP; }
        }
        if (!rc && n != n0) {
            if (he[n - 2] != 0.0) { set_error("pg_syevd_dev: padding row did not stay decoupled (e = %g)", he[n - 2]); rc = PG_EHIP; }
            double top = 0.0, nrm = 0.0;   // Gershgorin bound and max-norm of the leading n0 x n0 tridiagonal
            for (int i = 0; i < n0; i++) {
                top = std::max(top, hd[i] + (i > 0 ? fabs(he[i - 1]) : 0.0) + (i + 1 < n0 ? fabs(he[i]) : 0.0));
                nrm = std::max(nrm, std::max(fabs(hd[i]), i + 1 < n0 ? fabs(he[i]) : 0.0));
            }
            // strictly above every eigenvalue, and of the matrix's own magnitude: a constant here (it was max(1, |top|)) would set the
            // scale of T for the divide & conquer and drown a K of norm << 1 in its deflation tolerances
            hd[n - 1] = top + (nrm > 0.0 ? nrm : 1.0);
        }
    }
    mark("tridiagonalise");
    if (!rc) rc = stedc_device(ctx, n, hd.data(), he.data(), ev, wk, &Z);
    mark("divide&conquer");
    if (!rc && n > 1) rc = backtransform_device(ctx, n, w.Vall, w.tau, Z, G, T, W, W2);
    mark("back-transform");
    if (!rc) {
        hipError_t e1 = hipMemcpyAsync(dev_ev, ev.data(), (size_t)n0 * 8, hipMemcpyHostToDevice, st);
        if (e1 != hipSuccess) rc = PG_EHIP;
        if (!rc) {
            finalize_kernel<<<(unsigned)(((size_t)n0 * n0 + 255) / 256), 256, 0, st>>>(n0, n, Z, dev_ev, U, evals, U64);
            if (evals64 && hipMemcpyAsync(evals64, dev_ev, (size_t)n0 * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = PG_EHIP;
            if (hipGetLastError() != hipSuccess) rc = PG_EHIP;
        }
        if (rc == PG_EHIP) set_error("pg_syevd_dev: output stage failed");
    }
    mark("outputs");
    cleanup();   // synchronises the stream: outputs are complete on return
    mark("free");
    return rc;
}

// ---- two-stage path (csrc/sb2.hip): dense -> band (GEMM-bound) -> tridiagonal (bulge chasing) -> divide & conquer -> Q1 (Q2 Z) ----
constexpr int PG_RETRY_ONESTAGE = 1000;    // internal: a panel of the band reduction could not be factored by CholeskyQR2
static int syevd_twostage(pg_ctx *ctx, int n, const float *K, float *evals, float *U, double *evals64, double *U64)
{
    hipStream_t st = ctx->stream;
    const bool timing = getenv("PG_SYEVD_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const double t = now();
        fprintf(stderr, "[pg_syevd_dev 2-stage n=%d] %-16s %8.1f ms\n", n, what, (t - t_prev) * 1e3);
        t_prev = t;
    };
    Sb2Work sw;
    StedcWork wk;
    double *A = nullptr, *dd = nullptr, *de = nullptr, *dev_ev = nullptr;
    int rc = PG_OK;
    DevArena arena;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(st);
        g_arena = &arena;
        for (double *p : {A, dd, de, dev_ev}) dev_free(p);
        sb2_free(sw);
        stedc_free(wk);
        g_arena = nullptr;
        if (arena.base && arena.base != static_cast<char *>(ctx->arena)) (void)hipFree(arena.base);      // (the context's arena stays)
        arena = DevArena{};
    };
    {   // everything below out of one allocation (when the device refuses it, the buffers are allocated one by one as before)
        const size_t nn = (size_t)n * n;
        arena.cap = sb2_bytes(n) + 8 * (2 * nn + (size_t)n * (DC_LEAF + 1) + ((size_t)n / UV_ROWS + 2) * n + 64 * (size_t)n) + ((size_t)1 << 20);
        // kept up to 16 GiB (n <= ~14 000): on some boxes hipMalloc of the 8 GB a solve at n = 10 000 needs took 0.12 - 0.33 s, also for memory
        // this process had just freed (profiles/r04: "allocate 329 ms"); larger arenas (200 GB at n = 50 000) are returned at once
        constexpr size_t KEEP_MAX = (size_t)16 << 30;
        if (arena.cap <= KEEP_MAX) {
            // small solves take the context's arena (grown on demand, freed with the context)
            if (ctx->arena_bytes < arena.cap) {
                if (ctx->arena) { (void)hipStreamSynchronize(st); (void)hipFree(ctx->arena); ctx->arena = nullptr; ctx->arena_bytes = 0; }
                if (hipMalloc(&ctx->arena, arena.cap) == hipSuccess) ctx->arena_bytes = arena.cap;
                else { (void)hipGetLastError(); ctx->arena = nullptr; }
            }
            arena.base = static_cast<char *>(ctx->arena);
            if (arena.base) arena.cap = ctx->arena_bytes; else arena = DevArena{};
        } else if (hipMalloc(reinterpret_cast<void **>(&arena.base), arena.cap) != hipSuccess) { (void)hipGetLastError(); arena = DevArena{}; }
    }
    g_arena = &arena;
    rc = alloc_d(&A, (size_t)n * n);
    if (!rc) rc = alloc_d(&dd, (size_t)n);
    if (!rc) rc = alloc_d(&de, (size_t)n);
    if (!rc) rc = alloc_d(&dev_ev, (size_t)n);
    if (!rc) rc = sb2_alloc(n, sw);
    if (!rc) {
        // three of the divide & conquer's four n x n buffers are buffers of the reduction that are idle by then: A (done with once the
        // band is extracted), the reflectors of stage 2 (done with once their blocks are built) and the transposed-panel space of stage 1
        // — 10 n^2 doubles in all instead of 13 (hipMalloc of tens of GB is not free: 2.6 s for 94 GB at n = 30 000)
        wk.borrowed = true;
        wk.Tp = A; wk.Um = sw.VV; wk.Qa = sw.Wws;
        rc = stedc_alloc(n, wk);
    }
    g_arena = nullptr;
    if (rc) {
        cleanup();
        // not enough device memory for this path's 10 n^2 doubles: the one-stage path needs about half
        if (rc == PG_ENOMEM) { (void)hipGetLastError(); return PG_RETRY_ONESTAGE; }
        return rc;
    }
    mark("allocate");
    sym_from_lower_kernel<<<(unsigned)(((size_t)n * n + 255) / 256), 256, 0, st>>>(n, n, K, A);
    rc = sy2sb_device(ctx, n, A, sw);
    mark("dense->band");
    std::vector<double> hd(n), he(n, 0.0), ev;
    int hfail[4] = {0, 0, 0, 0};
    for (int attempt = 0; attempt < 2 && !rc; attempt++) {
        // second attempt: the stationary bulge-chasing kernel needs every workgroup resident at once; should a wait of its have expired
        // (CUs masked or taken by someone else), the band is still in A and the kernel that carries the rows through memory takes over
        bool stationary = false;
        rc = sb2st_device(ctx, n, A, dd, de, sw, attempt == 0, &stationary);
        if (rc) break;
        if (hipMemcpyAsync(hd.data(), dd, (size_t)n * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(he.data(), de, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipMemcpyAsync(hfail, sw.fail, sizeof(hfail), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { set_error("pg_syevd_dev: reading T back failed: %s", hipGetErrorString(hipGetLastError())); rc = PG_EHIP; break; }
        if (!(hfail[1] && !hfail[0] && stationary)) break;
        if (timing) fprintf(stderr, "[pg_syevd_dev 2-stage n=%d] a wait of the stationary bulge chasing expired (sweep %d, block %d): again with the rows through memory\n", n, hfail[2], hfail[3]);
        if (hipMemsetAsync(sw.fail + 1, 0, 3 * sizeof(int), st) != hipSuccess) { rc = PG_EHIP; break; }
    }
    mark("band->tridiag");
    if (!rc && (hfail[0] || hfail[1])) {
        if (timing) fprintf(stderr, "[pg_syevd_dev 2-stage n=%d] flags %d %d: falling back to the one-stage reduction\n", n, hfail[0], hfail[1]);
        cleanup();
        return PG_RETRY_ONESTAGE;
    }
    // the reflector blocks of the second back-transformation need only the reflectors: enqueued here, they are built while the divide
    // & conquer's host part solves the leaves (2.4 ms at n = 10 000 with the chip idle; the blocks take 2.2).  (A second stream beside the
    // divide & conquer did the same for a slower version of that kernel; creating and destroying it cost ~2 ms per solve.)
    bool prepared = false;
    if (!rc) { rc = bt2_prep_device(ctx, n, sw, st); prepared = (rc == PG_OK); }
    double *Z = nullptr;
    if (!rc) rc = stedc_device(ctx, n, hd.data(), he.data(), ev, wk, &Z);
    mark("divide&conquer");
    if (!rc) rc = bt2_device(ctx, n, Z, sw, prepared);
    mark("back-transform 2");
    if (!rc) rc = bt1_device(ctx, n, Z, sw);
    mark("back-transform 1");
    if (!rc) {
        hipError_t e1 = hipMemcpyAsync(dev_ev, ev.data(), (size_t)n * 8, hipMemcpyHostToDevice, st);
        if (e1 != hipSuccess) rc = PG_EHIP;
        if (!rc) {
            finalize_kernel<<<(unsigned)(((size_t)n * n + 255) / 256), 256, 0, st>>>(n, n, Z, dev_ev, U, evals, U64);
            if (evals64 && hipMemcpyAsync(evals64, dev_ev, (size_t)n * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = PG_EHIP;
            if (hipGetLastError() != hipSuccess) rc = PG_EHIP;
        }
        if (rc == PG_EHIP) set_error("pg_syevd_dev: output stage failed");
    }
    mark("outputs");
    cleanup();
    mark("free");
    return rc;
}

#ifndef PG_SYEVD2_MIN_N
#define PG_SYEVD2_MIN_N 768      // profiles/r03_other_sizes.txt: the two-stage path is ahead from here on, first call of a process included
                                 // (768: 15 vs 20 ms, first call 17 vs 24; 1 940: 32 vs 51; 10 000: 299 vs 561)
#endif
#ifndef PG_SYEVD2_MAX_N
#define PG_SYEVD2_MAX_N 65535    // no limit of its own: work space 10 n^2 doubles (200 GB at n = 50 000); when the device cannot give that, the one-stage path is taken
#endif
extern "C" int pg_syevd_dev(pg_ctx *ctx, int64_t n64, const float *K, float *evals, float *U, double *evals64, double *U64)
{
    PG_REQUIRE(ctx && K && (evals || evals64), "pg_syevd_dev: NULL argument");
    PG_REQUIRE(n64 >= 1 && n64 <= 65535, "pg_syevd_dev: n=%lld out of range", (long long)n64);
    PG_HIP(hipSetDevice(ctx->device));
    const int n0 = (int)n64;
    // PG_SYEVD_STAGES=1 | 2 forces a path (tests, A/B timing); by default the size decides
    const char *force = getenv("PG_SYEVD_STAGES");
    bool two = n0 >= PG_SYEVD2_MIN_N && n0 <= PG_SYEVD2_MAX_N;
    if (force && force[0] == '1') two = false;
    if (force && force[0] == '2') two = n0 >= 3 * SB_B;
    if (two) {
        const int rc = syevd_twostage(ctx, n0, K, evals, U, evals64, U64);
        if (rc != PG_RETRY_ONESTAGE) return rc;
    }
    return syevd_onestage(ctx, n0, K, evals, U, evals64, U64);
}

// ---- test hooks of the two-stage pieces -------------------------------------------------------------------------------------------
// stage 1 alone: K (n x n float32, lower triangle read) -> Aband (n x n fp64: the band |i - j| <= 64 of the result is meaningful);
// if Z (n x n fp64) is given it is replaced by Q1 Z.  flags[4] (host): [0] panel factorisation flag.
extern "C" int pgx_sb2_stage1_dev(pg_ctx *ctx, int64_t n64, const float *K, double *Aband, double *Z, int *flags)
{
    PG_REQUIRE(ctx && K && Aband && flags && n64 >= 3 * SB_B && n64 <= 65535, "pgx_sb2_stage1_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    const int n = (int)n64;
    Sb2Work sw;
    int rc = sb2_alloc(n, sw);
    if (rc) return rc;
    sym_from_lower_kernel<<<(unsigned)(((size_t)n * n + 255) / 256), 256, 0, ctx->stream>>>(n, n, K, Aband);
    rc = sy2sb_device(ctx, n, Aband, sw);
    if (!rc && Z) rc = bt1_device(ctx, n, Z, sw);
    if (!rc && hipMemcpyAsync(flags, sw.fail, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PG_EHIP;
    (void)hipStreamSynchronize(ctx->stream);
    sb2_free(sw);
    return rc;
}

// stage 2 alone: Aband (n x n fp64, only its band |i - j| <= 64, lower part, is read) -> d (n), e (n - 1) on the device;
// if Z is given it is replaced by Q2 Z.
extern "C" int pgx_sb2_stage2_dev(pg_ctx *ctx, int64_t n64, const double *Aband, double *d, double *e, double *Z, int *flags)
{
    PG_REQUIRE(ctx && Aband && d && e && flags && n64 >= 3 && n64 <= 65535, "pgx_sb2_stage2_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    const int n = (int)n64;
    Sb2Work sw;
    int rc = sb2_alloc(n, sw);
    if (rc) return rc;
    if (hipMemsetAsync(sw.fail, 0, 4 * sizeof(int), ctx->stream) != hipSuccess) rc = PG_EHIP;
    if (!rc) rc = sb2st_device(ctx, n, Aband, d, e, sw);
    if (!rc && Z) rc = bt2_device(ctx, n, Z, sw);
    if (!rc && hipMemcpyAsync(flags, sw.fail, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = PG_EHIP;
    (void)hipStreamSynchronize(ctx->stream);
    sb2_free(sw);
    return rc;
}

// debugging aid: p = host-mapped memory (pg_host_alloc) of >= 4 ints, or NULL; the bulge-chasing kernel then records (sweep, step, phase)
extern "C" int pgx_sb2_set_debug(void *p) { sb2_set_debug((int *)p); return PG_OK; }
