// dgemm.hpp — fp64 MFMA GEMM (v_mfma_f64_16x16x4_f64) used by the eigensolver: the syr2k trailing update
// of the Householder tridiagonalisation, the divide-and-conquer merges and the back-transformation.
//
//   C (MxN, row-major, ldc) = alpha * op(A) * B + beta * C
//   TA = false : A is M x K row-major (lda)            TA = true : A is stored K x M row-major (lda)
//   B is K x N row-major (ldb)
//
// 128x128x8 tile per 256-thread workgroup (4 waves as 2x2, each 4x4 MFMA tiles of 16x16), LDS tiles kept
// K-major ([k][m], [k][n]) so a wave's operand read is 16 consecutive doubles per k-row; rows padded by 16
// doubles so the four k-rows a ds_read_b64 touches fall on disjoint bank halves.  f64 MFMA lane maps:
// A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], C/D col = l&15, row = (l>>4) + 4*reg.
#pragma once
#include "common.hpp"

#include <algorithm>

namespace pg {

typedef double doublex4 __attribute__((ext_vector_type(4)));

#ifndef PG_DBK
#define PG_DBK 8
#endif
constexpr int DBM = 128, DBN = 128, DBK = PG_DBK, DPAD = 16, DPASS = DBK / 8;   // staging works in passes of 8 k-rows

struct DgemmParams {
    long long M, N, K, lda, ldb, ldc;
    const double *A, *B;
    double *C;
    double alpha, beta;
    int vecA, vecB;   // operand base and leading dimension allow 16-byte loads
    int lower;        // 1: C is symmetric and only its lower triangle is needed: tiles strictly above the diagonal are skipped
    int ksplit;       // > 1: blockIdx.y owns a K range and writes a partial MxN slab to ws (summed by splitk_reduce_kernel)
    long long kchunk;
    double *ws;
};

template <bool TA>
__global__ __launch_bounds__(256, 2) void dgemm_kernel(DgemmParams gp)
{
    __shared__ double As[2][DBK][DBM + DPAD];
    __shared__ double Bs[2][DBK][DBN + DPAD];
    const int tiles_n = (int)((gp.N + DBN - 1) / DBN);
    const long long m0 = (long long)(blockIdx.x / tiles_n) * DBM, n0 = (long long)(blockIdx.x % tiles_n) * DBN;
    if (gp.lower && n0 >= m0 + DBM) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;

    doublex4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.0;

    const long long kbeg_ = (gp.ksplit > 1) ? (long long)blockIdx.y * gp.kchunk : 0;
    const long long KEND = (gp.ksplit > 1) ? ((kbeg_ + gp.kchunk < gp.K) ? kbeg_ + gp.kchunk : gp.K) : gp.K;
    double ra[DPASS][4], rb[DPASS][4];
    // interior tiles with 16-byte-aligned operands take two double2 loads per operand; edges go element-wise
    const bool fastB = gp.vecB && (n0 + DBN <= gp.N);
    const bool fastA = gp.vecA && (m0 + DBM <= gp.M);
    auto load4d = [&](const double *p, double (&r)[4]) {
        const double2 u = *reinterpret_cast<const double2 *>(p), w = *reinterpret_cast<const double2 *>(p + 2);
        r[0] = u.x; r[1] = u.y; r[2] = w.x; r[3] = w.y;
    };
    auto gload = [&](long long kbase) {
#pragma unroll
        for (int ps = 0; ps < DPASS; ps++) {
            const long long k0 = kbase + 8 * ps;
            // B tile: 8 rows (k) x 128 cols: thread -> row tid/32, 4 consecutive cols
            {
                const long long kr = k0 + (tid >> 5), col = n0 + (tid & 31) * 4;
                if (fastB && kr < KEND) load4d(gp.B + kr * gp.ldb + col, rb[ps]);
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) rb[ps][q] = (kr < KEND && col + q < gp.N) ? gp.B[kr * gp.ldb + col + q] : 0.0;
                }
            }
            if (TA) {
                const long long kr = k0 + (tid >> 5), col = m0 + (tid & 31) * 4;
                if (fastA && kr < KEND) load4d(gp.A + kr * gp.lda + col, ra[ps]);
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) ra[ps][q] = (kr < KEND && col + q < gp.M) ? gp.A[kr * gp.lda + col + q] : 0.0;
                }
            } else {
                // A tile: 128 rows (m) x 8 (k): thread -> row tid/2, 4 consecutive k
                const long long row = m0 + (tid >> 1), kc = k0 + (tid & 1) * 4;
                if (fastA && kc + 3 < KEND) load4d(gp.A + row * gp.lda + kc, ra[ps]);
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++) ra[ps][q] = (row < gp.M && kc + q < KEND) ? gp.A[row * gp.lda + kc + q] : 0.0;
                }
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < DPASS; ps++) {
#pragma unroll
            for (int q = 0; q < 4; q++) Bs[buf][8 * ps + (tid >> 5)][(tid & 31) * 4 + q] = rb[ps][q];
            if (TA) {
#pragma unroll
                for (int q = 0; q < 4; q++) As[buf][8 * ps + (tid >> 5)][(tid & 31) * 4 + q] = ra[ps][q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) As[buf][8 * ps + (tid & 1) * 4 + q][tid >> 1] = ra[ps][q];
            }
        }
    };

    const long long kbeg = (gp.ksplit > 1) ? (long long)blockIdx.y * gp.kchunk : 0;
    const long long kend = (gp.ksplit > 1) ? ((kbeg + gp.kchunk < gp.K) ? kbeg + gp.kchunk : gp.K) : gp.K;
    const int KT = (int)((kend - kbeg + DBK - 1) / DBK);
    gload(kbeg);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < KT) gload(kbeg + (long long)(kt + 1) * DBK);
#pragma unroll
        for (int kk = 0; kk < DBK; kk += 4) {
            const int kr = kk + (lane >> 4);
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = As[buf][kr][wm * 64 + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = Bs[buf][kr][wn * 64 + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const long long col = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const long long row = m0 + wm * 64 + i * 16 + (lane >> 4) + 4 * e;
                if (row < gp.M && col < gp.N) {
                    if (gp.ksplit > 1) {
                        gp.ws[((size_t)blockIdx.y * gp.M + row) * gp.N + col] = acc[i][j][e];
                    } else {
                        double *c = gp.C + row * gp.ldc + col;
                        double v = gp.alpha * acc[i][j][e];
                        if (gp.beta != 0.0) v += gp.beta * (*c);
                        *c = v;
                    }
                }
            }
        }
}

__global__ void splitk_reduce_kernel(long long M, long long N, int ksplit, const double *ws, double alpha, double beta, double *C, long long ldc)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    double s = 0.0;
    for (int z = 0; z < ksplit; z++) s += ws[(size_t)z * M * N + idx];   // fixed order: deterministic
    double *c = C + (idx / N) * ldc + (idx % N);
    double v = alpha * s;
    if (beta != 0.0) v += beta * (*c);
    *c = v;
}

inline int dgemm(pg_ctx *ctx, bool transA, long long M, long long N, long long K, double alpha, const double *A, long long lda,
                 const double *B, long long ldb, double beta, double *C, long long ldc, bool lower_only = false)
{
    if (M <= 0 || N <= 0) return PG_OK;
    DgemmParams gp{M, N, K, lda, ldb, ldc, A, B, C, alpha, beta, 0, 0, lower_only ? 1 : 0, 1, K, nullptr};
    gp.vecA = ((uintptr_t)A % 16 == 0) && (lda % 2 == 0);
    gp.vecB = ((uintptr_t)B % 16 == 0) && (ldb % 2 == 0);
    const long long tiles = ((M + DBM - 1) / DBM) * ((N + DBN - 1) / DBN);
    // skinny output with a long K (V'V, V'Z of the back-transformation): too few tiles to fill 256 CUs -> split K
    int ksplit = 1;
    if (tiles < 256 && K >= 1024) {
        ksplit = (int)std::min<long long>((768 + tiles - 1) / tiles, K / 256);
        if (ksplit < 2) ksplit = 1;
    }
    if (ksplit > 1) {
        long long kchunk = (K + ksplit - 1) / ksplit;
        kchunk = (kchunk + DBK - 1) / DBK * DBK;            // keep 16-byte alignment of the K offset
        ksplit = (int)((K + kchunk - 1) / kchunk);
        int rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, (size_t)ksplit * M * N * sizeof(double));
        if (rc) return rc;
        gp.ksplit = ksplit; gp.kchunk = kchunk; gp.ws = (double *)ctx->scratch;
    }
    dim3 grid((unsigned)tiles, (unsigned)ksplit);
    if (transA) dgemm_kernel<true><<<grid, 256, 0, ctx->stream>>>(gp);
    else dgemm_kernel<false><<<grid, 256, 0, ctx->stream>>>(gp);
    PG_HIP(hipGetLastError());
    if (ksplit > 1) {
        splitk_reduce_kernel<<<(unsigned)((M * N + 255) / 256), 256, 0, ctx->stream>>>(M, N, ksplit, gp.ws, alpha, beta, C, ldc);
        PG_HIP(hipGetLastError());
    }
    return PG_OK;
}

}  // namespace pg
