// dgemm.hpp — fp64 MFMA GEMM (v_mfma_f64_16x16x4_f64) used by the eigensolver: the rank-2b trailing updates of the band
// reduction and of the Householder tridiagonalisation, the divide-and-conquer merges and the back-transformations.
//
//   C (MxN, row-major, ldc) = alpha * op(A) * op(B) + beta * C
//   TA = false : A is M x K row-major (lda)            TA = true : A is stored K x M row-major (lda)
//   TB = false : B is K x N row-major (ldb)            TB = true : B is stored N x K row-major (ldb)
//   kxorB      : B's k index is XOR-ed with kxorB (a multiple of 8): the operand [V W] read as [W V]
//   batch      : blockIdx.z = batch index z; operands advance by strideA/B/C elements per z; the LAST batch element may
//                have fewer rows (M_last) and a shorter K (K_last) — the clipped bottom block of a wavefront of reflector blocks
//
//   symA       : (TA = false) A is symmetric and only its LOWER triangle (plus the full 128 x 128 diagonal tiles) holds valid data:
//                k-chunks right of a row tile's diagonal tile are read transposed, A[row][k] = A[k][row]
//
// (32 WMI)x(32 WNI)x8 tile per 256-thread workgroup (4 waves as 2x2, each WMI x WNI MFMA tiles of 16x16; 4: 128 rows / columns,
// 2: 64 for outputs with M <= 64 / N <= 64), LDS tiles kept K-major ([k][m], [k][n]) so a wave's operand read is 16 consecutive doubles
// per k-row; rows padded by 16 doubles so the four k-rows a ds_read_b64 touches fall on disjoint bank halves.  f64 MFMA lane maps:
// A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], C/D col = l&15, row = (l>>4) + 4*reg.
#pragma once
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace pg {

typedef double doublex4 __attribute__((ext_vector_type(4)));
typedef double doublex2 __attribute__((ext_vector_type(2)));

#ifndef PG_DBK
#define PG_DBK 8
#endif
constexpr int DBK = PG_DBK, DPAD = 16, DPASS = DBK / 8;   // staging works in passes of 8 k-rows

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
    // extensions (all zero for a plain call)
    int kxorB;                              // TB only
    int nbatch;                             // >= 1
    long long strideA, strideB, strideC;    // elements per batch index
    long long M_last, K_last;               // dimensions of batch element nbatch - 1 (0: same as M / K)
    int symA;                               // TA = false only; needs 128-row tiles (WMI = 4) and lda = row stride of the symmetric matrix
    int vecC;                               // C base, ldc and strideC allow 16-byte accesses
    long long *stamps;                      // diagnostics: in-kernel time stamps of one workgroup (nullptr: none)
    int tune;                               // diagnostics (PG_DGEMM_TUNE): bit 3 in-kernel stamps, bit 5 one workgroup per CU, bit 7 DMA issued in one block
};

template <bool TA, bool TB, int WMI, int WNI, bool SYM>
__global__ __launch_bounds__(256, 2) void dgemm_kernel(DgemmParams gp)
{
    constexpr int DBM = 32 * WMI, DBN = 32 * WNI;
    __shared__ struct __attribute__((aligned(16))) { double As[2][DBK][DBM + DPAD]; double Bs[2][DBK][DBN + DPAD]; } sm;   // one block: the staged epilogue reuses it
    auto &As = sm.As;
    auto &Bs = sm.Bs;
    const int z = blockIdx.z;
    const bool lastz = (z == gp.nbatch - 1);
    const long long M = (lastz && gp.M_last > 0) ? gp.M_last : gp.M;
    const long long Kfull = (lastz && gp.K_last > 0) ? gp.K_last : gp.K;
    const double *Ap = gp.A + (long long)z * gp.strideA, *Bp = gp.B + (long long)z * gp.strideB;
    double *Cp = gp.C + (long long)z * gp.strideC;
    const int tiles_n = (int)((gp.N + DBN - 1) / DBN);
    const long long m0 = (long long)(blockIdx.x / tiles_n) * DBM, n0 = (long long)(blockIdx.x % tiles_n) * DBN;
    if (m0 >= M) return;
    if (gp.lower && n0 >= m0 + DBM) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;

    doublex4 acc[WMI][WNI];
#pragma unroll
    for (int i = 0; i < WMI; i++)
#pragma unroll
        for (int j = 0; j < WNI; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.0;

    const long long kbeg = (gp.ksplit > 1) ? (long long)blockIdx.y * gp.kchunk : 0;
    const long long KEND = (gp.ksplit > 1) ? ((kbeg + gp.kchunk < Kfull) ? kbeg + gp.kchunk : Kfull) : Kfull;
    double ra[DPASS][4], rb[DPASS][4];
    // interior tiles with 16-byte-aligned operands take two double2 loads per operand; edges go element-wise
    const bool fastB = gp.vecB && (n0 + DBN <= gp.N);
    const bool fastA = gp.vecA && (m0 + DBM <= M);
    auto load4d = [&](const double *p, double (&r)[4]) {
        const double2 u = *reinterpret_cast<const double2 *>(p), w = *reinterpret_cast<const double2 *>(p + 2);
        r[0] = u.x; r[1] = u.y; r[2] = w.x; r[3] = w.y;
    };
    auto gload = [&](long long kbase) {
#pragma unroll
        for (int ps = 0; ps < DPASS; ps++) {
            const long long k0 = kbase + 8 * ps;
            if (TB) {
                // B tile: 128 rows (n) x 8 (k): thread -> row tid/2, 4 consecutive k (k index XOR-ed as a block of 4)
                const long long row = n0 + (tid >> 1), kc = k0 + (tid & 1) * 4, kp = kc ^ (long long)gp.kxorB;
                if (DBN == 128 || (tid >> 1) < DBN) {
                    if (fastB && kc + 3 < KEND) load4d(Bp + row * gp.ldb + kp, rb[ps]);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; q++) rb[ps][q] = (row < gp.N && kc + q < KEND) ? Bp[row * gp.ldb + kp + q] : 0.0;
                    }
                }
            } else {
                // B tile: 8 rows (k) x 128 cols: thread -> row tid/32, 4 consecutive cols
                const long long kr = k0 + (tid >> 5), col = n0 + (tid & 31) * 4, kp = kr ^ (long long)gp.kxorB;
                if (DBN == 128 || (tid & 31) * 4 < DBN) {
                    if (fastB && kr < KEND) load4d(Bp + kp * gp.ldb + col, rb[ps]);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; q++) rb[ps][q] = (kr < KEND && col + q < gp.N) ? Bp[kp * gp.ldb + col + q] : 0.0;
                    }
                }
            }
            if (TA) {
                const long long kr = k0 + (tid >> 5), col = m0 + (tid & 31) * 4;
                if (DBM == 128 || (tid & 31) * 4 < DBM) {
                    if (fastA && kr < KEND) load4d(Ap + kr * gp.lda + col, ra[ps]);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; q++) ra[ps][q] = (kr < KEND && col + q < M) ? Ap[kr * gp.lda + col + q] : 0.0;
                    }
                }
            } else if (SYM && k0 >= m0 + DBM) {
                // symmetric A, k-chunk right of the diagonal tile: A[row][k] = A[k][row], read like a transposed operand
                const long long kr = k0 + (tid >> 5), col = m0 + (tid & 31) * 4;
                if ((tid & 31) * 4 < DBM) {
                    if (fastA && kr < KEND) load4d(Ap + kr * gp.lda + col, ra[ps]);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; q++) ra[ps][q] = (kr < KEND && col + q < M) ? Ap[kr * gp.lda + col + q] : 0.0;
                    }
                }
            } else {
                // A tile: DBM rows (m) x 8 (k): thread -> row tid/2, 4 consecutive k
                const long long row = m0 + (tid >> 1), kc = k0 + (tid & 1) * 4;
                if (DBM == 128 || (tid >> 1) < DBM) {
                    if (fastA && kc + 3 < KEND) load4d(Ap + row * gp.lda + kc, ra[ps]);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; q++) ra[ps][q] = (row < M && kc + q < KEND) ? Ap[row * gp.lda + kc + q] : 0.0;
                    }
                }
            }
        }
    };
    auto lstore = [&](int buf, long long kbase) {
#pragma unroll
        for (int ps = 0; ps < DPASS; ps++) {
            if (TB) {
                if (DBN == 128 || (tid >> 1) < DBN) {
#pragma unroll
                    for (int q = 0; q < 4; q++) Bs[buf][8 * ps + (tid & 1) * 4 + q][tid >> 1] = rb[ps][q];
                }
            } else {
                if (DBN == 128 || (tid & 31) * 4 < DBN) {
#pragma unroll
                    for (int q = 0; q < 4; q++) Bs[buf][8 * ps + (tid >> 5)][(tid & 31) * 4 + q] = rb[ps][q];
                }
            }
            if (TA || (SYM && kbase + 8 * ps >= m0 + DBM)) {
                if (DBM == 128 || (tid & 31) * 4 < DBM) {
#pragma unroll
                    for (int q = 0; q < 4; q++) As[buf][8 * ps + (tid >> 5)][(tid & 31) * 4 + q] = ra[ps][q];
                }
            } else {
                if (DBM == 128 || (tid >> 1) < DBM) {
#pragma unroll
                    for (int q = 0; q < 4; q++) As[buf][8 * ps + (tid & 1) * 4 + q][tid >> 1] = ra[ps][q];
                }
            }
        }
    };

    const int KT = (int)((KEND - kbeg + DBK - 1) / DBK);
    if (KT > 0) {
        gload(kbeg);
        lstore(0, kbeg);
    }
    __syncthreads();
    for (int kt = 0; kt < KT; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < KT) gload(kbeg + (long long)(kt + 1) * DBK);
#pragma unroll
        for (int kk = 0; kk < DBK; kk += 4) {
            const int kr = kk + (lane >> 4);
            double a[WMI], b[WNI];
#pragma unroll
            for (int i = 0; i < WMI; i++) a[i] = As[buf][kr][wm * (16 * WMI) + i * 16 + (lane & 15)];
#pragma unroll
            for (int j = 0; j < WNI; j++) b[j] = Bs[buf][kr][wn * (16 * WNI) + j * 16 + (lane & 15)];
#pragma unroll
            for (int i = 0; i < WMI; i++)
#pragma unroll
                for (int j = 0; j < WNI; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(buf ^ 1, kbeg + (long long)(kt + 1) * DBK);
        __syncthreads();
    }
    // ---- staged epilogue (full 128 x 128 tiles): C moves in whole 1 KB rows, 16 bytes per lane, through LDS — 32 rows at a time in the
    // operands' space — and meets the accumulators there.  Lane-wise 8-byte accesses in the accumulator layout (row = (lane >> 4) + 4 e)
    // made a beta = 1 update of depth 128 cost as much again as the product itself (8192 x 8192 x 128: 0.66 ms against 0.39 with beta = 0).
    if (WMI == 4 && WNI == 4 && gp.ksplit <= 1 && gp.vecC && m0 + DBM <= M && n0 + DBN <= gp.N) {
        constexpr int SP = DBN + DPAD;                    // pitch of a staged row: 144 doubles (32 rows = the 36 864 bytes of As + Bs)
        double *stage = reinterpret_cast<double *>(&sm);
        const bool rd = gp.beta != 0.0;
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
            double *Cg = Cp + (m0 + 32 * ch) * gp.ldc + n0;
            if (rd) {
                double2 cv[8];
#pragma unroll
                for (int ps = 0; ps < 8; ps++) cv[ps] = *reinterpret_cast<const double2 *>(Cg + (long long)(4 * ps + wave) * gp.ldc + 2 * lane);
#pragma unroll
                for (int ps = 0; ps < 8; ps++) *reinterpret_cast<double2 *>(stage + (4 * ps + wave) * SP + 2 * lane) = cv[ps];
                __syncthreads();
            }
            if (wm == (ch >> 1)) {                        // this chunk's rows belong to the waves of tile row ch / 2: MFMA tiles 2 (ch & 1), + 1
#pragma unroll
                for (int ii = 0; ii < 2; ii++)
#pragma unroll
                    for (int j = 0; j < WNI; j++)
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            double *sp = stage + (ii * 16 + (lane >> 4) + 4 * e) * SP + wn * 64 + j * 16 + (lane & 15);
                            const double a0 = (ch & 1) ? acc[2 + ii][j][e] : acc[ii][j][e];
                            double v = gp.alpha * a0;
                            if (rd) v += gp.beta * (*sp);
                            *sp = v;
                        }
            }
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < 8; ps++)
                *reinterpret_cast<double2 *>(Cg + (long long)(4 * ps + wave) * gp.ldc + 2 * lane) = *reinterpret_cast<const double2 *>(stage + (4 * ps + wave) * SP + 2 * lane);
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < WMI; i++)
#pragma unroll
        for (int j = 0; j < WNI; j++) {
            const long long col = n0 + wn * (16 * WNI) + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const long long row = m0 + wm * (16 * WMI) + i * 16 + (lane >> 4) + 4 * e;
                if (row < M && col < gp.N) {
                    if (gp.ksplit > 1) {
                        gp.ws[((size_t)blockIdx.y * gp.M + row) * gp.N + col] = acc[i][j][e];
                    } else {
                        double *c = Cp + row * gp.ldc + col;
                        double v = gp.alpha * acc[i][j][e];
                        if (gp.beta != 0.0) v += gp.beta * (*c);
                        *c = v;
                    }
                }
            }
        }
}

// ==== LDS-DMA ring kernel =========================================================================================================
// The same 128 x (32 WNI) tile and MFMA arrangement, but the operands reach LDS by global_load_lds_dwordx4 (no staging registers, no
// ds_write pass) into a ring of RS = 4 slots of RBK = 8 k-rows: the DMA of chunk t + 3 is issued while chunk t is multiplied, so
// three chunks (~6 000 cycles of fp64 MFMA) cover the memory latency — the register-staged kernel above prefetches ONE chunk (~2 000
// cycles, less than a loaded HBM round trip) and stalls its whole workgroup at every barrier.  One raw s_barrier per chunk, counted
// vmcnt waits (the DMA stays in flight across barriers).  B is always K x N row-major ("k-major"); A is k-major (AKM: stored K x M)
// or m-major (stored M x K); SYM: A symmetric from its lower triangle — chunks left of / on the diagonal tile are read m-major,
// chunks right of it k-major from the transposed position.
// LDS images (all conflict-free for the ds_read_b64 fragment reads, lane = (l & 15) element x (l >> 4) k-row):
//   k-major operand: row k of the chunk = 1 KB (128 doubles; 512 B for a 64-wide B), 16-byte chunk c of the row holds global chunk
//                    c ^ 8 (k & 1): odd k-rows are rotated by 128 bytes, so the two k-rows one lane group reads sit on different bank halves
//   m-major operand: row m = 64 bytes (8 k), 16-byte chunk c holds global k-pair c ^ ((m >> 2) & 3)
// The swizzles are applied on the GLOBAL address of each lane's 16 bytes (the LDS side of an LDS-DMA is lane-linear).
// The C tile rides the same ring: after the last operand chunk the next virtual chunks are the tile's 8 groups of 16 rows (16 KB = one
// slot), prefetched three iterations ahead while the last products run; each group is updated in LDS from the accumulators and leaves
// in whole rows, 16 bytes per lane.  (The depth-128 update of the band reduction spent as long in its epilogue as in its products.)
constexpr int RBK = 8, RS = 4;

__device__ __forceinline__ void ring_wait_vm(int n)
{
    switch (n) {       // s_waitcnt needs an immediate
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void ring_dma16(const double *src, unsigned char *dst)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

// in-kernel time stamps of ONE workgroup (PG_DGEMM_TUNE bit 3; blockIdx.x == 300): where a tile's time goes — diagnostics only
extern long long *g_ring_stamp_buf;      // device buffer of 64 stamps (syevd.hip: pgx_ring_stamps), nullptr until asked for
#define RING_STAMP(ix) do { if (stamp) gp.stamps[ix] = __builtin_amdgcn_s_memtime(); } while (0)

template <bool AKM, bool SYM, int WNI>
__global__ __launch_bounds__(256, 2) void dgemm_ring_kernel(DgemmParams gp)
{
    const bool stamp = gp.stamps != nullptr && blockIdx.x == 300 && blockIdx.y == 0 && threadIdx.x == 0;
    RING_STAMP(0);
    constexpr int DBM = 128, DBN = 32 * WNI;
    constexpr int BROW = DBN * 8;                       // bytes of one k-row of the B image (and of one row of a staged C group)
    constexpr int SLOT_A = 8192, SLOT = SLOT_A + RBK * BROW;
    constexpr int NI = (WNI == 4) ? 4 : 3;              // DMA instructions per wave and operand chunk
    constexpr int NC = (WNI == 4) ? 4 : 2;              // ... per staged C group, and global stores per wave and group
    extern __shared__ __attribute__((aligned(1024))) unsigned char ring[];      // RS slots (dynamic: the launch may add idle LDS, PG_DGEMM_TUNE bit 5)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const long long M = gp.M, N = gp.N;
    // tile of this workgroup: XCD-contiguous remap of the launch index (workgroups b and b + 8 share an XCD's L2), then row-major —
    // or, for a symmetric update, the enumeration of the lower triangle
    const int tiles_m = (int)((M + DBM - 1) / DBM), tiles_n = (int)((N + DBN - 1) / DBN);
    const int T = (gp.lower == 1) ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    int tm, tn;
    {
        const int b = blockIdx.x, q = T / 8, r = T % 8, xcd = b % 8;
        const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
        if (gp.lower == 1) {
            int t_ = (int)((sqrt(8.0 * (double)lid + 1.0) - 1.0) * 0.5);
            while ((t_ + 1) * (t_ + 2) / 2 <= lid) t_++;
            while (t_ * (t_ + 1) / 2 > lid) t_--;
            tm = t_; tn = lid - t_ * (t_ + 1) / 2;
        } else { tm = lid / tiles_n; tn = lid % tiles_n; }
    }
    const long long m0 = (long long)tm * DBM, n0 = (long long)tn * DBN;
    if (gp.lower == 2 && n0 >= m0 + DBM) return;          // uniform over the workgroup, before any barrier
    const long long kbeg = (gp.ksplit > 1) ? (long long)blockIdx.y * gp.kchunk : 0;
    const long long kend = (gp.ksplit > 1) ? ((kbeg + gp.kchunk < gp.K) ? kbeg + gp.kchunk : gp.K) : gp.K;
    const int NCH = (kend > kbeg) ? (int)((kend - kbeg) / RBK) : 0;
    const int ktail = (kend > kbeg) ? (int)((kend - kbeg) % RBK) : 0;
    const double *Ap = gp.A, *Bp = gp.B;
    double *Cp = gp.C;
    long long ldc = gp.ldc;
    double alpha = gp.alpha, beta = gp.beta;
    if (gp.ksplit > 1) { Cp = gp.ws + (size_t)blockIdx.y * gp.M * gp.N; ldc = gp.N; alpha = 1.0; beta = 0.0; }
    const bool staged = (gp.ksplit > 1 ? (gp.N % 2 == 0) : (gp.vecC != 0)) && m0 + DBM <= M && n0 + DBN <= N;
    const bool rdC = staged && beta != 0.0;
    const int NG = staged ? 8 : 0, NV = NCH + NG;

    doublex4 acc[4][WNI];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < WNI; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0.0;

    // ---- DMA sources of this lane (element offsets at chunk 0; every chunk adds its own k offset)
    const long long colA_max = (M - 1) & ~1LL, colB_max = (N - 1) & ~1LL;
    long long offA_km[2], offA_mk[2], offB[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int a = 2 * wave + u;                                  // A instruction: k-row a (k-major) / rows 16 a .. (m-major)
        long long col = m0 + 2 * (lane ^ (8 * (a & 1)));
        col = col < colA_max ? col : colA_max;
        offA_km[u] = (long long)a * gp.lda + col;
        const int r = 16 * a + (lane >> 2), c = lane & 3;
        long long row = m0 + r;
        row = row < M ? row : M - 1;
        offA_mk[u] = row * gp.lda + 2 * (c ^ ((r >> 2) & 3));
        if (WNI == 4) {
            long long cb = n0 + 2 * (lane ^ (8 * (a & 1)));
            cb = cb < colB_max ? cb : colB_max;
            offB[u] = cb;                                            // + ((k0 + a) ^ kxor) ldb
        } else {
            const int rowb = 2 * wave + (lane >> 5);                 // one instruction: k-rows 2 wave, 2 wave + 1
            long long cb = n0 + 2 * ((lane & 31) ^ (8 * (rowb & 1)));
            cb = cb < colB_max ? cb : colB_max;
            offB[u] = cb;
        }
    }
    auto chunk_km = [&](long long k0) { return SYM ? (k0 >= m0 + DBM) : AKM; };
    auto issue_ops = [&](int x) {
        const long long k0 = kbeg + (long long)x * RBK;
        unsigned char *sl = ring + (x & (RS - 1)) * SLOT;
        const bool km = chunk_km(k0);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int a = 2 * wave + u;
            const double *src = km ? Ap + k0 * gp.lda + offA_km[u] : Ap + offA_mk[u] + k0;
            ring_dma16(src, sl + a * 1024);
        }
        if (WNI == 4) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int a = 2 * wave + u;
                ring_dma16(Bp + (((k0 + a) ^ (long long)gp.kxorB)) * gp.ldb + offB[u], sl + SLOT_A + a * 1024);
            }
        } else {
            const int rowb = 2 * wave + (lane >> 5);
            ring_dma16(Bp + (((k0 + rowb) ^ (long long)gp.kxorB)) * gp.ldb + offB[0], sl + SLOT_A + wave * 1024);
        }
    };
    auto issue_c = [&](int g, int slot) {                            // rows 16 g .. 16 g + 15 of the C tile -> slot
        unsigned char *sl = ring + slot * SLOT;
        if (WNI == 4) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const int row = 4 * wave + rr;
                ring_dma16(Cp + (m0 + 16 * g + row) * ldc + n0 + 2 * (lane ^ (8 * (row & 1))), sl + row * BROW);
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int row = 4 * wave + 2 * rr + (lane >> 5);
                ring_dma16(Cp + (m0 + 16 * g + row) * ldc + n0 + 2 * ((lane & 31) ^ (8 * (row & 1))), sl + (4 * wave + 2 * rr) * BROW);
            }
        }
    };
    // one DMA instruction of virtual chunk x (operand chunk: A piece 0, 1, then B; C group: its rows): issued BETWEEN the products of the
    // chunk being multiplied — an MFMA holds the wave's issue port for 8 of its 64 cycles, so a DMA's ~120 issue cycles disappear in the
    // shadow of the products; issued in one block before them they were 480 of a chunk's 3 200 cycles (in-kernel stamps, r4)
    auto issue_piece = [&](int x, int piece) {
        if (x < NCH) {
            const long long k0 = kbeg + (long long)x * RBK;
            unsigned char *sl = ring + (x & (RS - 1)) * SLOT;
            if (piece < 2) {
                const bool km = chunk_km(k0);
                const int a = 2 * wave + piece;
                const double *src = km ? Ap + k0 * gp.lda + offA_km[piece] : Ap + offA_mk[piece] + k0;
                ring_dma16(src, sl + a * 1024);
            } else if (WNI == 4) {
                const int a = 2 * wave + (piece - 2);
                ring_dma16(Bp + (((k0 + a) ^ (long long)gp.kxorB)) * gp.ldb + offB[piece - 2], sl + SLOT_A + a * 1024);
            } else if (piece == 2) {
                const int rowb = 2 * wave + (lane >> 5);
                ring_dma16(Bp + (((k0 + rowb) ^ (long long)gp.kxorB)) * gp.ldb + offB[0], sl + SLOT_A + wave * 1024);
            }
        } else if (x < NV && rdC) {
            const int g = x - NCH;
            unsigned char *sl = ring + (x & (RS - 1)) * SLOT;
            if (WNI == 4) {
                const int row = 4 * wave + piece;
                ring_dma16(Cp + (m0 + 16 * g + row) * ldc + n0 + 2 * (lane ^ (8 * (row & 1))), sl + row * BROW);
            } else if (piece < 2) {
                const int row = 4 * wave + 2 * piece + (lane >> 5);
                ring_dma16(Cp + (m0 + 16 * g + row) * ldc + n0 + 2 * ((lane & 31) ^ (8 * (row & 1))), sl + (4 * wave + 2 * piece) * BROW);
            }
        }
    };
    auto cnt = [&](int x) { return x < NCH ? NI : ((x < NV && rdC) ? NC : 0); };
    auto issue = [&](int x) {
        if (x < NCH) issue_ops(x);
        else if (x < NV && rdC) issue_c(x - NCH, x & (RS - 1));
    };
    // products of one chunk from its slot
    auto frag_a = [&](const unsigned char *sl, bool km, int ks, int i) {
        const int k = 4 * ks + (lane >> 4), ml = wm * 64 + 16 * i + (lane & 15);
        const int off = km ? k * 1024 + ((ml * 8) ^ ((k & 1) * 128)) : ml * 64 + (((k >> 1) ^ ((ml >> 2) & 3)) * 16) + (k & 1) * 8;
        return *reinterpret_cast<const double *>(sl + off);
    };
    auto frag_b = [&](const unsigned char *sl, int ks, int j) {
        const int k = 4 * ks + (lane >> 4), nl = wn * (16 * WNI) + 16 * j + (lane & 15);
        return *reinterpret_cast<const double *>(sl + SLOT_A + k * BROW + ((nl * 8) ^ ((k & 1) * 128)));
    };
    // the 32 (16) products of one chunk from its slot; xi >= 0: the DMA instructions of virtual chunk xi go out between them
    auto mfma_chunk = [&](const unsigned char *sl, bool km, int xi) {
        double a0[4], b0[WNI], a1[4], b1[WNI];
#pragma unroll
        for (int i = 0; i < 4; i++) a0[i] = frag_a(sl, km, 0, i);
#pragma unroll
        for (int j = 0; j < WNI; j++) b0[j] = frag_b(sl, 0, j);
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < WNI; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (xi >= 0) issue_piece(xi, i);
            if (i == 1) {      // the second k-step's fragments: in flight under the remaining products of the first
#pragma unroll
                for (int i2 = 0; i2 < 4; i2++) a1[i2] = frag_a(sl, km, 1, i2);
#pragma unroll
                for (int j = 0; j < WNI; j++) b1[j] = frag_b(sl, 1, j);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < WNI; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: the first three chunks are on their way before anything else happens
    issue(0); issue(1); issue(2);
    RING_STAMP(1);
    if (ktail > 0) {
        // the last, partial chunk by hand into slot 3 (its rows past the end are zeros), multiplied first; chunk 3's DMA is issued behind
        // the barrier of iteration 0, i.e. after every wave has read this
        const long long k0 = kbeg + (long long)NCH * RBK;
        unsigned char *sl = ring + 3 * SLOT;
        const bool km = chunk_km(k0);
        for (int idx = tid; idx < RBK * DBM; idx += 256) {
            int k, ml;
            if (km) { k = idx >> 7; ml = idx & 127; } else { ml = idx >> 3; k = idx & 7; }
            double v = 0.0;
            if (k < ktail && m0 + ml < M) v = km ? Ap[(k0 + k) * gp.lda + m0 + ml] : Ap[(m0 + ml) * gp.lda + k0 + k];
            const int off = km ? k * 1024 + ((ml * 8) ^ ((k & 1) * 128)) : ml * 64 + (((k >> 1) ^ ((ml >> 2) & 3)) * 16) + (k & 1) * 8;
            *reinterpret_cast<double *>(sl + off) = v;
        }
        for (int idx = tid; idx < RBK * DBN; idx += 256) {
            const int k = idx / DBN, nl = idx % DBN;
            double v = 0.0;
            if (k < ktail && n0 + nl < N) v = Bp[((k0 + k) ^ (long long)gp.kxorB) * gp.ldb + n0 + nl];
            *reinterpret_cast<double *>(sl + SLOT_A + k * BROW + ((nl * 8) ^ ((k & 1) * 128))) = v;
        }
        __syncthreads();          // (drains the three DMAs above: once per tile)
        mfma_chunk(sl, km, -1);
    }
    // ---- main loop: one barrier per chunk
    for (int t = 0; t < NCH; t++) {
        if (t == 8) RING_STAMP(32);
        // steady state (two operand chunks younger than the one waited for, no prefetch in flight): an immediate — the computed count
        // below is a 30-way compare chain of ~400 cycles, paid per chunk
        if (t + 2 < NCH) {
            if (NI == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else ring_wait_vm(cnt(t + 1) + cnt(t + 2));
        if (t == 8) RING_STAMP(33);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t == 0) RING_STAMP(2);
        if (t == 1) RING_STAMP(3);
        if (t == 8) RING_STAMP(4);
        if (gp.tune & 128) issue(t + 3);          // A/B: the DMA in one block before the products (as until r4 call 9)
        if (t == 8) RING_STAMP(34);
        if (t == 8) RING_STAMP(35);
        mfma_chunk(ring + (t & (RS - 1)) * SLOT, chunk_km(kbeg + (long long)t * RBK), (gp.tune & 128) ? -1 : t + 3);
        if (t == 8) RING_STAMP(36);
        if (t == 9) RING_STAMP(37);
    }
    RING_STAMP(5);
    if (staged) {
        // ---- the C tile through the ring, 16 rows per step
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const int t = NCH + g;
            unsigned char *sl = ring + (t & (RS - 1)) * SLOT;
            // younger than group g's DMA: the DMAs of groups g + 1, g + 2 and the stores of groups g - 3 .. g - 1
            const int st = NC * ((g >= 3) ? 3 : g);
            ring_wait_vm(cnt(t + 1) + cnt(t + 2) + st);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            RING_STAMP(8 + 2 * g);
            issue(t + 3);
            if (wm == (g >> 2)) {
                double cv[WNI][4];
#pragma unroll
                for (int j = 0; j < WNI; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) cv[j][e] = 0.0;
                if (rdC) {
#pragma unroll
                    for (int j = 0; j < WNI; j++)
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int rl = (lane >> 4) + 4 * e, col = wn * (16 * WNI) + 16 * j + (lane & 15);
                            cv[j][e] = beta * *reinterpret_cast<const double *>(sl + rl * BROW + ((col * 8) ^ ((rl & 1) * 128)));
                        }
                }
#pragma unroll
                for (int j = 0; j < WNI; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int rl = (lane >> 4) + 4 * e, col = wn * (16 * WNI) + 16 * j + (lane & 15);
                        *reinterpret_cast<double *>(sl + rl * BROW + ((col * 8) ^ ((rl & 1) * 128))) = alpha * acc[g & 3][j][e] + cv[j][e];
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the LDS writes are done; no fence: it would drain the DMA and the stores
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (WNI == 4) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int row = 4 * wave + rr;
                    const double2 v = *reinterpret_cast<const double2 *>(sl + row * BROW + ((lane * 16) ^ ((row & 1) * 128)));
                    *reinterpret_cast<double2 *>(Cp + (m0 + 16 * g + row) * ldc + n0 + 2 * lane) = v;
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int row = 4 * wave + 2 * rr + (lane >> 5);
                    const double2 v = *reinterpret_cast<const double2 *>(sl + row * BROW + (((lane & 31) * 16) ^ ((row & 1) * 128)));
                    *reinterpret_cast<double2 *>(Cp + (m0 + 16 * g + row) * ldc + n0 + 2 * (lane & 31)) = v;
                }
            }
            RING_STAMP(9 + 2 * g);
        }
        RING_STAMP(6);
        if (stamp) gp.stamps[7] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    // ---- edge tiles: straight from the accumulators, element-wise
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < WNI; j++) {
            const long long col = n0 + wn * (16 * WNI) + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const long long row = m0 + wm * 64 + i * 16 + (lane >> 4) + 4 * e;
                if (row < M && col < N) {
                    double *c = Cp + row * ldc + col;
                    double v = alpha * acc[i][j][e];
                    if (beta != 0.0) v += beta * (*c);
                    *c = v;
                }
            }
        }
}

static __global__ void splitk_reduce_kernel(long long M, long long N, int ksplit, const double *ws, double alpha, double beta, double *C, long long ldc)
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

// full descriptor form; dgemm() below is the plain call
struct DgemmDesc {
    bool transA = false, transB = false;
    long long M = 0, N = 0, K = 0;
    double alpha = 1.0, beta = 0.0;
    const double *A = nullptr, *B = nullptr;
    double *C = nullptr;
    long long lda = 0, ldb = 0, ldc = 0;
    bool lower_only = false;
    int kxorB = 0;
    int nbatch = 1;
    long long strideA = 0, strideB = 0, strideC = 0, M_last = 0, K_last = 0;
    bool allow_splitk = true;
    bool symA = false;
    // split-K products whose caller sums the slices itself (fused with what follows): no reduction kernel is launched;
    // *partials receives {slices, slab base (slice z at base + z M N, row-major M x N)} — slices == 1: the product went to C as usual
    struct Partials { int slices = 1; const double *ws = nullptr; } *partials = nullptr;
};

inline int dgemm_ex(pg_ctx *ctx, const DgemmDesc &d)
{
    if (d.M <= 0 || d.N <= 0 || d.nbatch <= 0) return PG_OK;
    DgemmParams gp{};
    gp.M = d.M; gp.N = d.N; gp.K = d.K; gp.lda = d.lda; gp.ldb = d.ldb; gp.ldc = d.ldc;
    gp.A = d.A; gp.B = d.B; gp.C = d.C; gp.alpha = d.alpha; gp.beta = d.beta;
    gp.lower = d.lower_only ? 1 : 0; gp.ksplit = 1; gp.kchunk = d.K; gp.ws = nullptr;
    gp.kxorB = d.kxorB; gp.nbatch = d.nbatch;
    gp.strideA = d.strideA; gp.strideB = d.strideB; gp.strideC = d.strideC; gp.M_last = d.M_last; gp.K_last = d.K_last;
    gp.vecA = ((uintptr_t)d.A % 16 == 0) && (d.lda % 2 == 0) && (d.strideA % 2 == 0);
    gp.vecB = ((uintptr_t)d.B % 16 == 0) && (d.ldb % 2 == 0) && (d.strideB % 2 == 0);
    gp.symA = (d.symA && !d.transA) ? 1 : 0;
    gp.vecC = ((uintptr_t)d.C % 16 == 0) && (d.ldc % 2 == 0) && (d.strideC % 2 == 0);
    const int tune_env = getenv("PG_DGEMM_TUNE") ? atoi(getenv("PG_DGEMM_TUNE")) : 0;
    gp.tune = tune_env;
    gp.stamps = (tune_env & 8) ? g_ring_stamp_buf : nullptr;
    const bool small_m = d.M <= 64 && !d.symA, small_n = d.N <= 64 && !d.lower_only;
    const int dbm = small_m ? 64 : 128, dbn = small_n ? 64 : 128;
    const long long tiles = ((d.M + dbm - 1) / dbm) * ((d.N + dbn - 1) / dbn);
    // skinny output with a long K (V'V, V'Z of the back-transformation, the panel Grams): too few tiles to fill 256 CUs -> split K
    int ksplit = 1;
    if (d.allow_splitk && d.nbatch == 1 && tiles < 256 && d.K >= 1024) {
        // at least 128 k per slice (a 64 x 64 Gram of 7400 rows: 28 slices of 264 k 26 us, 57 of 128 k 17 us; 64 k no better, 32 worse)
        ksplit = (int)std::min<long long>((768 + tiles - 1) / tiles, d.K / 128);
        if (ksplit < 2) ksplit = 1;
        const int ks_mode = getenv("PG_DGEMM_KSPLIT_FILL") ? atoi(getenv("PG_DGEMM_KSPLIT_FILL")) : 1;
        if (ks_mode && tiles >= 16) {
            // long-K shapes with many row tiles (X = A22 V): two workgroups per CU are resident, so tiles x slices should fill whole
            // rounds of 2 x CUs workgroups; among the slice counts with >= 256 k each take the best fill, the fewest slices at a tie
            const long long slots = 2LL * ctx->num_cu;
            double best = -1.0; int best_ks = 1;
            for (int ks = 1; ks <= 16 && d.K / ks >= 256; ks++) {
                const long long wg = tiles * ks, rounds = (wg + slots - 1) / slots;
                const double fill = (double)wg / (double)(rounds * slots);
                if (fill > best + 0.03) { best = fill; best_ks = ks; }
            }
            ksplit = best_ks;
        }
    }
    if (ksplit > 1) {
        long long kchunk = (d.K + ksplit - 1) / ksplit;
        kchunk = (kchunk + DBK - 1) / DBK * DBK;            // keep 16-byte alignment of the K offset
        ksplit = (int)((d.K + kchunk - 1) / kchunk);
        int rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, (size_t)ksplit * d.M * d.N * sizeof(double));
        if (rc) return rc;
        gp.ksplit = ksplit; gp.kchunk = kchunk; gp.ws = (double *)ctx->scratch;
    }
    dim3 grid((unsigned)tiles, (unsigned)ksplit, (unsigned)d.nbatch);
    hipStream_t st = ctx->stream;
    // LDS-DMA ring kernel: 128-row tiles, B k-major, 16-byte aligned operands, M >= 2 and N >= 2 (pairs are clamped, not masked)
    const bool ring_off = getenv("PG_DGEMM_RING") && atoi(getenv("PG_DGEMM_RING")) == 0;     // A/B timing and tests (read per call: tests toggle it)
    if (!ring_off && !small_m && !d.transB && d.nbatch == 1 && gp.vecA && gp.vecB && d.M >= 2 && d.N >= 2 && d.K >= 1 && (d.kxorB % 8) == 0 &&
        (ksplit == 1 || gp.kchunk % RBK == 0)) {
        const long long tm_ = (d.M + 127) / 128, tn_ = (d.N + dbn - 1) / dbn;
        // lower triangle of a square C: the triangle's tiles are enumerated; of a rectangular one (a tile column of an update): every
        // tile of the rectangle is launched and those above the diagonal leave at once (gp.lower == 2)
        const bool tri = d.lower_only && tm_ == tn_ && dbn == 128;
        gp.lower = tri ? 1 : (d.lower_only ? 2 : 0);
        const long long T = tri ? tm_ * (tm_ + 1) / 2 : tm_ * tn_;
        dim3 rgrid((unsigned)T, (unsigned)ksplit, 1);
        constexpr int LDS4 = RS * (8192 + RBK * 1024), LDS2 = RS * (8192 + RBK * 512);
        {   // the 128-wide variants take 68 KB of dynamic LDS: a per-DEVICE function attribute, set on a device's first call
            static std::atomic<unsigned> done_mask{0};
            const unsigned bit = 1u << (ctx->device & 31);
            if (!(done_mask.load(std::memory_order_acquire) & bit)) {
                PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dgemm_ring_kernel<false, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
                PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dgemm_ring_kernel<true, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
                PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dgemm_ring_kernel<false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4));
                done_mask.fetch_or(bit, std::memory_order_release);
            }
        }
        const int xl = (gp.tune & 32) ? 40960 : 0;      // diagnostics: + 40 KB, so that one workgroup per CU is resident
        if (xl) {
            PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dgemm_ring_kernel<false, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4 + xl));
            PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dgemm_ring_kernel<true, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4 + xl));
            PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dgemm_ring_kernel<false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4 + xl));
        }
        if (gp.symA) {
            if (small_n) dgemm_ring_kernel<false, true, 2><<<rgrid, 256, LDS2, st>>>(gp);
            else dgemm_ring_kernel<false, true, 4><<<rgrid, 256, LDS4 + xl, st>>>(gp);
        } else if (d.transA) {
            if (small_n) dgemm_ring_kernel<true, false, 2><<<rgrid, 256, LDS2, st>>>(gp);
            else dgemm_ring_kernel<true, false, 4><<<rgrid, 256, LDS4 + xl, st>>>(gp);
        } else {
            if (small_n) dgemm_ring_kernel<false, false, 2><<<rgrid, 256, LDS2, st>>>(gp);
            else dgemm_ring_kernel<false, false, 4><<<rgrid, 256, LDS4 + xl, st>>>(gp);
        }
        PG_HIP(hipGetLastError());
        if (d.partials) { d.partials->slices = ksplit; d.partials->ws = (ksplit > 1) ? gp.ws : nullptr; }
        if (ksplit > 1 && !d.partials) {
            splitk_reduce_kernel<<<(unsigned)((d.M * d.N + 255) / 256), 256, 0, st>>>(d.M, d.N, ksplit, gp.ws, d.alpha, d.beta, d.C, d.ldc);
            PG_HIP(hipGetLastError());
        }
        return PG_OK;
    }
#define PG_DG_LAUNCH(TA_, TB_)                                                               \
    do {                                                                                     \
        if (small_m && small_n) dgemm_kernel<TA_, TB_, 2, 2, false><<<grid, 256, 0, st>>>(gp);      \
        else if (small_m) dgemm_kernel<TA_, TB_, 2, 4, false><<<grid, 256, 0, st>>>(gp);            \
        else if (small_n) dgemm_kernel<TA_, TB_, 4, 2, false><<<grid, 256, 0, st>>>(gp);            \
        else dgemm_kernel<TA_, TB_, 4, 4, false><<<grid, 256, 0, st>>>(gp);                         \
    } while (0)
    if (gp.symA) {
        if (small_n) dgemm_kernel<false, false, 4, 2, true><<<grid, 256, 0, st>>>(gp);
        else dgemm_kernel<false, false, 4, 4, true><<<grid, 256, 0, st>>>(gp);
    } else if (d.transA && d.transB) PG_DG_LAUNCH(true, true);
    else if (d.transA) PG_DG_LAUNCH(true, false);
    else if (d.transB) PG_DG_LAUNCH(false, true);
    else PG_DG_LAUNCH(false, false);
#undef PG_DG_LAUNCH
    PG_HIP(hipGetLastError());
    if (d.partials) { d.partials->slices = ksplit; d.partials->ws = (ksplit > 1) ? gp.ws : nullptr; }
    if (ksplit > 1 && !d.partials) {
        splitk_reduce_kernel<<<(unsigned)((d.M * d.N + 255) / 256), 256, 0, st>>>(d.M, d.N, ksplit, gp.ws, d.alpha, d.beta, d.C, d.ldc);
        PG_HIP(hipGetLastError());
    }
    return PG_OK;
}

inline int dgemm(pg_ctx *ctx, bool transA, long long M, long long N, long long K, double alpha, const double *A, long long lda,
                 const double *B, long long ldb, double beta, double *C, long long ldc, bool lower_only = false)
{
    DgemmDesc d;
    d.transA = transA; d.M = M; d.N = N; d.K = K; d.alpha = alpha; d.beta = beta;
    d.A = A; d.lda = lda; d.B = B; d.ldb = ldb; d.C = C; d.ldc = ldc; d.lower_only = lower_only;
    return dgemm_ex(ctx, d);
}

}  // namespace pg
