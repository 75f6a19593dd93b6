// rotate_geno.hip — H2 fast path for GENOTYPE columns: X <- U'X when every column of the SNP block
// takes at most three equally spaced values (hard-called genotypes 0/1/2, raw or centred/standardised — what
// lmm.pygemma is fed by every caller in the reference: experiments/*/run_*.py, tests/test_pygemma.py:184-192).
//
//   x_g = v0_g + dx_g * code_g,  code in {0,1,2}   =>   U'x_g = v0_g * (U'1) + dx_g * (U' code_g)
//
// The codes are exact in bf16, so U' code needs only U split into three bf16 planes U = U1 + U2 + U3 (8 + 8 + 8
// significant bits): every product code*U_s is exact in fp32 and the three partial GEMMs accumulate into ONE fp32
// accumulator on the bf16 MFMA pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate; 3 passes => 5.3x fewer
// matrix cycles than the fp32 path).  Error class = fp32 accumulation, the same as the fp32-MFMA kernel and as the
// reference's sgemm (lmm/lmm.py:244); the split drops < 2^-24 |U|.  U'1 is taken in fp64.
//
// Layout: both operands K-contiguous ("NT" GEMM): Gt [p][ldk] bf16 codes (SNP-major), Up [n][3*KT*GBK] bf16 with the
// three planes of each 64-sample K-tile interleaved, so the kernel is a plain GEMM over K' = 3K whose A tile index is
// kt'/3.  128x128 tile, 4 waves x (2x2) 32x32 MFMA tiles, BK = 64, LDS rows padded to 144 B (conflict-free
// ds_read_b128), register-staged double buffering.
#include "common.hpp"

namespace pg {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#ifndef PG_GBK
#define PG_GBK 64
#endif
#ifndef PG_GBLK
#define PG_GBLK 2
#endif
constexpr int GBM = 128, GBN = 128, GBK = PG_GBK, GROWB = GBK * 2 + 16;   // LDS row bytes (padded)
constexpr int GCH = GBK / 8, GRPP = 256 / GCH, GNH = 128 / GRPP;          // 16-B chunks per row, rows per staging pass, passes

__device__ __forceinline__ unsigned short f32_to_bf16_rn(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);       // round to nearest even (inputs are finite)
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// U (n x n, row stride ldU, eigenvector k in column k) -> Up[k][kt][plane][j] (bf16) and colsum[k] = sum_i U[i][k] (fp64)
__global__ __launch_bounds__(256) void split_u_kernel(long long n, long long ldU, const float *U, unsigned short *Up, long long ldp, double *colsum)
{
    __shared__ float tile[64][65];
    const long long k0 = (long long)blockIdx.x * 64, i0 = (long long)blockIdx.y * 64;   // eigen index block, sample block (= K-tile)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const long long i = i0 + r, k = k0 + tx;
        tile[r][tx] = (i < n && k < n) ? U[i * ldU + k] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {      // r: eigen index within block, tx: sample within K-tile
        const long long k = k0 + r;
        if (k >= n) continue;
        const float u = tile[tx][r];
        const unsigned short h1 = f32_to_bf16_rn(u);
        const float r1 = u - bf16_to_f32(h1);
        const unsigned short h2 = f32_to_bf16_rn(r1);
        const float r2 = r1 - bf16_to_f32(h2);
        const unsigned short h3 = f32_to_bf16_rn(r2);
        unsigned short *dst = Up + k * ldp + (long long)blockIdx.y * 3 * GBK;
        dst[tx] = h1; dst[GBK + tx] = h2; dst[2 * GBK + tx] = h3;
    }
    // column sums (fp64), one wave per 16 eigen indices, deterministic order over the 64 samples, atomics across K-tiles avoided:
    // each (k, K-tile) partial is written to colsum workspace by the caller's reduce (see launch): here accumulate via atomicAdd-free path
    if (ty == 0) {
        double s = 0.0;
        for (int r = 0; r < 64; r++) s += (double)tile[r][tx];
        if (k0 + tx < n) colsum[(long long)blockIdx.y * n + k0 + tx] = s;     // partial per K-tile; reduced by colsum_reduce_kernel
    }
}
__global__ void colsum_reduce_kernel(long long n, int kt, double *colsum)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double s = 0.0;
    for (int t = 0; t < kt; t++) s += colsum[(long long)t * n + k];
    colsum[(long long)kt * n + k] = s;   // final sums live after the partials
}

// Genotype detection + encoding in three passes, all parallel over rows and columns:
//   minmax : per column lowest / highest value (order-independent atomics on an order-preserving int key)
//   encode : code = 0 (lowest), 2 (highest), 1 (anything else: must sit at the midpoint within 8 ulp, else the block
//            is not a genotype block), transposed to SNP-major bf16 Gt [p][ldk] through a 32x32 LDS tile; pad zeroed
//   params : v0 = lowest, dx = (highest - lowest)/2  ->  x = v0 + dx*code for one-, two- and three-valued columns alike
__device__ __forceinline__ int f2key(float f) { int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

__global__ void minmax_init_kernel(long long p, int *kmin, int *kmax)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < p) { kmin[g] = 0x7FFFFFFF; kmax[g] = (int)0x80000000; }
}
__global__ __launch_bounds__(256) void minmax_geno_kernel(long long n, long long p, const float *X, long long ldX, int *kmin, int *kmax, int *flag)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long g = (long long)blockIdx.x * 64 + tx;
    if (g >= p) return;
    const long long i0 = (long long)blockIdx.y * 256;
    int lo = 0x7FFFFFFF, hi = (int)0x80000000;
    bool bad = false;
    for (long long i = i0 + ty; i < i0 + 256 && i < n; i += 4) {
        const float x = X[i * ldX + g];
        if (!(fabsf(x) <= 3.0e38f)) bad = true;
        const int k = f2key(x);
        lo = k < lo ? k : lo; hi = k > hi ? k : hi;
    }
    if (bad) atomicOr(flag, 1);
    atomicMin(&kmin[g], lo);
    atomicMax(&kmax[g], hi);
}
__global__ __launch_bounds__(256) void encode_geno_kernel(long long n, long long p, const float *X, long long ldX, const int *kmin, const int *kmax,
                                                          unsigned short *Gt, long long ldk, int *flag)
{
    __shared__ unsigned short tile[32][34];
    const long long g0 = (long long)blockIdx.x * 32, i0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long g = g0 + tx;
    const float lo = (g < p) ? key2f(kmin[g]) : 0.0f, hv = (g < p) ? key2f(kmax[g]) : 0.0f;
    const float mid = lo + 0.5f * (hv - lo), tol = 8.0f * 1.1920929e-7f * fmaxf(fabsf(lo), fabsf(hv));
    bool bad = false;
    for (int r = ty; r < 32; r += 8) {
        const long long i = i0 + r;
        unsigned short code = 0;
        if (i < n && g < p) {
            const float x = X[i * ldX + g];
            if (x == lo) code = 0;
            else if (x == hv) code = 0x4000;                       // bf16 2.0
            else { code = 0x3F80; if (!(fabsf(x - mid) <= tol)) bad = true; }   // bf16 1.0
        }
        tile[r][tx] = code;
    }
    if (bad) atomicOr(flag, 1);
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long long gg = g0 + r, i = i0 + tx;
        if (gg < p && i < ldk) Gt[gg * ldk + i] = tile[tx][r];
    }
}
__global__ void params_geno_kernel(long long p, const int *kmin, const int *kmax, float *v0, float *dx)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p) return;
    const float lo = key2f(kmin[g]), hv = key2f(kmax[g]);
    v0[g] = lo; dx[g] = 0.5f * (hv - lo);
}

struct GenoParams {
    long long n, p, ldx, ldk, ldp;
    const unsigned short *Gt, *Up;
    const float *v0, *dx;
    const double *colsum;
    float *Xr;
    int tiles_m, tiles_n, KT3;
};

__global__ __launch_bounds__(256, PG_GBLK) void rotate_geno_kernel(GenoParams gp)
{
    __shared__ __attribute__((aligned(16))) unsigned char As[2][GBM * GROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][GBN * GROWB];
    const int T = gp.tiles_m * gp.tiles_n;
    const int b = blockIdx.x;
    const int q = T / 8, r = T % 8, xcd = b % 8;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    const int per_group = 8 * gp.tiles_n;
    const int grp = lid / per_group, first_m = grp * 8;
    const int gsz = (gp.tiles_m - first_m) < 8 ? (gp.tiles_m - first_m) : 8;
    const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
    const long long m0 = (long long)tm * GBM, n0 = (long long)tn * GBN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // staging: 128 rows x 128 B per operand per stage = 1024 x 16 B; 256 threads x 4
    const int srow = tid / GCH, schunk = tid % GCH;     // rows srow + GRPP*h, 16-byte chunk schunk
    // The genotype tile of a K-tile serves all three U planes: it is staged once per K-tile (buffer (kt3/3)&1),
    // only the U-plane tile changes every stage — a third less LDS-write traffic, which is what bounds this loop.
    uint4 ra[GNH], rb[GNH];
    auto gload = [&](int kt3) {
        const long long kb = (long long)kt3 * GBK;                // B: plane-interleaved U tiles
#pragma unroll
        for (int h = 0; h < GNH; h++) {
            const long long rown = n0 + srow + GRPP * h;
            rb[h] = (rown < gp.n) ? *reinterpret_cast<const uint4 *>(gp.Up + rown * gp.ldp + kb + schunk * 8) : make_uint4(0, 0, 0, 0);
        }
        if (kt3 % 3 == 0) {
            const long long ka = (long long)(kt3 / 3) * GBK;      // A: genotype codes of K-tile kt3/3
#pragma unroll
            for (int h = 0; h < GNH; h++) {
                const long long rowm = m0 + srow + GRPP * h;
                ra[h] = (rowm < gp.p) ? *reinterpret_cast<const uint4 *>(gp.Gt + rowm * gp.ldk + ka + schunk * 8) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto lstore = [&](int kt3) {
#pragma unroll
        for (int h = 0; h < GNH; h++) *reinterpret_cast<uint4 *>(&Bs[kt3 & 1][(srow + GRPP * h) * GROWB + schunk * 16]) = rb[h];
        if (kt3 % 3 == 0) {
#pragma unroll
            for (int h = 0; h < GNH; h++) *reinterpret_cast<uint4 *>(&As[(kt3 / 3) & 1][(srow + GRPP * h) * GROWB + schunk * 16]) = ra[h];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < gp.KT3; kt++) {
        const int buf = kt & 1, abuf = (kt / 3) & 1;
        if (kt + 1 < gp.KT3) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 16) {
            // 32x32x16 operand: lane l holds row (l & 31), k = kk + 8*(l >> 5) .. +7  (16 bytes)
            const int koff = (kk + 8 * (lane >> 5)) * 2;
            bf16x8 a[2], bb[2];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = *reinterpret_cast<const bf16x8 *>(&As[abuf][(wm * 64 + i * 32 + (lane & 31)) * GROWB + koff]);
#pragma unroll
            for (int j = 0; j < 2; j++) bb[j] = *reinterpret_cast<const bf16x8 *>(&Bs[buf][(wn * 64 + j * 32 + (lane & 31)) * GROWB + koff]);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < gp.KT3) lstore(kt + 1);
        __syncthreads();
    }
    // epilogue: Xr[g][k] = v0_g * (U'1)_k + dx_g * acc   (fp64 combine, one rounding to fp32); pad columns zero
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const long long col = n0 + wn * 64 + j * 32 + (lane & 31);
            const double ck = (col < gp.n) ? gp.colsum[col] : 0.0;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const long long row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < gp.p && col < gp.ldx) {
                    const double v = (col < gp.n) ? fma((double)gp.dx[row], (double)acc[i][j][e], (double)gp.v0[row] * ck) : 0.0;
                    gp.Xr[row * gp.ldx + col] = (float)v;
                }
            }
        }
}

}  // namespace pg

using namespace pg;

// Prepare U once per eigendecomposition: bf16 planes + fp64 column sums.  Uprep must hold pg_geno_prep_bytes(n) bytes.
extern "C" size_t pg_geno_prep_bytes(int64_t n)
{
    const long long kt = (n + GBK - 1) / GBK;
    const size_t planes = (size_t)n * kt * 3 * GBK * 2;
    const size_t sums = (size_t)(kt + 1) * n * 8;
    return ((planes + 255) & ~(size_t)255) + sums + 256;
}
extern "C" int pg_geno_prep_dev(pg_ctx *ctx, int64_t n, const float *U, int64_t ldU, void *Uprep)
{
    PG_REQUIRE(ctx && U && Uprep && n > 0 && ldU >= n, "pg_geno_prep_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    const long long kt = (n + GBK - 1) / GBK, ldp = kt * 3 * GBK;
    unsigned short *Up = (unsigned short *)Uprep;
    double *colsum = (double *)((char *)Uprep + (((size_t)n * ldp * 2 + 255) & ~(size_t)255));
    split_u_kernel<<<dim3((unsigned)((n + 63) / 64), (unsigned)kt), 256, 0, ctx->stream>>>(n, ldU, U, Up, ldp, colsum);
    colsum_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(n, (int)kt, colsum);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// Rotate a block of genotype columns.  work must hold pg_geno_work_bytes(n, p).  *is_geno_host receives 1 when the block
// qualified and Xr was written, 0 when it did not (Xr untouched: the caller falls back to pg_rotate_dev).  Synchronises.
extern "C" size_t pg_geno_work_bytes(int64_t n, int64_t p)
{
    const long long ldk = (n + GBK - 1) / GBK * GBK;
    return (((size_t)p * ldk * 2 + 255) & ~(size_t)255) + (size_t)p * 16 + 512;
}
extern "C" int pg_rotate_geno_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const float *X, int64_t ldX, float *Xr, int64_t ldx,
                                  void *work, int *is_geno_host)
{
    PG_REQUIRE(ctx && Uprep && X && Xr && work && is_geno_host, "pg_rotate_geno_dev: NULL argument");
    PG_REQUIRE(n > 0 && p > 0 && ldX >= p && ldx >= n && ldx <= (n + 127) / 128 * 128, "pg_rotate_geno_dev: bad shape");
    PG_HIP(hipSetDevice(ctx->device));
    const long long kt = (n + GBK - 1) / GBK, ldk = kt * GBK, ldp = kt * 3 * GBK;
    unsigned short *Gt = (unsigned short *)work;
    char *tail = (char *)work + (((size_t)p * ldk * 2 + 255) & ~(size_t)255);
    float *v0 = (float *)tail, *dx = v0 + p;
    int *kmin = (int *)(dx + p), *kmax = kmin + p;
    int *flag = kmax + p;
    PG_HIP(hipMemsetAsync(flag, 0, 4, ctx->stream));
    minmax_init_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, kmin, kmax);
    minmax_geno_kernel<<<dim3((unsigned)((p + 63) / 64), (unsigned)((n + 255) / 256)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, flag);
    encode_geno_kernel<<<dim3((unsigned)((p + 31) / 32), (unsigned)((ldk + 31) / 32)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, Gt, ldk, flag);
    params_geno_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, kmin, kmax, v0, dx);
    PG_HIP(hipGetLastError());
    int hflag = 0;
    PG_HIP(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    *is_geno_host = hflag ? 0 : 1;
    if (hflag) return PG_OK;
    GenoParams gp{};
    gp.n = n; gp.p = p; gp.ldx = ldx; gp.ldk = ldk; gp.ldp = ldp;
    gp.Gt = Gt; gp.Up = (const unsigned short *)Uprep;
    gp.colsum = (const double *)((const char *)Uprep + (((size_t)n * ldp * 2 + 255) & ~(size_t)255)) + (size_t)kt * n;
    gp.v0 = v0; gp.dx = dx; gp.Xr = Xr;
    gp.tiles_m = (int)((p + GBM - 1) / GBM); gp.tiles_n = (int)((n + GBN - 1) / GBN); gp.KT3 = (int)(3 * kt);
    const long long T = (long long)gp.tiles_m * gp.tiles_n;
    PG_REQUIRE(T < (1LL << 31), "pg_rotate_geno_dev: too many tiles");
    rotate_geno_kernel<<<dim3((unsigned)T), 256, 0, ctx->stream>>>(gp);
    PG_HIP(hipGetLastError());
    return PG_OK;
}
