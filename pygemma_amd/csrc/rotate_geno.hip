// rotate_geno.hip — H2 on the fp16 MFMA pipe: X <- U'X for a block of SNP columns, 16x the fp32-MFMA rate.
//
// U is split once into two fp16 planes, S*U = H1 + H2 + e, S the power of two that puts S*max|U| in [2^14, 2^15) (keeps H2
// out of the fp16 subnormals), round-to-nearest at both steps: |e| <= 2^-23 |S*U| worst case (2^-24 typical) — within one bit of float32's own rounding of U
// (measured: the split adds 0.6x the error U already carries from that rounding, 1/20 of the fp32 accumulation error).
//
// (1) GENOTYPE columns — at most three equally spaced values (hard calls 0/1/2, raw or centred/standardised: what every
//     caller in the reference feeds lmm.pygemma, experiments/*/run_*.py, tests/test_pygemma.py:184-192):
//         x_g = v0_g + dx_g * code_g,  code in {0,1,2}   =>   U'x_g = v0_g * (U'1) + dx_g * (U' code_g)
//     The codes are exact in fp16, every product code*H is exact in fp32, and the two partial GEMMs accumulate into ONE fp32
//     accumulator (K' = 2K).  U'1 in fp64, combine in fp64, one rounding to fp32.
// (2) a genotype column may hold ONE other value o_g anywhere (missing calls imputed with the column mean, as the reference's
//     callers do: experiments/benchmarks/benchmarks.py:243-244):  x_g = v0_g + dx_g * code_g + (o_g - v0_g) * ind_g  (code 0
//     where ind = 1); the block takes a second, accumulating pass of the same GEMM on the 0/1 indicator plane.
// (3) any other FINITE block (imputed dosages, arbitrary float X): X itself in two fp16 planes, s_g x = X1 + X2 + e (s_g a
//     per-column power of two, |e| <= 2^-23 |s_g x|): pass 1 on X1, pass 2 accumulates X2.
// (4) a block with a NaN/Inf is left to the fp32-MFMA kernel (pg_rotate_dev), whose propagation is the reference's sgemm's.
// Error class of (1)-(3) = fp32 accumulation, the same as the fp32-MFMA kernel and the reference's sgemm (lmm/lmm.py:244).
//
// Layout: both operands K-contiguous ("NT" GEMM): Gt [p][ldk] fp16 (SNP-major), Up [n][2*KT*64] fp16 with the two planes of
// each 64-sample K-tile interleaved, so the kernel is a plain GEMM over K' = 2K whose A tile index is kt'/2.  Kernel design:
// see rotate_geno_kernel.
#include "common.hpp"

#include <type_traits>

namespace pg {

// Device-side choice of the rotation path (pg_rotate_auto_dev): every candidate kernel of a block is enqueued and looks at the
// flags the detect pass left — flag[0] bit 0: not genotype-valued, bit 1: NaN/Inf present; flag[1]: some column has an imputed
// value — so the host never waits for them.
enum { COND_ALWAYS = 0, COND_PASS1 = 1, COND_PASS2 = 2, COND_INDICATOR = 3, COND_SPLIT = 4, COND_FP32 = 5, COND_GENO = 6 };
__device__ __forceinline__ bool run_cond(const int *flag, int mode)
{
    if (!flag || mode == COND_ALWAYS) return true;
    const int f0 = flag[0], f1 = flag[1];
    switch (mode) {
        case COND_PASS1: return !(f0 & 2);
        case COND_PASS2: return !(f0 & 2) && ((f0 & 1) || f1);
        case COND_INDICATOR: return f0 == 0 && f1;
        case COND_SPLIT: return (f0 & 3) == 1;
        case COND_GENO: return f0 == 0;
        default: return (f0 & 2) != 0;
    }
}

typedef float floatx4 __attribute__((ext_vector_type(4)));

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GBK = 64;   // samples per K-tile: the LDS image is 128-byte rows = 64 fp16

__device__ __forceinline__ unsigned short f32_to_f16_rn(float f)
{
    const _Float16 h = (_Float16)f;        // v_cvt_f16_f32: round to nearest even
    unsigned short b;
    __builtin_memcpy(&b, &h, 2);
    return b;
}
__device__ __forceinline__ float f16_to_f32(unsigned short b)
{
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}

// max |U| as an int key (order-independent), then the power-of-two scale S: S*max|U| in (2^14, 2^15]
__global__ void absmax_kernel(long long n, long long ldU, const float *U, int *key)
{
    int m = 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n * n; idx += (long long)gridDim.x * blockDim.x) {
        const int b = __float_as_int(U[(idx / n) * ldU + idx % n]) & 0x7FFFFFFF;
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) atomicMax(key, m);
}
__global__ void scale_kernel(const int *key, float *scale)
{
    const int e = ((*key) >> 23) & 0xFF;                 // biased exponent of max|U| (NaN/inf keys are rejected by the caller's flag)
    int se = 127 + 14 - (e - 127);                       // S = 2^(14 - floor(log2 max)): S*max in [2^14, 2^15)
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    scale[0] = __int_as_float(se << 23);
    scale[1] = __int_as_float((254 - se) << 23);         // 1/S
}

// U (n x n, row stride ldU, eigenvector k in column k) -> Up[k][kt][plane][j] (fp16 planes of S*U) and colsum[k] = sum_i U[i][k] (fp64)
__global__ __launch_bounds__(256) void split_u_kernel(long long n, long long ldU, const float *U, unsigned short *Up, long long ldp, double *colsum,
                                                      const float *scale)
{
    __shared__ float tile[64][65];
    const long long k0 = (long long)blockIdx.x * 64, i0 = (long long)blockIdx.y * 64;   // eigen index block, sample block (= K-tile)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float S = scale[0];
    for (int r = ty; r < 64; r += 4) {
        const long long i = i0 + r, k = k0 + tx;
        tile[r][tx] = (i < n && k < n) ? U[i * ldU + k] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {      // r: eigen index within block, tx: sample within K-tile
        const long long k = k0 + r;
        if (k >= n) continue;
        const float u = tile[tx][r] * S;                     // exact (power of two)
        const unsigned short h1 = f32_to_f16_rn(u);
        const float r1 = u - f16_to_f32(h1);                 // exact
        const unsigned short h2 = f32_to_f16_rn(r1);
        unsigned short *dst = Up + k * ldp + (long long)blockIdx.y * 2 * GBK;
        dst[tx] = h1; dst[GBK + tx] = h2;
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
//            is not a genotype block), transposed to SNP-major fp16 Gt [p][ldk] through a 32x32 LDS tile; pad zeroed
//   params : v0 = lowest, dx = (highest - lowest)/2  ->  x = v0 + dx*code for one-, two- and three-valued columns alike
__device__ __forceinline__ int f2key(float f) { int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

constexpr int OTHER_EMPTY = 0x7FC00001;   // a NaN payload: never a data value (NaNs disqualify the block)
__global__ void minmax_init_kernel(long long p, int *kmin, int *kmax, int *other)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < p) { kmin[g] = 0x7FFFFFFF; kmax[g] = (int)0x80000000; other[g] = OTHER_EMPTY; }
}
template <class T>
__global__ __launch_bounds__(256) void minmax_geno_kernel(long long n, long long p, const T *X, long long ldX, int *kmin, int *kmax, int *flag)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long g = (long long)blockIdx.x * 64 + tx;
    if (g >= p) return;
    const long long i0 = (long long)blockIdx.y * 256;
    int lo = 0x7FFFFFFF, hi = (int)0x80000000;
    bool bad = false;
    for (long long i = i0 + ty; i < i0 + 256 && i < n; i += 4) {
        const float x = (float)X[i * ldX + g];
        if (!(fabsf(x) <= 3.0e38f)) bad = true;
        const int k = f2key(x);
        lo = k < lo ? k : lo; hi = k > hi ? k : hi;
    }
    if (bad) atomicOr(flag, 2);       // NaN / Inf somewhere in the block
    atomicMin(&kmin[g], lo);
    atomicMax(&kmax[g], hi);
}
template <class T, bool I8 = false>      // I8: the codes as int8 0/1/2 with row stride ldk BYTES (the int8 kernel's operand), else fp16 0.0/1.0/2.0
__global__ __launch_bounds__(256) void encode_geno_kernel(long long n, long long p, const T *X, long long ldX, const int *kmin, const int *kmax,
                                                          int *other, unsigned short *Gt, long long ldk, int *flag)
{
    // tile: 64 SNPs x 128 samples — 256-byte row segments on the way in (64 floats) and on the way out (128 fp16 codes)
    __shared__ unsigned short tile[128][66];
    const long long g0 = (long long)blockIdx.x * 64, i0 = (long long)blockIdx.y * 128;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long g = g0 + tx;
    const float lo = (g < p) ? key2f(kmin[g]) : 0.0f, hv = (g < p) ? key2f(kmax[g]) : 0.0f;
    const float mid = lo + 0.5f * (hv - lo), tol = 8.0f * 1.1920929e-7f * fmaxf(fabsf(lo), fabsf(hv));
    bool bad = false, any_ind = false;
#pragma unroll 8
    for (int r = ty; r < 128; r += 4) {
        const long long i = i0 + r;
        unsigned short code = 0;
        if (i < n && g < p) {
            const float x = (float)X[i * ldX + g];
            if (x == lo) code = 0;
            else if (x == hv) code = I8 ? 2 : 0x4000;              // fp16 2.0
            else if (fabsf(x - mid) <= tol) code = I8 ? 1 : 0x3C00;   // fp16 1.0
            else {                                                 // the column's one other value (every occurrence the same bits)
                const int xb = __float_as_int(x);
                const int prev = atomicCAS(&other[g], OTHER_EMPTY, xb);
                if (prev != OTHER_EMPTY && prev != xb) bad = true;
                any_ind = true;
            }
        }
        tile[r][tx] = code;
    }
    if (bad) atomicOr(flag, 1);
    if (any_ind) atomicOr(flag + 1, 1);
    __syncthreads();
    // thread -> two consecutive samples of one SNP row: 64 threads x 4 bytes = one 256-byte segment
    for (int r = ty; r < 64; r += 4) {
        const long long gg = g0 + r, i = i0 + 2 * tx;
        if (gg < p && i < ldk) {   // ldk is a multiple of 64: i even and i < ldk  =>  i + 1 < ldk
            if (I8) *reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(Gt) + gg * ldk + i) = (unsigned short)(tile[2 * tx][r] | (tile[2 * tx + 1][r] << 8));
            else *reinterpret_cast<unsigned *>(Gt + gg * ldk + i) = (unsigned)tile[2 * tx][r] | ((unsigned)tile[2 * tx + 1][r] << 16);
        }
    }
}

// general float columns: s x = X1 + X2 (two fp16 planes, round to nearest twice), s = the power of two that puts the
// column's largest magnitude in [2^14, 2^15); SNP-major planes like Gt
__device__ __forceinline__ float split_scale(float lo, float hv)
{
    const float m = fmaxf(fabsf(lo), fabsf(hv));
    if (!(m > 0.0f)) return 1.0f;
    int e = (int)((__float_as_uint(m) >> 23) & 0xFF);          // biased exponent (subnormal maxima: e = 0 -> clamp below)
    int se = 127 + 14 - (e - 127);
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __int_as_float(se << 23);
}
template <class T>
__global__ __launch_bounds__(256) void split_x_kernel(long long n, long long p, const T *X, long long ldX, const int *kmin, const int *kmax,
                                                      unsigned short *X1, unsigned short *X2, long long ldk, const int *cond = nullptr)
{
    __shared__ unsigned short t1[128][66], t2[128][66];
    if (!run_cond(cond, COND_SPLIT)) return;
    const long long g0 = (long long)blockIdx.x * 64, i0 = (long long)blockIdx.y * 128;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long g = g0 + tx;
    const float sc = (g < p) ? split_scale(key2f(kmin[g]), key2f(kmax[g])) : 1.0f;
#pragma unroll 8
    for (int r = ty; r < 128; r += 4) {
        const long long i = i0 + r;
        unsigned short h1 = 0, h2 = 0;
        if (i < n && g < p) {
            const float x = (float)X[i * ldX + g] * sc;          // exact: power of two
            h1 = f32_to_f16_rn(x);
            h2 = f32_to_f16_rn(x - f16_to_f32(h1));
        }
        t1[r][tx] = h1; t2[r][tx] = h2;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const long long gg = g0 + r, i = i0 + 2 * tx;
        if (gg < p && i < ldk) {
            *reinterpret_cast<unsigned *>(X1 + gg * ldk + i) = (unsigned)t1[2 * tx][r] | ((unsigned)t1[2 * tx + 1][r] << 16);
            *reinterpret_cast<unsigned *>(X2 + gg * ldk + i) = (unsigned)t2[2 * tx][r] | ((unsigned)t2[2 * tx + 1][r] << 16);
        }
    }
}
__global__ void params_split_kernel(long long p, const int *kmin, const int *kmax, float *v0, float *dx, float *dlt, const int *cond = nullptr)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p || !run_cond(cond, COND_SPLIT)) return;
    const float inv = 1.0f / split_scale(key2f(kmin[g]), key2f(kmax[g]));   // exact
    v0[g] = 0.0f; dx[g] = inv; dlt[g] = inv;
}

// indicator plane (fp16 0/1) of the columns' other value, SNP-major like Gt; only run for blocks that have one
template <class T, bool I8 = false>      // I8: int8 0/1 with row stride ldk bytes
__global__ __launch_bounds__(256) void indicator_geno_kernel(long long n, long long p, const T *X, long long ldX, const int *other,
                                                             unsigned short *Gi, long long ldk, const int *cond = nullptr)
{
    __shared__ unsigned short tile[32][34];
    if (!run_cond(cond, COND_INDICATOR)) return;
    const long long g0 = (long long)blockIdx.x * 32, i0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long g = g0 + tx;
    const int ob = (g < p) ? other[g] : OTHER_EMPTY;
    for (int r = ty; r < 32; r += 8) {
        const long long i = i0 + r;
        tile[r][tx] = (i < n && g < p && ob != OTHER_EMPTY && __float_as_int((float)X[i * ldX + g]) == ob) ? (I8 ? 1 : 0x3C00) : 0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long long gg = g0 + r, i = i0 + tx;
        if (gg < p && i < ldk) {
            if (I8) reinterpret_cast<signed char *>(Gi)[gg * ldk + i] = (signed char)tile[tx][r];
            else Gi[gg * ldk + i] = tile[tx][r];
        }
    }
}
__global__ void params_geno_kernel(long long p, const int *kmin, const int *kmax, const int *other, float *v0, float *dx, float *dlt)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p) return;
    const float lo = key2f(kmin[g]), hv = key2f(kmax[g]);
    v0[g] = lo; dx[g] = 0.5f * (hv - lo);
    dlt[g] = (other[g] == OTHER_EMPTY) ? 0.0f : __int_as_float(other[g]) - lo;   // exact or one rounding: folded in fp64 below
}

// PLINK .bed block (SNP-major, 4 samples per byte, sample k of a byte in bits 2k..2k+1: 00 hom A1, 01 missing, 10 het,
// 11 hom A2) -> code plane, indicator plane of the missing calls and v0 = 0, dx = 1, delta = column mean of the called
// genotypes (what SimpleImputer(strategy='mean') puts there, experiments/benchmarks/benchmarks.py:243-244).
// Dosage = copies of A2 (pysnptools count_A1=False, benchmarks.py:233) or of A1 (count_a1).  One workgroup per SNP.
template <bool I8>       // I8: code and indicator planes as int8 with row stride ldk8 bytes
__global__ __launch_bounds__(256) void decode_bed_kernel(long long n, long long p, const unsigned char *bed, long long ldb, int count_a1,
                                                         unsigned short *Gt, unsigned short *Gi, long long ldk, long long ldk8, float *v0, float *dx, float *dlt, int *flag)
{
    __shared__ unsigned cnt[3];
    const long long g = blockIdx.x;
    const unsigned char *row = bed + g * ldb;
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned n1 = 0, n2 = 0, nm = 0;
    const long long span = I8 ? ldk8 : ldk;
    for (long long i = threadIdx.x; i < span; i += blockDim.x) {
        unsigned short code = 0, ind = 0;
        if (i < n) {
            const unsigned c = (row[i >> 2] >> (2 * (i & 3))) & 3u;
            if (c == 1u) { ind = I8 ? 1 : 0x3C00; nm++; }
            else if (c == 2u) { code = I8 ? 1 : 0x3C00; n1++; }
            else if ((c == 3u) != (count_a1 != 0)) { code = I8 ? 2 : 0x4000; n2++; }     // hom A2 counts 2 (A2 dosage) / hom A1 counts 2 (A1 dosage)
        }
        if (I8) { reinterpret_cast<signed char *>(Gt)[g * ldk8 + i] = (signed char)code; reinterpret_cast<signed char *>(Gi)[g * ldk8 + i] = (signed char)ind; }
        else { Gt[g * ldk + i] = code; Gi[g * ldk + i] = ind; }
    }
    atomicAdd(&cnt[0], n1); atomicAdd(&cnt[1], n2); atomicAdd(&cnt[2], nm);   // integer sums: order-independent
    __syncthreads();
    if (threadIdx.x == 0) {
        const double called = (double)n - (double)cnt[2];
        v0[g] = 0.0f; dx[g] = 1.0f;
        dlt[g] = (cnt[2] > 0 && called > 0) ? (float)(((double)cnt[0] + 2.0 * (double)cnt[1]) / called) : 0.0f;
        if (cnt[2] > 0) atomicOr(flag + 1, 1);
    }
}

template <class T>
__global__ void cast_to_f32_kernel(long long n, long long p, const T *X, long long ldX, float *Xf, long long ldXf)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p) return;
    for (long long i = blockIdx.y; i < n; i += gridDim.y) Xf[i * ldXf + g] = (float)X[i * ldX + g];
}

struct GenoParams {
    long long n, p, ldx, ldk, ldp;
    const unsigned short *Gt, *Up;
    const float *v0, *dx;       // v0 == nullptr: the accumulate pass (Xr += dx * acc)
    const double *colsum;
    float *Xr;
    int tiles_m, tiles_n, KT;   // KT = K-tiles of 64 samples; stages = 2*KT (two U planes per K-tile)
    const float *scale;         // {S, 1/S}
    const int *cond;            // device-side predication (pg_rotate_auto_dev): block flags of the detect pass, or nullptr
    int cmode;
};

// LDS image of a 128-row x 64-k fp16 tile: plain 128-byte rows (what the LDS-DMA writes: a wave instruction fills
// 1 KB = 8 consecutive rows, lane L -> row L/8, 16-byte chunk L%8), with the chunk index XOR-swizzled by (row>>1)&7.
// The swizzle is applied on the GLOBAL source address of the DMA (the LDS side stays lane-linear) and again on the
// read address, so the 16 rows a ds_read_b128 lane group touches fall on 16 different bank quads.
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// The GEMM.  256 x 256 tile per 512-thread workgroup, 8 waves as 4 x 2, each 64 (SNPs) x 128 (eigen indices) = 4 x 8
// v_mfma_f32_16x16x32_f16 tiles; K-tile 64; stage t = U plane t&1 of K-tile t>>1 (K' = 2K).
// Staging by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into two genotype buffers + three
// U-plane buffers of 32 KB = 160 KB of LDS, DMA for stage t+2 in flight across the barriers of stage t.  Waits are counted
// (`s_waitcnt vmcnt(#issued this stage)` retires everything older) and the barriers are raw s_barrier: __syncthreads()
// would drain the DMA at every barrier.
// Ping-pong: waves 0-3 ("early") and 4-7 ("late") sit one per SIMD (workgroup waves go to SIMDs round-robin) and run half
// a stage apart: while one wave of a SIMD issues its 64 MFMAs the other issues DMA and reads fragments.  The genotype
// fragments are read once per K-tile and kept in registers for both planes.  The U fragments of the second k-half are read during the MFMA phase (each replaces
// its predecessor as soon as that one's MFMAs are issued: 32 fragment registers instead of 64), so a U buffer stays busy
// through the late waves' MFMA phase; the DMA duties are therefore split by group: the LATE waves stage the whole U plane
// (8 instructions per wave and stage, issued in their memory phase, i.e. after their own last read of the buffer they
// refill), the EARLY waves stage the genotype tile (4 per wave and stage), each group with its own counted wait.
//   phase 2t   : early mem(t)   [A half of tile T+1|T+2; reads] | late mfma(t-1) [rolling reads of B[(t-1)%3]]
//   phase 2t+1 : early mfma(t)                                  | late mem(t)    [B(t+2) -> B[(t-1)%3]; reads]
#ifndef PG_GENO_GRP
#define PG_GENO_GRP 4
#endif
template <int MF>      // MFMA shape: 16 = v_mfma_f32_16x16x32_f16 (shipped), 32 = v_mfma_f32_32x32x16_f16 (measured alternative: see launch_geno_gemm)
__global__ __launch_bounds__(512, 1) void rotate_geno_kernel(GenoParams gp)
{
    if (!run_cond(gp.cond, gp.cmode)) return;      // uniform over the grid: every wave leaves before any barrier
    constexpr int TBUF = 256 * 128;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const Bs = lds, *const As = lds + 3 * TBUF;
    const int T = gp.tiles_m * gp.tiles_n;
    const int b = blockIdx.x;
    const int q = T / 8, r = T % 8, xcd = b % 8;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    const int per_group = PG_GENO_GRP * gp.tiles_n;
    const int grp = lid / per_group, first_m = grp * PG_GENO_GRP;
    const int gsz = (gp.tiles_m - first_m) < PG_GENO_GRP ? (gp.tiles_m - first_m) : PG_GENO_GRP;
    const int tm = first_m + (lid % per_group) % gsz, tn = (lid % per_group) / gsz;
    const long long m0 = (long long)tm * 256, n0 = (long long)tn * 256;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves, each 64 (SNPs) x 128 (eigen indices)
    const int KT2 = 2 * gp.KT;

    // 128 accumulator registers either way: 4 x 8 tiles of 16 x 16 (4 floats per lane) or 2 x 4 tiles of 32 x 32 (16 per lane)
    floatx4 acc[MF == 16 ? 4 : 1][MF == 16 ? 8 : 1];
    floatx16 acc32[MF == 32 ? 2 : 1][MF == 32 ? 4 : 1];
    if (MF == 16) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int e = 0; e < 4; e++) acc[MF == 16 ? i : 0][MF == 16 ? j : 0][e] = 0.0f;
    } else {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc32[MF == 32 ? i : 0][MF == 32 ? j : 0][e] = 0.0f;
    }

    // early wave w (0..3) fills rows 64w .. 64w+63 of the genotype tile, late wave w (4..7) rows 64(w-4) .. of the U-plane
    // tile: 8 instructions of 8 rows; rows past the end of the operand are clamped (their outputs are never stored)
    const bool late = wave >= 4;
    const int lrow = lane >> 3, lchunk = lane & 7;
    const unsigned char *gsrc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int row = (wave & 3) * 64 + 8 * t + lrow;
        long long rm = m0 + row, rn = n0 + row;
        rm = rm < gp.p ? rm : gp.p - 1;
        rn = rn < gp.n ? rn : gp.n - 1;
        gsrc[t] = (late ? reinterpret_cast<const unsigned char *>(gp.Up + rn * gp.ldp)
                        : reinterpret_cast<const unsigned char *>(gp.Gt + rm * gp.ldk)) + swz(row, lchunk) * 16;
    }
    auto dmaB = [&](int stage, int buf) {            // late waves only
        unsigned char *dst = Bs + buf * TBUF + ((wave & 3) * 64) * 128;
#pragma unroll
        for (int t = 0; t < 8; t++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc[t] + (size_t)stage * GBK * 2),
                                             (__attribute__((address_space(3))) void *)(dst + 8 * t * 128), 16, 0, 0);
    };
    auto dmaA = [&](int ktile, int half) {           // early waves only: rows 32*half .. 32*half+31 of the wave's 64 rows
        unsigned char *dst = As + (ktile & 1) * TBUF + ((wave & 3) * 64 + 32 * half) * 128;
#pragma unroll
        for (int t = 0; t < 4; t++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc[4 * half + t] + (size_t)ktile * GBK * 2),
                                             (__attribute__((address_space(3))) void *)(dst + 8 * t * 128), 16, 0, 0);
    };
    // prologue: K-tile 0 (genotypes + both planes) and the first half of the genotypes of K-tile 1
    if (late) {
        dmaB(0, 0); dmaB(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        dmaA(0, 0); dmaA(0, 1);
        if (gp.KT > 1) { dmaA(1, 0); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();
    int b0 = 0;
    halfx8 fa[2][4];
    for (int ktile = 0; ktile < gp.KT; ktile++) {
      const unsigned char *Acur = As + (ktile & 1) * TBUF;
#pragma unroll
      for (int pl = 0; pl < 2; pl++) {
        const int kt = 2 * ktile + pl;
        const unsigned char *Bcur = Bs + b0 * TBUF;
        const int b2 = (b0 >= 1) ? b0 - 1 : 2;            // (t + 2) % 3
        bool issued = false;
        // ---------------- memory phase
        if (late) { if (kt + 2 < KT2) { dmaB(kt + 2, b2); issued = true; } }
        else if (pl == 0) { if (ktile + 1 < gp.KT) { dmaA(ktile + 1, 1); issued = true; } }   // second half of K-tile T+1
        else { if (ktile + 2 < gp.KT) { dmaA(ktile + 2, 0); issued = true; } }              // first half of K-tile T+2
        halfx8 fb[8];
        const int chunk0 = (MF == 16) ? (lane >> 4) : (lane >> 5);
        if (MF == 16) {
            if (pl == 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int row = wm * 64 + i * 16 + (lane & 15);
                        fa[ks][i] = *reinterpret_cast<const halfx8 *>(Acur + row * 128 + swz(row, 4 * ks + chunk0) * 16);
                    }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int row = wn * 128 + j * 16 + (lane & 15);
                fb[j] = *reinterpret_cast<const halfx8 *>(Bcur + row * 128 + swz(row, chunk0) * 16);
            }
        } else {
            // 32 x 32 x 16: lane l holds row l & 31 and the eight k of chunk 2 s + (l >> 5) of k-step s (four k-steps of 16 per K-tile);
            // fa[s >> 1][2 (s & 1) + i] = genotype rows 32 i .. of k-step s;  fb[4 (s & 1) + j] = U rows 32 j .. of k-step s (s = 0, 1 here)
            if (pl == 0) {
#pragma unroll
                for (int st = 0; st < 4; st++)
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        const int row = wm * 64 + i * 32 + (lane & 31);
                        fa[st >> 1][2 * (st & 1) + i] = *reinterpret_cast<const halfx8 *>(Acur + row * 128 + swz(row, 2 * st + chunk0) * 16);
                    }
            }
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int row = wn * 128 + j * 32 + (lane & 31);
                    fb[4 * st + j] = *reinterpret_cast<const halfx8 *>(Bcur + row * 128 + swz(row, 2 * st + chunk0) * 16);
                }
        }
        // everything this wave issued before this stage has landed
        if (!issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (late) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- MFMA phase
        __builtin_amdgcn_s_setprio(1);
        if (MF == 16) {
            // k-half 0 column by column; each U fragment is replaced by its k-half-1 successor as soon as its four MFMAs are
            // issued (the read hides behind the remaining ones), then k-half 1
#pragma unroll
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc[MF == 16 ? i : 0][MF == 16 ? j : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[j], acc[MF == 16 ? i : 0][MF == 16 ? j : 0], 0, 0, 0);
                const int row = wn * 128 + j * 16 + (lane & 15);
                fb[j] = *reinterpret_cast<const halfx8 *>(Bcur + row * 128 + swz(row, 4 + chunk0) * 16);
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc[MF == 16 ? i : 0][MF == 16 ? j : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1][i], fb[j], acc[MF == 16 ? i : 0][MF == 16 ? j : 0], 0, 0, 0);
        } else {
            // k-steps 0 and 1 column by column, each U fragment replaced by its successor two k-steps on as soon as its two MFMAs are
            // issued; then k-steps 2 and 3.  32 MFMAs of 32 x 32 x 16 per stage instead of 64 of 16 x 16 x 32: with operands in registers
            // the matrix pipe sustains 1 800 TF on the former and 1 285 on the latter (tools/probe_mfma_sustained.hip).
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int i = 0; i < 2; i++)
                        acc32[MF == 32 ? i : 0][MF == 32 ? j : 0] =
                            __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][2 * st + i], fb[4 * st + j], acc32[MF == 32 ? i : 0][MF == 32 ? j : 0], 0, 0, 0);
                    const int row = wn * 128 + j * 32 + (lane & 31);
                    fb[4 * st + j] = *reinterpret_cast<const halfx8 *>(Bcur + row * 128 + swz(row, 2 * (st + 2) + chunk0) * 16);
                    __builtin_amdgcn_sched_barrier(0);      // keep the read HERE, behind the MFMAs still to be issued (the scheduler sank it to its use)
                }
#pragma unroll
            for (int st = 0; st < 2; st++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
                        acc32[MF == 32 ? i : 0][MF == 32 ? j : 0] =
                            __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][2 * st + i], fb[4 * st + j], acc32[MF == 32 ? i : 0][MF == 32 ? j : 0], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        b0 = (b0 == 2) ? 0 : b0 + 1;
      }
    }
    if (!late) __builtin_amdgcn_s_barrier();
    const double invS = (double)gp.scale[1];
    const bool accum = gp.v0 == nullptr;
    auto put = [&](long long row, long long col, double ck, float a) {
        if (row < gp.p && col < gp.ldx) {
            float *dst = gp.Xr + row * gp.ldx + col;
            const double base = accum ? (double)(*dst) : (double)gp.v0[row] * ck;
            const double v = (col < gp.n) ? fma((double)gp.dx[row] * invS, (double)a, base) : 0.0;
            *dst = (float)v;
        }
    };
    if (MF == 16) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const long long col = n0 + wn * 128 + j * 16 + (lane & 15);
                const double ck = (col < gp.n) ? gp.colsum[col] : 0.0;
#pragma unroll
                for (int e = 0; e < 4; e++) put(m0 + wm * 64 + i * 16 + 4 * (lane >> 4) + e, col, ck, acc[MF == 16 ? i : 0][MF == 16 ? j : 0][e]);
            }
    } else {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const long long col = n0 + wn * 128 + j * 32 + (lane & 31);
                const double ck = (col < gp.n) ? gp.colsum[col] : 0.0;
#pragma unroll
                for (int e = 0; e < 16; e++)
                    put(m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col, ck, acc32[MF == 32 ? i : 0][MF == 32 ? j : 0][e]);
            }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same rotation on the int8 matrix pipe (r4): genotype CODES only (path (1) above and .bed blocks; indicator and split-plane
// passes stay on fp16).  v_mfma_i32_16x16x64_i8 takes the cycles of v_mfma_f32_16x16x32_f16 for twice the K
// (profiles/r03_mfma_sustained.txt: 2 620 TOP/s sustained against 1 285 TF), the codes 0/1/2 are exact in int8 and the integer
// accumulation is EXACT, so all of the error sits in the representation of U:
//     U[i][k] = 2^(E_k - 22) * (q_ik + e),  q_ik = rint(U[i][k] * 2^(22 - E_k)) = 65536 d2 + 256 d1 + d0,  d in [-128, 127],  |e| <= 1/2
// a 24-bit fixed point per EIGENVECTOR (2^E_k = the power of two at or below max_i |U[i][k]|, one higher when the top digit would
// overflow): |error of U[i][k]| <= 2^-24 * 2^(E_k+1) — float32's own rounding for entries within a factor two of the column's largest,
// k bits coarser for entries 2^-k below it.  For delocalised eigenvectors (entries ~ N(0, 1/n), maximum ~ 5 sigma) the error of U'x is
// ~ 0.5x what the fp32 accumulation of the fp16 kernel (and of the reference's sgemm) carries; tools/geno_accuracy.py measures both.
// Three planes = three GEMM passes where fp16 x 2 needs two at half the rate: 0.75 of the MFMA work.
//
// The planes cannot share an accumulator (different weights) and three accumulator sets do not fit the registers, so the planes are
// laid out as OUTPUT COLUMNS: the plane matrix Up8 has 256 rows per tile of 85 eigen indices — rows [0, 85) digit d2, [85, 170) d1,
// [170, 255) d0, row 255 zero — and the kernel is a plain 256 x 256 int8 GEMM tile over K-tiles of 128 samples (128-byte rows: the
// LDS image, swizzle, DMA shapes and fragment reads are byte-for-byte those of the fp16 kernel).  The three digits of an output meet
// in the epilogue, through the LDS the main loop has released: i32 accumulators -> LDS, then per output 65536 a2 + 256 a1 + a0 in
// fp64 (exact), times dx_g * 2^(E_k - 22), plus v0_g * (U'1)_k, one rounding to float32 — and row-contiguous 340-byte stores.
// Every stage now brings a new genotype tile (no second plane to reuse it for): the early waves stage the whole 32 KB tile of stage
// s+1 during stage s (8 DMA instructions per wave, waited at the END of their MFMA phase), into the buffer whose last readers — the
// late waves' memory phase of stage s-1 — finished one barrier earlier.
// Measured and rejected (r4): ONE barrier per stage (late waves stage the plane tile right after the barrier, nothing forces the phases
// to alternate, both waves of a SIMD may issue MFMAs together): 27.0 ms against 23.7 per step — the DMA issue blocks on the full
// vector-memory queue for ~900 cycles, and in that form it sits in front of the wave's own MFMAs instead of under its partner's.
// An L2 prefetch (each sharer of a tile touching its 1/8 or 1/4 of the K-slice 2 or 4 stages ahead, one byte per line): 24.4 / 24.7 ms
// against 24.2 — the DMAs do not wait on one another's fills.  Wave priorities (none / MFMA phase / + DMA issue): 24.3 / 24.2 / 24.1.
// The genotype tile nibble-packed (two codes per byte, unpacked by 48 VALU per wave and stage; 48 KB per stage, a third genotype buffer, no
// wait for that tile any more): 24.05 against 23.95 — the time is set by the MFMA phases (one wave alone issues a 16 x 16 x 64 MFMA every
// 20 cycles, not 16: 1 300 cycles per phase; with the unpacking beside it 1 490), not by the bytes.  v_mfma_i32_32x32x32_i8 in the same
// loop (one wave fills the pipe with it in isolation): 30.8 ms — 45 cycles per MFMA and 1 500 cycles for the partner wave to issue its 8
// DMAs: while a wave streams 32 x 32 MFMAs the other wave of the SIMD hardly issues at all (the fp16 kernel's 32 x 32 form lost the same way).
typedef int intx4 __attribute__((ext_vector_type(4)));
constexpr int GBK8 = 128;      // samples per K-tile (int8): 128-byte rows again
constexpr int I8_EIG = 85;     // eigen indices per 256-row tile of the plane matrix (3 x 85 = 255)

struct GenoI8Params {
    long long n, p, ldx, ldk8;
    const signed char *Gt8, *Up8;
    const float *v0, *dx;            // v0 == nullptr: the accumulate pass (Xr += dx * U'plane)
    const double *colsum, *wscale;   // wscale[k] = 2^(E_k - 22)
    float *Xr;
    int tiles_m, tiles_n, KT;
    int nchunks, nc_base, nc_rem;    // the n-tiles go in nchunks chunks (the first nc_rem of nc_base + 1 tiles, the rest of nc_base): see the tile order
    const int *cond;
    int cmode;
};

#ifdef PG_GENO_STAMPS      // tools/geno_i8_stamps.py: s_memtime at the phase boundaries of stages 8 .. 23 of workgroup 0, waves 0 (early) and 4 (late)
__device__ long long g_geno_stamps[2][16][6];
#define GENO_STAMP(slot)                                                                                   \
    do {                                                                                                   \
        if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && kt >= 8 && kt < 24)                         \
            g_geno_stamps[wave >> 2][kt - 8][slot] = (long long)__builtin_amdgcn_s_memtime();              \
    } while (0)
#else
#define GENO_STAMP(slot) do { } while (0)
#endif
#ifndef PG_GENO_AUX
#define PG_GENO_AUX 1      // cache-policy bits of the LDS-DMA loads (sc0 = 1, nt = 2): measured 0 / 1 / 2 / 3 -> rotation 24.9 / 24.6 / 37.3 / 37.7 ms per step (nt gives up the L2 reuse between the workgroups of an XCD)
#endif
#ifndef PG_GENO_PRIO
#define PG_GENO_PRIO 1      // 0: no wave priorities, 1: the MFMA phase at priority 1, 2: and the DMA issue of the memory phase at 3
#endif
__global__ __launch_bounds__(512, 1) void rotate_geno_i8_kernel(GenoI8Params gp)
{
    if (!run_cond(gp.cond, gp.cmode)) return;
    constexpr int TBUF = 256 * 128;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const Bs = lds, *const As = lds + 3 * TBUF;
    // Tile order.  Outermost: chunks of n-tiles small enough for their plane rows to stay in the 256 MB memory-side cache while ALL m-tiles
    // pass over them (every XCD is in the same chunk at the same time; the misses of an XCD's L2 on the planes are then served from that
    // cache instead of HBM).  Inside a chunk, as in the fp16 kernel: each XCD (workgroup id mod 8) takes a contiguous range of the order
    // [group of PG_GENO_GRP m-tiles][n-tile][m-tile in group], so the 32 workgroups resident on an XCD share 4 genotype and 8 plane tiles.
    // A chunk's tile count is padded to a multiple of 8 workgroups; the pad workgroups leave at once.
    int b = blockIdx.x, tn0 = 0, nc = gp.nc_base + (gp.nc_rem > 0 ? 1 : 0);
    for (int c = 0; c < gp.nchunks; c++) {
        nc = gp.nc_base + (c < gp.nc_rem ? 1 : 0);
        const int slots = (gp.tiles_m * nc + 7) / 8 * 8;
        if (b < slots) break;
        b -= slots; tn0 += nc;
    }
    const int Tc = gp.tiles_m * nc, qc = (Tc + 7) / 8;
    const int lid = (b % 8) * qc + b / 8;
    if (lid >= Tc) return;                             // pad slot (uniform over the workgroup, before any barrier)
    const int per_group = PG_GENO_GRP * nc;
    const int grp = lid / per_group, first_m = grp * PG_GENO_GRP;
    const int gsz = (gp.tiles_m - first_m) < PG_GENO_GRP ? (gp.tiles_m - first_m) : PG_GENO_GRP;
    const int tm = first_m + (lid % per_group) % gsz, tn = tn0 + (lid % per_group) / gsz;
    const long long m0 = (long long)tm * 256, r0 = (long long)tn * 256, k0 = (long long)tn * I8_EIG;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;     // wave: uniform, and the compiler is told so
    const int wm = wave >> 1, wn = wave & 1;          // 4 x 2 waves, each 64 (SNPs) x 128 (plane rows)

    intx4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][j][e] = 0;

    const bool late = wave >= 4;
    const int lrow = lane >> 3, lchunk = lane & 7;
    // DMA sources as a uniform 64-bit base (scalar registers; advanced by the stage) + a 32-bit per-lane offset that never changes:
    // the saddr form of global_load_lds, no vector address arithmetic in the loop (24 VALU instructions per wave and stage before —
    // each of them takes an issue slot from the partner wave's MFMAs)
    const unsigned char *const gbase = late ? reinterpret_cast<const unsigned char *>(gp.Up8 + r0 * gp.ldk8)      // the plane matrix is padded to whole tiles
                                            : reinterpret_cast<const unsigned char *>(gp.Gt8 + m0 * gp.ldk8);
    unsigned voff[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int row = (wave & 3) * 64 + 8 * t + lrow;
        const long long rmax = gp.p - 1 - m0;        // rows past the end of the genotype block are clamped (their outputs are never stored)
        const long long rr = late ? row : (row < rmax ? row : rmax);
        voff[t] = (unsigned)(rr * gp.ldk8) + swz(row, lchunk) * 16;
    }
    auto dmaB = [&](int stage, int buf) {            // late waves only
        unsigned char *dst = Bs + buf * TBUF + ((wave & 3) * 64) * 128;
        const unsigned char *src = gbase + (size_t)stage * GBK8;
#pragma unroll
        for (int t = 0; t < 8; t++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff[t]),
                                             (__attribute__((address_space(3))) void *)(dst + 8 * t * 128), 16, 0, PG_GENO_AUX);
    };
    auto dmaA = [&](int stage) {                     // early waves only: the wave's 64 rows of the genotype tile
        unsigned char *dst = As + (stage & 1) * TBUF + ((wave & 3) * 64) * 128;
        const unsigned char *src = gbase + (size_t)stage * GBK8;
#pragma unroll
        for (int t = 0; t < 8; t++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff[t]),
                                             (__attribute__((address_space(3))) void *)(dst + 8 * t * 128), 16, 0, PG_GENO_AUX);
    };
    if (late) {
        dmaB(0, 0);
        if (gp.KT > 1) { dmaB(1, 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        dmaA(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const int chunk0 = lane >> 4;
    if (late) __builtin_amdgcn_s_barrier();
    int b0 = 0;
    for (int kt = 0; kt < gp.KT; kt++) {
        const unsigned char *Acur = As + (kt & 1) * TBUF;
        const unsigned char *Bcur = Bs + b0 * TBUF;
        const int b2 = (b0 >= 1) ? b0 - 1 : 2;            // (kt + 2) % 3
        bool issued = false;
        // ---------------- memory phase
        GENO_STAMP(0);
#if PG_GENO_PRIO == 2
        __builtin_amdgcn_s_setprio(3);
#endif
        if (late) { if (kt + 2 < gp.KT) { dmaB(kt + 2, b2); issued = true; } }
        else if (kt + 1 < gp.KT) dmaA(kt + 1);
#if PG_GENO_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        intx4 fa[2][4], fb[8];
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = wm * 64 + i * 16 + (lane & 15);
                fa[ks][i] = *reinterpret_cast<const intx4 *>(Acur + row * 128 + swz(row, 4 * ks + chunk0) * 16);
            }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int row = wn * 128 + j * 16 + (lane & 15);
            fb[j] = *reinterpret_cast<const intx4 *>(Bcur + row * 128 + swz(row, chunk0) * 16);
        }
        if (late) {         // everything this wave issued before this stage has landed
            if (issued) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        GENO_STAMP(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GENO_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GENO_STAMP(3);
        // ---------------- MFMA phase
#if PG_GENO_PRIO >= 1
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int j = 0; j < 8; j++) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[0][i], fb[j], acc[i][j], 0, 0, 0);
            const int row = wn * 128 + j * 16 + (lane & 15);
            fb[j] = *reinterpret_cast<const intx4 *>(Bcur + row * 128 + swz(row, 4 + chunk0) * 16);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[1][i], fb[j], acc[i][j], 0, 0, 0);
#if PG_GENO_PRIO >= 1
        __builtin_amdgcn_s_setprio(0);
#endif
        GENO_STAMP(4);
        if (!late) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the genotype tile of the next stage is in LDS
        GENO_STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        b0 = (b0 == 2) ? 0 : b0 + 1;
    }
    if (!late) __builtin_amdgcn_s_barrier();
    // ---------------- epilogue: the three digits of each output meet through LDS (all staging buffers are free now)
    const bool accum = gp.v0 == nullptr;           // the indicator pass: Xr += delta_g * U'ind_g
    constexpr int LDW = 260;                       // words per staged row: the 4 row groups of a store (rows 4 apart) x 16 columns fall on 64 different banks
    int *const stg = reinterpret_cast<int *>(lds);
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        if ((wm >> 1) == half) {
            const int rb = (wm & 1) * 64;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 8; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        stg[(rb + i * 16 + 4 * (lane >> 4) + e) * LDW + wn * 128 + j * 16 + (lane & 15)] = acc[i][j][e];
        }
        __syncthreads();
        for (int idx = tid; idx < 128 * I8_EIG; idx += 512) {
            const int rr = idx / I8_EIG, ee = idx - rr * I8_EIG;
            const long long row = m0 + half * 128 + rr, col = k0 + ee;
            if (row < gp.p && col < gp.n) {
                const int *s = stg + rr * LDW + ee;
                const double P = (double)s[0] * 65536.0 + (double)s[I8_EIG] * 256.0 + (double)s[2 * I8_EIG];     // exact
                float *dst = gp.Xr + row * gp.ldx + col;
                const double base = accum ? (double)(*dst) : (double)gp.v0[row] * gp.colsum[col];
                *dst = (float)fma((double)gp.dx[row] * gp.wscale[col], P, base);
            }
        }
        __syncthreads();
    }
    if (tn == gp.tiles_n - 1) {                    // the pad columns [n, ldx) of the rotated rows are zero (the association kernel reads whole rows)
        const int padw = (int)(gp.ldx - gp.n);
        for (int idx = tid; idx < 256 * padw; idx += 512) {
            const long long row = m0 + idx / padw;
            if (row < gp.p) gp.Xr[row * gp.ldx + gp.n + idx % padw] = 0.0f;
        }
    }
}

// per-eigenvector exponent and the three int8 digit planes of U (see rotate_geno_i8_kernel)
__global__ __launch_bounds__(256) void colmax_kernel(long long n, long long ldU, const float *U, int *key)
{
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long k = (long long)blockIdx.x * 64 + tx, i0 = (long long)blockIdx.y * 256;
    if (k >= n) return;
    int m = 0;
    for (long long i = i0 + ty; i < i0 + 256 && i < n; i += 4) {
        const int bts = __float_as_int(U[i * ldU + k]) & 0x7FFFFFFF;
        m = bts > m ? bts : m;
    }
    atomicMax(&key[k], m);
}
__global__ void i8_scale_kernel(long long n, const int *key, float *qscale, double *wscale)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float mx = __int_as_float(key[k]);
    int E = ((key[k] >> 23) & 0xFF) - 127;                 // floor(log2 max)
    if (E == 128) { qscale[k] = 0.0f; wscale[k] = __longlong_as_double(0x7FF8000000000000LL); return; }   // a NaN / Inf in eigenvector k: its outputs are NaN, as in the fp16 and fp32 kernels
    if (!(mx > 0.0f) || E < -100) { qscale[k] = 1.0f; wscale[k] = 1.0; return; }     // an all-zero (or vanishing) column: q = 0
    if (ldexpf(mx, 22 - E) > 8355711.0f) E++;              // 127 * 65536 + 127 * 256 + 127: the top digit must stay <= 127
    qscale[k] = ldexpf(1.0f, 22 - E);
    wscale[k] = ldexp(1.0, E - 22);
}
__global__ __launch_bounds__(256) void split_u_i8_kernel(long long n, long long ldU, const float *U, signed char *Up8, long long ldk8, const float *qscale)
{
    __shared__ float tile[64][65];
    const long long k0 = (long long)blockIdx.x * 64, i0 = (long long)blockIdx.y * 64;   // eigen index block, sample block
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const long long i = i0 + r, k = k0 + tx;
        tile[r][tx] = (i < n && k < n) ? U[i * ldU + k] : 0.0f;
    }
    __syncthreads();
    if (i0 + tx >= ldk8) return;
    for (int r = ty; r < 64; r += 4) {      // r: eigen index within the block, tx: sample
        const long long k = k0 + r;
        if (k >= n) continue;
        const int qv = (int)rintf(tile[tx][r] * qscale[k]);          // the product is exact (power of two), |q| <= 127.5 * 65536
        const int d0 = ((qv + 128) & 255) - 128, q1 = (qv - d0) >> 8;
        const int d1 = ((q1 + 128) & 255) - 128, d2 = (q1 - d1) >> 8;
        signed char *dst = Up8 + ((k / I8_EIG) * 256 + k % I8_EIG) * ldk8 + i0 + tx;
        dst[0] = (signed char)d2; dst[(long long)I8_EIG * ldk8] = (signed char)d1; dst[(long long)2 * I8_EIG * ldk8] = (signed char)d0;
    }
}
}  // namespace pg

using namespace pg;

// Layout of the prepared U (one allocation of pg_geno_prep_bytes(n) bytes): fp16 planes | per-K-tile and final fp64 column sums |
// {S, 1/S} + max|U| key | int8 digit planes (whole 256-row tiles of 85 eigen indices) | 2^(E_k - 22) fp64 | 2^(22 - E_k) fp32 | max keys
struct PrepLayout {
    long long kt, ldp, kt8, ldk8, tiles8;
    size_t sums, scale, planes8, wscale, qscale, keys, total;
};
static PrepLayout prep_layout(long long n)
{
    PrepLayout L;
    L.kt = (n + GBK - 1) / GBK; L.ldp = L.kt * 2 * GBK;
    L.kt8 = (n + GBK8 - 1) / GBK8; L.ldk8 = L.kt8 * GBK8;
    L.tiles8 = (n + I8_EIG - 1) / I8_EIG;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    L.sums = up((size_t)n * L.ldp * 2);
    L.scale = L.sums + (size_t)(L.kt + 1) * n * 8;
    L.planes8 = up(L.scale + 256);
    L.wscale = up(L.planes8 + (size_t)L.tiles8 * 256 * L.ldk8);
    L.qscale = up(L.wscale + (size_t)n * 8);
    L.keys = up(L.qscale + (size_t)n * 4);
    L.total = up(L.keys + (size_t)n * 4);
    return L;
}
// PG_GENO_I8=0 keeps genotype codes on the fp16 x 2 kernel (A/B, tests); read per call
static bool geno_i8_enabled()
{
    const char *e = getenv("PG_GENO_I8");
    return !(e && atoi(e) == 0);
}

// Prepare U once per eigendecomposition: two fp16 planes of S*U + fp64 column sums + {S, 1/S}, and the three int8 digit planes with
// their per-eigenvector scales.  Uprep must hold pg_geno_prep_bytes(n) bytes.
extern "C" size_t pg_geno_prep_bytes(int64_t n)
{
    return prep_layout(n).total;
}
extern "C" int pg_geno_prep_dev(pg_ctx *ctx, int64_t n, const float *U, int64_t ldU, void *Uprep)
{
    PG_REQUIRE(ctx && U && Uprep && n > 0 && ldU >= n, "pg_geno_prep_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    const PrepLayout L = prep_layout(n);
    const long long kt = L.kt, ldp = L.ldp;
    unsigned short *Up = (unsigned short *)Uprep;
    double *colsum = (double *)((char *)Uprep + L.sums);
    float *scale = (float *)((char *)Uprep + L.scale);
    int *key = (int *)(scale + 2);
    PG_HIP(hipMemsetAsync(key, 0, 4, ctx->stream));
    absmax_kernel<<<1024, 256, 0, ctx->stream>>>(n, ldU, U, key);
    scale_kernel<<<1, 1, 0, ctx->stream>>>(key, scale);
    split_u_kernel<<<dim3((unsigned)((n + 63) / 64), (unsigned)kt), 256, 0, ctx->stream>>>(n, ldU, U, Up, ldp, colsum, scale);
    colsum_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(n, (int)kt, colsum);
    // int8 digit planes: pad rows (row 255 of every tile, eigen indices past n) and pad samples stay zero
    signed char *Up8 = (signed char *)Uprep + L.planes8;
    double *wscale = (double *)((char *)Uprep + L.wscale);
    float *qscale = (float *)((char *)Uprep + L.qscale);
    int *keys = (int *)((char *)Uprep + L.keys);
    PG_HIP(hipMemsetAsync(Up8, 0, (size_t)L.tiles8 * 256 * L.ldk8, ctx->stream));
    PG_HIP(hipMemsetAsync(keys, 0, (size_t)n * 4, ctx->stream));
    colmax_kernel<<<dim3((unsigned)((n + 63) / 64), (unsigned)((n + 255) / 256)), 256, 0, ctx->stream>>>(n, ldU, U, keys);
    i8_scale_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(n, keys, qscale, wscale);
    split_u_i8_kernel<<<dim3((unsigned)((n + 63) / 64), (unsigned)((L.ldk8 + 63) / 64)), 256, 0, ctx->stream>>>(n, ldU, U, Up8, L.ldk8, qscale);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// Pass 1 on the int8 kernel (Gt holds int8 codes, row stride ldk8 bytes)
static int launch_geno_i8(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const void *Gt8, const float *v0, const float *dx, float *Xr,
                          int64_t ldx, const int *cond, int cmode)
{
    const PrepLayout L = prep_layout(n);
    GenoI8Params gp{};
    gp.n = n; gp.p = p; gp.ldx = ldx; gp.ldk8 = L.ldk8;
    gp.Gt8 = (const signed char *)Gt8; gp.Up8 = (const signed char *)Uprep + L.planes8;
    gp.colsum = (const double *)((const char *)Uprep + L.sums) + (size_t)L.kt * n;
    gp.wscale = (const double *)((const char *)Uprep + L.wscale);
    gp.v0 = v0; gp.dx = dx; gp.Xr = Xr;
    gp.tiles_m = (int)((p + 255) / 256); gp.tiles_n = (int)L.tiles8; gp.KT = (int)L.kt8;
    // chunks of n-tiles whose plane rows fit PG_GENO_CHUNK_MB (default 100 MB of the 256 MB memory-side cache; 0: one chunk)
    const char *ce = getenv("PG_GENO_CHUNK_MB");
    const long long chunk_mb = ce ? atoll(ce) : 100;
    long long ncmax = chunk_mb > 0 ? (chunk_mb << 20) / (256 * L.ldk8) : L.tiles8;
    ncmax = ncmax < 8 ? 8 : ncmax;
    gp.nchunks = (int)((L.tiles8 + ncmax - 1) / ncmax);
    gp.nc_base = (int)(L.tiles8 / gp.nchunks); gp.nc_rem = (int)(L.tiles8 % gp.nchunks);
    long long T = 0;
    for (int c = 0; c < gp.nchunks; c++) T += ((long long)gp.tiles_m * (gp.nc_base + (c < gp.nc_rem ? 1 : 0)) + 7) / 8 * 8;
    PG_REQUIRE(T < (1LL << 31) && L.ldk8 < (1LL << 24), "genotype rotation: too many tiles");      // 32-bit lane offsets: 256 rows x ldk8 bytes
    constexpr int WLDS = 5 * 256 * 128;
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rotate_geno_i8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WLDS));
    gp.cond = cond; gp.cmode = cond ? cmode : COND_ALWAYS;
    rotate_geno_i8_kernel<<<dim3((unsigned)T), 512, WLDS, ctx->stream>>>(gp);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// i8codes: Gt / Gi hold int8 genotype codes / indicators for the int8 kernel (the caller encoded them that way: geno_i8_enabled());
// with block flags (cond) the fp16 kernel still takes both passes of a split-plane block, whose planes have overwritten Gt and Gi.
static int launch_geno_gemm(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const unsigned short *Gt, const unsigned short *Gi,
                            const float *v0, const float *dx, const float *dlt, float *Xr, int64_t ldx, bool second_pass, const int *cond = nullptr,
                            bool i8codes = false)
{
    const PrepLayout L = prep_layout(n);
    const long long kt = L.kt, ldk = kt * GBK, ldp = L.ldp;
    if (i8codes) {      // genotype block: codes, then (if any column holds an imputed value) the indicator plane, both int8
        int rc = launch_geno_i8(ctx, n, p, Uprep, Gt, v0, dx, Xr, ldx, cond, COND_GENO);
        if (rc) return rc;
        if (second_pass) {
            rc = launch_geno_i8(ctx, n, p, Uprep, Gi, nullptr, dlt, Xr, ldx, cond, COND_INDICATOR);
            if (rc) return rc;
        }
        if (!cond) return PG_OK;
    }
    GenoParams gp{};
    gp.n = n; gp.p = p; gp.ldx = ldx; gp.ldk = ldk; gp.ldp = ldp;
    gp.Gt = Gt; gp.Up = (const unsigned short *)Uprep;
    const double *sums = (const double *)((const char *)Uprep + L.sums);
    gp.colsum = sums + (size_t)kt * n;
    gp.scale = (const float *)((const char *)Uprep + L.scale);
    gp.v0 = v0; gp.dx = dx; gp.Xr = Xr;
    constexpr int WLDS = 5 * 256 * 128;
    // per device (the attribute belongs to the function object of the current device; contexts of several GPUs share this process)
    // PG_GENO_MFMA=32 selects the 32 x 32 x 16 instantiation (A/B, tests).  With operands in registers the matrix pipe sustains 1 800 TF
    // on that shape against 1 285 on 16 x 16 x 32 (tools/probe_mfma_sustained.hip; one wave per SIMD alone: 1 780 against 900), but in this
    // kernel it is the slower one: 38.9 ms against 31.2 per 100 000 SNPs, MFMA pipe 0.56 busy against 0.74 (tools/ab_rotate_mfma.sh).
    // A wave of 32 x 32 MFMAs saturates the pipe by itself, so the two waves of a SIMD can no longer fill each other's issue gaps and the
    // memory phase (LDS-DMA of 48 KB per stage and CU) is exposed instead of hidden.
    const bool mf16 = !(getenv("PG_GENO_MFMA") && atoi(getenv("PG_GENO_MFMA")) == 32);
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rotate_geno_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, WLDS));
    PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rotate_geno_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, WLDS));
    gp.tiles_m = (int)((p + 255) / 256); gp.tiles_n = (int)((n + 255) / 256); gp.KT = (int)kt;
    const long long T = (long long)gp.tiles_m * gp.tiles_n;
    PG_REQUIRE(T < (1LL << 31), "genotype rotation: too many tiles");
    gp.cond = cond; gp.cmode = cond ? (i8codes ? COND_SPLIT : COND_PASS1) : COND_ALWAYS;
    if (mf16) rotate_geno_kernel<16><<<dim3((unsigned)T), 512, WLDS, ctx->stream>>>(gp);
    else rotate_geno_kernel<32><<<dim3((unsigned)T), 512, WLDS, ctx->stream>>>(gp);
    if (second_pass) {   // Xr += delta * U'ind (or the second plane of a split block)
        gp.Gt = Gi; gp.v0 = nullptr; gp.dx = dlt; gp.cmode = cond ? (i8codes ? COND_SPLIT : COND_PASS2) : COND_ALWAYS;
        if (mf16) rotate_geno_kernel<16><<<dim3((unsigned)T), 512, WLDS, ctx->stream>>>(gp);
        else rotate_geno_kernel<32><<<dim3((unsigned)T), 512, WLDS, ctx->stream>>>(gp);
    }
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// Rotate a block of genotype columns.  work must hold pg_geno_work_bytes(n, p).  *is_geno_host receives 1 when the block
// qualified and Xr was written, 0 when it did not (Xr untouched: the caller falls back to pg_rotate_dev).  Synchronises.
extern "C" size_t pg_geno_work_bytes(int64_t n, int64_t p)
{
    const long long ldk = (n + GBK - 1) / GBK * GBK;
    return 2 * (((size_t)p * ldk * 2 + 255) & ~(size_t)255) + (size_t)p * 24 + 512;   // codes, indicator plane, v0/dx/delta/min/max/other
}
template <class T>
static int rotate_geno_any(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const T *X, int64_t ldX, float *Xr, int64_t ldx,
                           void *work, int *is_geno_host)
{
    PG_REQUIRE(ctx && Uprep && X && Xr && work && is_geno_host, "pg_rotate_geno_dev: NULL argument");
    PG_REQUIRE(n > 0 && p > 0 && ldX >= p && ldx >= n && ldx <= (n + 127) / 128 * 128, "pg_rotate_geno_dev: bad shape");
    PG_HIP(hipSetDevice(ctx->device));
    const long long kt = (n + GBK - 1) / GBK, ldk = kt * GBK, ldp = kt * 2 * GBK;
    const size_t plane = ((size_t)p * ldk * 2 + 255) & ~(size_t)255;
    unsigned short *Gt = (unsigned short *)work, *Gi = (unsigned short *)((char *)work + plane);
    char *tail = (char *)work + 2 * plane;
    float *v0 = (float *)tail, *dx = v0 + p, *dlt = dx + p;
    int *kmin = (int *)(dlt + p), *kmax = kmin + p, *other = kmax + p;
    int *flag = other + p;           // flag[0]: not a genotype block; flag[1]: some column holds an other (imputed) value
    PG_HIP(hipMemsetAsync(flag, 0, 8, ctx->stream));
    minmax_init_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, kmin, kmax, other);
    minmax_geno_kernel<T><<<dim3((unsigned)((p + 63) / 64), (unsigned)((n + 255) / 256)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, flag);
    const bool i8 = geno_i8_enabled();
    const long long ldk8 = prep_layout(n).ldk8;
    if (i8) encode_geno_kernel<T, true><<<dim3((unsigned)((p + 63) / 64), (unsigned)(ldk8 / 128)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, other, Gt, ldk8, flag);
    else encode_geno_kernel<T><<<dim3((unsigned)((p + 63) / 64), (unsigned)((ldk + 127) / 128)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, other, Gt, ldk, flag);
    params_geno_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, kmin, kmax, other, v0, dx, dlt);
    PG_HIP(hipGetLastError());
    int hflag[2] = {0, 0};
    PG_HIP(hipMemcpyAsync(hflag, flag, 8, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    if (hflag[0] & 2) { *is_geno_host = 0; return PG_OK; }     // NaN / Inf: the fp32 kernel's propagation is the reference's
    if (hflag[0] & 1) {                                        // finite, not genotype-valued: X itself in two fp16 planes
        *is_geno_host = 2;
        split_x_kernel<T><<<dim3((unsigned)((p + 63) / 64), (unsigned)((ldk + 127) / 128)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, Gt, Gi, ldk);
        params_split_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, kmin, kmax, v0, dx, dlt);
        PG_HIP(hipGetLastError());
        return launch_geno_gemm(ctx, n, p, Uprep, Gt, Gi, v0, dx, dlt, Xr, ldx, true);
    }
    *is_geno_host = 1;
    if (hflag[1]) {
        if (i8) indicator_geno_kernel<T, true><<<dim3((unsigned)((p + 31) / 32), (unsigned)(ldk8 / 32)), 256, 0, ctx->stream>>>(n, p, X, ldX, other, Gi, ldk8);
        else indicator_geno_kernel<T><<<dim3((unsigned)((p + 31) / 32), (unsigned)((ldk + 31) / 32)), 256, 0, ctx->stream>>>(n, p, X, ldX, other, Gi, ldk);
    }
    return launch_geno_gemm(ctx, n, p, Uprep, Gt, Gi, v0, dx, dlt, Xr, ldx, hflag[1] != 0, nullptr, i8);
}

extern "C" int pg_rotate_geno_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const float *X, int64_t ldX, float *Xr, int64_t ldx,
                                  void *work, int *is_geno_host)
{
    return rotate_geno_any<float>(ctx, n, p, Uprep, X, ldX, Xr, ldx, work, is_geno_host);
}

namespace pg {
__global__ void path_code_kernel(const int *flag, int *path)
{
    const int f0 = flag[0];
    path[0] = (f0 & 2) ? 0 : ((f0 & 1) ? 2 : 1);
}
int rotate_fp32_cond(pg_ctx *ctx, long long n, long long p, const float *U, long long ldU, const float *X, long long ldX, float *Xr,
                     long long ldx, const int *cond, int cmode);      // rotate.hip
}  // namespace pg

// The rotation of a float32 block without a host decision: the detect pass leaves its flags on the device and every candidate
// kernel (genotype GEMM, indicator pass, split planes + second pass, fp32-MFMA fallback for NaN/Inf blocks) is enqueued
// predicated on them (run_cond) — the path that applies does the work, the others leave at once.  Same outputs as
// pg_rotate_geno_dev followed by the caller's fallback to pg_rotate_dev, but nothing synchronises the stream (the flag read-back
// of pg_rotate_geno_dev costs ~0.7 ms of idle GPU per 16 384-SNP block at n = 10 000).  path_dev (optional, device int) receives
// 1 / 2 / 0 like *is_geno.
template <class T>
static int rotate_auto_any(pg_ctx *ctx, int64_t n, int64_t p, const float *U, int64_t ldU, const void *Uprep, const T *X, int64_t ldX,
                           float *Xr, int64_t ldx, void *work, int *path_dev)
{
    PG_REQUIRE(ctx && Uprep && X && Xr && work, "pg_rotate_auto_dev: NULL argument");
    PG_REQUIRE(n > 0 && p > 0 && ldX >= p && ldx >= n && ldx <= (n + 127) / 128 * 128, "pg_rotate_auto_dev: bad shape");
    PG_HIP(hipSetDevice(ctx->device));
    const long long kt = (n + GBK - 1) / GBK, ldk = kt * GBK;
    const size_t plane = ((size_t)p * ldk * 2 + 255) & ~(size_t)255;
    unsigned short *Gt = (unsigned short *)work, *Gi = (unsigned short *)((char *)work + plane);
    char *tail = (char *)work + 2 * plane;
    float *v0 = (float *)tail, *dx = v0 + p, *dlt = dx + p;
    int *kmin = (int *)(dlt + p), *kmax = kmin + p, *other = kmax + p;
    int *flag = other + p;
    PG_HIP(hipMemsetAsync(flag, 0, 8, ctx->stream));
    minmax_init_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, kmin, kmax, other);
    minmax_geno_kernel<T><<<dim3((unsigned)((p + 63) / 64), (unsigned)((n + 255) / 256)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, flag);
    const bool i8 = geno_i8_enabled();
    const long long ldk8 = prep_layout(n).ldk8;
    if (i8) encode_geno_kernel<T, true><<<dim3((unsigned)((p + 63) / 64), (unsigned)(ldk8 / 128)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, other, Gt, ldk8, flag);
    else encode_geno_kernel<T><<<dim3((unsigned)((p + 63) / 64), (unsigned)((ldk + 127) / 128)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, other, Gt, ldk, flag);
    params_geno_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, kmin, kmax, other, v0, dx, dlt);
    // genotype block with an imputed value: indicator plane; finite non-genotype block: X in two fp16 planes (overwrites Gt, Gi, params)
    if (i8) indicator_geno_kernel<T, true><<<dim3((unsigned)((p + 31) / 32), (unsigned)(ldk8 / 32)), 256, 0, ctx->stream>>>(n, p, X, ldX, other, Gi, ldk8, flag);
    else indicator_geno_kernel<T><<<dim3((unsigned)((p + 31) / 32), (unsigned)((ldk + 31) / 32)), 256, 0, ctx->stream>>>(n, p, X, ldX, other, Gi, ldk, flag);
    split_x_kernel<T><<<dim3((unsigned)((p + 63) / 64), (unsigned)((ldk + 127) / 128)), 256, 0, ctx->stream>>>(n, p, X, ldX, kmin, kmax, Gt, Gi, ldk, flag);
    params_split_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, kmin, kmax, v0, dx, dlt, flag);
    PG_HIP(hipGetLastError());
    int rc = launch_geno_gemm(ctx, n, p, Uprep, Gt, Gi, v0, dx, dlt, Xr, ldx, true, flag, i8);
    if (rc) return rc;
    if constexpr (std::is_same<T, float>::value) {       // NaN / Inf block (float input only): the reference's propagation
        PG_REQUIRE(U && ldU >= n, "pg_rotate_auto_dev: U is needed for the fp32 fallback");
        rc = rotate_fp32_cond(ctx, n, p, U, ldU, X, ldX, Xr, ldx, flag, COND_FP32);
        if (rc) return rc;
    }
    if (path_dev) {
        path_code_kernel<<<1, 1, 0, ctx->stream>>>(flag, path_dev);
        PG_HIP(hipGetLastError());
    }
    return PG_OK;
}
extern "C" int pg_rotate_auto_dev(pg_ctx *ctx, int64_t n, int64_t p, const float *U, int64_t ldU, const void *Uprep, const float *X, int64_t ldX,
                                  float *Xr, int64_t ldx, void *work, int *path_dev)
{
    return rotate_auto_any<float>(ctx, n, p, U, ldU, Uprep, X, ldX, Xr, ldx, work, path_dev);
}
// 8-bit X (always finite): genotype path or split planes, nothing else; no host read-back either
extern "C" int pg_rotate_auto_i8_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const void *X8, int is_unsigned, int64_t ldX,
                                     float *Xr, int64_t ldx, void *work, int *path_dev)
{
    return is_unsigned ? rotate_auto_any<unsigned char>(ctx, n, p, nullptr, 0, Uprep, (const unsigned char *)X8, ldX, Xr, ldx, work, path_dev)
                       : rotate_auto_any<signed char>(ctx, n, p, nullptr, 0, Uprep, (const signed char *)X8, ldX, Xr, ldx, work, path_dev);
}

// The same for X held as 8-bit integers (genotype matrices are often stored that way; the reference casts any dtype to
// float32, lmm/lmm.py:121-122, so the values are the same): 4x fewer bytes to upload and to scan.
extern "C" int pg_rotate_geno_i8_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const void *X8, int is_unsigned, int64_t ldX,
                                     float *Xr, int64_t ldx, void *work, int *is_geno_host)
{
    return is_unsigned ? rotate_geno_any<unsigned char>(ctx, n, p, Uprep, (const unsigned char *)X8, ldX, Xr, ldx, work, is_geno_host)
                       : rotate_geno_any<signed char>(ctx, n, p, Uprep, (const signed char *)X8, ldX, Xr, ldx, work, is_geno_host);
}

// ... and for float64 X (numpy's default dtype; the reference's X.astype(np.float32), lmm/lmm.py:121-122, is a round-to-nearest
// conversion per element — the same one the kernels apply when they read the block): no host-side copy of the matrix.
extern "C" int pg_rotate_geno_f64_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const double *X, int64_t ldX, float *Xr, int64_t ldx,
                                      void *work, int *is_geno_host)
{
    return rotate_geno_any<double>(ctx, n, p, Uprep, X, ldX, Xr, ldx, work, is_geno_host);
}
extern "C" int pg_cast_f64_f32_dev(pg_ctx *ctx, int64_t n, int64_t p, const double *X, int64_t ldX, float *Xf, int64_t ldXf)
{
    PG_REQUIRE(ctx && X && Xf && n > 0 && p > 0 && ldX >= p && ldXf >= p, "pg_cast_f64_f32_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((p + 255) / 256), (unsigned)(n < 65535 ? n : 65535));
    cast_to_f32_kernel<double><<<grid, 256, 0, ctx->stream>>>(n, p, X, ldX, Xf, ldXf);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// float32 image of an 8-bit block (the fallback input of pg_rotate_dev / pg_transpose_dev for blocks that do not qualify)
extern "C" int pg_cast_i8_f32_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *X8, int is_unsigned, int64_t ldX, float *Xf, int64_t ldXf)
{
    PG_REQUIRE(ctx && X8 && Xf && n > 0 && p > 0 && ldX >= p && ldXf >= p, "pg_cast_i8_f32_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((p + 255) / 256), (unsigned)(n < 65535 ? n : 65535));
    if (is_unsigned) cast_to_f32_kernel<unsigned char><<<grid, 256, 0, ctx->stream>>>(n, p, (const unsigned char *)X8, ldX, Xf, ldXf);
    else cast_to_f32_kernel<signed char><<<grid, 256, 0, ctx->stream>>>(n, p, (const signed char *)X8, ldX, Xf, ldXf);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// Rotate a block of PLINK .bed genotypes (device copy of the packed bytes: p rows of ldb >= ceil(n/4) bytes, SNP-major as in
// the file).  Missing calls take the mean of the called genotypes of their SNP.  Same outputs as pg_rotate_geno_dev.
extern "C" int pg_rotate_bed_dev(pg_ctx *ctx, int64_t n, int64_t p, const void *Uprep, const unsigned char *bed, int64_t ldb, int count_a1,
                                 float *Xr, int64_t ldx, void *work)
{
    PG_REQUIRE(ctx && Uprep && bed && Xr && work, "pg_rotate_bed_dev: NULL argument");
    PG_REQUIRE(n > 0 && p > 0 && ldb >= (n + 3) / 4 && ldx >= n && ldx <= (n + 127) / 128 * 128, "pg_rotate_bed_dev: bad shape");
    PG_HIP(hipSetDevice(ctx->device));
    const long long kt = (n + GBK - 1) / GBK, ldk = kt * GBK;
    const size_t plane = ((size_t)p * ldk * 2 + 255) & ~(size_t)255;
    unsigned short *Gt = (unsigned short *)work, *Gi = (unsigned short *)((char *)work + plane);
    char *tail = (char *)work + 2 * plane;
    float *v0 = (float *)tail, *dx = v0 + p, *dlt = dx + p;
    int *flag = (int *)(dlt + p) + 3 * p;
    PG_HIP(hipMemsetAsync(flag, 0, 8, ctx->stream));
    const bool i8 = geno_i8_enabled();
    const long long ldk8 = prep_layout(n).ldk8;
    if (i8) decode_bed_kernel<true><<<(unsigned)p, 256, 0, ctx->stream>>>(n, p, bed, ldb, count_a1, Gt, Gi, ldk, ldk8, v0, dx, dlt, flag);
    else decode_bed_kernel<false><<<(unsigned)p, 256, 0, ctx->stream>>>(n, p, bed, ldb, count_a1, Gt, Gi, ldk, ldk8, v0, dx, dlt, flag);
    PG_HIP(hipGetLastError());
    int hflag[2] = {0, 0};
    PG_HIP(hipMemcpyAsync(hflag, flag, 8, hipMemcpyDeviceToHost, ctx->stream));
    PG_HIP(hipStreamSynchronize(ctx->stream));
    return launch_geno_gemm(ctx, n, p, Uprep, Gt, Gi, v0, dx, dlt, Xr, ldx, hflag[1] != 0, nullptr, i8);
}

#ifdef PG_GENO_STAMPS
extern "C" int pgx_geno_stamps(long long *out192)
{
    return hipMemcpyFromSymbol(out192, HIP_SYMBOL(pg::g_geno_stamps), sizeof(long long) * 192) == hipSuccess ? 0 : -1;
}
#endif
