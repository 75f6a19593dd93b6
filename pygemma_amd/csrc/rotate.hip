// rotate.hip — H2: the eigen-rotation  X <- U' X  (lmm/lmm.py:243-246; OpenBLAS sgemm in the reference)
// as an fp32-MFMA GEMM on gfx950, writing the SNP-major layout the association kernels read.
//
//   Xr[g][k] = sum_i X[i][g] * U[i][k]        M = p (SNPs g), N = n (eigen index k), K = n (samples i)
//
// Both operands are K-major in memory (X is (n,p) row-major, U is (n,n) row-major with eigenvector k in
// column k), which is exactly the operand order v_mfma_f32_32x32x2_f32 wants from LDS tiles stored
// [k][m] / [k][n]: lane l reads A[k = l>>5][m = l&31] and B[k = l>>5][n = l&31] — 32 consecutive floats
// per half-wave, conflict-free ds_read_b32, and the global->LDS copies are straight 16-byte row segments.
// fp32 in / fp32 accumulate: bit-for-bit a k-ordered fmaf chain, i.e. the reference's own precision class.
//
// Tile: 128(M) x 128(N) x 16(K) per 256-thread workgroup, 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32;
// register-staged global loads for tile t+1 issued before the MFMAs of tile t, LDS double-buffered, one
// barrier per K-tile.  blockIdx -> tile map is XCD-aware (blocks b and b+8 share an XCD, so the bijective
// remap hands each XCD a contiguous run of tiles) and grouped 8 tile-rows deep so that the ~64 tiles an XCD
// works on at once share their X and U panels in its 4 MiB L2.
#include "common.hpp"

namespace pg {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int RBM = 128, RBN = 128, RBK = 16, RGROUP = 8;

// out[g][k] = scale * sum_{i<kdim} X[i*ldX + g] * U[i*ldU + k]   g < p (M), k < ncol (N); columns [ncol, ldx) zeroed
struct RotParams {
    long long kdim, ncol, p, ldx, ldU, ldX;
    const float *U, *X;
    float *Xr;
    float scale;
    int tiles_m, tiles_n;
    int lower;   // syrk mode (square output, X == U): only tiles on or below the diagonal are computed, the rest is mirrored
    const int *cond;   // pg_rotate_auto_dev: run only if (cond[0] & 2), i.e. the block holds a NaN/Inf (nullptr: always)
};

template <int VEC>
__device__ __forceinline__ float4 load4(const float *base, long long row, long long ld, long long col, long long nrow, long long ncol)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrow) {
        const float *p = base + row * ld + col;
        if (VEC == 4 && col + 3 < ncol) {
            v = *reinterpret_cast<const float4 *>(p);
        } else {
            if (col < ncol) v.x = p[0];
            if (col + 1 < ncol) v.y = p[1];
            if (col + 2 < ncol) v.z = p[2];
            if (col + 3 < ncol) v.w = p[3];
        }
    }
    return v;
}

template <int VEC>
__global__ __launch_bounds__(256, 2) void rotate_kernel(RotParams rp)
{
    __shared__ float As[2][RBK][RBM];
    __shared__ float Bs[2][RBK][RBN];
    if (rp.cond && !(rp.cond[0] & 2)) return;      // uniform over the grid
    // ---- XCD-aware, grouped tile order
    const int T = rp.lower ? rp.tiles_m * (rp.tiles_m + 1) / 2 : rp.tiles_m * rp.tiles_n;
    const int b = blockIdx.x;
    const int q = T / 8, r = T % 8, xcd = b % 8;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + b / 8;
    int tm, tn;
    if (rp.lower) {      // lid -> (tm, tn), tn <= tm, row-major over the lower triangle of the tile grid
        tm = (int)((sqrtf(8.0f * (float)lid + 1.0f) - 1.0f) * 0.5f);
        while (tm * (tm + 1) / 2 > lid) tm--;
        while ((tm + 1) * (tm + 2) / 2 <= lid) tm++;
        tn = lid - tm * (tm + 1) / 2;
    } else {
        const int per_group = RGROUP * rp.tiles_n;
        const int grp = lid / per_group;
        const int first_m = grp * RGROUP;
        const int gsz = (rp.tiles_m - first_m) < RGROUP ? (rp.tiles_m - first_m) : RGROUP;
        tm = first_m + (lid % per_group) % gsz;
        tn = (lid % per_group) / gsz;
    }
    const long long m0 = (long long)tm * RBM, n0 = (long long)tn * RBN;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = tid >> 5, lcol = (tid & 31) * 4;   // global->LDS: 8 rows x 32 float4 per pass

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    const int KT = (int)((rp.kdim + RBK - 1) / RBK);
    float4 ra[2], rb[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        ra[h] = load4<VEC>(rp.X, h * 8 + lrow, rp.ldX, m0 + lcol, rp.kdim, rp.p);
        rb[h] = load4<VEC>(rp.U, h * 8 + lrow, rp.ldU, n0 + lcol, rp.kdim, rp.ncol);
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
        *reinterpret_cast<float4 *>(&As[0][h * 8 + lrow][lcol]) = ra[h];
        *reinterpret_cast<float4 *>(&Bs[0][h * 8 + lrow][lcol]) = rb[h];
    }
    __syncthreads();
    for (int kt = 0; kt < KT; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < KT) {
            const long long k0 = (long long)(kt + 1) * RBK;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                ra[h] = load4<VEC>(rp.X, k0 + h * 8 + lrow, rp.ldX, m0 + lcol, rp.kdim, rp.p);
                rb[h] = load4<VEC>(rp.U, k0 + h * 8 + lrow, rp.ldU, n0 + lcol, rp.kdim, rp.ncol);
            }
        }
#pragma unroll
        for (int kk = 0; kk < RBK; kk += 2) {
            const int kr = kk + (lane >> 5);
            const float a0 = As[buf][kr][wm * 64 + (lane & 31)];
            const float a1 = As[buf][kr][wm * 64 + 32 + (lane & 31)];
            const float b0 = Bs[buf][kr][wn * 64 + (lane & 31)];
            const float b1 = Bs[buf][kr][wn * 64 + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < KT) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                *reinterpret_cast<float4 *>(&As[buf ^ 1][h * 8 + lrow][lcol]) = ra[h];
                *reinterpret_cast<float4 *>(&Bs[buf ^ 1][h * 8 + lrow][lcol]) = rb[h];
            }
        }
        __syncthreads();
    }
    // ---- C/D map of the 32x32 f32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const long long col = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const long long row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < rp.p && col < rp.ldx) {
                    const float v = rp.scale == 1.0f ? acc[i][j][e] : rp.scale * acc[i][j][e];
                    rp.Xr[row * rp.ldx + col] = v;
                    if (rp.lower && tn < tm && col < rp.ncol) rp.Xr[col * rp.ldx + row] = v;   // mirror of an off-diagonal tile
                }
            }
        }
}

}  // namespace pg

using namespace pg;

static int launch_tn(pg_ctx *ctx, long long kdim, long long ncol, long long p, const float *U, long long ldU, const float *X,
                     long long ldX, float *out, long long ldx, float scale, bool lower = false, const int *cond = nullptr)
{
    static_assert(RBM == RBN, "the syrk tile enumeration assumes square tiles");
    RotParams rp{};
    rp.kdim = kdim; rp.ncol = ncol; rp.p = p; rp.ldx = ldx; rp.ldU = ldU; rp.ldX = ldX; rp.U = U; rp.X = X; rp.Xr = out; rp.scale = scale;
    rp.tiles_m = (int)((p + RBM - 1) / RBM);
    rp.tiles_n = (int)((ncol + RBN - 1) / RBN);
    rp.lower = lower ? 1 : 0;
    rp.cond = cond;
    const long long T = lower ? (long long)rp.tiles_m * (rp.tiles_m + 1) / 2 : (long long)rp.tiles_m * rp.tiles_n;
    PG_REQUIRE(T < (1LL << 31), "rotate: too many tiles; process SNPs in batches");
    const bool vec = (ldU % 4 == 0) && (ldX % 4 == 0) && (((uintptr_t)U | (uintptr_t)X) % 16 == 0);
    if (vec) rotate_kernel<4><<<dim3((unsigned)T), 256, 0, ctx->stream>>>(rp);
    else rotate_kernel<1><<<dim3((unsigned)T), 256, 0, ctx->stream>>>(rp);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

namespace pg {
int rotate_fp32_cond(pg_ctx *ctx, long long n, long long p, const float *U, long long ldU, const float *X, long long ldX, float *Xr,
                     long long ldx, const int *cond, int)
{
    return launch_tn(ctx, n, n, p, U, ldU, X, ldX, Xr, ldx, 1.0f, false, cond);
}
}  // namespace pg

extern "C" int pg_rotate_dev(pg_ctx *ctx, int64_t n, int64_t p, const float *U, int64_t ldU, const float *X, int64_t ldX,
                             float *Xr, int64_t ldx)
{
    PG_REQUIRE(ctx && U && X && Xr, "pg_rotate_dev: NULL argument");
    PG_REQUIRE(n > 0 && p > 0 && ldU >= n && ldX >= p && ldx >= n && ldx <= (n + 127) / 128 * 128,
               "pg_rotate_dev: bad shape n=%lld p=%lld ldU=%lld ldX=%lld ldx=%lld (need n <= ldx <= roundup(n,128))",
               (long long)n, (long long)p, (long long)ldU, (long long)ldX, (long long)ldx);
    PG_HIP(hipSetDevice(ctx->device));
    return launch_tn(ctx, n, n, p, U, ldU, X, ldX, Xr, ldx, 1.0f);
}

extern "C" int pg_kinship_dev(pg_ctx *ctx, int64_t n, int64_t p_k, const float *Gt, int64_t ldg, float *K)
{
    PG_REQUIRE(ctx && Gt && K && n > 0 && p_k > 0 && ldg >= n, "pg_kinship_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    // K[a][b] = (1/p_k) sum_g Gt[g][a] Gt[g][b]: a syrk — tiles on or below the diagonal on the MFMA pipe, the rest mirrored
    // (bit-symmetric: K[a][b] and K[b][a] are the same fma chain over g of commuting products)
    return launch_tn(ctx, p_k, n, n, Gt, ldg, Gt, ldg, K, n, (float)(1.0 / (double)p_k), true);
}

namespace pg {
// per-column mean and 1/sd of G (n x p row-major), fp64: mu = mean, sd = sqrt(mean((x - mu)^2)) (numpy's np.std, ddof = 0),
// sd == 0 -> 1 (experiments/animal_gwas/run_gwas.py:46-49).  blockDim (64, 4): 64 columns per block, rows strided over y.
constexpr int CS_ROWG = 16;     // row groups per 64-column block (4: 2.07 ms at n = 10 000, p = 20 000: 2 x 2 500 dependent loads per thread)
__global__ __launch_bounds__(64 * CS_ROWG) void colstats_kernel(long long n, long long p, const float *G, long long ldG, double *mu, double *isd)
{
    __shared__ double red[CS_ROWG][64];
    const long long g = (long long)blockIdx.x * 64 + threadIdx.x;
    const int ty = threadIdx.y;
    auto total = [&]() { double t = red[0][threadIdx.x]; for (int k = 1; k < CS_ROWG; k++) t += red[k][threadIdx.x]; return t; };
    double s = 0.0;
    if (g < p) for (long long i = ty; i < n; i += CS_ROWG) s += (double)G[i * ldG + g];
    red[ty][threadIdx.x] = s;
    __syncthreads();
    const double m = total() / (double)n;
    __syncthreads();
    double v = 0.0;
    if (g < p) for (long long i = ty; i < n; i += CS_ROWG) { const double t = (double)G[i * ldG + g] - m; v = fma(t, t, v); }
    red[ty][threadIdx.x] = v;
    __syncthreads();
    if (ty == 0 && g < p) {
        const double var = total() / (double)n;
        double sd = sqrt(var);
        if (sd == 0.0) sd = 1.0;
        mu[g] = m; isd[g] = 1.0 / sd;
    }
}
__global__ __launch_bounds__(256) void standardize_t_kernel(long long n, long long p, const float *G, long long ldG, const double *mu,
                                                            const double *isd, float *Zt, long long ldz)
{
    __shared__ float tile[32][33];
    const long long g0 = (long long)blockIdx.x * 32, i0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long long i = i0 + r, g = g0 + tx;
        float v = 0.0f;
        if (i < n && g < p) {
            const double x = (double)G[i * ldG + g];
            v = mu ? (float)((x - mu[g]) * isd[g]) : (float)x;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long long g = g0 + r, i = i0 + tx;
        if (g < p && i < ldz) Zt[g * ldz + i] = tile[tx][r];
    }
}
}  // namespace pg

// N3: K = Z Z' / p straight from the (n x p) genotype matrix as the reference's callers build it
// (experiments/animal_gwas/run_gwas.py:46-56): column standardisation on the device (fp64 mean and population sd, sd == 0 -> 1),
// then the lower-triangle syrk above.  K (n x n row-major float32, both triangles) can feed pg_syevd_dev directly.
extern "C" int pg_kinship_geno_dev(pg_ctx *ctx, int64_t n, int64_t p, const float *G, int64_t ldG, int standardize, float *K)
{
    PG_REQUIRE(ctx && G && K && n > 0 && p > 0 && ldG >= p, "pg_kinship_geno_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    const long long ldz = (n + 63) / 64 * 64;
    const size_t off_stats = ((size_t)p * ldz * 4 + 255) & ~(size_t)255;
    int rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, off_stats + (size_t)p * 16);
    if (rc) return rc;
    float *Zt = (float *)ctx->scratch;
    double *mu = (double *)((char *)ctx->scratch + off_stats), *isd = mu + p;
    if (standardize) {
        colstats_kernel<<<dim3((unsigned)((p + 63) / 64)), dim3(64, CS_ROWG), 0, ctx->stream>>>(n, p, G, ldG, mu, isd);
        PG_HIP(hipGetLastError());
    }
    dim3 grid((unsigned)((p + 31) / 32), (unsigned)((ldz + 31) / 32));
    standardize_t_kernel<<<grid, 256, 0, ctx->stream>>>(n, p, G, ldG, standardize ? mu : nullptr, isd, Zt, ldz);
    PG_HIP(hipGetLastError());
    return pg_kinship_dev(ctx, n, p, Zt, ldz, K);
}
