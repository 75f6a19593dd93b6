// assoc.hip — the per-SNP LMM association operator on gfx950 (MI355X).
//
// Replaces, per SNP g, everything under calculate() in the reference (lmm/lmm.py:461-495):
//   calc_lambda_restricted (pygemma_model.pyx:64-194) -> precompute_mat (:880-1053), newton (:1349-1416),
//   scipy.optimize.brentq (:176-182), the *_overload scalars (:1656-1698, :1813-1830),
//   calc_beta_vg_ve_restricted_overload (:1514-1537), F = (beta/se)^2 and scipy.stats.f.sf (lmm.py:471,482).
//
// Mapping: ONE 64-lane wavefront per SNP.  Lane l owns elements i = l, l+64, ... of the n-vectors
// (coalesced reads of the SNP-major genotype row and of the packed fixed rows {d_i, w_i*, y_i});
// every Gram entry is a per-lane fma chain followed by a 64-lane xor-butterfly (strides 1,2,..,32),
// so all lanes hold the same f64 sums and run the O(m^3) sweeps / Brent / Newton logic uniformly.
// Precision follows SURVEY.md Appendix A: h = 1/(lam*d+1) in three separately rounded f32 ops,
// Grams and sweeps in f64, every exported quadratic form rounded to f32, f32 lambda iterates.
// The arithmetic order is mirrored exactly by oracle/pygemma_oracle.c (order=1) so that GPU and
// oracle agree bit-for-bit; compile with -ffp-contract=off (every fma below is explicit).
//
// What is shared between SNPs is hoisted: the decade lambdas 10^-5..10^5 are the same for every SNP,
// so h tables, the SNP-independent Gram entries (W,y block), trace terms and log|H| at those 11
// lambdas are computed once (setup kernel); per SNP the scan only accumulates the 2(c+2) entries
// that involve x.  Evaluations at SNP-specific lambdas (Brent, Newton, final) need the full Grams.
#include "common.hpp"

#include <cmath>

namespace pg {

constexpr int NLAM = 11;  // decade lambdas 10^-5 .. 10^5
#ifndef PG_ASSOC_WPB
#define PG_ASSOC_WPB 2    // wavefronts (= SNPs) per workgroup
#endif
// Two wavefronts per workgroup (r3; it was 4).  A SNP whose Newton iteration runs to the reference's 100-step cap (pyx:1411) keeps its
// whole workgroup's slots and LDS until it is done: in a 4-SNP workgroup three finished wavefronts idled behind each straggler.
// Measured on one box, same inputs (weak-signal phenotype, 43 Newton evaluations per SNP on average, max 106 | default phenotype,
// 6 evaluations for every SNP):   4 per workgroup 0.381 of the fp64 roof | 33.75 ms;   2: 0.464 | 33.8 ms;   1: 0.469 | 34.16 ms.
// A persistent work queue inside the kernel (wavefronts pulling SNP indices from an atomic counter) reaches 0.479 on the weak-signal
// phenotype too, but its loop costs registers (VGPR spills 58 -> 90 at c = 5) and 9 % on the default one: the hardware dispatcher
// refilling small workgroups does the same job for free.
constexpr int WPB = PG_ASSOC_WPB;
#define PG_MINV 1e-35f    // pygemma_model.pyx:39
// tuning knobs (overridable with -D for A/B builds)
#ifndef PG_FUSE_PQ_MAX
#define PG_FUSE_PQ_MAX 36   // fuse the P and Q powers in one pass while (c+2)(c+3)/2 <= this
#endif
#ifndef PG_FUSE_PQR_MAX
#define PG_FUSE_PQR_MAX 28  // fuse all three powers (Newton evaluations)
#endif
#ifndef PG_PFD
#define PG_PFD 2            // prefetch ring depth
#endif
#ifndef PG_CHUNK_MIN_NP
#define PG_CHUNK_MIN_NP 80   // more Gram entries than this (c >= 11): slot-chunked passes (measured: c=10 is still faster un-chunked)
#endif
#ifndef PG_SCAN_NV
#define PG_SCAN_NV 96        // decade-scan accumulators per pass (x-entries of as many lambdas as fit)
#endif
#ifndef PG_SWEEP_LDS_MIN_SLOTS
#define PG_SWEEP_LDS_MIN_SLOTS 3   // lanes owning this many Gram entries or more exchange the sweeps' pivot column through LDS
#endif
#ifndef PG_WAVES
#define PG_WAVES 2          // waves per SIMD the register allocation is held to
#endif

// Per-wave dynamic LDS of assoc_kernel — ONE definition for the kernel's carve-up and the host's launch size, so the two
// cannot drift apart: [piv: 3 M doubles when the sweeps exchange their pivot column through LDS] | xent[NLAM][2M] doubles |
// evs[NLAM] | d1s[NLAM] | lls[NLAM] | vals[n_vals], rounded up to 16 bytes.
__host__ __device__ constexpr size_t assoc_per_wave_bytes(int M, bool sweep_lds, size_t evalout_bytes, int n_vals)
{
    return (((sweep_lds ? (size_t)3 * M * 8 : 0) + (size_t)NLAM * 2 * M * 8 + NLAM * evalout_bytes + 2 * NLAM * 4 + (size_t)n_vals * 4) + 15) & ~(size_t)15;
}

struct AssocParams {
    int n, npad, c, niter, nu, grid, rowf;  // nu = n - (c+1); rowf = floats per fixed row (multiple of 4)
    long long p, ldx;
    const float *xr;
    const float *fixed;    // [npad][rowf]: d, w_0..w_{c-1}, y, zero pad; rows >= n all zero
    float *htab;           // [NLAM][npad]   (0 for i >= n)
    double *fixg;          // [NLAM][2][NP]  level-0 P,Q over all columns with x := 0
    double *t1tab;         // [NLAM]
    float *ldHtab;         // [NLAM]
    float lam11[NLAM];
    float logl_c;          // f32(f32(.5nu*ln(.5nu/pi)) - .5nu)   (pyx:1821-1822)
    const int *leaf, *node, *level, *chunk;
    int n_leaf, n_level, n_chunk, n_vals;
    float *beta, *se, *tau, *lam;
    double *F;
    unsigned long long *stats;
    unsigned *trace;       // optional: per SNP, fast evaluations | full (Newton) evaluations << 16 (pg_assoc_set_eval_trace)
    // N2 (LRT instantiations only): ML lambda and log-likelihood per SNP; ml_c = f32(f32((n/2) ln(n/2pi)) - (n/2)) (pyx:1552-1554)
    int lrt, nhalf;
    float ml_c;
    float *lalt, *lamalt;
};

// quadratic forms of one evaluation; sh, shh = the level-0 (un-projected) traces sum h, sum h^2 the ML functions use
struct EvalOut { float yPy, yPPy, yPPPy, trP, trPP, ld, Pxx_c, Pyx_c, sh, shh; };

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float hinv_f32(float lam, float d)
{   // pyx:903  1.0/(lam*eigenVals + 1.0) on float32 arrays: three separately rounded f32 operations.
    // The reciprocal: v_rcp_f32 plus ONE Newton step in fma form is the correctly rounded 1/x for every normal x whose
    // reciprocal is normal (checked exhaustively against __fdiv_rn over all 2^23 mantissas, tools/rcpcheck.hip) — 3
    // instructions instead of the 11 of the IEEE division expansion; anything else (x >= 2^126, inf, NaN; x >= 1 here)
    // takes the full division.
    const float x = __fadd_rn(__fmul_rn(lam, d), 1.0f);
    const float r0 = __builtin_amdgcn_rcpf(x);
    float r = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
    if (__builtin_expect(!(x < 0x1p126f), 0)) r = __fdiv_rn(1.0f, x);
    return r;
}

// numpy's SIMD float32 log (see oracle/pygemma_oracle.c: orc_np_logf) — bit-exact restatement
__device__ __forceinline__ float np_logf(float x)
{
    const float p1 = 9.999999999999998702752e-01f, p2 = 2.112677543073053063722e+00f,
                p3 = 1.480000633576506585156e+00f, p4 = 3.808837741388407920751e-01f,
                p5 = 2.589979117907922693523e-02f;
    const float q1 = 2.612677543073109236779e+00f, q2 = 2.453006071784736363091e+00f,
                q3 = 9.864942958519418960339e-01f, q4 = 1.546476374983906719538e-01f,
                q5 = 5.875095403124574342950e-03f;
    // anything but a positive finite argument (an eigenvalue handed over as inf / NaN / negative): numpy's special values
    if (__builtin_expect(!(x > 0.0f) || x == INFINITY, 0)) return (x != x) ? x : (x < 0.0f ? NAN : (x == 0.0f ? -INFINITY : x));
    unsigned u = __float_as_uint(x);
    int e = (int)((u >> 23) & 0xff) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    float ef = (float)e;
    if (m <= 0.70710678118f) { m = __fadd_rn(m, m); ef = __fsub_rn(ef, 1.0f); }
    float t = __fsub_rn(m, 1.0f);
    float den = fmaf(q5, t, q4);
    den = fmaf(den, t, q3); den = fmaf(den, t, q2); den = fmaf(den, t, q1); den = fmaf(den, t, 1.0f);
    float num = fmaf(p5, t, p4);
    num = fmaf(num, t, p3); num = fmaf(num, t, p2); num = fmaf(num, t, p1); num = fmaf(num, t, 0.0f);
    float poly = __fdiv_rn(num, den);
    return fmaf(ef, 0.693147180559945309417232121458176568f, poly);
}

__device__ __forceinline__ double bfly(double v)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// log|H| = float(np.log(lam*eigenVals + 1.0).sum())  (pyx:972): numpy's float32 pairwise sum
// (8-accumulator leaves of <= 128 elements, binary recursion, sequential across 8192-element
// chunks) reproduced with the host-built plan: 8 lanes per leaf, tree levels in parallel.
// vals: per-wave LDS scratch of n_vals floats.  All lanes return the same value.
__device__ float device_logdet_H(const AssocParams &pr, float lam, int lane, float *vals)
{
    const int sub = lane & 7, grp = lane >> 3;
    auto elem = [&](int i) -> float {
        float d = pr.fixed[(size_t)i * pr.rowf];
        return np_logf(__fadd_rn(__fmul_rn(lam, d), 1.0f));
    };
    for (int l0 = 0; l0 < pr.n_leaf; l0 += 8) {
        const int lf = l0 + grp;
        int start = 0, len = 0;
        if (lf < pr.n_leaf) { start = pr.leaf[2 * lf]; len = pr.leaf[2 * lf + 1]; }
        float res = 0.0f;
        if (len >= 8) {
            const int body = len - (len & 7);
            float r = elem(start + sub);
            for (int i = 8; i < body; i += 8) r = __fadd_rn(r, elem(start + i + sub));
            // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) : xor butterfly inside the 8-lane group
            r = __fadd_rn(r, __shfl_xor(r, 1, 64));
            r = __fadd_rn(r, __shfl_xor(r, 2, 64));
            r = __fadd_rn(r, __shfl_xor(r, 4, 64));
            res = r;
            for (int i = body; i < len; i++) res = __fadd_rn(res, elem(start + i));
        } else {
            // the shuffles above must be executed by whole groups only; groups are uniform in `len`
            for (int i = 0; i < len; i++) res = __fadd_rn(res, elem(start + i));
        }
        if (lf < pr.n_leaf && sub == 0) vals[lf] = res;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int h = 0; h < pr.n_level; h++) {
        const int b = pr.level[h], e = pr.level[h + 1];
        for (int k = b + lane; k < e; k += 64) {
            float a = vals[pr.node[2 * k]], c = vals[pr.node[2 * k + 1]];
            vals[pr.n_leaf + k] = __fadd_rn(a, c);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    float tot = vals[pr.chunk[0]];
    for (int k = 1; k < pr.n_chunk; k++) tot = __fadd_rn(tot, vals[pr.chunk[k]]);
    return tot;
}

// ------------------------------------------------------------------------------------------------
// scalar REML functions — same statements, same widths as the C Cython generates
__device__ __forceinline__ float logl_f(const AssocParams &pr, float yPy, float ldH, float ld)
{   // pyx:1813-1830 (first two statements folded into pr.logl_c on the host; logdet_Wt_W = 0.0)
    float r = pr.logl_c;
    r = (float)((double)r + (0.5 * (double)0.0f));
    r = (float)((double)r - (0.5 * (double)ldH));
    r = (float)((double)r - (0.5 * (double)ld));
    r = (float)((double)r - ((0.5 * (double)pr.nu) * log((double)yPy)));
    return r;
}
__device__ __forceinline__ float d1_f(const AssocParams &pr, float lam, float yPy, float yPPy, float trP)
{   // pyx:1656-1669
    float yT = (PG_MINV > yPy) ? PG_MINV : yPy;
    float nc_tr = __fsub_rn((float)pr.nu, trP);
    float r = (float)(-0.5 * (double)__fdiv_rn(nc_tr, lam));
    float t = (0.0f > yPPy) ? 0.0f : yPPy;
    float g = __fdiv_rn(__fsub_rn(yT, t), lam);
    r = (float)((double)r + ((0.5 * (double)pr.nu) * (double)g) / (double)yT);
    return r;
}
__device__ __forceinline__ float d2_f(const AssocParams &pr, float lam, float yPy, float yPPy, float yPPPy,
                                      float trP, float trPP)
{   // pyx:1675-1698
    float a = (PG_MINV > yPy) ? PG_MINV : yPy;
    float b = (PG_MINV > yPPy) ? PG_MINV : yPPy;
    float e = (PG_MINV > yPPPy) ? PG_MINV : yPPPy;
    double lam2 = (double)lam * (double)lam;           // pow(lam, 2.0): exact product of two f32
    float ae = __fadd_rn(a, e);
    float G2 = (float)(((double)ae - 2.0 * (double)b) / lam2);
    float G1 = __fdiv_rn(__fsub_rn(a, b), lam);
    float nct = __fadd_rn((float)pr.nu, trPP);
    float r = (float)((0.5 * ((double)nct - 2.0 * (double)trP)) / lam2);
    float G2a = __fmul_rn(G2, a);
    r = (float)((double)r - ((double)pr.nu * ((double)G2a - (0.5 * (double)G1) * (double)G1)) / ((double)a * (double)a));
    return r;
}

// N2: the ML (non-restricted) scalars — likelihood_lambda pyx:1542-1562, likelihood_derivative1_lambda pyx:1567-1581,
// likelihood_derivative2_lambda pyx:1586-1603 — same statements and widths as oracle orc_ml_logl / orc_ml_d1 / orc_ml_d2
// (the reference takes the quadratic forms from float32 NumPy helpers; here they come from the sweeps: tolerance parity,
// see oracle/pygemma_oracle.c).
__device__ __forceinline__ float ml_logl_f(const AssocParams &pr, float yPy, float ldH)
{
    float r = pr.ml_c;                                                       // pyx:1552-1554 folded on the host
    r = __fsub_rn(r, __fmul_rn(0.5f, ldH));                                  // pyx:1556
    const float t = (PG_MINV > yPy) ? PG_MINV : yPy;
    r = (float)((double)r - (double)pr.nhalf * log((double)t));              // pyx:1558
    return r;
}
__device__ __forceinline__ float ml_d1_f(const AssocParams &pr, float lam, float yPy, float yPPy, float sh)
{
    float r = __fmul_rn(-0.5f, __fdiv_rn(__fsub_rn((float)pr.n, sh), lam));  // pyx:1574
    const float num = (PG_MINV > yPPy) ? PG_MINV : yPPy, den = (PG_MINV > yPy) ? PG_MINV : yPy;
    r = (float)((double)r + ((double)pr.nhalf * (1.0 - (double)__fdiv_rn(num, den))) / (double)lam);   // pyx:1579
    return r;
}
__device__ __forceinline__ float ml_d2_f(const AssocParams &pr, float lam, float yPy, float yPPy, float yPPPy, float sh, float shh)
{
    const float a = (PG_MINV > yPy) ? PG_MINV : yPy, b = (PG_MINV > yPPy) ? PG_MINV : yPPy, e = (PG_MINV > yPPPy) ? PG_MINV : yPPPy;
    const float G2 = (float)(((double)__fadd_rn(a, e) - 2.0 * (double)b) / (double)__fmul_rn(lam, lam));   // pyx:1597
    const float G1 = __fdiv_rn(__fsub_rn(a, b), lam);                                                    // pyx:1598
    const float t = __fmul_rn(0.5f, __fsub_rn(__fadd_rn((float)pr.n, shh), __fmul_rn(2.0f, sh)));        // pyx:1600
    float r = (float)((double)t / ((double)lam * (double)lam));
    r = (float)((double)r - ((0.5 * (double)pr.n) * ((2.0 * (double)G2) - (double)__fdiv_rn(__fmul_rn(G1, G1), a))) / (double)a);   // pyx:1601
    return r;
}

// ------------------------------------------------------------------------------------------------
template <int C> struct Shape {
    static constexpr int M = C + 2;             // columns of W* = [W | x | y]
    static constexpr int NP = M * (M + 1) / 2;  // lower-triangle entries
    static constexpr int SLOTS = (NP + 63) / 64;               // Gram entries owned per lane
    static constexpr int NV0 = NP < 64 ? NP : 64, NV1 = NP > 64 ? NP - 64 : 0;
    // entries of slot sl: [64 sl, 64 sl + slot_nv(sl)).  CHUNKED: one slot and one power per pass over n (64 accumulators),
    // which fits 2 waves/SIMD at any c; below PG_CHUNK_MIN_NP all entries of one or more powers share a pass.
    static constexpr int slot_nv(int sl) { return (NP - 64 * sl) < 64 ? (NP - 64 * sl) : 64; }
    static constexpr bool CHUNKED = NP > PG_CHUNK_MIN_NP;
    static constexpr bool SWEEP_LDS = SLOTS >= PG_SWEEP_LDS_MIN_SLOTS;   // pivot column of the sweeps through per-wave LDS (3 M doubles)
    // scan: decade lambdas handled per pass
    // (two reduce-scatters of <= 64 and <= 32 values; PG_SCAN_NV caps the per-pass accumulators: 2 VGPRs each
    // — measured: 96 is best for c >= 3, 64 for c <= 2 where one more lambda per pass starts to spill)
    static constexpr int SCAN_NV = (M <= 4 && PG_SCAN_NV > 64) ? 64 : PG_SCAN_NV;
    static constexpr int G = (SCAN_NV / (2 * M)) < 1 ? 1 : ((SCAN_NV / (2 * M)) > 11 ? 11 : (SCAN_NV / (2 * M)));
    // powers accumulated per pass over n: fused while the accumulators fit the register budget of 2 waves/SIMD
    static constexpr bool FUSE_PQ = NP <= PG_FUSE_PQ_MAX, FUSE_PQR = NP <= PG_FUSE_PQR_MAX;
};
__host__ __device__ constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }  // r >= c
__host__ __device__ constexpr int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__device__ __forceinline__ double dmaxf(double a, float b) { return ((double)b > a) ? (double)b : a; }

// 64-lane sums of NVP per-lane partials, one value index per lane ("reduce-scatter"): recursive halving with
// ascending strides 1,2,4,.. — lane l keeps the values whose index agrees with its low bits and adds its
// partner's partials of them — then plain butterfly steps for the strides >= NVP.  Every value goes through
// exactly the additions of the xor-butterfly (strides 1..32) the oracle's order=1 mode performs, so the bits
// are the same, at ~1/6 of the adds and shuffles.  Returns the total of value index (lane % NVP).
template <int NVP>
__device__ __forceinline__ double reduce_scatter(double (&v)[NVP], int lane)
{
    static_assert(NVP >= 1 && NVP <= 64 && (NVP & (NVP - 1)) == 0, "power of two");
#pragma unroll
    for (int s = 1; s < NVP; s <<= 1) {
        const bool b = (lane & s) != 0;
#pragma unroll
        for (int m = 0; m < NVP / (2 * s); m++) {
            const double keep = b ? v[2 * m + 1] : v[2 * m];
            const double send = b ? v[2 * m] : v[2 * m + 1];
            v[m] = keep + __shfl_xor(send, s, 64);
        }
    }
    double x = v[0];
#pragma unroll
    for (int s = NVP; s < 64; s <<= 1) x += __shfl_xor(x, s, 64);
    return x;
}

// NP per-lane partials (lower-triangle entries) -> one owned entry per lane and slot: slot 0 holds entry
// (lane % NVP0) of entries [0,64), slot 1 entry 64 + (lane % NVP1).
template <int C, int SL = 0>
__device__ __forceinline__ void scatter_entries(const double (&acc)[Shape<C>::NP], int lane, double (&out)[Shape<C>::SLOTS])
{
    constexpr int NV = Shape<C>::slot_nv(SL), NVP = next_pow2(NV);
    double t[NVP];
#pragma unroll
    for (int k = 0; k < NVP; k++) t[k] = (k < NV) ? acc[64 * SL + k] : 0.0;
    out[SL] = reduce_scatter<NVP>(t, lane);
    if constexpr (SL + 1 < Shape<C>::SLOTS) scatter_entries<C, SL + 1>(acc, lane, out);
}

// which Gram entries a lane owns
template <int C> struct Own {
    int e[Shape<C>::SLOTS], r[Shape<C>::SLOTS], c[Shape<C>::SLOTS];
    bool valid[Shape<C>::SLOTS];
    __device__ __forceinline__ void init(int lane)
    {
#pragma unroll
        for (int sl = 0; sl < Shape<C>::SLOTS; sl++) {
            const int nv = Shape<C>::slot_nv(sl), k = lane % next_pow2(nv);
            valid[sl] = k < nv;
            e[sl] = valid[sl] ? 64 * sl + k : 0;
            int rr = 0;
            while ((rr + 1) * (rr + 2) / 2 <= e[sl]) rr++;
            r[sl] = rr; c[sl] = e[sl] - rr * (rr + 1) / 2;
        }
    }
};

// value of Gram entry `es` (per-lane index) / `eq` (wave-uniform index) from its owning lane
template <int SLOTS>
__device__ __forceinline__ double gather(const double (&X)[SLOTS], int es)
{
    double v = __shfl(X[0], es & 63, 64);
#pragma unroll
    for (int sl = 1; sl < SLOTS; sl++) { const double w = __shfl(X[sl], es & 63, 64); v = ((es >> 6) == sl) ? w : v; }
    return v;
}

__device__ __forceinline__ void wave_lds_sync();

// The c_tot sweeps of precompute_mat (pyx:947-963 / :1007-1036) with the Gram entries spread over the lanes:
// every lane updates the entry (r,c) it owns, fetching the pivot-column operands from their owners; arithmetic per
// entry identical to oracle sweeps(order=1).  Scalars (traces, pivots) are wave-uniform.
template <int C, bool FULL>
__device__ __forceinline__ void lane_sweeps(double (&P)[Shape<C>::SLOTS], double (&Q)[Shape<C>::SLOTS], double (&R)[Shape<C>::SLOTS],
                                            const Own<C> &own, double t1, double t2, int lane, EvalOut &o, double *piv)
{
    constexpr int M = Shape<C>::M, NP = Shape<C>::NP, SLOTS = Shape<C>::SLOTS;
    // With three or more slots per lane (c >= 14) the pivot column goes through the wave's LDS (piv: 3 M doubles) instead of
    // lane shuffles: a shuffle-gather costs SLOTS permutes per operand, 6 SLOTS^2 per step — pure data movement either way.
    constexpr bool VIA_LDS = Shape<C>::SWEEP_LDS;
    if (own.valid[0] && own.e[0] == 0) P[0] = dmaxf(P[0], PG_MINV);
    double trP = t1, trPP = t2, apiv = 1.0;
    for (int i = 1; i < M; i++) {
        const int q = i - 1, eqq = tri(q, q);
        double a, b, e = 0.0;
        double ur[SLOTS], uc[SLOTS], vr[SLOTS], vc[SLOTS], wr[SLOTS], wc[SLOTS];
        bool act[SLOTS];
        if constexpr (VIA_LDS) {
#pragma unroll
            for (int sl = 0; sl < SLOTS; sl++)
                if (own.valid[sl] && own.c[sl] == q) {       // owners of column q publish it (replicas write the same value)
                    piv[own.r[sl]] = P[sl]; piv[M + own.r[sl]] = Q[sl];
                    if (FULL) piv[2 * M + own.r[sl]] = R[sl];
                }
            wave_lds_sync();
            a = piv[q]; b = piv[M + q];
            if (FULL) e = piv[2 * M + q];
#pragma unroll
            for (int sl = 0; sl < SLOTS; sl++) {
                act[sl] = own.valid[sl] && own.c[sl] >= i;
                const int rr = act[sl] ? own.r[sl] : q, cc = act[sl] ? own.c[sl] : q;
                ur[sl] = piv[rr]; uc[sl] = piv[cc];
                vr[sl] = piv[M + rr]; vc[sl] = piv[M + cc];
                if (FULL) { wr[sl] = piv[2 * M + rr]; wc[sl] = piv[2 * M + cc]; }
            }
            wave_lds_sync();                                  // all reads done before the next step's writes
        } else {
            a = gather<SLOTS>(P, eqq); b = gather<SLOTS>(Q, eqq);
            e = FULL ? gather<SLOTS>(R, eqq) : 0.0;
#pragma unroll
            for (int sl = 0; sl < SLOTS; sl++) {
                act[sl] = own.valid[sl] && own.c[sl] >= i;       // r >= c >= i
                const int er = act[sl] ? tri(own.r[sl], q) : 0, ec = act[sl] ? tri(own.c[sl], q) : 0;
                ur[sl] = gather<SLOTS>(P, er); uc[sl] = gather<SLOTS>(P, ec);
                vr[sl] = gather<SLOTS>(Q, er); vc[sl] = gather<SLOTS>(Q, ec);
                if (FULL) { wr[sl] = gather<SLOTS>(R, er); wc[sl] = gather<SLOTS>(R, ec); }
            }
        }
        if (lane == q) apiv = a;
        const double a2 = a * a, ia = -1.0 / a, ba2 = b / a2;
        if (FULL) {
            const double ba = b / a;
            trPP = (trPP + ba * ba) - 2 * (e / a);
            const double a3 = a2 * a, b2 = b * b;
            const double cR = (e / a2) - (b2 / a3);
#pragma unroll
            for (int sl = 0; sl < SLOTS; sl++)
                if (act[sl]) {
                    double s1 = fma(cR * uc[sl], ur[sl], R[sl]);
                    double s2 = fma(ia * uc[sl], wr[sl], 0.0); s2 = fma(ia * wc[sl], ur[sl], s2);
                    double s3 = fma(ia * vc[sl], vr[sl], 0.0);
                    double s4 = fma(ba2 * uc[sl], vr[sl], 0.0); s4 = fma(ba2 * vc[sl], ur[sl], s4);
                    double rn = ((s1 + s2) + s3) + s4;
                    if (own.r[sl] == i && own.c[sl] == i) rn = dmaxf(rn, PG_MINV);
                    R[sl] = rn;
                }
        }
        trP = trP - b / a;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++)
            if (act[sl]) {
                double s1 = fma(ba2 * uc[sl], ur[sl], Q[sl]);
                double s2 = fma(ia * uc[sl], vr[sl], 0.0); s2 = fma(ia * vc[sl], ur[sl], s2);
                double qn = s1 + s2;
                double pn = fma(ia * uc[sl], ur[sl], P[sl]);
                if (own.r[sl] == i && own.c[sl] == i) { qn = dmaxf(qn, PG_MINV); pn = dmaxf(pn, PG_MINV); }
                Q[sl] = qn; P[sl] = pn;
            }
        if (i == M - 2) { o.Pxx_c = (float)gather<SLOTS>(P, tri(M - 2, M - 2)); o.Pyx_c = (float)gather<SLOTS>(P, tri(M - 1, M - 2)); }
    }
    // logdet_Wt_H_inv_W (pyx:957/1029): f32 accumulator += log(pivot), pivots in order; the M-1 logs are taken
    // at once (lane q holds pivot q), the f32 accumulation stays sequential
    const double lg = log(apiv);
    float ld = 0.0f;
    for (int q = 0; q < M - 1; q++) ld = (float)((double)ld + __shfl(lg, q, 64));
    o.yPy = (float)gather<SLOTS>(P, NP - 1);
    o.yPPy = (float)gather<SLOTS>(Q, NP - 1);
    o.yPPPy = FULL ? (float)gather<SLOTS>(R, NP - 1) : 0.0f;
    o.trP = (float)trP;
    o.trPP = FULL ? (float)trPP : 0.0f;
    o.ld = ld;
    o.sh = (float)t1;
    o.shh = FULL ? (float)t2 : 0.0f;
}

// One element's operands: the packed fixed row (d, w_0..w_{C-1}, y) as 16-byte vectors plus x_i.
// Loads are unconditional so that the next element's loads can be issued before the current element's arithmetic, and no value
// is selected afterwards: rows >= n of the table hold w = y = 0 and d = +inf, which makes h = 1/(lam d + 1) exactly 0 there
// (x = inf takes the division fallback of hinv_f32: 1/inf = 0), so every product of a pad element is 0 whatever x is read
// (the x index is clamped to n-1: a caller's row need not have a pad).
template <int C> struct Elem {
    static constexpr int S4 = (C + 2 + 3) / 4;
    float4 row[S4];
    float x;
};
template <int C, bool HASX = true>
__device__ __forceinline__ void load_elem(const AssocParams &pr, const float *xrow, int i, Elem<C> &e)
{
    const float4 *src = reinterpret_cast<const float4 *>(pr.fixed + (size_t)i * pr.rowf);
#pragma unroll
    for (int k = 0; k < Elem<C>::S4; k++) e.row[k] = src[k];
    const int ic = min(i, pr.n - 1);
    e.x = HASX ? xrow[ic] : 0.0f;
}
// software pipeline with a ring of PFD register sets (compile-time slots: no register copies, no scratch)
constexpr int PFD = PG_PFD;
template <class E, int PF = PFD, class LoadF, class BodyF>
__device__ __forceinline__ void pipelined(int niter, LoadF &&ld, BodyF &&body)
{
    E buf[PF];
#pragma unroll
    for (int d = 0; d < PF; d++) ld(buf[d], d < niter ? d : niter - 1);
    for (int it = 0; it < niter; it += PF) {
#pragma unroll
        for (int d = 0; d < PF; d++) {
            if (it + d < niter) body(buf[d], it + d);
            const int nx = it + d + PF;
            ld(buf[d], nx < niter ? nx : niter - 1);
        }
    }
}
template <int C>
__device__ __forceinline__ void unpack_elem(const AssocParams &pr, const Elem<C> &e, int i, float &d, float (&col)[C + 2])
{
    float buf[4 * Elem<C>::S4];
#pragma unroll
    for (int k = 0; k < Elem<C>::S4; k++) {
        buf[4 * k] = e.row[k].x; buf[4 * k + 1] = e.row[k].y; buf[4 * k + 2] = e.row[k].z; buf[4 * k + 3] = e.row[k].w;
    }
    d = buf[0];
#pragma unroll
    for (int j = 0; j < C; j++) col[j] = buf[1 + j];
    col[C] = e.x;
    col[C + 1] = buf[C + 1];
}

// One pass over the n elements accumulating the level-0 Gram powers selected by MASK (1: P = W*'H^-1 W*,
// 2: Q = W*'H^-2 W*, 4: R = W*'H^-3 W*; t1 rides with P, t2 with R), then the reduce-scatter to the owning lanes.
template <int C, int MASK, bool HASX>
__device__ __forceinline__ void gram_pass(const AssocParams &pr, const float *xrow, float lam, int lane, float *htab_out,
                                          double (&Po)[Shape<C>::SLOTS], double (&Qo)[Shape<C>::SLOTS], double (&Ro)[Shape<C>::SLOTS],
                                          double &t1, double &t2)
{
    constexpr int M = Shape<C>::M, NP = Shape<C>::NP;
    constexpr bool DP = (MASK & 1) != 0, DQ = (MASK & 2) != 0, DR = (MASK & 4) != 0;
    double P[DP ? NP : 1], Q[DQ ? NP : 1], R[DR ? NP : 1];
#pragma unroll
    for (int k = 0; k < NP; k++) { if constexpr (DP) P[k] = 0.0; if constexpr (DQ) Q[k] = 0.0; if constexpr (DR) R[k] = 0.0; }
    double s1 = 0.0, s2 = 0.0;
    pipelined<Elem<C>>(pr.niter, [&](Elem<C> &e, int it) { load_elem<C, HASX>(pr, xrow, it * 64 + lane, e); },
                       [&](const Elem<C> &cur, int it) {
        const int i = it * 64 + lane;
        float d, colf[M];
        unpack_elem<C>(pr, cur, i, d, colf);
        const float h = hinv_f32(lam, d);        // pad rows: d = +inf -> h = 0
        if (htab_out) htab_out[i] = h;
        const double hd = (double)h;
        double col[M], a[M];
#pragma unroll
        for (int j = 0; j < M; j++) { col[j] = (double)colf[j]; a[j] = hd * col[j]; }
        if (DP || DQ) {
#pragma unroll
            for (int j = 0; j < M; j++)
#pragma unroll
                for (int k = 0; k <= j; k++) {
                    if constexpr (DP) P[tri(j, k)] = fma(a[j], col[k], P[tri(j, k)]);
                    if constexpr (DQ) Q[tri(j, k)] = fma(a[j], a[k], Q[tri(j, k)]);
                }
        }
        if (DP) s1 += hd;
        if constexpr (DR) {
            const double h2 = hd * hd;
#pragma unroll
            for (int j = 0; j < M; j++) {
                const double gj = h2 * col[j];
#pragma unroll
                for (int k = 0; k <= j; k++) R[tri(j, k)] = fma(gj, a[k], R[tri(j, k)]);
            }
            s2 = fma(hd, hd, s2);
        }
    });
    if constexpr (DP) { scatter_entries<C>(P, lane, Po); t1 = bfly(s1); }
    if constexpr (DQ) scatter_entries<C>(Q, lane, Qo);
    if constexpr (DR) { scatter_entries<C>(R, lane, Ro); t2 = bfly(s2); }
}

// One pass over the n elements for the entries of ONE slot and ONE power (PW = 1: P, 2: Q, 3: R): the per-entry arithmetic
// and summation order are those of gram_pass (every entry is its own fma chain), only the grouping into passes differs.
template <int C, int PW, int SL, bool HASX>
__device__ __forceinline__ void gram_pass_slot(const AssocParams &pr, const float *xrow, float lam, int lane, float *htab_out, double &out, double &tsum)
{
    constexpr int M = Shape<C>::M, E0 = 64 * SL, NV = Shape<C>::slot_nv(SL), NVP = next_pow2(NV);
    double acc[NVP];
#pragma unroll
    for (int k = 0; k < NVP; k++) acc[k] = 0.0;
    double s = 0.0;
    // wide rows (c > 20): one register set in flight instead of two, the element ring would push the slot's 64 accumulators to scratch
    pipelined<Elem<C>, (M > 22 ? 1 : PFD)>(pr.niter, [&](Elem<C> &e, int it) { load_elem<C, HASX>(pr, xrow, it * 64 + lane, e); },
                       [&](const Elem<C> &cur, int it) {
        const int i = it * 64 + lane;
        float d, colf[M];
        unpack_elem<C>(pr, cur, i, d, colf);
        const float h = hinv_f32(lam, d);        // pad rows: d = +inf -> h = 0
        if (htab_out) htab_out[i] = h;
        const double hd = (double)h, h2 = hd * hd;
        double col[M], a[M], g[PW == 3 ? M : 1];
#pragma unroll
        for (int j = 0; j < M; j++) { col[j] = (double)colf[j]; a[j] = hd * col[j]; if constexpr (PW == 3) g[j] = h2 * col[j]; }
#pragma unroll
        for (int j = 0; j < M; j++)
#pragma unroll
            for (int k = 0; k <= j; k++) {
                const int e = tri(j, k);
                if (e >= E0 && e < E0 + NV) {
                    if constexpr (PW == 1) acc[e - E0] = fma(a[j], col[k], acc[e - E0]);
                    else if constexpr (PW == 2) acc[e - E0] = fma(a[j], a[k], acc[e - E0]);
                    else acc[e - E0] = fma(g[j], a[k], acc[e - E0]);
                }
            }
        if constexpr (SL == 0 && PW == 1) s += hd;
        if constexpr (SL == 0 && PW == 3) s = fma(hd, hd, s);
    });
    out = reduce_scatter<NVP>(acc, lane);
    if constexpr (SL == 0 && (PW == 1 || PW == 3)) tsum = bfly(s);
}
template <int C, int PW, bool HASX, int SL = 0>
__device__ __forceinline__ void slot_passes(const AssocParams &pr, const float *xrow, float lam, int lane, float *htab_out,
                                            double (&X)[Shape<C>::SLOTS], double &tsum)
{
    gram_pass_slot<C, PW, SL, HASX>(pr, xrow, lam, lane, SL == 0 ? htab_out : nullptr, X[SL], tsum);
    if constexpr (SL + 1 < Shape<C>::SLOTS) slot_passes<C, PW, HASX, SL + 1>(pr, xrow, lam, lane, htab_out, X, tsum);
}

// Level-0 Grams at an arbitrary lambda, then the sweeps.
template <int C, bool FULL>
__device__ __forceinline__ void eval_specific(const AssocParams &pr, const float *xrow, float lam, int lane, const Own<C> &own, EvalOut &o,
                                              double *piv)
{
    constexpr int SLOTS = Shape<C>::SLOTS;
    double P[SLOTS], Q[SLOTS], R[SLOTS], t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int sl = 0; sl < SLOTS; sl++) R[sl] = 0.0;
    if constexpr (Shape<C>::CHUNKED) {
        double tq = 0.0;
        slot_passes<C, 1, true>(pr, xrow, lam, lane, nullptr, P, t1);
        slot_passes<C, 2, true>(pr, xrow, lam, lane, nullptr, Q, tq);
        if (FULL) slot_passes<C, 3, true>(pr, xrow, lam, lane, nullptr, R, t2);
    } else if constexpr (FULL && Shape<C>::FUSE_PQR) {
        gram_pass<C, 7, true>(pr, xrow, lam, lane, nullptr, P, Q, R, t1, t2);
    } else if constexpr (Shape<C>::FUSE_PQ) {
        gram_pass<C, 3, true>(pr, xrow, lam, lane, nullptr, P, Q, R, t1, t2);
        if (FULL) gram_pass<C, 4, true>(pr, xrow, lam, lane, nullptr, P, Q, R, t1, t2);
    } else {
        gram_pass<C, 1, true>(pr, xrow, lam, lane, nullptr, P, Q, R, t1, t2);
        gram_pass<C, 2, true>(pr, xrow, lam, lane, nullptr, P, Q, R, t1, t2);
        if (FULL) gram_pass<C, 4, true>(pr, xrow, lam, lane, nullptr, P, Q, R, t1, t2);
    }
    lane_sweeps<C, FULL>(P, Q, R, own, t1, t2, lane, o, piv);
}

// ------------------------------------------------------------------------------------------------
// setup: one wavefront per decade lambda t — h table, SNP-independent level-0 Gram entries (x := 0),
// t1 and log|H| at lam11[t].
template <int C>
__global__ __launch_bounds__(64) void setup_tabs_kernel(AssocParams pr)
{
    constexpr int NP = Shape<C>::NP, SLOTS = Shape<C>::SLOTS;
    extern __shared__ float smem_f[];
    const int t = blockIdx.x, lane = threadIdx.x;
    const float lam = pr.lam11[t];
    Own<C> own;
    own.init(lane);
    double P[SLOTS], Q[SLOTS], R[SLOTS], t1 = 0.0, t2 = 0.0;
    if constexpr (Shape<C>::CHUNKED) {
        double tq = 0.0;
        slot_passes<C, 1, false>(pr, nullptr, lam, lane, pr.htab + (size_t)t * pr.npad, P, t1);
        slot_passes<C, 2, false>(pr, nullptr, lam, lane, nullptr, Q, tq);
    } else if constexpr (Shape<C>::FUSE_PQ) {
        gram_pass<C, 3, false>(pr, nullptr, lam, lane, pr.htab + (size_t)t * pr.npad, P, Q, R, t1, t2);
    } else {
        gram_pass<C, 1, false>(pr, nullptr, lam, lane, pr.htab + (size_t)t * pr.npad, P, Q, R, t1, t2);
        gram_pass<C, 2, false>(pr, nullptr, lam, lane, nullptr, P, Q, R, t1, t2);
    }
#pragma unroll
    for (int sl = 0; sl < SLOTS; sl++)
        if (own.valid[sl]) {   // replicas write the same value
            pr.fixg[((size_t)t * 2 + 0) * NP + own.e[sl]] = P[sl];
            pr.fixg[((size_t)t * 2 + 1) * NP + own.e[sl]] = Q[sl];
        }
    float ldH = device_logdet_H(pr, lam, lane, smem_f);
    if (lane == 0) { pr.t1tab[t] = t1; pr.ldHtab[t] = ldH; }
}

// scan accumulation: only the Gram entries that involve x, for GG decade lambdas at once; the totals go to
// xent[t][2M] in the wave's LDS (entry k <= C: P(x,k); C+1: P(y,x); M+k likewise for Q)
template <int C, int GG>
__device__ __forceinline__ void scan_accumulate(const AssocParams &pr, const float *xrow, int t0, int lane, double *xent)
{
    constexpr int M = Shape<C>::M, NV = GG * 2 * M;
    constexpr int NV0 = NV < 64 ? NV : 64, NV1 = NV > 64 ? NV - 64 : 0;      // values reduced by the first / second reduce-scatter
    constexpr int NVP0 = next_pow2(NV0), NVP1 = next_pow2(NV1 > 0 ? NV1 : 1);
    static_assert(NV <= 128, "scan group too large");
    double acc[NV0 == 64 ? 64 + NVP1 : NVP0];
#pragma unroll
    for (int k = 0; k < (NV0 == 64 ? 64 + NVP1 : NVP0); k++) acc[k] = 0.0;
    struct SE { Elem<C> e; float h[GG]; };
    pipelined<SE>(pr.niter, [&](SE &q, int it) {
        const int i = it * 64 + lane;
        load_elem<C>(pr, xrow, i, q.e);
#pragma unroll
        for (int g = 0; g < GG; g++) q.h[g] = pr.htab[(size_t)(t0 + g) * pr.npad + i];
    }, [&](const SE &q, int it) {
        const int i = it * 64 + lane;
        float d, colf[M];
        unpack_elem<C>(pr, q.e, i, d, colf);
        // z_k = x * col_k (k <= C: the covariates and x itself) and z_{C+1} = y * x are exact in f64 (24 x 24 bits) and do not
        // depend on lambda: P(x,k) += h z_k and Q(x,k) += h^2 z_k add the same exact products (h x)(w_k), (h x)(h w_k) that
        // the Gram pass of a specific lambda forms from a_j = h col_j — same bits, 6 multiplies fewer per lambda.
        const double xd = (double)colf[C];
        double z[M];
#pragma unroll
        for (int k = 0; k <= C; k++) z[k] = xd * (double)colf[k];
        z[C + 1] = (double)colf[C + 1] * xd;
#pragma unroll
        for (int g = 0; g < GG; g++) {
            const double hd = (double)q.h[g];
            const double h2 = hd * hd;                       // exact
            // row x (index C): columns k <= C ; row y (index C+1): column x
#pragma unroll
            for (int k = 0; k <= C + 1; k++) {
                acc[g * 2 * M + k] = fma(hd, z[k], acc[g * 2 * M + k]);
                acc[g * 2 * M + M + k] = fma(h2, z[k], acc[g * 2 * M + M + k]);
            }
        }
    });
    {
        double t[NVP0];
#pragma unroll
        for (int k = 0; k < NVP0; k++) t[k] = acc[k];
        const double tot = reduce_scatter<NVP0>(t, lane);
        const int idx = lane % NVP0;
        if (idx < NV0 && lane < NVP0) xent[t0 * 2 * M + idx] = tot;
    }
    if constexpr (NV1 > 0) {
        double t[NVP1];
#pragma unroll
        for (int k = 0; k < NVP1; k++) t[k] = acc[64 + k];
        const double tot = reduce_scatter<NVP1>(t, lane);
        const int idx = lane % NVP1;
        if (idx < NV1 && lane < NVP1) xent[t0 * 2 * M + 64 + idx] = tot;
    }
}
template <int C, int T0>
__device__ __forceinline__ void scan_all(const AssocParams &pr, const float *xrow, int lane, double *xent)
{
    constexpr int G = Shape<C>::G;
    if constexpr (T0 < NLAM) {
        constexpr int GG = (NLAM - T0) < G ? (NLAM - T0) : G;
        scan_accumulate<C, GG>(pr, xrow, T0, lane, xent);
        scan_all<C, T0 + GG>(pr, xrow, lane, xent);
    }
}

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// scipy.optimize.brentq (scipy/optimize/Zeros/brentq.c), xtol=2e-12, rtol=0.1, maxiter=100 (pyx:176-182).
// f(a), f(b) are the decade-scan values (the reference re-evaluates them: same inputs, same results).
template <class Fn>
__device__ __forceinline__ double brentq_dev(Fn &&f, double xa, double xb, double fa, double fb, int maxiter = 100)
{
    const double xtol = 2e-12, rtol = 0.1;
    double xpre = xa, xcur = xb, xblk = 0., fpre = fa, fcur = fb, fblk = 0., spre = 0., scur = 0., sbis, delta, stry, dpre, dblk;
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    for (int i = 0; i < maxiter; i++) {
        if (fpre != 0 && fcur != 0 && (__builtin_signbit(fpre) != __builtin_signbit(fcur))) {
            xblk = xpre; fblk = fpre; spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2;
        sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                dpre = (fpre - fcur) / (xpre - xcur);
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            double lim = 3 * fabs(sbis) - delta;
            double mn = fabs(spre) < lim ? fabs(spre) : lim;
            if (2 * fabs(stry) < mn) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else { spre = sbis; scur = sbis; }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = f(xcur);
    }
    return xcur;
}

// newton (pyx:1349-1416), all f32: root <- root - d1/d2; stops on the sign test (:1392), on leaving [l0, l1] (returns the
// previous iterate, :1398-1404), on NaN/Inf (:1406), on r_eps < 1e-5 or iteration > 100 (:1411).  eval(lam, e) fills the
// full (three-power) quadratic forms at lam.
template <class Eval>
__device__ __forceinline__ float newton_dev(const AssocParams &pr, float lroot, float l0, float l1, Eval &&eval)
{
    int iteration = 0;
    for (;;) {
        EvalOut e;
        eval(lroot, e);
        const float d1 = d1_f(pr, lroot, e.yPy, e.yPPy, e.trP);
        const float d2 = d2_f(pr, lroot, e.yPy, e.yPPy, e.yPPPy, e.trP, e.trPP);
        const float ratio = __fdiv_rn(d1, d2);
        // np.sign(ratio)*np.sign(d1)*np.sign(d2) <= 0.0   (NaN compares False)
        const bool any_nan = (ratio != ratio) || (d1 != d1) || (d2 != d2);
        const float sr = (float)((ratio > 0) - (ratio < 0)), s1 = (float)((d1 > 0) - (d1 < 0)), s2 = (float)((d2 > 0) - (d2 < 0));
        if (!any_nan && sr * s1 * s2 <= 0.0f) break;
        const float lnew = __fsub_rn(lroot, ratio);
        const float r_eps = (float)(fabs((double)__fsub_rn(lnew, lroot)) / fabs((double)lroot));
        if (lnew < l0) break;
        if (lnew > l1) break;
        if (isnan(lnew) || isinf(lnew)) break;
        lroot = lnew;
        if ((double)r_eps < 1e-5 || iteration > 100) break;
        iteration++;
    }
    return lroot;
}

template <int C, bool LRT>
// two waves per SIMD for EVERY c (PG_WAVES = 2): a lone wave cannot issue fp64 VALU at the pipe's rate (measured 2.3x faster at 2 than at
// 1).  From c = 4 on the instantiations are capped at 256 VGPRs with spills to scratch (c = 5: 58 VGPR + 35 SGPR spills, 196 B; c = 10:
// 107, 320 B; c = 14: 276 B) — all of them in the per-evaluation outer loops; the Gram loops themselves are scratch-free (checked on the
// shipped code object: VERDICT r3)
__global__ __launch_bounds__(64 * WPB, PG_WAVES) void assoc_kernel(AssocParams pr)
{
    constexpr int M = Shape<C>::M, NP = Shape<C>::NP, SLOTS = Shape<C>::SLOTS;
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * WPB + wave;
    if (g >= pr.p) return;  // whole wavefront leaves; no workgroup barrier is used anywhere in this kernel
    // per-wave LDS: xent[NLAM][2M] doubles | evs[NLAM] | d1s[NLAM] | lls[NLAM] | vals[n_vals]
    constexpr size_t piv_bytes = Shape<C>::SWEEP_LDS ? (size_t)3 * M * 8 : 0;
    unsigned char *base = smem + (size_t)wave * assoc_per_wave_bytes(M, Shape<C>::SWEEP_LDS, sizeof(EvalOut), pr.n_vals);
    double *piv = reinterpret_cast<double *>(base);          // pivot column exchange of the sweeps (c >= 14 only)
    base += piv_bytes;
    double *xent = reinterpret_cast<double *>(base);
    EvalOut *evs = reinterpret_cast<EvalOut *>(base + (size_t)NLAM * 2 * M * 8);
    float *d1s = reinterpret_cast<float *>(evs + NLAM);
    float *lls = d1s + NLAM;
    float *vals = lls + NLAM;
    const float *xrow = pr.xr + (size_t)g * pr.ldx;

    Own<C> own;
    own.init(lane);
    // ---- decade scan: x-dependent Gram entries at the 11 shared lambdas
    scan_all<C, 0>(pr, xrow, lane, xent);
    wave_lds_sync();
    for (int t = 0; t < NLAM; t++) {
        double P[SLOTS], Q[SLOTS], R[SLOTS];
        const double *fp = pr.fixg + ((size_t)t * 2 + 0) * NP, *fq = pr.fixg + ((size_t)t * 2 + 1) * NP;
        const double *xe = xent + t * 2 * M;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) {
            const int r = own.r[sl], c = own.c[sl];
            const bool isx = (r == C) || (r == C + 1 && c == C);
            const int k = (r == C) ? c : C + 1;
            P[sl] = isx ? xe[k] : fp[own.e[sl]];
            Q[sl] = isx ? xe[M + k] : fq[own.e[sl]];
            R[sl] = 0.0;
        }
        EvalOut e;
        lane_sweeps<C, false>(P, Q, R, own, pr.t1tab[t], 0.0, lane, e, piv);
        if (lane == 0) {
            evs[t] = e;
            d1s[t] = d1_f(pr, pr.lam11[t], e.yPy, e.yPPy, e.trP);
            lls[t] = logl_f(pr, e.yPy, pr.ldHtab[t], e.ld);
        }
    }
    wave_lds_sync();
    // ---- candidate selection (pyx:109-117 / :144-152): start from the two boundaries
    float best_l = lls[0], best_lambda;
    EvalOut best_e;
    if (best_l < lls[NLAM - 1]) { best_l = lls[NLAM - 1]; best_lambda = pr.lam11[NLAM - 1]; best_e = evs[NLAM - 1]; }
    else { best_lambda = pr.lam11[0]; best_e = evs[0]; }
    unsigned n_fast = 0, n_full = 0;
    if (pr.grid) {
        for (int t = 0; t < NLAM - 1; t++)   // pyx:119-130, k = -5..4, strict '>'
            if (lls[t] > best_l) { best_l = lls[t]; best_lambda = pr.lam11[t]; best_e = evs[t]; }
    } else {
        for (int k = 0; k < NLAM - 1; k++) {  // pyx:154-192
            const float f0 = d1s[k], f1 = d1s[k + 1];
            if (__builtin_signbit(f0) == __builtin_signbit(f1)) continue;   // copysignf product < 0 (pyx:174)
            const float l0 = pr.lam11[k], l1 = pr.lam11[k + 1];
            double root = brentq_dev(
                [&](double x) -> double {
                    EvalOut e;
                    const float lf = (float)x;      // pyx:1631: np.float32_t lam
                    eval_specific<C, false>(pr, xrow, lf, lane, own, e, piv);
                    n_fast++;
                    return (double)d1_f(pr, lf, e.yPy, e.yPPy, e.trP);
                },
                (double)l0, (double)l1, (double)f0, (double)f1);
            // newton (pyx:1349-1416), all f32
            const float lroot = newton_dev(pr, (float)root, l0, l1, [&](float lf, EvalOut &e) {
                eval_specific<C, true>(pr, xrow, lf, lane, own, e, piv);
                n_full++;
            });
            EvalOut e;
            eval_specific<C, false>(pr, xrow, lroot, lane, own, e, piv);   // pyx:186
            n_fast++;
            const float ldH = device_logdet_H(pr, lroot, lane, vals);
            const float ll = logl_f(pr, e.yPy, ldH, e.ld);       // pyx:188
            if (ll > best_l) { best_l = ll; best_lambda = lroot; best_e = e; }
        }
    }
    if constexpr (LRT) {
        // ---- N2: lambda_ML = calc_lambda(eigenVals, Y, [W, x]) (lmm/lmm.py:22-84) and likelihood_lambda at it (pyx:1542-1562).
        // Candidates in the reference's order: 1e-5, 1e5, then per decade with a sign change of d logL / d lambda one root by
        // brentq(rtol = 0.1, maxiter = 5000) + scipy.optimize.newton(fprime = d2, rtol = 1e-5, tol = 1.48e-8, maxiter = 10);
        // np.argmax keeps the first maximum (a NaN wins).  The decade values reuse the scan's quadratic forms.
        float ml_l = ml_logl_f(pr, evs[0].yPy, pr.ldHtab[0]), ml_lam = pr.lam11[0];
        {
            const float l1 = ml_logl_f(pr, evs[NLAM - 1].yPy, pr.ldHtab[NLAM - 1]);
            if (ml_l == ml_l && (l1 != l1 || l1 > ml_l)) { ml_l = l1; ml_lam = pr.lam11[NLAM - 1]; }
        }
        for (int k = 0; k < NLAM - 1; k++) {
            const float f0 = ml_d1_f(pr, pr.lam11[k], evs[k].yPy, evs[k].yPPy, evs[k].sh);
            const float f1 = ml_d1_f(pr, pr.lam11[k + 1], evs[k + 1].yPy, evs[k + 1].yPPy, evs[k + 1].sh);
            if (f0 != f0 || f1 != f1) continue;
            const float s0 = (float)((f0 > 0) - (f0 < 0)), s1 = (float)((f1 > 0) - (f1 < 0));
            if (!(s0 * s1 < 0.0f)) continue;                                 // np.sign(f0) * np.sign(f1) < 0
            double p0 = brentq_dev(
                [&](double x) -> double {
                    EvalOut e;
                    const float lf = (float)x;
                    eval_specific<C, false>(pr, xrow, lf, lane, own, e, piv);
                    n_fast++;
                    return (double)ml_d1_f(pr, lf, e.yPy, e.yPPy, e.sh);
                },
                (double)pr.lam11[k], (double)pr.lam11[k + 1], (double)f0, (double)f1, 5000);
            double pn = p0;
            for (int itr = 0; itr < 10; itr++) {
                EvalOut e;
                const float lf = (float)p0;
                eval_specific<C, true>(pr, xrow, lf, lane, own, e, piv);
                n_full++;
                const float fval = ml_d1_f(pr, lf, e.yPy, e.yPPy, e.sh);
                if (fval == 0.0f) { pn = p0; break; }
                const float fder = ml_d2_f(pr, lf, e.yPy, e.yPPy, e.yPPPy, e.sh, e.shh);
                if (fder == 0.0f) { pn = p0; break; }
                pn = p0 - (double)__fdiv_rn(fval, fder);
                if (fabs(pn - p0) <= 1.48e-8 + 1e-5 * fabs(p0)) break;
                p0 = pn;
            }
            if (ml_l == ml_l) {
                EvalOut e;
                const float lf = (float)pn;
                eval_specific<C, false>(pr, xrow, lf, lane, own, e, piv);
                n_fast++;
                const float l = ml_logl_f(pr, e.yPy, device_logdet_H(pr, lf, lane, vals));
                if (l != l || l > ml_l) { ml_l = l; ml_lam = lf; }
            }
        }
        if (lane == 0) { pr.lalt[g] = ml_l; pr.lamalt[g] = ml_lam; }
    }
    // ---- beta, se, tau, F (pyx:1529-1537, lmm.py:471)
    if (lane == 0) {
        const float b = __fdiv_rn(best_e.Pyx_c, best_e.Pxx_c);
        const float pxx = (PG_MINV > best_e.Pxx_c) ? PG_MINV : best_e.Pxx_c;
        const float sb = (float)(__dsqrt_rn((double)best_e.yPy) / ((double)sqrtf(pxx) * __dsqrt_rn((double)(pr.nu))));
        const float ta = __fdiv_rn((float)pr.nu, best_e.yPy);
        const double t = (double)__fdiv_rn(b, sb);
        pr.beta[g] = b; pr.se[g] = sb; pr.tau[g] = ta; pr.lam[g] = best_lambda; pr.F[g] = t * t;
        if (pr.stats) { atomicAdd(&pr.stats[0], (unsigned long long)n_fast); atomicAdd(&pr.stats[1], (unsigned long long)n_full); }
        if (pr.trace) pr.trace[g] = n_fast | (n_full << 16);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
template <int C, bool LRT = false>
static int launch_assoc(pg_ctx *ctx, AssocParams &pr)
{
    // launch size from the instantiation's own Shape (the same constexpr the kernel indexes with), checked on the host
    // against the device limit before anything is launched
    const size_t lds = (size_t)WPB * assoc_per_wave_bytes(Shape<C>::M, Shape<C>::SWEEP_LDS, sizeof(EvalOut), pr.n_vals);
    static_assert(Shape<C>::M == C + 2, "Shape<C>::M");
    PG_REQUIRE(pr.n_vals >= 0 && lds <= 160 * 1024, "assoc: c = %d, n_vals = %d need %zu bytes of LDS per workgroup", C, pr.n_vals, lds);
    setup_tabs_kernel<C><<<NLAM, 64, (size_t)pr.n_vals * 4 + 16, ctx->stream>>>(pr);
    PG_HIP(hipGetLastError());
    if (lds > 64 * 1024)
        PG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&assoc_kernel<C, LRT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long nblk = (pr.p + WPB - 1) / WPB;
    assoc_kernel<C, LRT><<<dim3((unsigned)nblk), 64 * WPB, lds, ctx->stream>>>(pr);
    PG_HIP(hipGetLastError());
    return PG_OK;
}


// The association kernel is instantiated per covariate count (and once more with the LRT search, N2); the instantiations
// are split over translation units so that they compile in parallel:
//   assoc.hip          c = 1..15            assoc_hi.hip      c = 16..PG_MAX_COVARIATES
//   assoc_lrt.hip      LRT, c = 0..15       assoc_lrt_hi.hip  LRT, c = 16..PG_MAX_COVARIATES
// (c = 0 exists for the LRT's null model at c = 1: the covariates alone, run as "c - 1 covariates + the last one as the SNP").
int launch_assoc_hi(pg_ctx *ctx, AssocParams &pr);
int launch_assoc_lrt(pg_ctx *ctx, AssocParams &pr);
int launch_assoc_lrt_hi(pg_ctx *ctx, AssocParams &pr);
#define PG_CASE(CC) case CC: return launch_assoc<CC, PG_CASE_LRT>(ctx, pr);
#if defined(PG_ASSOC_PART) && PG_ASSOC_PART == 1
#define PG_CASE_LRT false
int launch_assoc_hi(pg_ctx *ctx, AssocParams &pr)
{
    switch (pr.c) {
        PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20) PG_CASE(21) PG_CASE(22) PG_CASE(23) PG_CASE(24) PG_CASE(25)
        PG_CASE(26) PG_CASE(27) PG_CASE(28) PG_CASE(29) PG_CASE(30)
        default: return PG_ENOTSUP;
    }
}
#elif defined(PG_ASSOC_PART) && PG_ASSOC_PART == 2
#define PG_CASE_LRT true
int launch_assoc_lrt(pg_ctx *ctx, AssocParams &pr)
{
    switch (pr.c) {
        PG_CASE(0) PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
        PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15)
        default: return launch_assoc_lrt_hi(ctx, pr);
    }
}
#elif defined(PG_ASSOC_PART) && PG_ASSOC_PART == 3
#define PG_CASE_LRT true
int launch_assoc_lrt_hi(pg_ctx *ctx, AssocParams &pr)
{
    switch (pr.c) {
        PG_CASE(16) PG_CASE(17) PG_CASE(18) PG_CASE(19) PG_CASE(20) PG_CASE(21) PG_CASE(22) PG_CASE(23) PG_CASE(24) PG_CASE(25)
        PG_CASE(26) PG_CASE(27) PG_CASE(28) PG_CASE(29) PG_CASE(30)
        default: return PG_ENOTSUP;
    }
}
#else
#define PG_CASE_LRT false
static int launch_assoc_any(pg_ctx *ctx, AssocParams &pr)
{
    if (pr.lrt) return launch_assoc_lrt(ctx, pr);
    switch (pr.c) {
#ifdef PG_ONLY_C
        PG_CASE(PG_ONLY_C)
#else
        PG_CASE(1) PG_CASE(2) PG_CASE(3) PG_CASE(4) PG_CASE(5) PG_CASE(6) PG_CASE(7) PG_CASE(8) PG_CASE(9) PG_CASE(10)
        PG_CASE(11) PG_CASE(12) PG_CASE(13) PG_CASE(14) PG_CASE(15)
#endif
        default: return launch_assoc_hi(ctx, pr);
    }
}
#ifdef PG_ONLY_C   // single-c development builds (A/B timing): the LRT pair c, c-1 in this translation unit, nothing else
#undef PG_CASE_LRT
#define PG_CASE_LRT true
int launch_assoc_lrt(pg_ctx *ctx, AssocParams &pr)
{
    switch (pr.c) { PG_CASE(PG_ONLY_C) case PG_ONLY_C - 1: return launch_assoc<PG_ONLY_C - 1, true>(ctx, pr); default: return PG_ENOTSUP; }
}
int launch_assoc_hi(pg_ctx *, AssocParams &) { return PG_ENOTSUP; }
#endif
#endif
#undef PG_CASE

#ifndef PG_ASSOC_PART
// ------------------------------------------------------------------------------------------------
// scipy.stats.f.sf(F, 1, dfd) (lmm.py:482): I_{w}(dfd/2, 1/2), w = dfd/(dfd+F) — same statements as
// oracle orc_fdist_sf (continued fraction, modified Lentz).
__device__ double betacf_dev(double a, double b, double x)
{
    const double FPMIN = 1e-300, EPS = 1e-16;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0, c = 1.0, dd = 1.0 - qab * x / qap, h;
    if (fabs(dd) < FPMIN) dd = FPMIN;
    dd = 1.0 / dd; h = dd;
    for (int m = 1; m <= 2000; m++) {
        int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        dd = 1.0 + aa * dd; if (fabs(dd) < FPMIN) dd = FPMIN;
        c = 1.0 + aa / c; if (fabs(c) < FPMIN) c = FPMIN;
        dd = 1.0 / dd; h *= dd * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        dd = 1.0 + aa * dd; if (fabs(dd) < FPMIN) dd = FPMIN;
        c = 1.0 + aa / c; if (fabs(c) < FPMIN) c = FPMIN;
        dd = 1.0 / dd;
        double del = dd * c;
        h *= del;
        if (fabs(del - 1.0) < EPS) break;
    }
    return h;
}
__global__ void fdist_sf_kernel(long long count, const double *F, double dfd, double *pval)
{
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count) return;
    double f = F[g], r;
    if (f != f) r = f;
    else if (f <= 0.0) r = 1.0;
    else if (isinf(f)) r = 0.0;
    else {
        double a = 0.5 * dfd, b = 0.5;
        double x = dfd / (dfd + f), omx = 1.0 - x;
        double lbeta = lgamma(a) + lgamma(b) - lgamma(a + b);
        double lbt = a * log(x) + b * log(omx) - lbeta;
        if (x < (a + 1.0) / (a + b + 2.0)) r = exp(lbt) * betacf_dev(a, b, x) / a;
        else r = 1.0 - exp(lbt) * betacf_dev(b, a, omx) / b;
    }
    pval[g] = r;
}

// packed fixed rows: d, w_0..w_{c-1}, y, zero pad; rows [n, npad) zero
__global__ void build_fixed_kernel(int n, int npad, int c, int ldw, int rowf, const float *d, const float *Wr, const float *yr, float *fixed)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    float *row = fixed + (size_t)i * rowf;
    for (int k = 0; k < rowf; k++) row[k] = 0.0f;
    if (i >= n) row[0] = INFINITY;      // h = 1/(lam*inf + 1) = 0: pad elements contribute exact zeros without any select
    if (i < n) {
        row[0] = d[i];
        for (int j = 0; j < c; j++) row[1 + j] = Wr[(size_t)i * ldw + j];
        row[c + 1] = yr[i];
    }
}

// (n x p row-major) -> SNP-major (p x ldx), 32x32 LDS tile transpose, pad zeroed
__global__ __launch_bounds__(256) void transpose_kernel(long long n, long long p, const float *X, long long ldX, float *Xr, long long ldx)
{
    __shared__ float tile[32][33];
    const long long g0 = (long long)blockIdx.x * 32, i0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        long long i = i0 + r, g = g0 + tx;
        tile[r][tx] = (i < n && g < p) ? X[(size_t)i * ldX + g] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        long long g = g0 + r, i = i0 + tx;
        if (g < p && i < ldx) Xr[(size_t)g * ldx + i] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// Inspection surface (the functions tests/test_pygemma.py:256-294 calls directly): precompute_mat (pyx:880) with
// every level exported, the *_overload scalars, newton.  One wavefront, any m = ctot+1 <= PG_MAX_COVARIATES + 2,
// written for checking, not for speed: level-0 Gram entries one at a time with exactly the arithmetic of
// gram_pass (per-lane fma chain over i = lane, lane+64, .., xor butterfly 1..32), then the sweeps run serially.
constexpr int IM = PG_MAX_COVARIATES + 2;

struct InspectParams {
    AssocParams pr;          // n, nu = n - ctot, logl_c, fixed = d (rowf = 1), the pairwise-sum plan
    int ctot, m, full;
    float lam, lam_min, lam_max;
    const float *d, *Wx, *y; // Wx: n x ctot row-major (the reference's np.c_[W, x]); y: n
    float *P3, *Q3, *R3;     // [m][m][m] indexed [row][level][col]; undefined entries NaN
    float *vecs;             // [5][m]: yt_Pi_y, yt_Pi_Pi_y, yt_Pi_Pi_Pi_y, tr_Pi, tr_Pi_Pi per level
    float *scal;             // logdet_Wt_H_inv_W, logdet_H, d1, d2 (full only), logL at the last level
    float *out;              // newton: the root
};

__device__ __forceinline__ float inspect_col(const InspectParams &ip, int j, int i)
{
    return (j < ip.ctot) ? ip.Wx[(size_t)i * ip.ctot + j] : ip.y[i];
}

// level-0 Grams into G[3][IM][IM] (lower triangles) + traces; all lanes hold the same values afterwards
__device__ void inspect_gram0(const InspectParams &ip, float lam, bool full, int lane, double (*G)[IM][IM], double *tr)
{
    const int n = ip.pr.n, m = ip.m;
    for (int j = 0; j < m; j++)
        for (int k = 0; k <= j; k++) {
            double P = 0.0, Q = 0.0, R = 0.0;
            for (int i = lane; i < n; i += 64) {
                const double hd = (double)hinv_f32(lam, ip.d[i]);
                const double cj = (double)inspect_col(ip, j, i), ck = (double)inspect_col(ip, k, i);
                const double aj = hd * cj, ak = hd * ck;
                P = fma(aj, ck, P);
                Q = fma(aj, ak, Q);
                if (full) { const double gj = (hd * hd) * cj; R = fma(gj, ak, R); }
            }
            P = bfly(P); Q = bfly(Q); R = bfly(R);
            if (lane == 0) { G[0][j][k] = P; G[1][j][k] = Q; G[2][j][k] = full ? R : 0.0; }
        }
    double s1 = 0.0, s2 = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double hd = (double)hinv_f32(lam, ip.d[i]);
        s1 += hd;
        s2 = fma(hd, hd, s2);
    }
    s1 = bfly(s1); s2 = bfly(s2);
    if (lane == 0) { tr[0] = s1; tr[1] = full ? s2 : 0.0; }
    wave_lds_sync();
}

// the c_tot sweeps (pyx:947-963 / :1007-1036) on lane 0, statement for statement the oracle's sweeps(order=1);
// exp: optional export of every level.  Returns the last level's forms in o.
__device__ void inspect_sweeps(const InspectParams &ip, bool full, double (*G)[IM][IM], const double *tr, bool exp, EvalOut &o)
{
    const int m = ip.m;
    double (*P)[IM] = G[0], (*Q)[IM] = G[1], (*R)[IM] = G[2];
    double trP = tr[0], trPP = tr[1];
    float ld = 0.0f;
    P[0][0] = dmaxf(P[0][0], PG_MINV);
    auto level_out = [&](int lv) {
        if (!exp) return;
        for (int r = lv; r < m; r++)
            for (int c = lv; c <= r; c++) {
                const size_t idx = ((size_t)r * m + lv) * m + c;
                ip.P3[idx] = (float)P[r][c]; ip.Q3[idx] = (float)Q[r][c];
                if (full) ip.R3[idx] = (float)R[r][c];
            }
        ip.vecs[0 * m + lv] = (float)P[m - 1][m - 1];
        ip.vecs[1 * m + lv] = (float)Q[m - 1][m - 1];
        if (full) ip.vecs[2 * m + lv] = (float)R[m - 1][m - 1];
        ip.vecs[3 * m + lv] = (float)trP;
        ip.vecs[4 * m + lv] = (float)trPP;
    };
    level_out(0);
    for (int i = 1; i < m; i++) {
        const int q = i - 1;
        const double a = P[q][q], b = Q[q][q], e = R[q][q];
        double u[IM], v[IM], w[IM];
        for (int r = i; r < m; r++) { u[r] = P[r][q]; v[r] = Q[r][q]; w[r] = R[r][q]; }
        if (full) {
            const double ba = b / a;
            trPP = trPP + ba * ba - 2 * (e / a);
            const double a2 = a * a, a3 = a2 * a, b2 = b * b;
            const double cR = (e / a2) - (b2 / a3), ia = -1.0 / a, ba2 = b / a2;
            for (int r = i; r < m; r++)
                for (int c = i; c <= r; c++) {
                    const double t1 = fma(cR * u[c], u[r], R[r][c]);
                    double t2 = fma(ia * u[c], w[r], 0.0); t2 = fma(ia * w[c], u[r], t2);
                    const double t3 = fma(ia * v[c], v[r], 0.0);
                    double t4 = fma(ba2 * u[c], v[r], 0.0); t4 = fma(ba2 * v[c], u[r], t4);
                    R[r][c] = ((t1 + t2) + t3) + t4;
                }
            R[i][i] = dmaxf(R[i][i], PG_MINV);
        }
        trP = trP - b / a;
        {
            const double a2 = a * a, al1 = b / a2, ia = -1.0 / a;
            for (int r = i; r < m; r++)
                for (int c = i; c <= r; c++) {
                    const double t1 = fma(al1 * u[c], u[r], Q[r][c]);
                    double t2 = fma(ia * u[c], v[r], 0.0); t2 = fma(ia * v[c], u[r], t2);
                    Q[r][c] = t1 + t2;
                }
            Q[i][i] = dmaxf(Q[i][i], PG_MINV);
        }
        ld = (float)((double)ld + log(a));
        {
            const double ia = -1.0 / a;
            for (int r = i; r < m; r++)
                for (int c = i; c <= r; c++) P[r][c] = fma(ia * u[c], u[r], P[r][c]);
            P[i][i] = dmaxf(P[i][i], PG_MINV);
        }
        if (!full) trPP = 0.0;
        level_out(i);
    }
    o.yPy = (float)P[m - 1][m - 1]; o.yPPy = (float)Q[m - 1][m - 1]; o.yPPPy = (float)R[m - 1][m - 1];
    o.trP = (float)trP; o.trPP = (float)trPP; o.ld = ld;
    o.Pxx_c = (m >= 2) ? (float)P[m - 2][m - 2] : 0.0f; o.Pyx_c = (m >= 2) ? (float)P[m - 1][m - 2] : 0.0f;
}

__global__ __launch_bounds__(64) void precompute_kernel(InspectParams ip)
{
    __shared__ double G[3][IM][IM];
    __shared__ double tr[2];
    __shared__ EvalOut eo;
    extern __shared__ float vals_dyn[];
    const int lane = threadIdx.x, m = ip.m;
    for (int idx = lane; idx < m * m * m; idx += 64) { ip.P3[idx] = NAN; ip.Q3[idx] = NAN; ip.R3[idx] = NAN; }
    for (int idx = lane; idx < 5 * m; idx += 64) ip.vecs[idx] = NAN;
    for (int idx = lane; idx < 3 * IM * IM; idx += 64) (&G[0][0][0])[idx] = 0.0;
    wave_lds_sync();
    inspect_gram0(ip, ip.lam, ip.full != 0, lane, G, tr);
    if (lane == 0) inspect_sweeps(ip, ip.full != 0, G, tr, true, eo);
    wave_lds_sync();
    const float ldH = device_logdet_H(ip.pr, ip.lam, lane, vals_dyn);
    if (lane == 0) {
        ip.scal[0] = eo.ld; ip.scal[1] = ldH;
        ip.scal[2] = d1_f(ip.pr, ip.lam, eo.yPy, eo.yPPy, eo.trP);
        ip.scal[3] = ip.full ? d2_f(ip.pr, ip.lam, eo.yPy, eo.yPPy, eo.yPPPy, eo.trP, eo.trPP) : NAN;
        ip.scal[4] = logl_f(ip.pr, eo.yPy, ldH, eo.ld);
        ip.scal[5] = (float)tr[0];                          // sum h   (un-projected trace of H^-1: the ML functions, pyx:1567-1603)
        ip.scal[6] = ip.full ? (float)tr[1] : NAN;          // sum h^2
    }
}

__global__ __launch_bounds__(64) void newton_kernel(InspectParams ip)
{
    __shared__ double G[3][IM][IM];
    __shared__ double tr[2];
    __shared__ EvalOut eo;
    const int lane = threadIdx.x;
    const float root = newton_dev(ip.pr, ip.lam, ip.lam_min, ip.lam_max, [&](float lf, EvalOut &e) {
        for (int idx = lane; idx < 3 * IM * IM; idx += 64) (&G[0][0][0])[idx] = 0.0;
        wave_lds_sync();
        inspect_gram0(ip, lf, true, lane, G, tr);
        if (lane == 0) inspect_sweeps(ip, true, G, tr, false, eo);
        wave_lds_sync();
        e = eo;
        wave_lds_sync();
    });
    if (lane == 0) ip.out[0] = root;
}

// the three *_overload scalars on caller-supplied quadratic forms (pyx:1813, :1656, :1675); args: lam, yPy, yPPy, yPPPy,
// trP, trPP, logdet_H, logdet_Wt_H_inv_W
__global__ void reml_scalars_kernel(AssocParams pr, const float *a, float *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = logl_f(pr, a[1], a[6], a[7]);
    out[1] = d1_f(pr, a[0], a[1], a[2], a[4]);
    out[2] = d2_f(pr, a[0], a[1], a[2], a[3], a[4], a[5]);
}

}  // namespace pg

using namespace pg;

extern "C" int pg_fdist_sf_dev(pg_ctx *ctx, int64_t count, const double *F, double dfd, double *pval)
{
    PG_REQUIRE(ctx && F && pval && count >= 0, "pg_fdist_sf_dev: bad arguments");
    if (count == 0) return PG_OK;
    PG_HIP(hipSetDevice(ctx->device));
    fdist_sf_kernel<<<(unsigned)((count + 255) / 256), 256, 0, ctx->stream>>>(count, F, dfd, pval);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

extern "C" int pg_transpose_dev(pg_ctx *ctx, int64_t n, int64_t p, const float *X, int64_t ldX, float *Xr, int64_t ldx)
{
    PG_REQUIRE(ctx && X && Xr && n > 0 && p > 0 && ldx >= n && ldX >= p, "pg_transpose_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)((p + 31) / 32), (unsigned)((ldx + 31) / 32));
    transpose_kernel<<<grid, 256, 0, ctx->stream>>>(n, p, X, ldX, Xr, ldx);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// fixed rows + tables + the kernel for one (covariates, SNP block); Wr is n x c with row stride ldw >= c (the LRT's null model
// passes the first c columns of a wider matrix)
static int assoc_run(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, int64_t ldw, const float *yr,
                     const float *Xr, int64_t ldx, int grid, float *beta, float *se, float *tau, float *lambda, double *F,
                     unsigned long long *stats_dev, bool lrt, float *lalt, float *lamalt)
{
    PG_HIP(hipSetDevice(ctx->device));
    int rc = build_npsum_plan(ctx, n);
    if (rc) return rc;

    AssocParams pr{};
    pr.n = (int)n; pr.npad = (int)((n + 63) / 64 * 64); pr.c = c; pr.niter = pr.npad / 64;
    pr.nu = (int)(n - c - 1); pr.grid = grid ? 1 : 0; pr.rowf = ((c + 2 + 3) / 4) * 4;
    pr.p = p; pr.ldx = ldx; pr.xr = Xr;
    const int M = c + 2, NP = M * (M + 1) / 2;
    rc = ensure(ctx, &ctx->fixed, &ctx->fixed_bytes, (size_t)pr.npad * pr.rowf * 4);
    if (rc) return rc;
    const size_t off_fixg = ((size_t)NLAM * pr.npad * 4 + 255) & ~(size_t)255;
    const size_t off_t1 = off_fixg + (size_t)NLAM * 2 * NP * 8;
    const size_t off_ldh = off_t1 + NLAM * 8;
    rc = ensure(ctx, &ctx->tabs, &ctx->tabs_bytes, off_ldh + NLAM * 4 + 256);
    if (rc) return rc;
    if (!ctx->stats) PG_HIP(hipMalloc(&ctx->stats, 16));
    pr.fixed = (const float *)ctx->fixed;
    pr.htab = (float *)ctx->tabs;
    pr.fixg = (double *)((char *)ctx->tabs + off_fixg);
    pr.t1tab = (double *)((char *)ctx->tabs + off_t1);
    pr.ldHtab = (float *)((char *)ctx->tabs + off_ldh);
    for (int k = -5; k <= 5; k++) pr.lam11[k + 5] = (float)pow(10.0, (double)(float)k);  // pyx:122,157-158
    {
        const int ctot = c + 1;
        float r = (float)((0.5 * (double)(n - ctot)) * std::log(0.5 * (double)(n - ctot) / M_PI));  // pyx:1821
        r = (float)((double)r - (0.5 * (double)(n - ctot)));                                          // pyx:1822
        pr.logl_c = r;
    }
    pr.lrt = lrt ? 1 : 0;
    pr.nhalf = (int)(n / 2);                                                                          // (n/2): C integer division
    {
        float r = (float)((double)(n / 2) * std::log((double)n / (2.0 * M_PI)));                      // pyx:1552
        pr.ml_c = r - (float)(n / 2);                                                                 // pyx:1554
    }
    pr.lalt = lalt; pr.lamalt = lamalt;
    pr.leaf = ctx->plan.d_leaf; pr.node = ctx->plan.d_node; pr.level = ctx->plan.d_level; pr.chunk = ctx->plan.d_chunk;
    pr.n_leaf = ctx->plan.n_leaf; pr.n_level = ctx->plan.n_level; pr.n_chunk = ctx->plan.n_chunk;
    pr.n_vals = ctx->plan.n_leaf + ctx->plan.n_node;
    pr.beta = beta; pr.se = se; pr.tau = tau; pr.lam = lambda; pr.F = F;
    pr.stats = stats_dev;
    pr.trace = lrt ? nullptr : ctx->eval_trace;

    build_fixed_kernel<<<(pr.npad + 255) / 256, 256, 0, ctx->stream>>>(pr.n, pr.npad, c, (int)ldw, pr.rowf, d, Wr, yr, (float *)ctx->fixed);
    PG_HIP(hipGetLastError());
    return launch_assoc_any(ctx, pr);
}

extern "C" int pg_assoc_dev(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                            const float *Xr, int64_t ldx, int grid, float *beta, float *se, float *tau, float *lambda,
                            double *F, double *pval, unsigned long long *stats_dev)
{
    PG_REQUIRE(ctx && d && Wr && yr && Xr && beta && se && tau && lambda && F, "pg_assoc_dev: NULL argument");
    PG_REQUIRE(n >= 2 && n < (1LL << 30) && p >= 0 && ldx >= n, "pg_assoc_dev: bad shape n=%lld p=%lld ldx=%lld", (long long)n, (long long)p, (long long)ldx);
    if (c < 1 || c > PG_MAX_COVARIATES) {
        set_error("pg_assoc_dev: c=%d covariates not supported by this build (1..%d)", c, PG_MAX_COVARIATES);
        return PG_ENOTSUP;
    }
    PG_REQUIRE(n - c - 1 > 0, "pg_assoc_dev: n - c - 1 must be positive");
    if (p == 0) return PG_OK;
    int rc = assoc_run(ctx, n, c, p, d, Wr, c, yr, Xr, ldx, grid, beta, se, tau, lambda, F, stats_dev, false, nullptr, nullptr);
    if (rc) return rc;
    if (pval) return pg_fdist_sf_dev(ctx, p, F, (double)(n - c - 1), pval);
    return PG_OK;
}

// Everything of the first pg_assoc_dev / pg_assoc_lrt_dev call of a context that blocks the host — the float32-sum plan (built on the
// host, uploaded with synchronous copies after a stream drain) and the scratch allocations — done ahead of time, so that a streaming
// caller's first batch does not stall between its rotation and its association kernel (r4: 6 ms per worker at n = 10 000).
extern "C" int pg_assoc_warm(pg_ctx *ctx, int64_t n, int c)
{
    PG_REQUIRE(ctx && n >= 2 && n < (1LL << 30), "pg_assoc_warm: bad arguments");
    if (c < 1 || c > PG_MAX_COVARIATES) {
        set_error("pg_assoc_warm: c=%d covariates not supported by this build (1..%d)", c, PG_MAX_COVARIATES);
        return PG_ENOTSUP;
    }
    PG_HIP(hipSetDevice(ctx->device));
    int rc = build_npsum_plan(ctx, n);
    if (rc) return rc;
    const int npad = (int)((n + 63) / 64 * 64), rowf = ((c + 2 + 3) / 4) * 4, M = c + 2, NP = M * (M + 1) / 2;
    rc = ensure(ctx, &ctx->fixed, &ctx->fixed_bytes, (size_t)npad * rowf * 4);
    if (rc) return rc;
    const size_t off_fixg = ((size_t)NLAM * npad * 4 + 255) & ~(size_t)255;
    rc = ensure(ctx, &ctx->tabs, &ctx->tabs_bytes, off_fixg + (size_t)NLAM * 2 * NP * 8 + NLAM * 8 + NLAM * 4 + 256);
    if (rc) return rc;
    if (!ctx->stats) PG_HIP(hipMalloc(&ctx->stats, 16));
    return PG_OK;
}

// Work-per-SNP trace for tail analysis (the Brent/Newton path is data-dependent, pyx:1349-1416: up to 101 Newton iterations):
// the next pg_assoc_dev calls of this context write, per SNP, fast evaluations | full evaluations << 16 into trace_dev
// (>= p entries); NULL switches it off.
extern "C" int pg_assoc_set_eval_trace(pg_ctx *ctx, unsigned *trace_dev)
{
    PG_REQUIRE(ctx, "pg_assoc_set_eval_trace: NULL ctx");
    ctx->eval_trace = trace_dev;
    return PG_OK;
}

namespace pg {
__global__ void gather_col_kernel(int n, int npad, const float *W, int ldw, int col, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) out[i] = (i < n) ? W[(size_t)i * ldw + col] : 0.0f;
}
// D_lrt = 2 (l_alt - l_null) on float32 scalars (lmm/lmm.py:283), p_lrt = chi2(1).sf(D_lrt) = erfc(sqrt(D/2)) (lmm.py:300 writes
// 1 - cdf: the same number above ~1e-16); all four columns widened to float64 like the frame's lambda column
__global__ void lrt_finish_kernel(long long p, const float *lalt, const float *lnull, double *o_lalt, double *o_lnull, double *o_D, double *o_p)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p) return;
    const float la = lalt[g], ln = lnull[0];
    const float D = __fmul_rn(2.0f, __fsub_rn(la, ln));
    const double Dd = (double)D;
    o_lalt[g] = (double)la; o_lnull[g] = (double)ln; o_D[g] = Dd;
    o_p[g] = (Dd != Dd) ? Dd : (Dd <= 0.0 ? 1.0 : erfc(sqrt(0.5 * Dd)));
}
}  // namespace pg

// ---- N2: pg_assoc_dev plus the likelihood-ratio columns the reference sketches and leaves commented out
// (lmm/lmm.py:137-141, 277-300) from its own ML functions (calc_lambda lmm.py:22-84; likelihood_lambda and derivatives
// pygemma_model.pyx:1542-1603).  Same kernel, instantiated with the ML lambda search appended: the decade scan is shared with the
// REML search (its quadratic forms serve both), roots are refined by brentq + Newton like the reference's scipy calls.
// l_null comes from the same kernel run on the covariates alone (c - 1 covariates + the last one in the SNP's place).
extern "C" int pg_assoc_lrt_dev(pg_ctx *ctx, int64_t n, int c, int64_t p, const float *d, const float *Wr, const float *yr,
                                const float *Xr, int64_t ldx, int grid, float *beta, float *se, float *tau, float *lambda,
                                double *F, double *pval, double *l_alt, double *l_null, double *D_lrt, double *p_lrt)
{
    PG_REQUIRE(ctx && d && Wr && yr && Xr && beta && se && tau && lambda && F && l_alt && l_null && D_lrt && p_lrt, "pg_assoc_lrt_dev: NULL argument");
    PG_REQUIRE(n >= 2 && n < (1LL << 30) && p >= 0 && ldx >= n, "pg_assoc_lrt_dev: bad shape");
    if (c < 1 || c > PG_MAX_COVARIATES) {
        set_error("pg_assoc_lrt_dev: c=%d covariates not supported by this build (1..%d)", c, PG_MAX_COVARIATES);
        return PG_ENOTSUP;
    }
    PG_REQUIRE(n - c - 1 > 0, "pg_assoc_lrt_dev: n - c - 1 must be positive");
    if (p == 0) return PG_OK;
    PG_HIP(hipSetDevice(ctx->device));
    const int64_t npad = (n + 63) / 64 * 64;
    // scratch: [x_null (npad f32) | null outputs: 4 f32 + 1 f64 | l_null f32 | lam_null f32 | l_alt (p f32) | lam_alt (p f32)]
    const size_t off_out = (size_t)npad * 4, off_la = off_out + 64, need = off_la + (size_t)p * 8 + 64;
    int rc = ensure(ctx, &ctx->scratch, &ctx->scratch_bytes, need);
    if (rc) return rc;
    char *sc = (char *)ctx->scratch;
    float *xnull = (float *)sc, *nb = (float *)(sc + off_out), *lnull = nb + 8, *lamnull = nb + 9;
    double *nF = (double *)(sc + off_out + 16);
    float *lalt = (float *)(sc + off_la), *lamalt = lalt + p;
    gather_col_kernel<<<(unsigned)((npad + 255) / 256), 256, 0, ctx->stream>>>((int)n, (int)npad, Wr, c, c - 1, xnull);
    PG_HIP(hipGetLastError());
    rc = assoc_run(ctx, n, c - 1, 1, d, Wr, c, yr, xnull, npad, grid, nb, nb + 1, nb + 2, nb + 3, nF, nullptr, true, lnull, lamnull);
    if (rc) return rc;
    rc = assoc_run(ctx, n, c, p, d, Wr, c, yr, Xr, ldx, grid, beta, se, tau, lambda, F, nullptr, true, lalt, lamalt);
    if (rc) return rc;
    lrt_finish_kernel<<<(unsigned)((p + 255) / 256), 256, 0, ctx->stream>>>(p, lalt, lnull, l_alt, l_null, D_lrt, p_lrt);
    PG_HIP(hipGetLastError());
    if (pval) return pg_fdist_sf_dev(ctx, p, F, (double)(n - c - 1), pval);
    return PG_OK;
}

// ---- inspection surface ----------------------------------------------------------------------------------------------
static int inspect_params(pg_ctx *ctx, int64_t n, int ctot, InspectParams &ip, const float *d)
{
    PG_REQUIRE(n >= 2 && n < (1LL << 30), "inspect: bad n");
    if (ctot < 1 || ctot > PG_MAX_COVARIATES + 1) {
        set_error("inspect: %d columns not supported by this build (1..%d)", ctot, PG_MAX_COVARIATES + 1);
        return PG_ENOTSUP;
    }
    PG_REQUIRE(n - ctot > 0, "inspect: n - c must be positive");
    PG_HIP(hipSetDevice(ctx->device));
    int rc = build_npsum_plan(ctx, n);
    if (rc) return rc;
    AssocParams &pr = ip.pr;
    pr.n = (int)n; pr.npad = (int)((n + 63) / 64 * 64); pr.c = ctot - 1; pr.niter = pr.npad / 64;
    pr.nu = (int)(n - ctot); pr.rowf = 1; pr.fixed = d;
    {
        float r = (float)((0.5 * (double)(n - ctot)) * std::log(0.5 * (double)(n - ctot) / M_PI));  // pyx:1821
        r = (float)((double)r - (0.5 * (double)(n - ctot)));                                          // pyx:1822
        pr.logl_c = r;
    }
    pr.leaf = ctx->plan.d_leaf; pr.node = ctx->plan.d_node; pr.level = ctx->plan.d_level; pr.chunk = ctx->plan.d_chunk;
    pr.n_leaf = ctx->plan.n_leaf; pr.n_level = ctx->plan.n_level; pr.n_chunk = ctx->plan.n_chunk;
    pr.n_vals = ctx->plan.n_leaf + ctx->plan.n_node;
    ip.ctot = ctot; ip.m = ctot + 1; ip.d = d;
    return PG_OK;
}

extern "C" int pg_precompute_mat_dev(pg_ctx *ctx, int64_t n, int ctot, float lam, const float *d, const float *Wx, const float *y,
                                     int full, float *P3, float *Q3, float *R3, float *vecs, float *scal)
{
    PG_REQUIRE(ctx && d && Wx && y && P3 && Q3 && R3 && vecs && scal, "pg_precompute_mat_dev: NULL argument");
    InspectParams ip{};
    int rc = inspect_params(ctx, n, ctot, ip, d);
    if (rc) return rc;
    ip.full = full ? 1 : 0; ip.lam = lam; ip.Wx = Wx; ip.y = y;
    ip.P3 = P3; ip.Q3 = Q3; ip.R3 = R3; ip.vecs = vecs; ip.scal = scal;
    precompute_kernel<<<1, 64, (size_t)ip.pr.n_vals * 4 + 16, ctx->stream>>>(ip);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

extern "C" int pg_newton_dev(pg_ctx *ctx, int64_t n, int ctot, float lam, float lam_min, float lam_max, const float *d,
                             const float *Wx, const float *y, float *root)
{
    PG_REQUIRE(ctx && d && Wx && y && root, "pg_newton_dev: NULL argument");
    InspectParams ip{};
    int rc = inspect_params(ctx, n, ctot, ip, d);
    if (rc) return rc;
    ip.full = 1; ip.lam = lam; ip.lam_min = lam_min; ip.lam_max = lam_max; ip.Wx = Wx; ip.y = y; ip.out = root;
    newton_kernel<<<1, 64, 0, ctx->stream>>>(ip);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

extern "C" int pg_reml_scalars_dev(pg_ctx *ctx, int64_t n, int ctot, const float *args8, float *out3)
{
    PG_REQUIRE(ctx && args8 && out3 && n - ctot > 0, "pg_reml_scalars_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    AssocParams pr{};
    pr.n = (int)n; pr.nu = (int)(n - ctot);
    {
        float r = (float)((0.5 * (double)(n - ctot)) * std::log(0.5 * (double)(n - ctot) / M_PI));
        r = (float)((double)r - (0.5 * (double)(n - ctot)));
        pr.logl_c = r;
    }
    reml_scalars_kernel<<<1, 1, 0, ctx->stream>>>(pr, args8, out3);
    PG_HIP(hipGetLastError());
    return PG_OK;
}

// N2 at the model level: likelihood_lambda (pyx:1542-1562), likelihood_derivative1_lambda (pyx:1567-1581),
// likelihood_derivative2_lambda (pyx:1586-1603) from the quadratic forms of precompute_mat — the statements the LRT kernel runs
__global__ void ml_scalars_kernel(AssocParams pr, const float *a, float *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = ml_logl_f(pr, a[1], a[6]);
    out[1] = ml_d1_f(pr, a[0], a[1], a[2], a[4]);
    out[2] = ml_d2_f(pr, a[0], a[1], a[2], a[3], a[4], a[5]);
}
extern "C" int pg_ml_scalars_dev(pg_ctx *ctx, int64_t n, const float *args7, float *out3)
{
    PG_REQUIRE(ctx && args7 && out3 && n >= 2 && n < (1LL << 30), "pg_ml_scalars_dev: bad arguments");
    PG_HIP(hipSetDevice(ctx->device));
    AssocParams pr{};
    pr.n = (int)n;
    pr.nhalf = (int)(n / 2);                                                     // (n/2): C integer division
    {
        float r = (float)((double)(n / 2) * std::log((double)n / (2.0 * M_PI)));  // pyx:1552
        pr.ml_c = r - (float)(n / 2);                                             // pyx:1554
    }
    ml_scalars_kernel<<<1, 1, 0, ctx->stream>>>(pr, args7, out3);
    PG_HIP(hipGetLastError());
    return PG_OK;
}
#else
}  // namespace pg
#endif  // PG_ASSOC_PART
