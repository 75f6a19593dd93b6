// probe_mfma_f64.hip — how does v_mfma_f64_16x16x4_f64 round?  D = C + sum_{k<4} A[i][k] B[k][j] is tested against candidate
// orders of IEEE operations on random operands with spread exponents (so that the order of the roundings shows).
// Lane maps (dgemm.hpp): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], C/D col = l&15, row = (l>>4) + 4*reg.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *a, const double *b, const d4 *c, d4 *d, int chain)
{
    const int l = threadIdx.x;
    d4 acc = c[l];
    for (int s = 0; s < chain; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s * 64 + l], b[s * 64 + l], acc, 0, 0, 0);
    d[l] = acc;
}
int main()
{
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::uniform_int_distribution<int> E(-20, 20);
    const int TR = 200, CH = 3;
    long bad[6] = {0, 0, 0, 0, 0, 0}, tot = 0;
    double *da, *db; d4 *dc, *dd;
    hipMalloc(&da, CH * 64 * 8); hipMalloc(&db, CH * 64 * 8); hipMalloc(&dc, 64 * 32); hipMalloc(&dd, 64 * 32);
    for (int t = 0; t < TR; t++) {
        double A[CH][16][4], B[CH][4][16], C[16][16], ha[CH * 64], hb[CH * 64], hc[64][4], hd[64][4];
        for (int s = 0; s < CH; s++)
            for (int i = 0; i < 16; i++) for (int kk = 0; kk < 4; kk++) { A[s][i][kk] = std::ldexp(U(rng), E(rng)); B[s][kk][i] = std::ldexp(U(rng), E(rng)); }
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) C[i][j] = std::ldexp(U(rng), E(rng));
        for (int s = 0; s < CH; s++) for (int l = 0; l < 64; l++) { ha[s * 64 + l] = A[s][l & 15][l >> 4]; hb[s * 64 + l] = B[s][l >> 4][l & 15]; }
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) hc[l][r] = C[(l >> 4) + 4 * r][l & 15];
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice); hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
        k<<<1, 64>>>(da, db, dc, dd, CH); hipDeviceSynchronize();
        hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
            const int i = (l >> 4) + 4 * r, j = l & 15;
            double h[6];
            for (int hyp = 0; hyp < 6; hyp++) {
                double acc = C[i][j];
                for (int s = 0; s < CH; s++) {
                    const double *a = A[s][i]; double bb[4] = {B[s][0][j], B[s][1][j], B[s][2][j], B[s][3][j]};
                    if (hyp == 0) for (int kk = 0; kk < 4; kk++) acc = std::fma(a[kk], bb[kk], acc);                 // ascending fma chain
                    else if (hyp == 1) for (int kk = 3; kk >= 0; kk--) acc = std::fma(a[kk], bb[kk], acc);            // descending fma chain
                    else if (hyp == 2) { __float128 e = acc; for (int kk = 0; kk < 4; kk++) e += (__float128)a[kk] * bb[kk]; acc = (double)e; }   // one rounding
                    else if (hyp == 3) { double p = std::fma(a[1], bb[1], a[0] * bb[0]), q = std::fma(a[3], bb[3], a[2] * bb[2]); acc = acc + (p + q); }
                    else if (hyp == 4) { for (int kk = 0; kk < 4; kk++) acc = acc + a[kk] * bb[kk]; }               // unfused mul + add
                    else { __float128 e = 0; for (int kk = 0; kk < 4; kk++) e += (__float128)a[kk] * bb[kk]; acc = acc + (double)e; }  // dot rounded, then add
                }
                h[hyp] = acc;
            }
            tot++;
            for (int hyp = 0; hyp < 6; hyp++) if (h[hyp] != hd[l][r]) bad[hyp]++;
        }
    }
    const char *nm[6] = {"ascending fma chain k=0..3", "descending fma chain k=3..0", "exact sum, one rounding per MFMA", "pairwise products then add",
                         "unfused multiply-add chain", "dot rounded once, then added to C"};
    for (int hyp = 0; hyp < 6; hyp++) printf("%-36s mismatches %ld / %ld\n", nm[hyp], bad[hyp], tot);
    return 0;
}
