// probe_f16_mfma32.hip — operand / accumulator lane maps of v_mfma_f32_32x32x16_f16 on gfx950, checked with exact small integers:
//   A: lane l holds row l & 31, k = 8 (l >> 5) + j (j = 0..7);  B: column l & 31, the same k;
//   D: lane l, element e -> column l & 31, row (e & 3) + 8 (e >> 2) + 4 (l >> 5).
// build: hipcc -O3 --offload-arch=gfx950 tools/probe_f16_mfma32.hip -o /tmp/probe_f16_mfma32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const halfx8 *a, const halfx8 *b, floatx16 *c)
{
    const int l = threadIdx.x;
    floatx16 acc;
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[l], b[l], acc, 0, 0, 0);
    c[l] = acc;
}
int main()
{
    const int M = 32, N = 32, K = 16;
    float A[M][K], B[K][N], Cref[M][N];
    for (int r = 0; r < M; r++) for (int kk = 0; kk < K; kk++) A[r][kk] = (float)((r * 7 + kk * 3) % 11 - 5);
    for (int kk = 0; kk < K; kk++) for (int c = 0; c < N; c++) B[kk][c] = (float)((kk * 5 + c * 2 + (kk * c) % 3) % 13 - 6);
    for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) { float s = 0; for (int kk = 0; kk < K; kk++) s += A[r][kk] * B[kk][c]; Cref[r][c] = s; }
    _Float16 ha[64][8], hb[64][8];
    for (int l = 0; l < 64; l++) for (int j = 0; j < 8; j++) { const int kk = 8 * (l >> 5) + j; ha[l][j] = (_Float16)A[l & 31][kk]; hb[l][j] = (_Float16)B[kk][l & 31]; }
    halfx8 *da, *db; floatx16 *dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 64 * 64);
    hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc); hipDeviceSynchronize();
    float hc[64][16]; hipMemcpy(hc, dc, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int e = 0; e < 16; e++) {
        const int col = l & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        if (hc[l][e] != Cref[row][col]) bad++;
    }
    printf("v_mfma_f32_32x32x16_f16: %d / 1024 accumulator mismatches with the hypothesised maps\n", bad);
    return bad != 0;
}
