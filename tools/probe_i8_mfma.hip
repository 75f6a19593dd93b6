// probe_i8_mfma.hip — determine the operand / accumulator lane maps of v_mfma_i32_32x32x32_i8 on gfx950 with exact data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const v4i* a, const v4i* b, v16i* c) {
    int l = threadIdx.x;
    v16i acc;
    for (int e = 0; e < 16; e++) acc[e] = 0;
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], b[l], acc, 0, 0, 0);
    c[l] = acc;
}
int main() {
    const int M = 32, N = 32, K = 32;
    signed char A[M][K], B[K][N];
    for (int r = 0; r < M; r++) for (int kk = 0; kk < K; kk++) A[r][kk] = (signed char)((r * 7 + kk * 3) % 11 - 5);
    for (int kk = 0; kk < K; kk++) for (int c = 0; c < N; c++) B[kk][c] = (signed char)((kk * 5 + c * 2 + (kk * c) % 3) % 13 - 6);
    int Cref[M][N];
    for (int r = 0; r < M; r++) for (int c = 0; c < N; c++) { int s = 0; for (int kk = 0; kk < K; kk++) s += A[r][kk] * B[kk][c]; Cref[r][c] = s; }
    for (int hyp = 0; hyp < 2; hyp++) {
        signed char ha[64][16], hb[64][16];
        for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
            int h = l >> 5, r = l & 31;
            int kk = (hyp == 0) ? (16 * h + j) : ((j < 8) ? (8 * h + j) : (16 + 8 * h + (j - 8)));
            ha[l][j] = A[r][kk]; hb[l][j] = B[kk][r];
        }
        v4i *da, *db; v16i* dc;
        hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 64 * 64);
        hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
        k<<<1, 64>>>(da, db, dc); hipDeviceSynchronize();
        int hc[64][16]; hipMemcpy(hc, dc, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int e = 0; e < 16; e++) {
            int col = l & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
            if (hc[l][e] != Cref[row][col]) bad++;
        }
        printf("hypothesis %d (k = %s): %d / 1024 accumulator mismatches with the f32 C/D map\n", hyp, hyp == 0 ? "16h+j" : "8h+j | 16+8h+j-8", bad);
    }
    return 0;
}
