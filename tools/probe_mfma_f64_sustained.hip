// probe_mfma_f64_sustained.hip — what the matrix pipe SUSTAINS on v_mfma_f64_16x16x4_f64 (the instruction of every GEMM of the fp64
// eigensolver) with operands in registers and nothing else going on, one and two waves per SIMD, random-ish operands, at the clock the
// chip holds under that load: the real ceiling of csrc/dgemm.hpp, to be read beside the 78.6 TF of the data sheet (128 flop/clk/CU at
// 2.4 GHz).  Also reports cycles per instruction from s_memtime (one wave) and the in-kernel clock (s_memtime / s_memrealtime).
// build: hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_f64_sustained.hip -o tools/probe_mfma_f64_sustained.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double doublex4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256, 4) void burn(long long iters, double *sink, long long *stamps)
{
    const int l = threadIdx.x & 63;
    double a = 0.37 + 0.001 * l, b = -0.21 + 0.002 * (l ^ 21);
    doublex4 c[NACC];
    for (int t = 0; t < NACC; t++) for (int e = 0; e < 4; e++) c[t][e] = 0.0;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (long long it = 0; it < iters; it++)
#pragma unroll
        for (int t = 0; t < NACC; t++) c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[t], 0, 0, 0);
    double out = 0.0;
    for (int t = 0; t < NACC; t++) out += c[t][0];
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 17) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
    if (out == 123456.789) sink[0] = out;
}

template <int NACC>
static void measure(int num_cu, int wg_per_cu, double *sink, long long *stamps)
{
    const int blocks = num_cu * wg_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    burn<NACC><<<blocks, 256>>>(20000, sink, stamps);
    CK(hipDeviceSynchronize());
    for (double target_ms : {1.0, 20.0}) {
        const long long iters = (long long)(target_ms * 1e-3 * 2.4e9 / (64.0 * NACC) / wg_per_cu);
        float ms = 0.0f;
        for (int rep = 0; rep < 4; rep++) {
            CK(hipEventRecord(e0));
            burn<NACC><<<blocks, 256>>>(iters, sink, stamps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        long long st[2];
        CK(hipMemcpy(st, stamps, sizeof(st), hipMemcpyDeviceToHost));
        const double flops = (double)iters * NACC * 2048.0 * 4.0 * blocks;
        printf("  %d accumulators, %d wave(s)/SIMD, %5.1f ms launch: %6.1f TF = %.3f of 78.6 | %.1f cycles per MFMA and wave = %.1f per SIMD | in-kernel clock %.2f GHz\n",
               NACC, wg_per_cu, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 78.6e12, (double)st[0] / ((double)iters * NACC), (double)st[0] / ((double)iters * NACC) / wg_per_cu,
               (double)st[0] / (double)st[1] * 0.1);
    }
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main()
{
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int num_cu = pr.multiProcessorCount;
    double *sink; long long *stamps;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stamps, 64));
    printf("%s, %d CUs: v_mfma_f64_16x16x4_f64 from registers\n", pr.gcnArchName, num_cu);
    // (launch bounds: 4 waves per SIMD must fit, i.e. <= 128 registers: up to 8 accumulators of 8 registers)
    for (int w : {1, 2, 3, 4}) { measure<8>(num_cu, w, sink, stamps); measure<4>(num_cu, w, sink, stamps); measure<2>(num_cu, w, sink, stamps); measure<1>(num_cu, w, sink, stamps); }
    return 0;
}
