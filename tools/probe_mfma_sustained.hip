// probe_mfma_sustained.hip — what the matrix pipe SUSTAINS on gfx950 for the instruction the genotype rotation uses
// (v_mfma_f32_16x16x32_f16) and for the int8 instructions of a 3-plane integer formulation, with operands in registers
// and nothing else going on: an upper bound for any kernel built on them, at the clock the chip holds under that load for as long as
// one rotation launch lasts (30 - 40 ms) and for a short burst.  A 3-plane int8 rotation issues 3 passes against the 2 of the
// fp16 x 2 kernel, so it can only win if int8 sustains more than 1.5x the fp16 rate.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_sustained.hip -o /tmp/probe_mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void burn(long long iters, float *sink)
{
    const int l = threadIdx.x & 63;
    float out = 0.0f;
    if (MODE == 0) {            // v_mfma_f32_16x16x32_f16
        halfx8 a, b;
        for (int q = 0; q < 8; q++) { a[q] = (_Float16)(0.001f * (l + q)); b[q] = (_Float16)(0.002f * (l - q)); }
        floatx4 c[8];
        for (int t = 0; t < 8; t++) for (int e = 0; e < 4; e++) c[t][e] = 0.0f;
        for (long long it = 0; it < iters; it++)
#pragma unroll
            for (int t = 0; t < 8; t++) c[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[t], 0, 0, 0);
        for (int t = 0; t < 8; t++) out += c[t][0];
    } else if (MODE == 1) {     // v_mfma_i32_16x16x64_i8
        v4i a, b;
        for (int q = 0; q < 4; q++) { a[q] = 0x01020301 * (l + q + 1); b[q] = 0x02010102 * (l + 2 * q + 1); }
        v4i c[8];
        for (int t = 0; t < 8; t++) for (int e = 0; e < 4; e++) c[t][e] = 0;
        for (long long it = 0; it < iters; it++)
#pragma unroll
            for (int t = 0; t < 8; t++) c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[t], 0, 0, 0);
        for (int t = 0; t < 8; t++) out += (float)c[t][0];
    } else if (MODE == 2) {     // v_mfma_f32_32x32x16_f16
        halfx8 a, b;
        for (int q = 0; q < 8; q++) { a[q] = (_Float16)(0.001f * (l + q)); b[q] = (_Float16)(0.002f * (l - q)); }
        floatx16 c[4];
        for (int t = 0; t < 4; t++) for (int e = 0; e < 16; e++) c[t][e] = 0.0f;
        for (long long it = 0; it < iters; it++)
#pragma unroll
            for (int t = 0; t < 4; t++) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[t], 0, 0, 0);
        for (int t = 0; t < 4; t++) out += c[t][0];
    } else {                    // v_mfma_i32_32x32x32_i8
        v4i a, b;
        for (int q = 0; q < 4; q++) { a[q] = 0x01020301 * (l + q + 1); b[q] = 0x02010102 * (l + 2 * q + 1); }
        v16i c[4];
        for (int t = 0; t < 4; t++) for (int e = 0; e < 16; e++) c[t][e] = 0;
        for (long long it = 0; it < iters; it++)
#pragma unroll
            for (int t = 0; t < 4; t++) c[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[t], 0, 0, 0);
        for (int t = 0; t < 4; t++) out += (float)c[t][0];
    }
    if (out == 123456.789f) sink[0] = out;      // never true: keeps the loop
}

template <int MODE>
static double run(int blocks, long long iters, float *sink)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    burn<MODE><<<blocks, 256>>>(iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms;
}

template <int MODE>
static void measure(const char *name, double ops_per_mfma, int mfma_per_iter, double peak_per_clk_cu, int num_cu, float *sink, int wg_per_cu)
{
    const int blocks = num_cu * wg_per_cu;          // workgroups of 4 waves: 2 per CU = 2 waves per SIMD, 1 per CU = one wave per SIMD issuing alone
    run<MODE>(blocks, 20000, sink);                 // warm-up
    const double ms0 = run<MODE>(blocks, 200000, sink);
    for (double target : {5.0, 40.0}) {
        const long long iters = (long long)(200000.0 * target / ms0);
        for (int rep = 0; rep < 3; rep++) (void)run<MODE>(blocks, iters, sink);
        const double ms = run<MODE>(blocks, iters, sink);       // the figure: a launch right after three of the same length
        const double ops = (double)iters * mfma_per_iter * ops_per_mfma * 4.0 * blocks;
        const double rate = ops / (ms * 1e-3);
        printf("%-28s %6.1f ms launch: %8.1f T(FL)OP/s sustained = clock %.2f GHz at the pipe's %g ops/clk/CU\n", name, ms, rate / 1e12,
               rate / (peak_per_clk_cu * num_cu) / 1e9, peak_per_clk_cu);
    }
}

int main(int argc, char **argv)
{
    const int wg_per_cu = (argc > 1) ? atoi(argv[1]) : 2;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int num_cu = pr.multiProcessorCount;
    float *sink;
    CK(hipMalloc(&sink, 64));
    printf("%s, %d CUs, %d wave(s) per SIMD\n", pr.gcnArchName, num_cu, wg_per_cu);
    // dense peaks of the guide: fp16 2.5 PF, int8 5 PF at 2.4 GHz on 256 CUs -> 4069 / 8138 ops per clock and CU
    measure<0>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, 8, 2.5e15 / 2.4e9 / 256, num_cu, sink, wg_per_cu);
    measure<2>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, 4, 2.5e15 / 2.4e9 / 256, num_cu, sink, wg_per_cu);
    measure<1>("v_mfma_i32_16x16x64_i8", 2.0 * 16 * 16 * 64, 8, 5.0e15 / 2.4e9 / 256, num_cu, sink, wg_per_cu);
    measure<3>("v_mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32, 4, 5.0e15 / 2.4e9 / 256, num_cu, sink, wg_per_cu);
    return 0;
}
