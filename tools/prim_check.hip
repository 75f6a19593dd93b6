// prim_check.hip — does the device round these primitives exactly like the host (IEEE RN)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(int n, const double* a, const double* b, const float* fa, const float* fb, double* o, float* fo) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    o[0*n+i] = sqrt(a[i]); o[1*n+i] = __dsqrt_rn(a[i]); o[2*n+i] = a[i] / b[i]; o[3*n+i] = log(a[i]);
    o[4*n+i] = fma(a[i], b[i], a[i]); o[5*n+i] = __drcp_rn(a[i]); o[6*n+i] = lgamma(a[i]*50.0); o[7*n+i] = exp(-a[i]);
    fo[0*n+i] = sqrtf(fa[i]); fo[1*n+i] = __fsqrt_rn(fa[i]); fo[2*n+i] = fa[i] / fb[i]; fo[3*n+i] = __fdiv_rn(fa[i], fb[i]);
}
int main() {
    const int n = 1 << 20; std::vector<double> a(n), b(n), o(8*n); std::vector<float> fa(n), fb(n), fo(4*n);
    srand(1); for (int i = 0; i < n; i++) { a[i] = exp((rand()/(double)RAND_MAX)*40-20); b[i] = exp((rand()/(double)RAND_MAX)*40-20); fa[i] = (float)a[i]; fb[i] = (float)b[i]; }
    double *da, *db, *d_o; float *dfa, *dfb, *dfo;
    hipMalloc(&da, n*8); hipMalloc(&db, n*8); hipMalloc(&d_o, 8*n*8); hipMalloc(&dfa, n*4); hipMalloc(&dfb, n*4); hipMalloc(&dfo, 4*n*4);
    hipMemcpy(da, a.data(), n*8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n*8, hipMemcpyHostToDevice);
    hipMemcpy(dfa, fa.data(), n*4, hipMemcpyHostToDevice); hipMemcpy(dfb, fb.data(), n*4, hipMemcpyHostToDevice);
    k<<<n/256, 256>>>(n, da, db, dfa, dfb, d_o, dfo); hipDeviceSynchronize();
    hipMemcpy(o.data(), d_o, 8*n*8, hipMemcpyDeviceToHost); hipMemcpy(fo.data(), dfo, 4*n*4, hipMemcpyDeviceToHost);
    long bad[12] = {0};
    for (int i = 0; i < n; i++) {
        bad[0] += o[0*n+i] != sqrt(a[i]); bad[1] += o[1*n+i] != sqrt(a[i]); bad[2] += o[2*n+i] != a[i]/b[i]; bad[3] += o[3*n+i] != log(a[i]);
        bad[4] += o[4*n+i] != fma(a[i], b[i], a[i]); bad[5] += o[5*n+i] != 1.0/a[i]; bad[6] += o[6*n+i] != lgamma(a[i]*50.0); bad[7] += o[7*n+i] != exp(-a[i]);
        bad[8] += fo[0*n+i] != sqrtf(fa[i]); bad[9] += fo[1*n+i] != sqrtf(fa[i]); bad[10] += fo[2*n+i] != fa[i]/fb[i]; bad[11] += fo[3*n+i] != fa[i]/fb[i];
    }
    const char* nm[12] = {"sqrt f64", "__dsqrt_rn", "div f64", "log f64", "fma f64", "__drcp_rn", "lgamma f64", "exp f64", "sqrtf", "__fsqrt_rn", "div f32", "__fdiv_rn"};
    for (int j = 0; j < 12; j++) printf("%-12s mismatches vs host: %ld / %d\n", nm[j], bad[j], n);
    return 0;
}
