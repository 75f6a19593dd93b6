// What one cross-stream dependency costs on this runtime: two streams hand a token back and forth through events (kernel on s1, record,
// s2 waits, kernel on s2, record, s1 waits ...), against the same kernels on ONE stream.  Decides whether a two-stream look-ahead in the
// band reduction (csrc/sb2.hip) can pay: it needs two such hand-offs per panel.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_event_pingpong.hip -o tools/probe_event_pingpong.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin_kernel(long long cycles, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 1000) *sink = 1;
}
int main()
{
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int N = 500;
    hipEvent_t ev[2 * N];
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (long long us : {0LL, 20LL, 100LL}) {
        const long long cyc = us * 100;   // wall_clock64: 100 MHz
        for (int mode = 0; mode < 3; mode++) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) {
                if (mode == 0) {            // one stream
                    spin_kernel<<<64, 256, 0, s1>>>(cyc, nullptr);
                    spin_kernel<<<1, 64, 0, s1>>>(cyc, nullptr);
                } else if (mode == 1) {     // ping-pong: big kernel on s1, small on s2, each waits for the other
                    spin_kernel<<<64, 256, 0, s1>>>(cyc, nullptr);
                    hipEventRecord(ev[2 * i], s1);
                    hipStreamWaitEvent(s2, ev[2 * i], 0);
                    spin_kernel<<<1, 64, 0, s2>>>(cyc, nullptr);
                    hipEventRecord(ev[2 * i + 1], s2);
                    hipStreamWaitEvent(s1, ev[2 * i + 1], 0);
                } else {                    // overlap: s2's small kernel depends on s1's PREVIOUS big kernel only (look-ahead shape)
                    hipEventRecord(ev[2 * i], s1);
                    spin_kernel<<<64, 256, 0, s1>>>(cyc, nullptr);
                    hipStreamWaitEvent(s2, ev[2 * i], 0);
                    spin_kernel<<<1, 64, 0, s2>>>(cyc, nullptr);
                    hipEventRecord(ev[2 * i + 1], s2);
                    hipStreamWaitEvent(s1, ev[2 * i + 1], 0);
                }
            }
            hipStreamSynchronize(s1); hipStreamSynchronize(s2);
            const double el = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("kernel %3lld us  mode %d (%s): %.1f us per iteration\n", us, mode,
                   mode == 0 ? "one stream, 2 kernels" : mode == 1 ? "two streams, strict ping-pong" : "two streams, small kernel beside the big one", el / N);
        }
    }
    return 0;
}
