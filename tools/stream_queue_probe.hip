// Which HIP streams of a process run side by side?  Creates N non-blocking streams, then for every pair launches a
// one-block kernel that spins for about a millisecond on each of the two and times the pair: ~1 ms = the two streams
// sit on different hardware queues, ~2 ms = they share one (GPU_MAX_HW_QUEUES per process) and run one after the other.
//   hipcc --offload-arch=gfx950 -O2 -o tools/stream_queue_probe tools/stream_queue_probe.hip && tools/stream_queue_probe [N]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && cycles < 0) *sink = 1;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    std::vector<hipStream_t> st(n + 1, nullptr);            // st[n]: the null stream
    for (int i = 0; i < n; i++)
        if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) { printf("stream %d: failed\n", i); return 1; }
    const long long ms1 = 100000;                           // wall_clock64 ticks at 100 MHz: 1 ms
    for (int i = 0; i <= n; i++) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], 1000, nullptr); }
    hipDeviceSynchronize();
    printf("time of a pair of 1 ms kernels, ms (row, column = stream in creation order; %d = the null stream)\n     ", n);
    for (int j = 0; j <= n; j++) printf("%5d", j);
    printf("\n");
    for (int i = 0; i <= n; i++) {
        printf("%5d", i);
        for (int j = 0; j <= n; j++) {
            if (j == i) { printf("    -"); continue; }
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[i], ms1, nullptr);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st[j], ms1, nullptr);
            hipStreamSynchronize(st[i]);
            hipStreamSynchronize(st[j]);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("%5.1f", ms);
        }
        printf("\n");
    }
    return 0;
}
