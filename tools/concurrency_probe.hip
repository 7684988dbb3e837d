// How many kernels of different streams run at the same time?  K streams, each with a chain of small spinning kernels
// (one block, a fixed number of device-clock ticks): the wall time of all chains over the time of one says how many ran
// side by side.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/concurrency_probe tools/concurrency_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (ticks < 0) *sink = 1;
}

int main() {
    int* sink;
    if (hipMalloc(&sink, 4) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int chain = 40;
    const long long ticks = 5000;     // 100 MHz clock: 50 us
    for (int blocks : {1, 256}) {
        double one = 0;
        for (int K : {1, 2, 4, 6, 8, 12, 16}) {
            std::vector<hipStream_t> st(K);
            for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (auto& s : st) hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s, 100, sink);
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < chain; i++)
                for (auto& s : st) hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s, ticks, sink);
            hipDeviceSynchronize();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (K == 1) one = ms;
            printf("blocks %3d  streams %2d: %7.3f ms for %d kernels per stream  (%.2f x one stream's time; %.1f us per kernel of a stream)\n",
                   blocks, K, ms, chain, ms / one, ms / chain * 1e3);
            for (auto& s : st) hipStreamDestroy(s);
        }
    }
    return 0;
}
