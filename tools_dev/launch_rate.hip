// developer microbenchmark (GPU box): how many kernel launches per second does the device complete, by number of streams and kernel length?
// The mixed-lidar secondary issues 3 x 20 short launches per mixed batch; this separates "the chip is full" from "the dispatch path is".
//   hipcc --offload-arch=gfx950 -O2 tools_dev/launch_rate.hip -o /tmp/launch_rate && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin_kernel(long long cycles, int *sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && threadIdx.x == 12345) *sink = 1;
}
int main() {
    const int N = 2000;
    int *sink;
    hipMalloc(&sink, 4);
    printf("| streams | kernel length us | workgroups x threads | launches/s | us per launch (all streams) | us per launch and stream |\n|---|---|---|---|---|---|\n");
    for (int us : {0, 5, 20, 50}) for (int wg : {1, 85}) for (int S : {1, 3, 6, 9}) {
        std::vector<hipStream_t> st(S);
        for (auto &s : st) hipStreamCreate(&s);
        const long long cyc = (long long)us * 100;   // wall_clock64 ticks at 100 MHz
        for (int i = 0; i < 64 * S; i++) spin_kernel<<<wg, 256, 0, st[i % S]>>>(cyc, sink);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N * S; i++) spin_kernel<<<wg, 256, 0, st[i % S]>>>(cyc, sink);
        hipDeviceSynchronize();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("| %d | %d | %d x 256 | %.0f | %.2f | %.2f |\n", S, us, wg, N * S / dt, dt / (N * S) * 1e6, dt / N * 1e6);
        for (auto &s : st) hipStreamDestroy(s);
    }
    return 0;
}
