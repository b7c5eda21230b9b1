// concurrency_probe.hip -- how many kernels does the device run at the same time?  N streams each launch one
// single-workgroup kernel that spins for ~1 ms; if k of them overlap, the wall time is ceil(N / k) ms.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/concurrency_probe.hip -o gpurun_out/concurrency_probe && GPU_MAX_HW_QUEUES=32 gpurun_out/concurrency_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks, int* sink)
{
    long long t0 = wall_clock64();
    int x = 0;
    while (wall_clock64() - t0 < ticks) x++;
    if (sink && x == -1) *sink = x;
}
int main()
{
    for (int n : {1, 2, 3, 4, 6, 8, 12, 16}) {
        std::vector<hipStream_t> st(n);
        for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000LL, nullptr);       // warm up
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 100000LL, nullptr);     // 1 ms at 100 MHz
        (void)hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%2d streams x 1 ms single-workgroup kernels: %.2f ms wall => ~%.1f concurrent\n", n, ms, n / ms);
        for (auto& s : st) (void)hipStreamDestroy(s);
    }
    return 0;
}
