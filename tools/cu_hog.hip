// cu_hog.hip -- a second PROCESS that holds most of the device's wave slots for a few seconds (test tool, not product).
//
//   build/cu_hog [--leave N] [--seconds S]
//
// Launches (wave slots of the device - N) single-wave workgroups that spin on the wall clock for S seconds, prints
// "resident <n> of <m>" once they are on the device (or after a second), and exits when they are done.  What
// tests/test_gpu_wide.py::test_wide_launch_survives_a_cu_hog_from_another_process runs next to a long NW call: the many-wave
// kernel's launch then fits only in part, notices (wide_kernels.hip: wide_all_resident), and the call still has to return
// the right distance -- the reference always returns (edlib.cpp:197-217).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

__global__ void __launch_bounds__(64) hog_kernel(unsigned* arrived, long long ticks)
{
    if (threadIdx.x == 0) __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = (long long)wall_clock64();                 // 100 MHz
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

int main(int argc, char** argv)
{
    int leave = 12; double seconds = 4.0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--leave")) leave = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--seconds")) seconds = atof(argv[i + 1]);
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "cu_hog: no device\n"); return 1; }
    const int slots = p.multiProcessorCount * (p.maxThreadsPerMultiProcessor / 64);
    const int wgs = slots - leave > 1 ? slots - leave : 1;
    unsigned* arrived = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&arrived), sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return 1;
    *arrived = 0;
    hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(64), 0, 0, arrived, (long long)(seconds * 1e8));
    if (hipGetLastError() != hipSuccess) { fprintf(stderr, "cu_hog: launch failed\n"); return 1; }
    const auto t0 = std::chrono::steady_clock::now();
    while (*(volatile unsigned*)arrived < (unsigned)wgs &&
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0)
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    printf("resident %u of %d\n", *(volatile unsigned*)arrived, wgs);
    fflush(stdout);
    (void)hipDeviceSynchronize();
    return 0;
}
