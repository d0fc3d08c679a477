// cu_hog.hip -- a second PROCESS that holds most of the device's wave slots for a few seconds (test tool, not product).
//
//   build/cu_hog [--leave N] [--seconds S]
//
// Launches (wave slots of the device - N) single-wave workgroups that spin on the wall clock for S seconds, prints
// "resident <n> of <m>" once they are on the device (or after a second), and exits when they are done.  What
// tests/test_gpu_wide.py::test_wide_launch_survives_a_cu_hog runs next to a long NW call: from another process (the device's
// scheduler then decides what shares the CUs) and, built as build/libcu_hog.so, from a second stream of the SAME process
// (cu_hog_start / cu_hog_wait: the two launches share the CUs, the many-wave kernel's launch fits only in part, notices --
// wide_kernels.hip: wide_all_resident -- and the call still has to return the right distance: the reference always
// returns, edlib.cpp:197-217).
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

static hipStream_t g_stream = nullptr;
static unsigned* g_arrived = nullptr;

// starts the hog on its own stream; returns the workgroups on the device after at most a second (*total = launched), -1 on error
extern "C" __attribute__((visibility("default"))) int cu_hog_start(int leave, double seconds, int* total)
{
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    const int slots = p.multiProcessorCount * (p.maxThreadsPerMultiProcessor / 64);
    const int wgs = slots - leave > 1 ? slots - leave : 1;
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!g_arrived && hipHostMalloc(reinterpret_cast<void**>(&g_arrived), sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return -1;
    *g_arrived = 0;
    hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(64), 0, g_stream, g_arrived, (long long)(seconds * 1e8));
    if (hipGetLastError() != hipSuccess) return -1;
    const auto t0 = std::chrono::steady_clock::now();
    while (*(volatile unsigned*)g_arrived < (unsigned)wgs &&
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0)
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (total) *total = wgs;
    return (int)*(volatile unsigned*)g_arrived;
}
extern "C" __attribute__((visibility("default"))) int cu_hog_wait(void)
{
    return g_stream && hipStreamSynchronize(g_stream) == hipSuccess ? 0 : 1;
}

#ifndef CU_HOG_LIBRARY
int main(int argc, char** argv)
{
    int leave = 12; double seconds = 4.0;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--leave")) leave = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--seconds")) seconds = atof(argv[i + 1]);
    }
    int total = 0;
    const int n = cu_hog_start(leave, seconds, &total);
    if (n < 0) { fprintf(stderr, "cu_hog: no device / launch failed\n"); return 1; }
    printf("resident %d of %d\n", n, total);
    fflush(stdout);
    return cu_hog_wait();
}
#endif
