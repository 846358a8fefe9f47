// Do stream memory operations work here, and how quickly does a stream blocked in
// hipStreamWaitValue32 start its next kernel after another kernel sets the flag from the device?
// build: hipcc -O2 --offload-arch=gfx950 streamvalue.hip -o streamvalue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_producer(uint32_t* flag, uint32_t value, long long* stamp, long long spin)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    stamp[0] = wall_clock64();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // keep running for a while, like a fused kernel that continues after publishing
    t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    stamp[2] = wall_clock64();
}
__global__ void k_consumer(long long* stamp) { stamp[1] = wall_clock64(); }

int main()
{
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    long long* stamp; CK(hipMalloc(&stamp, 64));
    for (int kind = 0; kind < 2; ++kind) {
        uint32_t* flag = nullptr;
        if (kind == 0) { if (hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) != hipSuccess) { printf("signal memory: not available\n"); continue; } }
        else CK(hipMalloc(&flag, 8));
        CK(hipMemset(flag, 0, 8));
        for (int rep = 1; rep <= 3; ++rep) {
            CK(hipMemset(stamp, 0, 64));
            hipError_t e = hipStreamWaitValue32(s1, flag, rep, hipStreamWaitValueGte, 0xffffffffu);
            if (e != hipSuccess) { printf("kind %d: hipStreamWaitValue32 -> %s\n", kind, hipGetErrorString(e)); break; }
            hipLaunchKernelGGL(k_consumer, dim3(1), dim3(64), 0, s1, stamp);
            hipLaunchKernelGGL(k_producer, dim3(1), dim3(64), 0, s0, flag, (uint32_t)rep, stamp, 2000LL /* 20 us at 100 MHz */);
            CK(hipStreamSynchronize(s0));
            CK(hipStreamSynchronize(s1));
            long long h[3]; CK(hipMemcpy(h, stamp, 24, hipMemcpyDeviceToHost));
            printf("kind %d (%s) rep %d: consumer started %.2f us after the flag store (producer ran on for %.2f us)\n",
                   kind, kind == 0 ? "signal memory" : "hipMalloc", rep, (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0);
        }
        // the other direction: a stream writes a value after its kernel, a running kernel could poll it
        hipError_t e = hipStreamWriteValue32(s1, flag, 77, 0);
        printf("kind %d: hipStreamWriteValue32 -> %s\n", kind, hipGetErrorString(e));
        CK(hipStreamSynchronize(s1));
        uint32_t v = 0; CK(hipMemcpy(&v, flag, 4, hipMemcpyDeviceToHost));
        printf("kind %d: flag after write = %u\n", kind, v);
    }
    return 0;
}
