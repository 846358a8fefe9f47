// standalone: dependent-issue latency (shader clocks) of the instructions on base16's serial chain, one wave alone
//   hipcc --offload-arch=gfx950 -O3 lat_bench.hip -o lat_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

__global__ void k(double* out, long long* ticks, double seed)
{
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c0 = 1e-9;
    long long t[16];
    int n = 0;
    // v_fma_f64 dependent chain
    t[n++] = clock64();
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c0));) }
    t[n++] = clock64();
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    t[n++] = clock64();
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));) }
    t[n++] = clock64();
    a = seed;
    int lo = __double2loint(a), hi = __double2hiint(a);
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(hi));) }
    t[n++] = clock64();
    // independent fma stream (issue rate): 4 chains
    double e0 = a, e1 = a + 1, e2 = a + 2, e3 = a + 3;
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                                       : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(b), "v"(c0));) }
    t[n++] = clock64();
    // readlane -> valu use -> readlane ...
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0" : "+v"(lo) : : "s20");) }
    t[n++] = clock64();
    // dependent mfma chain (accumulator), and mfma -> valu read -> mfma
    v4d acc = {a, a, a, a};
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(c0));) }
    t[n++] = clock64();
    for (int i = 0; i < 16; ++i) { REP64(asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n s_nop 15\n s_nop 3\n v_mul_f64 %1, %1, %3" : "+v"(acc), "+v"(b) : "v"(c0), "v"(acc[0]));) }
    t[n++] = clock64();
    // LDS write -> read round trip
    __shared__ double sh[64];
    double v = a;
    for (int i = 0; i < 16; ++i) { REP64(sh[threadIdx.x] = v; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); v = sh[threadIdx.x ^ 1] + 1.0;) }
    t[n++] = clock64();
    out[threadIdx.x] = a + e0 + e1 + e2 + e3 + lo + acc[0] + acc[3] + b + v;
    if (threadIdx.x == 0) for (int i = 0; i + 1 < n; ++i) ticks[i] = t[i + 1] - t[i];
}

int main()
{
    double* d; long long* dt; hipMalloc(&d, 64 * 8); hipMalloc(&dt, 16 * 8);
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dt, 1.5); hipDeviceSynchronize(); }
    long long t[16]; hipMemcpy(t, dt, sizeof(t), hipMemcpyDeviceToHost);
    const char* name[] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rsq_f64 dependent", "v_cndmask_b32 dependent",
                          "v_fma_f64 x4 independent (per instruction)", "v_readlane + dependent v_add (pair)", "mfma f64 16x16x4 dependent (acc)",
                          "mfma -> 20 nops -> valu -> mfma (triple)", "LDS write -> sync -> read + add (round trip)"};
    const double per[] = {1024, 1024, 1024, 1024, 4096, 1024, 1024, 1024, 1024};
    for (int i = 0; i < 9; ++i) printf("%-48s %.1f clocks\n", name[i], t[i] / per[i]);
    return 0;
}
