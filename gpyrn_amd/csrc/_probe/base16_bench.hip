// standalone: cycles of one base16() call (the 16-pivot chain) -- hipcc --offload-arch=gfx950 base16_bench.hip -o base16_bench
#include "../factor.hip"
#include <stdio.h>
void prof_begin(gprn_ctx*, int, hipStream_t) {}
void prof_end(gprn_ctx*) {}
int launch_tiles(gprn_ctx*, const TileTask*, size_t, double**, int, int, int, hipStream_t, int, Signal, Await) { return 0; }

__global__ void k_bench(double* A, double* Xg, long long* out, int* info, int reps)
{
    __shared__ double St[16 * PP], xd[16 * PP], line[64];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) St[(i / 16) * PP + (i % 16)] = A[i];
    __syncthreads();
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        for (int i = l; i < 256; i += 64) St[(i / 16) * PP + (i % 16)] = A[i];
        __builtin_amdgcn_wave_barrier();
        base16(St, xd, (gptr_t)Xg, 16, info, 0, 0, line);
    }
    long long t1 = clock64();
    if (l == 0) out[0] = (t1 - t0) / reps;
}

int main()
{
    double hA[256];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) hA[i * 16 + j] = (i == j) ? 20.0 : 1.0 / (1 + abs(i - j));
    double *dA, *dX; long long* dout; int* dinfo;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dX, sizeof(hA)); hipMalloc(&dout, 8); hipMalloc(&dinfo, 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, dA, dX, dout, dinfo, 200);
        hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, dout, 8, hipMemcpyDeviceToHost);
    double hX[256]; hipMemcpy(hX, dX, sizeof(hX), hipMemcpyDeviceToHost);
    printf("base16: %lld cycles per call (%.0f per pivot); X[0][0]=%g X[15][15]=%g\n", c, c / 16.0, hX[0], hX[255]);
    return 0;
}
