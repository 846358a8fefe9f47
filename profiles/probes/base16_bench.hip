// standalone: cycles of one base16() call (the 16-pivot chain) -- hipcc --offload-arch=gfx950 base16_bench.hip -o base16_bench
#include "../../gpyrn_amd/csrc/diag_tile.h"
#include <stdio.h>

template <int OLD>
__global__ void k_bench(double* A, double* Xg, long long* out, int* info, int reps)
{
    __shared__ double St[16 * PP], xd[16 * PP], line[64];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) St[(i / 16) * PP + (i % 16)] = A[i];
    __syncthreads();
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        for (int i = l; i < 256; i += 64) St[(i / 16) * PP + (i % 16)] = A[i];
        wave_lds_sync();
        if (OLD == 0) base16(St, xd, (gptr_t)Xg, 16, info, 0, 0, line);
        else wave_lds_sync();
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
    double ref[256];
    hipLaunchKernelGGL(k_bench<2>, dim3(1), dim3(64), 0, 0, dA, dX, dout, dinfo, 200);
    hipDeviceSynchronize();
    long long c_empty; hipMemcpy(&c_empty, dout, 8, hipMemcpyDeviceToHost);
    printf("empty loop (reload of the block): %lld ticks\n", c_empty);
    for (int old = 1; old >= 0; --old) {
        hipMemset(dX, 0xff, sizeof(hA));
        for (int rep = 0; rep < 2; ++rep) {
            if (old) hipLaunchKernelGGL(k_bench<1>, dim3(1), dim3(64), 0, 0, dA, dX, dout, dinfo, 200);
            else hipLaunchKernelGGL(k_bench<0>, dim3(1), dim3(64), 0, 0, dA, dX, dout, dinfo, 200);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, dout, 8, hipMemcpyDeviceToHost);
        double hX[256]; hipMemcpy(hX, dX, sizeof(hX), hipMemcpyDeviceToHost);
        double dmax = 0;
        for (int i = 0; i < 256; ++i) { if (old) ref[i] = hX[i]; else dmax = fmax(dmax, fabs(hX[i] - ref[i])); }
        printf("%s: %lld clock64 ticks per call (%.0f per pivot); X[0][0]=%g X[15][15]=%g X[15][0]=%g max|X - X_lanes|=%g\n",
               old ? "base16_lanes" : "base16 (mfma)", c, c / 16.0, hX[0], hX[255], hX[240], dmax);
    }
#ifdef BASE16_STAMPS
    long long st[4][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(b16_stamps), sizeof(st));
    for (int R = 0; R < 4; ++R) {
        printf("round %d: ", R);
        for (int i = 1; i < 8; ++i) printf("%s %lld  ", (const char*[]){"", "readlanes", "L", "X", "lds", "mfma1", "mfma2", "stores"}[i], st[R][i] - st[R][i - 1]);
        if (R < 3) printf("| round total %lld", st[R + 1][0] - st[R][0]);
        printf("\n");
    }
#endif
    return 0;
}
