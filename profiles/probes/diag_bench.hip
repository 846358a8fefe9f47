// standalone: shader-clock timeline of one diag_tile() call (potrf + inverse of a 128x128 tile), per wave and phase
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 diag_bench.hip -o diag_bench
#define DIAG_STAMPS
#include "../../gpyrn_amd/csrc/diag_tile.h"
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void k_bench(double* B, double* X, int ld, int* info, long long* total)
{
    __shared__ __attribute__((aligned(16))) double lds[DIAG_LDS_DOUBLES];
    const long long t0 = clock64();
    diag_tile(lds, (gptr_t)B, (gptr_t)X, ld, info, 0, 0);
    __syncthreads();
    if (threadIdx.x == 0) total[0] = clock64() - t0;
}

int main()
{
    const int n = 128;
    std::vector<double> A(n * n), A0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j) ? 3.0 : exp(-0.02 * (i - j) * (i - j));
    A0 = A;
    double *dB, *dX; int* dinfo; long long* dtot;
    hipMalloc(&dB, n * n * 8); hipMalloc(&dX, n * n * 8); hipMalloc(&dinfo, 4); hipMalloc(&dtot, 8);
    hipMemset(dinfo, 0, 4);
    long long tot = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dB, A0.data(), n * n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_bench, dim3(1), dim3(256), 0, 0, dB, dX, n, dinfo, dtot);
        hipDeviceSynchronize();
        hipMemcpy(&tot, dtot, 8, hipMemcpyDeviceToHost);
        printf("diag_tile: %lld ticks\n", tot);
    }
    long long st[4][NSB + 1][6];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(diag_stamps), sizeof(st));
    const long long z = st[0][NSB][0];
    printf("prologue (loads, publish): wave0 reaches barrier at +0; base16(0) on wave 3: %lld ticks; all past barrier %lld\n",
           st[3][NSB][2] - st[3][NSB][1], st[0][NSB][3] - z);
    printf("phase: per wave [work before M | wait M | work after M | wait E]\n");
    for (int kb = 0; kb < NSB; ++kb) {
        printf("kb %d (start +%6lld):", kb, st[0][kb][0] - z);
        for (int w = 0; w < 4; ++w)
            printf("  w%d [%5lld|%5lld|%5lld|%5lld]", w, st[w][kb][1] - st[w][kb][0], st[w][kb][2] - st[w][kb][1],
                   st[w][kb][3] - st[w][kb][2], st[w][kb][4] - st[w][kb][3]);
        printf("\n");
    }
    printf("end of last phase +%lld\n", st[0][NSB - 1][4] - z);
    // check
    std::vector<double> L(n * n), X(n * n);
    hipMemcpy(L.data(), dB, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, n * n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0, u = 0;
        for (int k = 0; k <= j; ++k) s += L[i * n + k] * L[j * n + k];
        for (int k = j; k <= i; ++k) u += L[i * n + k] * X[k * n + j];
        e1 = fmax(e1, fabs(s - A0[i * n + j])); e2 = fmax(e2, fabs(u - (i == j)));
    }
    printf("max |L L^T - A| = %.2e, max |L X - I| = %.2e\n", e1, e2);
    return 0;
}
