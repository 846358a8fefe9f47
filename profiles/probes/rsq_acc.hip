#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(double* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = 0.5 + 3.5 * (i + 0.37) / n;            // [0.5, 4)
    double y0 = __builtin_amdgcn_rsq(x);
    double hx = -0.5 * x;
    double y1 = y0 * fma(hx * y0, y0, 1.5);
    double y2 = y1 * fma(hx * y1, y1, 1.5);
    out[3 * i] = y0; out[3 * i + 1] = y1; out[3 * i + 2] = y2;
}
int main()
{
    const int n = 1 << 20;
    double* d; hipMalloc(&d, 3 * n * sizeof(double));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
    double* h = (double*)malloc(3 * n * sizeof(double));
    hipMemcpy(h, d, 3 * n * sizeof(double), hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        double x = 0.5 + 3.5 * (i + 0.37) / n;
        long double t = 1.0L / sqrtl((long double)x);
        e0 = fmax(e0, fabs((double)((h[3 * i] - t) / t)));
        e1 = fmax(e1, fabs((double)((h[3 * i + 1] - t) / t)));
        e2 = fmax(e2, fabs((double)((h[3 * i + 2] - t) / t)));
    }
    printf("v_rsq_f64 max rel err: seed %.3e, 1 Newton %.3e, 2 Newton %.3e\n", e0, e1, e2);
    return 0;
}
