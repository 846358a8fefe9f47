// standalone: 100 MHz timeline of k_tile_rows (the chain's two products), last workgroup, wave 0
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 rows_bench.hip -o rows_bench
#define ROWS_STAMPS
#include "../../gpyrn_amd/csrc/gemm_tile.hip"
#include <stdio.h>
#include <vector>
void prof_begin(gprn_ctx*, int, hipStream_t) {}
void prof_end(gprn_ctx*) {}
bool tab_rows(gprn_ctx*, double**, int, PtrArgs*) { return false; }

int main()
{
    const int ld = 4096, nb = 2;
    double *B[2], *X[2];
    std::vector<double> h((size_t)ld * 384, 0.001);
    double* hp[2 * GPRN_NBUF] = {0};
    for (int b = 0; b < nb; ++b) {
        hipMalloc(&B[b], h.size() * 8); hipMalloc(&X[b], h.size() * 8);
        hipMemcpy(B[b], h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(X[b], h.data(), h.size() * 8, hipMemcpyHostToDevice);
        hp[b * GPRN_NBUF + BUF_B] = B[b]; hp[b * GPRN_NBUF + BUF_X] = X[b];
    }
    double** d_p; hipMalloc(&d_p, sizeof(hp)); hipMemcpy(d_p, hp, sizeof(hp), hipMemcpyHostToDevice);
    unsigned* sig; hipMalloc(&sig, 64); hipMemset(sig, 0, 64);
    PtrArgs pa{};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int with_sig = 0; with_sig < 2; ++with_sig)
            for (int rep = 0; rep < 3; ++rep) {
                const int64_t a_off = (int64_t)128 * ld, b_off = mode == 0 ? 0 : a_off, c_off = mode == 0 ? a_off : a_off + 128;
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL((k_chain_l<false>), dim3(8, nb), dim3(512), 0, 0, (double* const*)d_p, pa, ld, a_off, b_off,
                                                  with_sig ? sig : nullptr, 1u + rep, nullptr, 0u, sig + 8, nullptr, 0u);
                else hipLaunchKernelGGL((k_chain_u<false>), dim3(36, nb), dim3(64), 0, 0, (double* const*)d_p, pa, ld, a_off, c_off,
                                        with_sig ? sig : nullptr, 1u + rep, nullptr, 0u, sig + 8, nullptr, 0u);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned long long st[8]; hipMemcpyFromSymbol(st, HIP_SYMBOL(rows_stamps), sizeof(st));
                if (rep == 2)
                    printf("mode %d signal %d: events %.1f us | in-kernel (us): wait %.2f  loads %.2f  barrier %.2f  mfma+store issue %.2f  stores done %.2f  signal %.2f  = %.2f\n",
                           mode, with_sig, ms * 1e3, (st[1] - st[0]) * 0.01, (st[2] - st[1]) * 0.01, (st[3] - st[2]) * 0.01, (st[4] - st[3]) * 0.01,
                           (st[5] - st[4]) * 0.01, (st[6] - st[5]) * 0.01, (st[6] - st[0]) * 0.01);
            }
    return 0;
}
