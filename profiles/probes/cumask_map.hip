// Which physical CUs does bit i of a hipExtStreamCreateWithCUMask mask select?  (gfx950, 8 XCDs x 32 CUs)
// build: hipcc -O2 --offload-arch=gfx950 cumask_map.hip -o cumask_map ; run: ./cumask_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
__global__ void k_where(uint32_t* out)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // spin a little so that workgroups spread over every CU the mask allows
    long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main()
{
    const int nwg = 4096;
    uint32_t* d; hipMalloc(&d, nwg * 8);
    std::vector<uint32_t> h(nwg * 2);
    for (int test = 0; test < 6; ++test) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const char* what = "";
        switch (test) {
        case 0: mask[0] = 0xffffffffu; what = "bits 0..31"; break;
        case 1: mask[0] = 0xffu; what = "bits 0..7"; break;
        case 2: mask[0] = 0x01010101u; what = "bits 0,8,16,24"; break;
        case 3: mask[7] = 0xffffffffu; what = "bits 224..255"; break;
        case 4: for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu; mask[0] = 0; what = "all but bits 0..31"; break;
        case 5: mask[0] = 0x1u; what = "bit 0"; break;
        }
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("mask create failed\n"); return 1; }
        hipMemsetAsync(d, 0xff, nwg * 8, s);
        hipLaunchKernelGGL(k_where, dim3(nwg), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
        std::set<uint32_t> cus; int per_xcc[8] = {0};
        std::set<uint32_t> per[8];
        for (int i = 0; i < nwg; ++i) {
            uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
            per[xcc & 7].insert((se << 8) | (sh << 4) | cu);
        }
        printf("%-20s:", what);
        for (int x = 0; x < 8; ++x) printf(" xcc%d:%zu", x, per[x].size());
        printf("\n");
        if (test == 1 || test == 5 || test == 2) {
            for (int x = 0; x < 8; ++x) for (auto v : per[x]) printf("   xcc%d se%u sh%u cu%u\n", x, v >> 8, (v >> 4) & 1, v & 0xf);
        }
        hipStreamDestroy(s);
    }
    return 0;
}
