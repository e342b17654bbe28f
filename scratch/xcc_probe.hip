// xcc_probe.hip -- which XCD / CU does workgroup b of a launch land on, with and without the CU mask the library's
// main stream uses (bits [reserve, ncu) set)?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
__global__ void probe(unsigned *out, int spin)
{
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    double x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = fma(x, 1.0000001, 1e-9);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid + (x == 1234.5 ? 1 : 0);
    }
}
int main(int argc, char **argv)
{
    const int reserve = argc > 1 ? atoi(argv[1]) : 32;
    const int G = argc > 2 ? atoi(argv[2]) : 2048;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    unsigned *d; hipMalloc(&d, G * 8);
    for (int pass = 0; pass < 2; pass++) {
        hipStream_t st;
        if (pass == 0) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        else {
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
            for (int i = reserve; i < ncu; i++) mask[i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&st, mask.size(), mask.data()) != hipSuccess) { printf("mask failed\n"); return 1; }
        }
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, st, d, 20000);
        hipStreamSynchronize(st);
        std::vector<unsigned> h(2 * G); hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        std::map<unsigned, int> perx; std::map<unsigned, std::map<unsigned, int>> cus;
        for (int b = 0; b < G; b++) { perx[h[2 * b] & 0xf]++; cus[h[2 * b] & 0xf][(h[2 * b + 1] >> 8) & 0xfff]++; }   // HW_ID: cu_id bits 11:8, sh 12, se 15:13
        printf("%s stream (%d CUs reported): workgroups per XCC:", pass ? "masked" : "plain", ncu);
        for (auto &kv : perx) printf("  x%u:%d(%zu cu)", kv.first, kv.second, cus[kv.first].size());
        printf("\n first 24 blocks -> xcc:");
        for (int b = 0; b < 24; b++) printf(" %u", h[2 * b] & 0xf);
        printf("\n");
    }
    return 0;
}
