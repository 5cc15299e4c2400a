// Which XCD does block b land on?  (speed-only assumption behind xcd_contiguous())
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* xcc, int lds_dummy) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xf);
    }
    if (lds_dummy == 12345) lds[threadIdx.x] = 1.f;
}
int main() {
    for (int n : {64, 2048, 6144, 32768}) {
        for (int ldsb : {1024, 28000, 50000}) {
            int* d; hipMalloc(&d, n * sizeof(int));
            hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(probe, dim3(n), dim3(256), ldsb, 0, d, 0);
            std::vector<int> h(n); hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
            int match = 0; for (int b = 0; b < n; ++b) match += (h[b] == (h[0] + b) % 8);
            int same8 = 0; for (int b = 8; b < n; ++b) same8 += (h[b] == h[b - 8]);
            printf("grid %6d lds %6d: first16 =", n, ldsb);
            for (int b = 0; b < 16; ++b) printf(" %d", h[b]);
            printf("  | xcc[b]==(xcc[0]+b)%%8: %.3f  xcc[b]==xcc[b-8]: %.3f\n", (double)match / n, (double)same8 / (n - 8));
            hipFree(d);
        }
    }
    return 0;
}
