// Is PCIe duplex reachable from HIP?  512 MiB up + 512 MiB down: back to back on one stream, concurrently on two streams,
// in chunks, linear vs pitched (hipMemcpy2DAsync) uploads, pinned (hipHostMalloc) vs registered (hipHostRegister) memory.
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/duplex_probe.hip -o tools/probes/duplex_probe.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)512 << 20;
    for (int mode = 0; mode < 2; ++mode) {
        char *h_in, *h_out;
        if (mode == 0) { CK(hipHostMalloc((void**)&h_in, N, 0)); CK(hipHostMalloc((void**)&h_out, N, 0)); }
        else {
            h_in = (char*)aligned_alloc(4096, N); h_out = (char*)aligned_alloc(4096, N);
            memset(h_in, 1, N); memset(h_out, 2, N);
            CK(hipHostRegister(h_in, N, hipHostRegisterDefault)); CK(hipHostRegister(h_out, N, hipHostRegisterDefault));
        }
        char *d_a, *d_b;
        CK(hipMalloc((void**)&d_a, N + (N >> 6))); CK(hipMalloc((void**)&d_b, N));
        hipStream_t s1, s2;
        CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        printf("%s host memory\n", mode == 0 ? "hipHostMalloc" : "hipHostRegister'ed");
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            CK(hipMemcpyAsync(d_a, h_in, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
            double t1 = now();
            CK(hipMemcpyAsync(h_out, d_b, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1));
            double t2 = now();
            printf("  sequential: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)  total %.2f\n", t1 - t0, N / (t1 - t0) / 1e6, t2 - t1, N / (t2 - t1) / 1e6, t2 - t0);
            t0 = now();
            CK(hipMemcpyAsync(d_a, h_in, N, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(h_out, d_b, N, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            printf("  concurrent whole copies on two streams: %.2f ms\n", now() - t0);
            const int nch = 16; const size_t C = N / nch;
            t0 = now();
            for (int k = 0; k < nch; ++k) {
                CK(hipMemcpyAsync(d_a + k * C, h_in + k * C, C, hipMemcpyHostToDevice, s1));
                CK(hipMemcpyAsync(h_out + k * C, d_b + k * C, C, hipMemcpyDeviceToHost, s2));
            }
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            printf("  concurrent, 16 chunks each: %.2f ms\n", now() - t0);
            // pitched upload as vt_volume_create does it: rows of 2048 B into a pitch of 2064 B
            const size_t W = 2048, P = 2064, rows = N / W;
            t0 = now();
            CK(hipMemcpy2DAsync(d_a, P, h_in, W, W, rows, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
            printf("  pitched H2D (hipMemcpy2DAsync, 2048 -> pitch 2064): %.2f ms\n", now() - t0);
            t0 = now();
            CK(hipMemcpy2DAsync(d_a, P, h_in, W, W, rows, hipMemcpyHostToDevice, s1));
            CK(hipMemcpyAsync(h_out, d_b, N, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            printf("  pitched H2D concurrent with linear D2H: %.2f ms\n", now() - t0);
            // progressive enqueue as a pipeline would do it: uploads two chunks ahead, each download enqueued only after the
            // matching upload has completed (host waits on an event)
            {
                hipEvent_t ev[16];
                for (int k = 0; k < nch; ++k) CK(hipEventCreate(&ev[k]));
                const size_t rows_c = rows / nch;
                for (int variant = 0; variant < 2; ++variant) {
                    t0 = now();
                    auto up = [&](int k) {
                        if (k >= nch) return;
                        if (variant == 0) CK(hipMemcpyAsync(d_a + k * C, h_in + k * C, C, hipMemcpyHostToDevice, s1));
                        else CK(hipMemcpy2DAsync(d_a + (size_t)k * rows_c * P, P, h_in + k * C, W, W, rows_c, hipMemcpyHostToDevice, s1));
                        CK(hipEventRecord(ev[k], s1));
                    };
                    up(0); up(1);
                    for (int k = 0; k < nch; ++k) {
                        CK(hipEventSynchronize(ev[k]));
                        CK(hipMemcpyAsync(h_out + k * C, d_b + k * C, C, hipMemcpyDeviceToHost, s2));
                        up(k + 2);
                    }
                    const double t_issue = now() - t0;
                    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
                    printf("  progressive pipeline (%s uploads): issue %.2f ms, total %.2f ms\n", variant ? "pitched" : "linear", t_issue, now() - t0);
                }
                for (int k = 0; k < nch; ++k) CK(hipEventDestroy(ev[k]));
            }
        }
        CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2)); CK(hipFree(d_a)); CK(hipFree(d_b));
        if (mode == 0) { CK(hipHostFree(h_in)); CK(hipHostFree(h_out)); }
        else { CK(hipHostUnregister(h_in)); CK(hipHostUnregister(h_out)); free(h_in); free(h_out); }
    }
    return 0;
}
