// Dev tool: shader clock seen by s_memtime (clock64) against the 100 MHz constant clock (wall_clock64)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long* out, int iters) {
    long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f;
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main() {
    long long* d; hipMalloc(&d, 64);
    for (int rep = 0; rep < 3; rep++)
    for (int blocks : {1, 256, 1024}) for (int iters : {100000, 2000000}) {
        spin<<<blocks, 256>>>(d, iters);
        long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("blocks %4d iters %8d: clock64 %lld wall(100MHz) %lld -> %.1f MHz, %.2f clk/iter\n", blocks, iters, h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / iters);
    }
    return 0;
}
