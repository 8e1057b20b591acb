// Latency microbenchmark (round 6 study): cycles per DEPENDENT instruction for one wave alone on a SIMD (the stepping kernels' regime).
// hipcc --offload-arch=gfx950 -O3 -o lat lat.hip && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int K> __global__ void k(double* out, long long* cyc, double a0, double b0, float fa0, const unsigned short* tab) {
    __shared__ int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 7 + 1) & 1023;
    __syncthreads();
    double a = a0 + threadIdx.x, b = b0; float fa = fa0 + threadIdx.x, fb = (float)b0; int idx = threadIdx.x;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    long long t0 = clock64();
    if (K == 0) for (int i = 0; i < N; i++) { a = __builtin_fma(a, b, b); }
    if (K == 1) for (int i = 0; i < N; i++) { fa = __builtin_fmaf(fa, fb, fb); }
    if (K == 2) for (int i = 0; i < N; i++) { a = __builtin_fma(a, b, b); a2 = __builtin_fma(a2, b, b); }
    if (K == 3) for (int i = 0; i < N; i++) { a = __builtin_fma(a, b, b); a2 = __builtin_fma(a2, b, b); a3 = __builtin_fma(a3, b, b); a4 = __builtin_fma(a4, b, b); }
    if (K == 4) for (int i = 0; i < N; i++) { idx = lds[idx]; }
    if (K == 5) for (int i = 0; i < N; i++) { fa = (float)a; a = (double)fa * b; }          // cvt f64->f32, cvt f32->f64 (+mul)
    if (K == 6) for (int i = 0; i < N; i++) { a = a > b ? a * b : a + b; }                      // cmp + select chain
    if (K == 7) for (int i = 0; i < N; i++) { a = 1.0 / a + b; }                                // full fp64 division
    if (K == 8) for (int i = 0; i < N; i++) { a = __builtin_sqrt(a) + b; }
    if (K == 9) for (int i = 0; i < N; i++) { idx = tab[idx & 1023] + i; }                     // dependent global (L2/L1) read
    if (K == 10) for (int i = 0; i < N; i++) { a = __builtin_amdgcn_rsq(a) + b; }
    if (K == 11) for (int i = 0; i < N; i++) { a = a + b; }
    if (K == 12) for (int i = 0; i < N; i++) { a = a * b; }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * 64] = a + a2 + a3 + a4 + fa + idx;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[K] = t1 - t0;
}
int main() {
    double* out; long long* cyc; unsigned short* tab;
    hipMalloc(&out, 64 * 1024 * 8); hipMallocManaged(&cyc, 16 * 8); hipMalloc(&tab, 2048);
    unsigned short h[1024]; for (int i = 0; i < 1024; i++) h[i] = (i * 13 + 5) & 1023; hipMemcpy(tab, h, 2048, hipMemcpyHostToDevice);
    const char* name[] = {"fma_f64 dependent", "fma_f32 dependent", "fma_f64 x2 independent chains (per pair)", "fma_f64 x4 chains (per quad)", "ds_read_b32 dependent", "cvt f64->f32 + cvt f32->f64 + mul_f64", "cmp_f64 + 2 ops + select", "1.0/a + b (f64)", "sqrt(a) + b (f64)", "global u16 dependent (cached)", "rsq_f64 + add", "add_f64 dependent", "mul_f64 dependent"};
#define RUN(K) hipLaunchKernelGGL(k<K>, dim3(1024), dim3(64), 0, 0, out, cyc, 1.0000001, 0.9999999, 1.0f, tab); hipDeviceSynchronize(); printf("%-48s %.1f cycles/iteration\n", name[K], (double)cyc[K] / N);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
    return 0;
}
