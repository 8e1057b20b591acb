// ks_math.h -- small fixed-size vector / matrix helpers shared by the gfx950 kernels.
// Everything is KS_HD (host+device) and templated on the real type so that the very same
// source can be lane-checked on the CPU (tests/native) and instantiated in fp64 on the GPU for
// algorithm-exactness checks.  The product kernels are the fp32 instantiation.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KS_HD __host__ __device__ __forceinline__
#define KS_UNROLL _Pragma("unroll")
#define KS_FN __host__ __device__ __attribute__((noinline))
#define KS_LDS __attribute__((address_space(3)))
#else
#define KS_HD inline
#define KS_UNROLL
#define KS_FN inline
#define KS_LDS
#endif

// Where the convex-hull tables of the collision code live.  Standard build (one object hull of a few hundred vertices beside the hand's):
// staged in LDS by the stepping kernels, KS_TAB = KS_LDS.  Multi-geom build (KS_MULTI_GEOM: the welded-piece objects, 2 - 3.5 k hull
// vertices, 50 - 105 KB of tables): read from global memory (L2), KS_TAB = generic.
#ifdef KS_MULTI_GEOM
#define KS_TAB
#else
#define KS_TAB KS_LDS
#endif

#include <math.h>
#include <string.h>

namespace ks {

template <typename T> struct Lim;
template <> struct Lim<float> { static constexpr float minval = 1e-30f; static constexpr float big = 3.0e38f; };
template <> struct Lim<double> { static constexpr double minval = 1e-300; static constexpr double big = 1e300; };

// bit pattern of a float and back (integer words kept in float-typed scratch slots)
KS_HD int float_bits(float f) {
    int i;
#if defined(__HIP_DEVICE_COMPILE__)
    i = __float_as_int(f);
#else
    memcpy(&i, &f, 4);
#endif
    return i;
}
KS_HD float bits_float(int i) {
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __int_as_float(i);
#else
    memcpy(&f, &i, 4);
#endif
    return f;
}

KS_HD float ksqrt(float x) { return sqrtf(x); }
KS_HD double ksqrt(double x) { return sqrt(x); }
KS_HD float kmin(float a, float b) { return a < b ? a : b; }
KS_HD double kmin(double a, double b) { return a < b ? a : b; }
KS_HD int kmin(int a, int b) { return a < b ? a : b; }
KS_HD int kctz(unsigned x) { return __builtin_ctz(x); }
KS_HD int kpopc(unsigned x) { return __builtin_popcount(x); }
// reciprocal square root: the hardware v_rsq_f32 (1 ulp) in fp32 device code
KS_HD float krsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __frsqrt_rn(x);
#else
    return 1.0f / sqrtf(x);
#endif
}
KS_HD double krsqrt(double x) { return 1.0 / sqrt(x); }
KS_HD float kabs(float x) { return fabsf(x); }
KS_HD double kabs(double x) { return fabs(x); }
KS_HD float ksin(float x) { return sinf(x); }
KS_HD double ksin(double x) { return sin(x); }
KS_HD float kcos(float x) { return cosf(x); }
KS_HD double kcos(double x) { return cos(x); }
KS_HD float kacos(float x) { return acosf(x); }
KS_HD double kacos(double x) { return acos(x); }
KS_HD float katan2(float y, float x) { return atan2f(y, x); }
KS_HD double katan2(double y, double x) { return atan2(y, x); }
KS_HD float kpow(float x, float y) { return powf(x, y); }
KS_HD double kpow(double x, double y) { return pow(x, y); }

template <typename T> KS_HD T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T> KS_HD void cross3(T* r, const T* a, const T* b) {
    T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> KS_HD void sub3(T* r, const T* a, const T* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
template <typename T> KS_HD void add3(T* r, const T* a, const T* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
template <typename T> KS_HD void copy3(T* r, const T* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
template <typename T> KS_HD void scl3(T* r, const T* a, T s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
template <typename T> KS_HD void addscl3(T* r, const T* a, T s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
template <typename T> KS_HD T norm3(const T* a) { return ksqrt(dot3(a, a)); }
// same degenerate-input convention as the oracle's normalize3
template <typename T> KS_HD T normalize3(T* a) {
    T n = norm3(a);
    if (n < T(1e-15)) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
    T inv = T(1) / n;
    a[0] *= inv; a[1] *= inv; a[2] *= inv;
    return n;
}
// r = R v  (R row-major 3x3)
template <typename T> KS_HD void mulRv(T* r, const T* R, const T* v) {
    T x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    T y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    T z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
// r = R^T v
template <typename T> KS_HD void mulRtv(T* r, const T* R, const T* v) {
    T x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    T y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    T z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
template <typename T> KS_HD void mulRR(T* r, const T* A, const T* B) {
    T t[9];
    KS_UNROLL
    for (int i = 0; i < 3; i++) {
        KS_UNROLL
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    }
    KS_UNROLL
    for (int i = 0; i < 9; i++) r[i] = t[i];
}
template <typename T> KS_HD void quat2mat(T* R, const T* q) {
    T w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
template <typename T> KS_HD void quatnormalize(T* q) {
    T n = ksqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < T(1e-15)) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    T inv = T(1) / n;
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
template <typename T> KS_HD void quatmul(T* r, const T* a, const T* b) {
    T t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    T t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    T t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    T t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
template <typename T> KS_HD T clampT(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }

// In-place dense Cholesky A = L L^T of an NxN symmetric matrix kept as full row-major storage in
// registers (static indexing after unrolling); only the lower triangle is read / written.
template <typename T, int N> KS_HD void chol_inplace(T* A) {
    KS_UNROLL
    for (int j = 0; j < N; j++) {
        T d = A[j * N + j];
        KS_UNROLL
        for (int k = 0; k < j; k++) d -= A[j * N + k] * A[j * N + k];
        d = d > T(1e-15) ? d : T(1e-15);
        T l = ksqrt(d), inv = T(1) / l;
        A[j * N + j] = l;
        KS_UNROLL
        for (int i = j + 1; i < N; i++) {
            T v = A[i * N + j];
            KS_UNROLL
            for (int k = 0; k < j; k++) v -= A[i * N + k] * A[j * N + k];
            A[i * N + j] = v * inv;
        }
    }
}
// element i (runtime) of a register array: a select chain, never a dynamically indexed (scratch) array
template <typename T, int N> KS_HD T pick(const T (&v)[N], int i) {
    T out = 0;
    KS_UNROLL
    for (int j = 0; j < N; j++) out = (j == i) ? v[j] : out;
    return out;
}

// x = (L L^T)^-1 b, L from chol_inplace (lower triangle of A)
template <typename T, int N> KS_HD void chol_solve(const T* L, const T* b, T* x) {
    T y[N];
    KS_UNROLL
    for (int i = 0; i < N; i++) {
        T v = b[i];
        KS_UNROLL
        for (int k = 0; k < i; k++) v -= L[i * N + k] * y[k];
        y[i] = v / L[i * N + i];
    }
    KS_UNROLL
    for (int i = N - 1; i >= 0; i--) {
        T v = y[i];
        KS_UNROLL
        for (int k = i + 1; k < N; k++) v -= L[k * N + i] * x[k];
        x[i] = v / L[i * N + i];
    }
}

}  // namespace ks
