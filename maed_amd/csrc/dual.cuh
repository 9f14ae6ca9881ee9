// Forward-mode dual numbers for the tiny closed-form geometry of the decoder tail and the loss (6D -> rotmat -> axis-angle,
// Rodrigues): a function written once as a template over its scalar type is evaluated on float for the value and on
// Dual<N> for the N partial derivatives.  Everything lives in registers; N <= 6.
#pragma once
#include <hip/hip_runtime.h>

template <int N>
struct Dual {
    float v;
    float d[N];
};

template <int N> __device__ __forceinline__ Dual<N> seed(float v, int k) {
    Dual<N> r; r.v = v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (i == k) ? 1.0f : 0.0f;
    return r;
}
template <typename S> __device__ __forceinline__ S constant(float v);
template <> __device__ __forceinline__ float constant<float>(float v) { return v; }
#define DUAL_CONSTANT(N) template <> __device__ __forceinline__ Dual<N> constant<Dual<N>>(float v) { return seed<N>(v, -1); }
DUAL_CONSTANT(3)
DUAL_CONSTANT(6)
#undef DUAL_CONSTANT

__device__ __forceinline__ float val(float a) { return a; }
template <int N> __device__ __forceinline__ float val(const Dual<N>& a) { return a.v; }

#define DUAL_LOOP for (int i = 0; i < N; ++i)
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; DUAL_LOOP r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; DUAL_LOOP r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; DUAL_LOOP r.d[i] = -a.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; DUAL_LOOP r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; const float inv = 1.0f / b.v; r.v = a.v * inv; DUAL_LOOP r.d[i] = (a.d[i] - r.v * b.d[i]) * inv; return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(float a, const Dual<N>& b) { Dual<N> r = b; r.v = a + b.v; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, float b) { Dual<N> r = a; r.v = a.v + b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, float b) { Dual<N> r = a; r.v = a.v - b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(float a, const Dual<N>& b) { Dual<N> r; r.v = a - b.v; DUAL_LOOP r.d[i] = -b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, float b) { Dual<N> r; r.v = a.v * b; DUAL_LOOP r.d[i] = a.d[i] * b; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(float a, const Dual<N>& b) { return b * a; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, float b) { return a * (1.0f / b); }

__device__ __forceinline__ float dsqrt(float a) { return sqrtf(a); }
template <int N> __device__ __forceinline__ Dual<N> dsqrt(const Dual<N>& a) { Dual<N> r; r.v = sqrtf(a.v); const float k = 0.5f / r.v; DUAL_LOOP r.d[i] = a.d[i] * k; return r; }
__device__ __forceinline__ float dsin(float a) { return sinf(a); }
template <int N> __device__ __forceinline__ Dual<N> dsin(const Dual<N>& a) { Dual<N> r; r.v = sinf(a.v); const float k = cosf(a.v); DUAL_LOOP r.d[i] = a.d[i] * k; return r; }
__device__ __forceinline__ float dcos(float a) { return cosf(a); }
template <int N> __device__ __forceinline__ Dual<N> dcos(const Dual<N>& a) { Dual<N> r; r.v = cosf(a.v); const float k = -sinf(a.v); DUAL_LOOP r.d[i] = a.d[i] * k; return r; }
__device__ __forceinline__ float datan2(float y, float x) { return atan2f(y, x); }
template <int N> __device__ __forceinline__ Dual<N> datan2(const Dual<N>& y, const Dual<N>& x) {
    Dual<N> r; r.v = atan2f(y.v, x.v); const float inv = 1.0f / (x.v * x.v + y.v * y.v);
    DUAL_LOOP r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * inv;
    return r;
}
// x.clamp_min(lo): the gradient is cut where the clamp is active (what ATen's clamp_min / F.normalize backward does)
__device__ __forceinline__ float clamp_min(float a, float lo) { return fmaxf(a, lo); }
template <int N> __device__ __forceinline__ Dual<N> clamp_min(const Dual<N>& a, float lo) { return a.v < lo ? seed<N>(lo, -1) : a; }
__device__ __forceinline__ float zero_if_nan(float a) { return a != a ? 0.0f : a; }
template <int N> __device__ __forceinline__ Dual<N> zero_if_nan(const Dual<N>& a) { return a.v != a.v ? seed<N>(0.0f, -1) : a; }
#undef DUAL_LOOP
