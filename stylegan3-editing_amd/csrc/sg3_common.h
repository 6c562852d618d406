// sg3_common.h -- shared host/device helpers for libsg3hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/sg3_ops.h"

namespace sg3 {

// thread-local last-error text behind sg3_last_error()
void set_error(const char* fmt, ...);

#define SG3_REQUIRE(cond, ...)                                   \
    do { if (!(cond)) { sg3::set_error(__VA_ARGS__); return SG3_BAD_ARG; } } while (0)

#define SG3_HIP_CHECK(expr)                                                        \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                           \
        sg3::set_error("%s failed: %s", #expr, hipGetErrorString(e_));             \
        return SG3_HIP_ERROR; } } while (0)

// after a kernel launch
#define SG3_LAUNCH_CHECK(name)                                                     \
    do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) {                \
        sg3::set_error("launch of %s failed: %s", name, hipGetErrorString(e_));    \
        return SG3_HIP_ERROR; } } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// floor division / modulus for possibly negative numerators (device + host)
__host__ __device__ static inline int floor_div(int a, int b) {
    int q = a / b; int r = a - q * b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}
__host__ __device__ static inline int ceil_div_s(int a, int b) { return -floor_div(-a, b); }

// storage type <-> fp32 compute type
template <typename T> struct io;
template <> struct io<float> {
    typedef float acc_t;
    __device__ static inline float ld(const float* p) { return *p; }
    __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct io<_Float16> {
    typedef float acc_t;
    __device__ static inline float ld(const _Float16* p) { return (float)*p; }
    __device__ static inline void st(_Float16* p, float v) { *p = (_Float16)v; }
};
template <> struct io<double> {
    typedef double acc_t;
    __device__ static inline double ld(const double* p) { return *p; }
    __device__ static inline void st(double* p, double v) { *p = v; }
};

} // namespace sg3
