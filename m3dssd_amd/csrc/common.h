// Internal helpers shared by the HIP translation units of libm3dssd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "m3dssd_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define M3D_LEAKY_SLOPE 0.01f

void m3d_set_error(const char *fmt, ...);

#define M3D_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            m3d_set_error(__VA_ARGS__); \
            return M3D_E_ARG;         \
        }                             \
    } while (0)

#define M3D_HIP(call)                                                             \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            m3d_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                          __FILE__, __LINE__);                                    \
            return M3D_E_HIP;                                                     \
        }                                                                         \
    } while (0)

#define M3D_LAUNCH_CHECK()                                                          \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            m3d_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                          __FILE__, __LINE__);                                      \
            return M3D_E_HIP;                                                       \
        }                                                                           \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int imin(int a, int b) { return a < b ? a : b; }

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : v * M3D_LEAKY_SLOPE; }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
