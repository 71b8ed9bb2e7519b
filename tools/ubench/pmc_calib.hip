// PMC calibration: what do FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count, per access width?
// MI355X_MICROARCH.md calibrates the gfx950 "x2" of FETCH_SIZE only for 16 B/lane streaming reads and calls every other width and
// WRITE_SIZE uncalibrated.  Every kernel below moves exactly `bytes` once through a buffer far larger than the 256 MB
// Infinity Cache (1 GiB), so nothing is served on-die; run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- pmc_calib.bin     (and again with WRITE_SIZE)
// and divide the counter by the bytes (tools/pmc_calib_report.py).  One kernel symbol per access pattern:
//   calib_load_b32 / _b64 / _b128       fully coalesced per-lane loads of 4 / 8 / 16 bytes (wave = 256 / 512 / 1024 contiguous bytes)
//   calib_load_b32_seg64                the F(4x4) patch pattern: 16 lanes x 4 bytes contiguous (64 B), the 4 lane groups 1 KB apart
//   calib_load_b128_line8               the wave-conv gather pattern: 8 lanes x 16 bytes = one 128-byte line, 8 lines per wave, scattered
//   calib_load_lds_b32 / _b128          buffer_load ... lds (LDS-DMA) of 4 / 16 bytes per lane
//   calib_store_b32 / _b128             coalesced stores of 4 / 16 bytes per lane
//   calib_store_b32_seg64               64-byte segments (16 lanes x 4 bytes), 1 KB apart
// build: hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o pmc_calib.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GRID 4096
#define TPB 256

template <typename T>
__device__ __forceinline__ void sink(float *out, T v);
template <> __device__ __forceinline__ void sink<float>(float *out, float v) { if (v == 123.456f) out[0] = v; }
template <> __device__ __forceinline__ void sink<f32x2>(float *out, f32x2 v) { if (v[0] + v[1] == 123.456f) out[0] = v[0]; }
template <> __device__ __forceinline__ void sink<f32x4>(float *out, f32x4 v) { if (v[0] + v[1] + v[2] + v[3] == 123.456f) out[0] = v[0]; }

template <typename T>
__device__ __forceinline__ void load_stream(const char *src, size_t bytes, float *out)
{
    const size_t n = bytes / sizeof(T), stride = (size_t)GRID * TPB;
    T acc = {};
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) acc += reinterpret_cast<const T *>(src)[i];
    sink<T>(out, acc);
}
__global__ __launch_bounds__(TPB) void calib_load_b32(const char *src, size_t bytes, float *out) { load_stream<float>(src, bytes, out); }
__global__ __launch_bounds__(TPB) void calib_load_b64(const char *src, size_t bytes, float *out) { load_stream<f32x2>(src, bytes, out); }
__global__ __launch_bounds__(TPB) void calib_load_b128(const char *src, size_t bytes, float *out) { load_stream<f32x4>(src, bytes, out); }

// 4 KB unit = 4 rows of 1 KB; wave-load j of a unit reads bytes [64 j, 64 j + 64) of each row: 16 lanes x 4 B contiguous per row
__global__ __launch_bounds__(TPB) void calib_load_b32_seg64(const char *src, size_t bytes, float *out)
{
    const size_t units = bytes / 4096, wave = ((size_t)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = (size_t)GRID * TPB / 64;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (size_t u = wave; u < units; u += nwaves)
#pragma unroll 4
        for (int j = 0; j < 16; ++j)
            acc += *reinterpret_cast<const float *>(src + u * 4096 + (size_t)(lane >> 4) * 1024 + j * 64 + (lane & 15) * 4);
    sink<float>(out, acc);
}
// 8 KB unit = 64 lines of 128 B; wave-load j reads lines {j, j + 8, ..., j + 56}: 8 lanes x 16 B per line, lines 1 KB apart
__global__ __launch_bounds__(TPB) void calib_load_b128_line8(const char *src, size_t bytes, float *out)
{
    const size_t units = bytes / 8192, wave = ((size_t)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = (size_t)GRID * TPB / 64;
    const int lane = threadIdx.x & 63;
    f32x4 acc = {};
    for (size_t u = wave; u < units; u += nwaves)
#pragma unroll 4
        for (int j = 0; j < 8; ++j)
            acc += *reinterpret_cast<const f32x4 *>(src + u * 8192 + (size_t)((lane >> 3) * 8 + j) * 128 + (lane & 7) * 16);
    sink<f32x4>(out, acc);
}

template <int BYTES>
__device__ __forceinline__ void load_lds_stream(const char *src, size_t bytes, float *out)
{
    __shared__ __attribute__((aligned(16))) float lds[TPB * 4];
    // 1 GiB does not fit a buffer descriptor's 32-bit offset arithmetic comfortably: one descriptor per 256 MiB window
    const size_t per_wave_instr = 64 * BYTES, n = bytes / per_wave_instr;
    const size_t wave = ((size_t)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = (size_t)GRID * TPB / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __attribute__((address_space(3))) float *lbase = (__attribute__((address_space(3))) float *)&lds[w * 64 * (BYTES / 4)];
    for (size_t i = wave; i < n; i += nwaves) {
        const char *p = src + i * per_wave_instr;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p), 0, (unsigned)per_wave_instr, 0x00020000);
        if constexpr (BYTES == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lbase, 4, (unsigned)lane * 4u, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lbase, 16, (unsigned)lane * 16u, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x] == 123.456f) out[0] = 1.f;
}
__global__ __launch_bounds__(TPB) void calib_load_lds_b32(const char *src, size_t bytes, float *out) { load_lds_stream<4>(src, bytes, out); }
__global__ __launch_bounds__(TPB) void calib_load_lds_b128(const char *src, size_t bytes, float *out) { load_lds_stream<16>(src, bytes, out); }

template <typename T>
__device__ __forceinline__ void store_stream(char *dst, size_t bytes, float v)
{
    const size_t n = bytes / sizeof(T), stride = (size_t)GRID * TPB;
    T val;
    if constexpr (sizeof(T) == 4) val = v; else if constexpr (sizeof(T) == 8) val = T{v, v}; else val = T{v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) reinterpret_cast<T *>(dst)[i] = val;
}
__global__ __launch_bounds__(TPB) void calib_store_b32(char *dst, size_t bytes, float v) { store_stream<float>(dst, bytes, v); }
__global__ __launch_bounds__(TPB) void calib_store_b128(char *dst, size_t bytes, float v) { store_stream<f32x4>(dst, bytes, v); }
__global__ __launch_bounds__(TPB) void calib_store_b32_seg64(char *dst, size_t bytes, float v)
{
    const size_t units = bytes / 4096, wave = ((size_t)blockIdx.x * TPB + threadIdx.x) >> 6, nwaves = (size_t)GRID * TPB / 64;
    const int lane = threadIdx.x & 63;
    for (size_t u = wave; u < units; u += nwaves)
#pragma unroll 4
        for (int j = 0; j < 16; ++j)
            *reinterpret_cast<float *>(dst + u * 4096 + (size_t)(lane >> 4) * 1024 + j * 64 + (lane & 15) * 4) = v;
}

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)
int main()
{
    const size_t bytes = (size_t)1 << 30;
    char *buf; float *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(K, ...) do { for (int rep = 0; rep < 3; ++rep) { hipEventRecord(e0); K<<<GRID, TPB>>>(__VA_ARGS__); hipEventRecord(e1); \
        CK(hipEventSynchronize(e1)); float ms; hipEventElapsedTime(&ms, e0, e1); \
        printf("%-24s %zu bytes  %.3f ms  %.2f TB/s\n", #K, bytes, ms, bytes / (ms * 1e-3) / 1e12); } } while (0)
    RUN(calib_load_b32, buf, bytes, out);
    RUN(calib_load_b64, buf, bytes, out);
    RUN(calib_load_b128, buf, bytes, out);
    RUN(calib_load_b32_seg64, buf, bytes, out);
    RUN(calib_load_b128_line8, buf, bytes, out);
    RUN(calib_load_lds_b32, buf, bytes, out);
    RUN(calib_load_lds_b128, buf, bytes, out);
    RUN(calib_store_b32, buf, bytes, 1.f);
    RUN(calib_store_b128, buf, bytes, 2.f);
    RUN(calib_store_b32_seg64, buf, bytes, 3.f);
    CK(hipDeviceSynchronize());
    return 0;
}
