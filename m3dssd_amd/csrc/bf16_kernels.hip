// HBM-bound helpers of the bf16 path (BASELINE.json configs[2]): 8 channels (16 bytes) per lane, fp32 arithmetic, one
// rounding to bf16 on the store.  NHWC views with a pixel stride (`*_cs`, in bf16 elements) so that channel slices of a
// concatenation buffer are read / written in place, exactly like the fp32 path.
//   m3d_stem_conv7x7_bf16 ... DLA.base_layer (pose_dla_dcn.py:336-340) from the fp32 NCHW image or the uint8 BGR frames
//   m3d_maxpool2x2_bf16 ..... Tree.downsample (pose_dla_dcn.py:306,316)
//   m3d_upsample2x_add_bf16 . IDAUp depthwise ConvTranspose2d(4, s2, p1) + skip add (pose_dla_dcn.py:536-538,550-552)
//   m3d_f32_to_bf16 ......... operand conversion (pooled ANAB keys / values)
//   m3d_softmax_rows_bf16 ... nn.Softmax(dim=-1) on the fp32 logits, probabilities written as bf16 (attention.py:208)
#include <stdlib.h>

#include "common.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void unpack8(const u32x4 u, float (&f)[8])
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(u[e] << 16);
        f[2 * e + 1] = __uint_as_float(u[e] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8])
{
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = pack2(f[2 * e], f[2 * e + 1]);
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem: 7x7, 3 -> 16, stride 1, pad 3; one thread = one output pixel x 16 channels on the packed-fp32 VALU (the image patch
// in LDS, weights through the scalar cache), bf16 NHWC store.  The uint8 form applies the reference's test-time Preprocess
// (lib/augmentations.py:44-57,472-501) in the loads, like m3d_stem_conv7x7_u8.
#define STEM_TH 8
#define STEM_TW 32
struct StemNorm {
    float mean[3], stds[3];
    int img_h, img_w;
};

template <bool U8>
__global__ __launch_bounds__(256) void stem_bf16_kernel(const void *__restrict__ img_, const float *__restrict__ wgt,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        __bf16 *__restrict__ out, int out_cs, int H, int W, StemNorm nm)
{
    constexpr int PH = STEM_TH + 6, PW = STEM_TW + 6;
    __shared__ float patch[3][PH][PW + 1];
    const int n = blockIdx.z, h0 = blockIdx.y * STEM_TH, w0 = blockIdx.x * STEM_TW;
    const float *im = static_cast<const float *>(img_) + (size_t)n * 3 * H * W;
    const unsigned char *frame = static_cast<const unsigned char *>(img_) + (size_t)n * nm.img_h * nm.img_w * 3;
    {
        constexpr int NE = 3 * PH * PW, NIT = (NE + 255) / 256;
        float v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int c = i / (PH * PW), r = (i / PW) % PH, q = i % PW;
            const int h = h0 + r - 3, w = w0 + q - 3;
            const bool inside = i < NE && h >= 0 && h < H && w >= 0 && w < W;
            if constexpr (U8) {
                float x = 0.f;
                if (inside && h < nm.img_h && w < nm.img_w) {      // plane c of the RGB tensor = BGR channel 2 - c of the frame
                    const int cb = 2 - c;
                    x = (float)frame[((size_t)h * nm.img_w + w) * 3 + cb];
                    x = x / 255.0f;
                    x = x - nm.mean[cb];
                    x = x / nm.stds[cb];
                } else if (inside) {                               // zero border of the padded frame: (0/255 - mean) / std
                    const int cb = 2 - c;
                    x = (0.0f - nm.mean[cb]) / nm.stds[cb];
                }
                v[k] = x;
            } else {
                v[k] = inside ? im[((size_t)c * H + h) * W + w] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < NE) patch[i / (PH * PW)][(i / PW) % PH][i % PW] = v[k];
        }
    }
    __syncthreads();
    const int ty = threadIdx.x / STEM_TW, tx = threadIdx.x % STEM_TW;
    f32x2 acc2[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc2[o] = f32x2{0.f, 0.f};
    for (int i = 0; i < 7; ++i) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = patch[c][ty + i][tx + j];
                const f32x2 vv = {v, v};
                const float *wp = wgt + ((i * 7 + j) * 3 + c) * 16;
#pragma unroll
                for (int o = 0; o < 8; ++o) acc2[o] = __builtin_elementwise_fma(vv, f32x2{wp[2 * o], wp[2 * o + 1]}, acc2[o]);
            }
        }
    }
    const int h = h0 + ty, w = w0 + tx;
    if (h < H && w < W) {
        __bf16 *op = out + ((size_t)(n * H + h) * W + w) * out_cs;
#pragma unroll
        for (int o8 = 0; o8 < 2; ++o8) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int o = o8 * 4 + e;
                f[2 * e] = leaky(acc2[o][0] * scale[2 * o] + shift[2 * o]);
                f[2 * e + 1] = leaky(acc2[o][1] * scale[2 * o + 1] + shift[2 * o + 1]);
            }
            global_store_u32x4_nop(op + o8 * 8, pack8(f));    // (store-data hazard found here by tools/check_isa_hazards.py: common.h)
        }
    }
}

extern "C" int m3d_stem_conv7x7_bf16(const void *img, int is_u8, int img_h, int img_w, const float *mean3, const float *stds3,
                                     const float *wgt, const float *scale, const float *shift, void *out, int out_cs, int N, int H,
                                     int W, m3d_stream_t stream)
{
    M3D_REQUIRE(img && wgt && scale && shift && out && out_cs % 8 == 0 && out_cs >= 16, "stem_bf16: bad arguments");
    StemNorm nm = {};
    if (is_u8) {
        M3D_REQUIRE(mean3 && stds3 && img_h >= 1 && img_w >= 1 && img_h <= H && img_w <= W, "stem_bf16: frame / normalisation arguments");
        for (int c = 0; c < 3; ++c) {
            M3D_REQUIRE(stds3[c] != 0.f, "stem_bf16: zero std");
            nm.mean[c] = mean3[c];
            nm.stds[c] = stds3[c];
        }
        nm.img_h = img_h; nm.img_w = img_w;
    }
    const dim3 grid(cdiv(W, STEM_TW), cdiv(H, STEM_TH), N);
    if (is_u8) hipLaunchKernelGGL(stem_bf16_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, img, wgt, scale, shift, (__bf16 *)out, out_cs, H, W, nm);
    else hipLaunchKernelGGL(stem_bf16_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, img, wgt, scale, shift, (__bf16 *)out, out_cs, H, W, nm);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void maxpool2x2_bf16_kernel(const __bf16 *__restrict__ in, int in_cs, __bf16 *__restrict__ out, int out_cs, int N, int H,
                                       int W, int C8)
{
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * Ho * Wo * C8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long long p = i / C8;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho), n = (int)(p / Ho);
        const __bf16 *b = in + ((size_t)(n * H + 2 * ho) * W + 2 * wo) * in_cs + c8 * 8;
        float a0[8], a1[8], a2[8], a3[8], r[8];
        unpack8(*reinterpret_cast<const u32x4 *>(b), a0);
        unpack8(*reinterpret_cast<const u32x4 *>(b + in_cs), a1);
        unpack8(*reinterpret_cast<const u32x4 *>(b + (size_t)W * in_cs), a2);
        unpack8(*reinterpret_cast<const u32x4 *>(b + (size_t)W * in_cs + in_cs), a3);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = fmaxf(fmaxf(a0[e], a1[e]), fmaxf(a2[e], a3[e]));
        *reinterpret_cast<u32x4 *>(out + ((size_t)(n * Ho + ho) * Wo + wo) * out_cs + c8 * 8) = pack8(r);
    }
}

extern "C" int m3d_maxpool2x2_bf16(const void *in, int in_cs, void *out, int out_cs, int N, int H, int W, int C, m3d_stream_t stream)
{
    M3D_REQUIRE(in && out && C % 8 == 0 && in_cs % 8 == 0 && out_cs % 8 == 0, "maxpool_bf16: C and strides must be x8");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2x2_bf16_kernel, dim3(imin(cdiv(total, 256), 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)in, in_cs, (__bf16 *)out, out_cs, N, H, W, C / 8);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// out[y][x] = sum over the 2x2 contributing inputs in[iy][ix] * w[ky][kx] (y = 2*iy - 1 + ky) + skip[y][x]; wgt fp32 [4][4][C]
template <typename IT>
__global__ void upsample2x_add_bf16_kernel(const __bf16 *__restrict__ in, int in_cs, const float *__restrict__ wgt,
                                           const __bf16 *__restrict__ skip, int skip_cs, __bf16 *__restrict__ out, int out_cs, int N,
                                           int H, int W, int C8)
{
    const int Ho = 2 * H, Wo = 2 * W, C = C8 * 8;
    const IT total = (IT)N * Ho * Wo * C8;
    for (IT i = (IT)blockIdx.x * (IT)blockDim.x + threadIdx.x; i < total; i += (IT)gridDim.x * (IT)blockDim.x) {
        const int c8 = (int)(i % C8);
        IT p = i / C8;
        const int x = (int)(p % Wo);
        p /= Wo;
        const int y = (int)(p % Ho), n = (int)(p / Ho);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int iy_hi = (y + 1) >> 1, ix_hi = (x + 1) >> 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int iy = iy_hi - a, ky = y + 1 - 2 * iy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ix = ix_hi - b, kx = x + 1 - 2 * ix;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                unpack8(*reinterpret_cast<const u32x4 *>(in + ((size_t)(n * H + iy) * W + ix) * in_cs + c8 * 8), v);
                const float *wp = wgt + (ky * 4 + kx) * C + c8 * 8;
                const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp), w1 = *reinterpret_cast<const f32x4 *>(wp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] += v[e] * w0[e]; acc[4 + e] += v[4 + e] * w1[e]; }
            }
        }
        const size_t o = (size_t)(n * Ho + y) * Wo + x;
        if (skip) {
            float s[8];
            unpack8(*reinterpret_cast<const u32x4 *>(skip + o * skip_cs + c8 * 8), s);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += s[e];
        }
        *reinterpret_cast<u32x4 *>(out + o * out_cs + c8 * 8) = pack8(acc);
    }
}

// The same, one output row per blockIdx.y and image per blockIdx.z: a thread = (pixel of a 256 / C8-pixel segment, 8 channels) -- no
// index divisions (the grid-stride form spends ~100 integer instructions per 8 channels on i / C8 / Wo / Ho: 0.084 ms for the 283 MB
// of a 128-channel 24x80 -> 48x160 step at bs 64 = 3.4 TB/s).  LOG2C8 = log2(C / 8).
template <int LOG2C8>
__global__ __launch_bounds__(256) void upsample2x_add_bf16_rows_kernel(const __bf16 *__restrict__ in, int in_cs, const float *__restrict__ wgt,
                                                                       const __bf16 *__restrict__ skip, int skip_cs, __bf16 *__restrict__ out,
                                                                       int out_cs, int H, int W)
{
    constexpr int C8 = 1 << LOG2C8, PX = 256 >> LOG2C8, C = C8 * 8;
    const int Ho = 2 * H, Wo = 2 * W;
    const int c8 = threadIdx.x & (C8 - 1);
    const int x = blockIdx.x * PX + (threadIdx.x >> LOG2C8), y = blockIdx.y, n = blockIdx.z;
    if (x >= Wo) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int iy_hi = (y + 1) >> 1, ix_hi = (x + 1) >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int iy = iy_hi - a, ky = y + 1 - 2 * iy;
        if (iy < 0 || iy >= H) continue;                         // (block-uniform)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ix = ix_hi - b, kx = x + 1 - 2 * ix;
            if (ix < 0 || ix >= W) continue;
            float v[8];
            unpack8(*reinterpret_cast<const u32x4 *>(in + ((size_t)(n * H + iy) * W + ix) * in_cs + c8 * 8), v);
            const float *wp = wgt + (ky * 4 + kx) * C + c8 * 8;
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp), w1 = *reinterpret_cast<const f32x4 *>(wp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += v[e] * w0[e]; acc[4 + e] += v[4 + e] * w1[e]; }
        }
    }
    const size_t o = (size_t)(n * Ho + y) * Wo + x;
    if (skip) {
        float s[8];
        unpack8(*reinterpret_cast<const u32x4 *>(skip + o * skip_cs + c8 * 8), s);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += s[e];
    }
    *reinterpret_cast<u32x4 *>(out + o * out_cs + c8 * 8) = pack8(acc);
}

// Row form with the weights in registers (round 5, second half): workgroup = one output row of one image, thread = (8 channels, a
// segment of the row); the 2 x 4 x 8 weights an output row needs are loaded ONCE per thread and the thread walks its segment in
// output PAIRS (2i, 2i + 1), which read input columns i - 1, i, i + 1 of two input rows -- a sliding window, 2 new 16-byte loads per
// pair.  Per output pixel: one input load, one skip load, one store, 32 FMAs (the form above: 4 input + 8 weight loads per pixel).
template <int LOG2C8>
__global__ __launch_bounds__(256) void upsample2x_add_bf16_seg_kernel(const __bf16 *__restrict__ in, int in_cs, const float *__restrict__ wgt,
                                                                      const __bf16 *__restrict__ skip, int skip_cs, __bf16 *__restrict__ out,
                                                                      int out_cs, int H, int W, int P)
{
    constexpr int C8 = 1 << LOG2C8, C = C8 * 8;
    const int c8 = threadIdx.x & (C8 - 1), seg = threadIdx.x >> LOG2C8;
    const int y = blockIdx.x, n = blockIdx.y;
    const int Wo = 2 * W;
    const int iy_hi = (y + 1) >> 1, ky_hi = y + 1 - 2 * iy_hi;              // rows iy_hi (tap ky_hi) and iy_hi - 1 (tap ky_hi + 2)
    float w[2][4][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const float *wp = wgt + ((ky_hi + 2 * a) * 4 + kx) * C + c8 * 8;
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(wp), w1 = *reinterpret_cast<const f32x4 *>(wp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { w[a][kx][e] = w0[e]; w[a][kx][4 + e] = w1[e]; }
        }
    const bool rok[2] = {iy_hi < H, iy_hi - 1 >= 0};                         // (block-uniform)
    const __bf16 *rin[2] = {in + ((size_t)(n * H + iy_hi) * W) * in_cs + c8 * 8, in + ((size_t)(n * H + iy_hi - 1) * W) * in_cs + c8 * 8};
    auto col = [&](int a, int ix) __attribute__((always_inline)) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (rok[a] && ix >= 0 && ix < W) v = *reinterpret_cast<const u32x4 *>(rin[a] + (size_t)ix * in_cs);
        return v;
    };
    const int i0 = seg * P, i1 = min(W, i0 + P);
    if (i0 >= i1) return;
    u32x4 cm[2], cc[2];                                                      // columns i - 1 and i of the two rows
#pragma unroll
    for (int a = 0; a < 2; ++a) { cm[a] = col(a, i0 - 1); cc[a] = col(a, i0); }
    const size_t orow = (size_t)(n * 2 * H + y) * Wo;
    for (int i = i0; i < i1; ++i) {
        u32x4 cp[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) cp[a] = col(a, i + 1);
        u32x4 s0 = {0u, 0u, 0u, 0u}, s1 = s0;
        if (skip) {
            s0 = *reinterpret_cast<const u32x4 *>(skip + (orow + 2 * i) * skip_cs + c8 * 8);
            s1 = *reinterpret_cast<const u32x4 *>(skip + (orow + 2 * i + 1) * skip_cs + c8 * 8);
        }
        float a0[8], a1[8];
        unpack8(s0, a0);
        unpack8(s1, a1);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float vm[8], vc[8], vp[8];
            unpack8(cm[a], vm);
            unpack8(cc[a], vc);
            unpack8(cp[a], vp);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a0[e] += vc[e] * w[a][1][e] + vm[e] * w[a][3][e];            // x = 2i:     column i tap 1, column i - 1 tap 3
                a1[e] += vp[e] * w[a][0][e] + vc[e] * w[a][2][e];            // x = 2i + 1: column i + 1 tap 0, column i tap 2
            }
        }
        global_store_u32x4_nop(out + (orow + 2 * i) * out_cs + c8 * 8, pack8(a0));
        global_store_u32x4_nop(out + (orow + 2 * i + 1) * out_cs + c8 * 8, pack8(a1));
#pragma unroll
        for (int a = 0; a < 2; ++a) { cm[a] = cc[a]; cc[a] = cp[a]; }
    }
}

extern "C" int m3d_upsample2x_add_bf16(const void *in, int in_cs, const float *wgt, const void *skip, int skip_cs, void *out,
                                       int out_cs, int N, int H, int W, int C, m3d_stream_t stream)
{
    M3D_REQUIRE(in && wgt && out && C % 8 == 0 && in_cs % 8 == 0 && out_cs % 8 == 0 && (!skip || skip_cs % 8 == 0),
                "upsample2x_add_bf16: C and strides must be x8");
    static const int rows_form = []() { const char *e = getenv("M3D_UPSAMPLE_ROWS"); return e ? atoi(e) : 1; }();
    if (rows_form == 1 && (C == 128 || C == 256 || C == 64) && N <= 65535) {
        const int nseg = 256 / (C / 8), P = cdiv(W, nseg);
        const dim3 grid(2 * H, N);
#define UPS_SEG(L2) hipLaunchKernelGGL(upsample2x_add_bf16_seg_kernel<L2>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)in, in_cs, \
                                       wgt, (const __bf16 *)skip, skip_cs, (__bf16 *)out, out_cs, H, W, P)
        if (C == 64) UPS_SEG(3);
        else if (C == 128) UPS_SEG(4);
        else UPS_SEG(5);
#undef UPS_SEG
        M3D_LAUNCH_CHECK();
        return M3D_OK;
    }
    if (rows_form && (C == 128 || C == 256 || C == 64) && 2 * H <= 65535 && N <= 65535) {     // M3D_UPSAMPLE_ROWS=2: a thread per pixel and 8 channels
        const int px = 256 / (C / 8);
        const dim3 grid(cdiv(2 * W, px), 2 * H, N);
#define UPS_ROWS(L2) hipLaunchKernelGGL(upsample2x_add_bf16_rows_kernel<L2>, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)in, in_cs, \
                                        wgt, (const __bf16 *)skip, skip_cs, (__bf16 *)out, out_cs, H, W)
        if (C == 64) UPS_ROWS(3);
        else if (C == 128) UPS_ROWS(4);
        else UPS_ROWS(5);
#undef UPS_ROWS
        M3D_LAUNCH_CHECK();
        return M3D_OK;
    }
    const long long total = (long long)N * 4 * H * W * (C / 8);
    // the int form's grid-stride increment (at most 16384 x 256) must not carry the index past 2^31 on its last step
    if (total < (1ll << 31) - 16384ll * 256)
        hipLaunchKernelGGL(upsample2x_add_bf16_kernel<int>, dim3(imin(cdiv(total, 256), 16384)), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16 *)in, in_cs, wgt, (const __bf16 *)skip, skip_cs, (__bf16 *)out, out_cs, N, H, W, C / 8);
    else
        hipLaunchKernelGGL(upsample2x_add_bf16_kernel<long long>, dim3(imin(cdiv(total, 256), 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)in, in_cs, wgt, (const __bf16 *)skip, skip_cs, (__bf16 *)out, out_cs, N, H, W, C / 8);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ void f32_to_bf16_kernel(const float *__restrict__ src, __bf16 *__restrict__ dst, long long n8)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(src + i * 8), b = *reinterpret_cast<const f32x4 *>(src + i * 8 + 4);
        const u32x4 r = {pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
        *reinterpret_cast<u32x4 *>(dst + i * 8) = r;
    }
}

extern "C" int m3d_f32_to_bf16(const float *src, void *dst, long long n, m3d_stream_t stream)
{
    M3D_REQUIRE(src && dst && n % 8 == 0 && n > 0, "f32_to_bf16: n must be a positive multiple of 8");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(imin(cdiv(n / 8, 256), 8192)), dim3(256), 0, (hipStream_t)stream, src, (__bf16 *)dst, n / 8);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Row softmax over the first `valid` fp32 columns (one wave per row), probabilities to bf16; columns [valid, out_cs) = 0.
__global__ __launch_bounds__(256) void softmax_rows_bf16_kernel(const float *__restrict__ x, int rows, int valid, int cs,
                                                                __bf16 *__restrict__ out, int out_cs)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * cs;
    constexpr int NV = 8;                       // up to 512 columns in registers
    float v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < valid ? xr[c] : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k] = lane + 64 * k < valid ? expf(v[k] - mx) : 0.f;
        s += v[k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float inv = 1.0f / s;
    __bf16 *orow = out + (size_t)row * out_cs;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < out_cs) orow[c] = (__bf16)(v[k] * inv);
    }
}

extern "C" int m3d_softmax_rows_bf16(const float *x, int rows, int valid, int cs, void *out, int out_cs, m3d_stream_t stream)
{
    M3D_REQUIRE(x && out && rows > 0 && valid > 0 && valid <= cs && valid <= out_cs && out_cs <= 512,
                "softmax_rows_bf16: bad arguments (at most 512 columns)");
    hipLaunchKernelGGL(softmax_rows_bf16_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, valid, cs,
                       (__bf16 *)out, out_cs);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
