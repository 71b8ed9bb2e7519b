// HBM-bound helpers of the backbone / up-sampling path (NHWC, 16-byte vector access along C):
// weight packing, layout changes at the op boundary, the 7x7 stem, 2x2 max-pool and the
// depthwise transposed-conv up-sampler fused with the IDA skip add.
#include <stdlib.h>

#include "common.h"

// ---------------------------------------------------------------------------------------
// [Cout, Cin, kh, kw] -> [Cout_pad, (i*kw + j)*Cin_pad + c]   (rows >= Cout and channels >= Cin are zero)
// (the source may be a channel slice [c0, c0 + Cin) of a [Cout, Ctot, kh, kw] tensor: one deformable group)
__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ p, int Cout, int Cout_pad,
                                   int Cin, int Cin_pad, int KK, int Ctot, int c0)
{
    const long long total = (long long)Cout_pad * KK * Cin_pad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin_pad);
        const int tap = (int)((i / Cin_pad) % KK);
        const int co = (int)(i / ((long long)Cin_pad * KK));
        p[i] = (co < Cout && c < Cin) ? w[((long long)co * Ctot + c0 + c) * KK + tap] : 0.f;
    }
}

int m3d_pack_conv_weight_slice(const float *w, int Ctot, int c0, float *packed, int Cout, int Cout_pad, int Cin, int Cin_pad,
                               int kh, int kw, m3d_stream_t stream)
{
    M3D_REQUIRE(w && packed && Cout > 0 && Cout_pad >= Cout && Cin_pad >= Cin && c0 >= 0 && c0 + Cin <= Ctot,
                "pack_conv_weight: bad arguments");
    const long long total = (long long)Cout_pad * kh * kw * Cin_pad;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(imin(cdiv(total, 256), 4096)), dim3(256), 0, (hipStream_t)stream, w,
                       packed, Cout, Cout_pad, Cin, Cin_pad, kh * kw, Ctot, c0);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_pack_conv_weight(const float *w, float *packed, int Cout, int Cout_pad, int Cin, int Cin_pad,
                                    int kh, int kw, m3d_stream_t stream)
{
    return m3d_pack_conv_weight_slice(w, Cin, 0, packed, Cout, Cout_pad, Cin, Cin_pad, kh, kw, stream);
}

// ---------------------------------------------------------------------------------------
// NCHW <-> NHWC through a 32x32 LDS tile (coalesced on both sides).
// (the source may be a channel slice [c0, c0 + C) of an image with Ctot channels)
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int C, int HW, int out_cs, int Ctot,
                                    int c0s)
{
    __shared__ float t[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        t[r][tx] = (c < C && p < HW) ? in[((size_t)n * Ctot + c0s + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (c < C && p < HW) out[((size_t)n * HW + p) * out_cs + c] = t[tx][r];
    }
}

__global__ void nhwc_to_nchw_kernel(const float *__restrict__ in, int in_cs, float *__restrict__ out, int C, int HW)
{
    __shared__ float t[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        t[r][tx] = (c < C && p < HW) ? in[((size_t)n * HW + p) * in_cs + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        if (c < C && p < HW) out[((size_t)n * C + c) * HW + p] = t[tx][r];
    }
}

int m3d_nchw_to_nhwc_slice(const float *in, int Ctot, int c0, float *out, int N, int C, int H, int W, int out_cs,
                           m3d_stream_t stream)
{
    M3D_REQUIRE(in && out && out_cs >= C && c0 >= 0 && c0 + C <= Ctot, "nchw_to_nhwc: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), N), dim3(256), 0, (hipStream_t)stream, in,
                       out, C, HW, out_cs, Ctot, c0);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_nchw_to_nhwc(const float *in, float *out, int N, int C, int H, int W, int out_cs,
                                m3d_stream_t stream)
{
    return m3d_nchw_to_nhwc_slice(in, C, 0, out, N, C, H, W, out_cs, stream);
}

extern "C" int m3d_nhwc_to_nchw(const float *in, int in_cs, float *out, int N, int C, int H, int W,
                                m3d_stream_t stream)
{
    M3D_REQUIRE(in && out && in_cs >= C, "nhwc_to_nchw: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), N), dim3(256), 0, (hipStream_t)stream, in,
                       in_cs, out, C, HW);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------
// Stem: 7x7, 3 -> 16, stride 1, pad 3 (model/pose_dla_dcn.py:336-340), NCHW image in, NHWC out.
// HBM-bound layer (5.9 MB in, 31.5 MB out per 384x1280 image).  One thread = one output pixel x
// 16 channels; the input patch of the 8x32-pixel tile sits in LDS, the 2352 weights are read
// through the scalar cache (wave-uniform index) so the VALU sees them as SGPR operands.
#define STEM_TH 8
#define STEM_TW 32
// Test-time input path of the reference fused into the load: uint8 BGR frames [N][img_h][img_w][3], zero
// border at the bottom / right up to H x W, then /255, -mean, /stds in float32 with mean / stds indexed by the BGR channel
// position, BGR -> RGB (lib/augmentations.py:36-57,138-160,472-501, lib/dataloader.py:943-950).  IEEE division, no contraction:
// bit-identical to the numpy arithmetic.
struct U8Norm {
    float mean[3], stds[3];       // indexed by BGR position, as the reference applies them
    int img_h, img_w;
};
__device__ __forceinline__ float u8_normalise(const unsigned char *__restrict__ frame, const U8Norm &nm, int h, int w, int c_rgb)
{
    const int cb = 2 - c_rgb;
    float v = (h < nm.img_h && w < nm.img_w) ? (float)frame[((size_t)h * nm.img_w + w) * 3 + cb] : 0.f;
    v = v / 255.0f;
    v = v - nm.mean[cb];
    return v / nm.stds[cb];
}

template <bool U8>
__global__ __launch_bounds__(256) void stem_conv7x7_kernel(const void *__restrict__ img_, const float *__restrict__ wgt,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift, float *__restrict__ out,
                                                           int out_cs, int H, int W, U8Norm nm)
{
    const float *img = static_cast<const float *>(img_);
    constexpr int PH = STEM_TH + 6, PW = STEM_TW + 6;
    __shared__ float patch[3][PH][PW + 1];
    const int n = blockIdx.z, h0 = blockIdx.y * STEM_TH, w0 = blockIdx.x * STEM_TW;
    const float *im = img + (size_t)n * 3 * H * W;
    const unsigned char *frame = static_cast<const unsigned char *>(img_) + (size_t)n * nm.img_h * nm.img_w * 3;
    {   // all loads of the thread in flight before the first LDS write
        constexpr int NE = 3 * PH * PW, NIT = (NE + 255) / 256;
        float v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int c = i / (PH * PW), r = (i / PW) % PH, q = i % PW;
            const int h = h0 + r - 3, w = w0 + q - 3;
            const bool inside = i < NE && h >= 0 && h < H && w >= 0 && w < W;      // outside: the conv's own zero padding
            if constexpr (U8) v[k] = inside ? u8_normalise(frame, nm, h, w, c) : 0.f;
            else v[k] = inside ? im[((size_t)c * H + h) * W + w] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < NE) patch[i / (PH * PW)][(i / PW) % PH][i % PW] = v[k];
        }
    }
    __syncthreads();
    const int ty = threadIdx.x / STEM_TW, tx = threadIdx.x % STEM_TW;
    // packed fp32: two output channels per v_pk_fma_f32 (the pixel value is broadcast, the weight pair is an SGPR pair) --
    // half the VALU instructions of a scalar-FMA loop, which ran at the non-packed VALU peak
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
                const float *wp = wgt + ((i * 7 + j) * 3 + c) * 16;   // uniform -> s_load
#pragma unroll
                for (int o = 0; o < 8; ++o) acc2[o] = __builtin_elementwise_fma(vv, f32x2{wp[2 * o], wp[2 * o + 1]}, acc2[o]);
            }
        }
    }
    float acc[16];
#pragma unroll
    for (int o = 0; o < 8; ++o) { acc[2 * o] = acc2[o][0]; acc[2 * o + 1] = acc2[o][1]; }
    const int h = h0 + ty, w = w0 + tx;
    if (h < H && w < W) {
        float *op = out + ((size_t)(n * H + h) * W + w) * out_cs;
#pragma unroll
        for (int o4 = 0; o4 < 4; ++o4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = leaky(acc[o4 * 4 + e] * scale[o4 * 4 + e] + shift[o4 * 4 + e]);
            *reinterpret_cast<f32x4 *>(op + o4 * 4) = v;
        }
    }
}

extern "C" int m3d_stem_conv7x7(const float *img_nchw, const float *wgt, const float *scale, const float *shift,
                                float *out, int out_cs, int N, int H, int W, m3d_stream_t stream)
{
    M3D_REQUIRE(img_nchw && wgt && scale && shift && out && out_cs % 4 == 0 && out_cs >= 16, "stem: bad arguments");
    hipLaunchKernelGGL(stem_conv7x7_kernel<false>, dim3(cdiv(W, STEM_TW), cdiv(H, STEM_TH), N), dim3(256), 0,
                       (hipStream_t)stream, (const void *)img_nchw, wgt, scale, shift, out, out_cs, H, W, U8Norm{});
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

static int fill_u8norm(U8Norm &nm, const float *mean3, const float *stds3, int img_h, int img_w, int H, int W)
{
    M3D_REQUIRE(mean3 && stds3, "preprocess: mean / stds are host pointers to 3 floats each");
    M3D_REQUIRE(img_h >= 1 && img_w >= 1 && img_h <= H && img_w <= W,
                "preprocess: the frame (%dx%d) must fit the padded size (%dx%d)", img_h, img_w, H, W);
    for (int c = 0; c < 3; ++c) {
        M3D_REQUIRE(stds3[c] != 0.f, "preprocess: zero std");
        nm.mean[c] = mean3[c];
        nm.stds[c] = stds3[c];
    }
    nm.img_h = img_h; nm.img_w = img_w;
    return M3D_OK;
}

extern "C" int m3d_stem_conv7x7_u8(const unsigned char *frames_bgr, int img_h, int img_w, const float *mean3, const float *stds3,
                                   const float *wgt, const float *scale, const float *shift, float *out, int out_cs, int N, int H,
                                   int W, m3d_stream_t stream)
{
    M3D_REQUIRE(frames_bgr && wgt && scale && shift && out && out_cs % 4 == 0 && out_cs >= 16, "stem_u8: bad arguments");
    U8Norm nm;
    const int rc = fill_u8norm(nm, mean3, stds3, img_h, img_w, H, W);
    if (rc != M3D_OK) return rc;
    hipLaunchKernelGGL(stem_conv7x7_kernel<true>, dim3(cdiv(W, STEM_TW), cdiv(H, STEM_TH), N), dim3(256), 0,
                       (hipStream_t)stream, (const void *)frames_bgr, wgt, scale, shift, out, out_cs, H, W, nm);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// The same input path on its own: [N][img_h][img_w][3] uint8 BGR -> [N][3][H][W] float32 RGB planes.
__global__ void preprocess_u8_kernel(const unsigned char *__restrict__ frames, float *__restrict__ out, int N, int H, int W,
                                     U8Norm nm)
{
    const long long total = (long long)N * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const long long t = i / W;
        const int h = (int)(t % H), n = (int)(t / H);
        const unsigned char *frame = frames + (size_t)n * nm.img_h * nm.img_w * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(((size_t)n * 3 + c) * H + h) * W + w] = u8_normalise(frame, nm, h, w, c);
    }
}

extern "C" int m3d_preprocess_u8(const unsigned char *frames_bgr, int N, int img_h, int img_w, const float *mean3,
                                 const float *stds3, float *out_nchw, int H, int W, m3d_stream_t stream)
{
    M3D_REQUIRE(frames_bgr && out_nchw && N >= 1, "preprocess_u8: bad arguments");
    U8Norm nm;
    const int rc = fill_u8norm(nm, mean3, stds3, img_h, img_w, H, W);
    if (rc != M3D_OK) return rc;
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(imin(cdiv(total, 256), 16384)), dim3(256), 0, (hipStream_t)stream, frames_bgr,
                       out_nchw, N, H, W, nm);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------
// level0: 3x3, 16 -> 16, stride 1, pad 1 at full resolution (model/pose_dla_dcn.py:341-342), NHWC in/out.
// (VALU reference variant, M3D_L0_VALU=1; the MFMA kernel below is the default: 0.211 vs 0.219 ms at bs=8)
// 16 output channels cannot fill a 32-wide MFMA tile (the igemm pads to 32 and wastes half the pipe) and the
// layer moves 0.5 GB at bs=8, so it runs as a direct convolution on the VALU like the stem: one thread = one output
// pixel x 16 channels, the (8+2)x(32+2)x16 input tile in LDS (pixel stride 20 floats: conflict-free b128 reads),
// the 2304 weights through the scalar cache (wave-uniform index -> SGPR operands of v_fma).
// wgt layout [(i*3 + j)*16 + cin][16 cout].
#define L0_TH 8
#define L0_TW 32
#define L0_PS 20
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const float *__restrict__ in, int in_cs,
                                                          const float *__restrict__ wgt, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, float *__restrict__ out,
                                                          int out_cs, int H, int W)
{
    constexpr int PH = L0_TH + 2, PW = L0_TW + 2;
    __shared__ __attribute__((aligned(16))) float tile[PH * PW * L0_PS];
    const int n = blockIdx.z, h0 = blockIdx.y * L0_TH, w0 = blockIdx.x * L0_TW;
    for (int i = threadIdx.x; i < PH * PW * 4; i += 256) {
        const int q = i & 3, p = i >> 2;
        const int r = p / PW, c = p - r * PW;
        const int h = h0 + r - 1, w = w0 + c - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (h >= 0 && h < H && w >= 0 && w < W)
            v = *reinterpret_cast<const f32x4 *>(in + ((size_t)(n * H + h) * W + w) * in_cs + q * 4);
        *reinterpret_cast<f32x4 *>(tile + p * L0_PS + q * 4) = v;
    }
    __syncthreads();
    const int ty = threadIdx.x / L0_TW, tx = threadIdx.x % L0_TW;
    f32x2 acc2[8];                             // packed fp32, two output channels per v_pk_fma_f32 (see the stem)
#pragma unroll
    for (int o = 0; o < 8; ++o) acc2[o] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float *px = tile + ((ty + i) * PW + tx + j) * L0_PS;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 x4 = *reinterpret_cast<const f32x4 *>(px + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float *wp = wgt + (((i * 3 + j) * 16) + q * 4 + e) * 16;   // uniform -> s_load
                    const f32x2 vv = {x4[e], x4[e]};
#pragma unroll
                    for (int o = 0; o < 8; ++o) acc2[o] = __builtin_elementwise_fma(vv, f32x2{wp[2 * o], wp[2 * o + 1]}, acc2[o]);
                }
            }
        }
    }
    float acc[16];
#pragma unroll
    for (int o = 0; o < 8; ++o) { acc[2 * o] = acc2[o][0]; acc[2 * o + 1] = acc2[o][1]; }
    const int h = h0 + ty, w = w0 + tx;
    if (h < H && w < W) {
        float *op = out + ((size_t)(n * H + h) * W + w) * out_cs;
#pragma unroll
        for (int o4 = 0; o4 < 4; ++o4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = leaky(acc[o4 * 4 + e] * scale[o4 * 4 + e] + shift[o4 * 4 + e]);
            *reinterpret_cast<f32x4 *>(op + o4 * 4) = v;
        }
    }
}


// MFMA variant of level0.  The VALU kernel above runs at the non-packed VALU peak (v_pk_fma_f32 issues at half rate, so
// packing buys ~5 %); v_mfma_f32_16x16x4_f32 has exactly the 16-wide N this layer needs (no padded half tile):
//   M = 16 consecutive pixels of a row, N = 16 couts, K = 4 channels per MFMA, 4 MFMAs per tap.
// Workgroup = 8 rows x 64 cols of output; the (8+2) x (64+2) x 16-channel input tile sits in LDS (64 B per pixel: the
// A-operand read of a 16-pixel tile is one contiguous 1 KB ds_read_b128), borders are zero-filled by buffer range checks.
// Lane (i = lane & 15, q = lane >> 4): A = channels 4q..4q+3 of pixel i (component t feeds MFMA t), B = W[tap][4q + t][cout i]
// held in 36 registers for the whole kernel; D: lane = cout, registers = pixels 4q..4q+3.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define L0M_TH 8
#define L0M_TW 64
__global__ __launch_bounds__(256) void conv3x3_c16_mfma_kernel(const float *__restrict__ in, int in_cs, unsigned in_bytes,
                                                               const float *__restrict__ wgt, const float *__restrict__ scale,
                                                               const float *__restrict__ shift, float *__restrict__ out,
                                                               int out_cs, int H, int W)
{
    // LDS tile: 64 bytes per pixel, row pitch PP = 72 pixels (66 used), the 16-byte channel quad q of pixel column c stored at
    // slot q ^ ((c >> 1) & 3).  ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): with
    // plain 64-byte pixels the A-fragment read of 16 pixels x 4 quads is 2-way conflicted in every group (PMC: 7.4 M conflict
    // cycles against 3.5 M active LDS cycles per launch); with this swizzle every group covers the 64 banks once for any starting
    // column, and a pitch that is a multiple of 8 pixels makes the swizzle term a per-lane constant per tap column.
    constexpr int PH = L0M_TH + 2, PW = L0M_TW + 2, PP = 72;
    __shared__ __attribute__((aligned(16))) float tile[PH * PP * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int n = blockIdx.z, h0 = blockIdx.y * L0M_TH, w0 = blockIdx.x * L0M_TW;
    // ---- B operand: 9 taps x (4 channels of this lane's k slot) for cout li --------------------------------------
    f32x4 bw[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
        for (int t = 0; t < 4; ++t) bw[t9][t] = wgt[((t9 * 16) + 4 * lq + t) * 16 + li];
    // ---- stage the input tile ---------------------------------------------------------------------------------------
    {
        // all loads of the thread are issued before the first LDS write (a rolled loop would serialise 11 memory latencies)
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(in, in_bytes);
        constexpr int NE = PH * PW * 4, NIT = (NE + 255) / 256;
        f32x4 v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + 256 * k;
            const int q = e & 3, p = e >> 2;
            const int r = p / PW, c = p - r * PW;
            const int h = h0 + r - 1, w = w0 + c - 1;
            const unsigned vo = (e < NE && h >= 0 && h < H && w >= 0 && w < W)
                                    ? ((unsigned)((n * H + h) * W + w) * (unsigned)in_cs + (unsigned)q * 4u) * 4u : M3D_BUF_OOB;
            v[k] = buf_load_f32x4(rin, vo, 0);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + 256 * k;
            const int q = e & 3, p = e >> 2;
            const int r = p / PW, c = p - r * PW;
            if (e < NE) *reinterpret_cast<f32x4 *>(tile + ((r * PP + c) * 4 + (q ^ ((c >> 1) & 3))) * 4) = v[k];
        }
    }
    __syncthreads();
    // ---- 9 taps x 8 pixel tiles (2 rows x 4 column tiles per wave) x 4 MFMAs -------------------------------------------
    f32x4 acc[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rr][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    int qoff[3];                                    // float offset of this lane's channel quad for tap column tj (column = 16*ct + li + tj)
#pragma unroll
    for (int tj = 0; tj < 3; ++tj) qoff[tj] = (lq ^ (((li + tj) >> 1) & 3)) * 4;
#pragma unroll
    for (int ti = 0; ti < 3; ++ti)
#pragma unroll
        for (int tj = 0; tj < 3; ++tj) {
            const f32x4 b = bw[ti * 3 + tj];
            // the 8 A fragments of the tap first, then t-major MFMAs: consecutive MFMAs hit different accumulators (a dependent
            // v_mfma_f32_16x16x4_f32 waits 40 cycles, an independent one issues every 32 -- MI355X_MICROARCH.md)
            f32x4 a[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    a[rr][ct] = *reinterpret_cast<const f32x4 *>(
                        tile + ((2 * wave + rr + ti) * PP + ct * 16 + li + tj) * 16 + qoff[tj]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        // operands swapped (A = weights, B = pixels): D[cout][pixel], so a lane ends up with 4 consecutive
                        // channels of ONE pixel and the epilogue stores 16 bytes per lane, 1 KB contiguous per instruction
                        acc[rr][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[t], a[rr][ct][t], acc[rr][ct], 0, 0, 0);
        }
    // ---- epilogue: lane = pixel li of the 16-pixel tile, registers = couts 4*lq + r -------------------------------------------
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + 4 * lq), sh = *reinterpret_cast<const f32x4 *>(shift + 4 * lq);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int h = h0 + 2 * wave + rr;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int w = w0 + ct * 16 + li;
            f32x4 v = acc[rr][ct] * sc + sh;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = leaky(v[e]);
            if (h < H && w < W) *reinterpret_cast<f32x4 *>(out + ((size_t)(n * H + h) * W + w) * out_cs + 4 * lq) = v;
        }
    }
}

extern "C" int m3d_conv3x3_c16(const float *in, int in_cs, const float *wgt, const float *scale, const float *shift,
                               float *out, int out_cs, int N, int H, int W, m3d_stream_t stream)
{
    M3D_REQUIRE(in && wgt && scale && shift && out && in_cs % 4 == 0 && out_cs % 4 == 0 && in_cs >= 16 && out_cs >= 16,
                "conv3x3_c16: bad arguments");
    M3D_REQUIRE((((uintptr_t)in | (uintptr_t)out | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0,
                "conv3x3_c16: in / out / scale / shift must be 16-byte aligned (float4 accesses)");
    static int valu = -1;                        // tuning knob (experiments only): M3D_L0_VALU=1 selects the VALU kernel
    if (valu < 0) { const char *e = getenv("M3D_L0_VALU"); valu = e ? atoi(e) : 0; }
    const long long in_bytes = (long long)N * H * W * in_cs * 4;
    if (!valu && in_bytes < (1ll << 31))
        hipLaunchKernelGGL(conv3x3_c16_mfma_kernel, dim3(cdiv(W, L0M_TW), cdiv(H, L0M_TH), N), dim3(256), 0, (hipStream_t)stream,
                           in, in_cs, (unsigned)in_bytes, wgt, scale, shift, out, out_cs, H, W);
    else
        hipLaunchKernelGGL(conv3x3_c16_kernel, dim3(cdiv(W, L0_TW), cdiv(H, L0_TH), N), dim3(256), 0, (hipStream_t)stream, in,
                           in_cs, wgt, scale, shift, out, out_cs, H, W);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------
// MaxPool2d(2, 2) (floor mode), float4 along channels.
__global__ void maxpool2x2_kernel(const float *__restrict__ in, int in_cs, float *__restrict__ out, int out_cs, int N,
                                  int H, int W, int C4)
{
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long p = i / C4;
        const int wo = (int)(p % Wo);
        p /= Wo;
        const int ho = (int)(p % Ho), n = (int)(p / Ho);
        const float *b = in + ((size_t)(n * H + 2 * ho) * W + 2 * wo) * in_cs + c4 * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4 *>(b);
        const f32x4 v01 = *reinterpret_cast<const f32x4 *>(b + in_cs);
        const f32x4 v10 = *reinterpret_cast<const f32x4 *>(b + (size_t)W * in_cs);
        const f32x4 v11 = *reinterpret_cast<const f32x4 *>(b + (size_t)W * in_cs + in_cs);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
        *reinterpret_cast<f32x4 *>(out + ((size_t)(n * Ho + ho) * Wo + wo) * out_cs + c4 * 4) = r;
    }
}

extern "C" int m3d_maxpool2x2(const float *in, int in_cs, float *out, int out_cs, int N, int H, int W, int C,
                              m3d_stream_t stream)
{
    M3D_REQUIRE(in && out && C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0, "maxpool: C and strides must be x4");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(imin(cdiv(total, 256), 8192)), dim3(256), 0, (hipStream_t)stream, in,
                       in_cs, out, out_cs, N, H, W, C / 4);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------
// Depthwise ConvTranspose2d(kernel 4, stride 2, padding 1, groups=C, no bias) + skip add
// (model/pose_dla_dcn.py:536-538,550-552).  out[y][x] += in[iy][ix] * w[ky][kx] with
// y = 2*iy - 1 + ky: each output pixel has exactly 2x2 contributing inputs.
// wgt layout [4][4][C] (ky, kx, c).
template <typename IT>
__global__ void upsample2x_add_kernel(const float *__restrict__ in, int in_cs, const float *__restrict__ wgt,
                                      const float *__restrict__ skip, int skip_cs, float *__restrict__ out, int out_cs,
                                      int N, int H, int W, int C4)
{
    const int Ho = 2 * H, Wo = 2 * W, C = C4 * 4;
    const IT total = (IT)N * Ho * Wo * C4;
    for (IT i = (IT)blockIdx.x * (IT)blockDim.x + threadIdx.x; i < total;
         i += (IT)gridDim.x * (IT)blockDim.x) {
        const int c4 = (int)(i % C4);
        IT p = i / C4;
        const int x = (int)(p % Wo);
        p /= Wo;
        const int y = (int)(p % Ho), n = (int)(p / Ho);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // ky = y + 1 - 2*iy in [0,3]  ->  iy in { (y+1)>>1, ((y+1)>>1) - 1 }
        const int iy_hi = (y + 1) >> 1, ix_hi = (x + 1) >> 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int iy = iy_hi - a, ky = y + 1 - 2 * iy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ix = ix_hi - b, kx = x + 1 - 2 * ix;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(in + ((size_t)(n * H + iy) * W + ix) * in_cs + c4 * 4);
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wgt + (ky * 4 + kx) * C + c4 * 4);
                acc += v * w;
            }
        }
        const size_t o = (size_t)(n * Ho + y) * Wo + x;
        if (skip) acc += *reinterpret_cast<const f32x4 *>(skip + o * skip_cs + c4 * 4);
        *reinterpret_cast<f32x4 *>(out + o * out_cs + c4 * 4) = acc;
    }
}

extern "C" int m3d_upsample2x_add(const float *in, int in_cs, const float *wgt, const float *skip, int skip_cs,
                                  float *out, int out_cs, int N, int H, int W, int C, m3d_stream_t stream)
{
    M3D_REQUIRE(in && wgt && out && C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0 && (!skip || skip_cs % 4 == 0),
                "upsample2x_add: C and strides must be x4");
    // (the row / segment form of the bf16 twin -- weights in registers, output pairs over a sliding window, csrc/bf16_kernels.hip -- was
    // ported and measured neutral at bs 8: 6.002 vs 6.005 ms per step; not kept)
    const long long total = (long long)N * 4 * H * W * (C / 4);
    // the int form's grid-stride increment (at most 8192 x 256) must not carry the index past 2^31 on its last step
    if (total < (1ll << 31) - 8192ll * 256)
        hipLaunchKernelGGL(upsample2x_add_kernel<int>, dim3(imin(cdiv(total, 256), 8192)), dim3(256), 0, (hipStream_t)stream, in,
                           in_cs, wgt, skip, skip_cs, out, out_cs, N, H, W, C / 4);      // (32-bit index arithmetic: no 64-bit divisions)
    else
        hipLaunchKernelGGL(upsample2x_add_kernel<long long>, dim3(imin(cdiv(total, 256), 8192)), dim3(256), 0, (hipStream_t)stream, in,
                       in_cs, wgt, skip, skip_cs, out, out_cs, N, H, W, C / 4);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---- cache warm-up ---------------------------------------------------------------------------------------------------------------
// One dword of every 128-byte line of [p, p + bytes): brings a weight tensor from HBM into the memory-side cache (and the L2 of the
// XCD that touched the line) just before a kernel whose operand lookahead covers an L2 / MALL round trip but not an HBM one
// (csrc/wino44_conv.hip: B fragments four transform positions = ~1400 cycles ahead).
__global__ __launch_bounds__(256) void cache_touch_kernel(const char *p, long long lines)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < lines) {
        unsigned v;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p + i * 128) : "memory");
    }
}

// Fed-input upload as a KERNEL (round 4): the frames of the NEXT batch sit in pinned host memory; `*src_slot` (an 8-byte word,
// itself in pinned host memory, written by the host before the graph is replayed) names them.  a few workgroups stream them over
// PCIe with 16-byte loads -- exactly once, no halo re-reads -- into the device buffer the next replay's stem reads.  Inside the
// captured graph on the side branch it needs no copy engine, no second stream and no events: on this platform an asynchronous
// hipMemcpyAsync next to the graph cost MORE step time than a serial one (tools/feed_probe.py).
__global__ __launch_bounds__(256) void upload_indirect_kernel(const void *const *src_slot, u32x4 *__restrict__ dst, long long bytes)
{
    const u32x4 *src = reinterpret_cast<const u32x4 *>(__builtin_nontemporal_load(src_slot));
    if (!src) return;
    const long long n16 = bytes >> 4, stride = (long long)gridDim.x * 256;
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 15))          // the last < 16 bytes
        reinterpret_cast<unsigned char *>(dst)[n16 * 16 + threadIdx.x] = reinterpret_cast<const unsigned char *>(src)[n16 * 16 + threadIdx.x];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * stride) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * stride < n16) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * stride < n16) dst[i + k * stride] = v[k];
    }
}

extern "C" int m3d_upload_indirect(const void *const *src_slot, void *dst, long long bytes, m3d_stream_t stream)
{
    M3D_REQUIRE(src_slot && dst && bytes >= 0 && ((uintptr_t)dst & 15) == 0, "upload_indirect: null pointer or misaligned destination");
    if (bytes == 0) return M3D_OK;
    // 8 workgroups (measured, tools/feed_probe.py, step = 6.07 ms resident): 1 -> 6.98 ms (the upload outlasts the forward),
    // 2 -> 6.39, 4 -> 6.19, 16 -> 6.19, 64 -> 6.24: a few workgroups keep enough loads in flight for PCIe and leave the CUs alone
    static const int wgs = []() { const char *e = getenv("M3D_UPLOAD_WGS"); const int v = e ? atoi(e) : 8; return v > 0 ? v : 8; }();
    hipLaunchKernelGGL(upload_indirect_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, src_slot, (u32x4 *)dst, bytes);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_cache_touch(const void *p, long long bytes, m3d_stream_t stream)
{
    M3D_REQUIRE(p && bytes >= 0, "cache_touch: null pointer");
    const long long lines = bytes / 128;
    if (lines == 0) return M3D_OK;
    M3D_REQUIRE(lines < (1ll << 31) * 256, "cache_touch: range too large");
    hipLaunchKernelGGL(cache_touch_kernel, dim3((unsigned)cdiv(lines, 256)), dim3(256), 0, (hipStream_t)stream, (const char *)p, lines);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
