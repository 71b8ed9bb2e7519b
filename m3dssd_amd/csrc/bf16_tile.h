// Pieces shared by the bf16 MFMA kernels (bf16_conv.hip, bf16_anab.hip): argument block, packing helpers and the
// accumulator epilogue (folded BatchNorm / bias, residual, LeakyReLU / sigmoid, bf16 / fp32 stores).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Bf16Args {
    const void *in;            // bf16 NHWC view, pixel stride in_cs elements
    const void *wgt;           // bf16 [Cout_pad][Kpad] (K = (i*kw + j)*Cin + c), zero padded
    void *out;
    const float *scale, *shift;
    const void *res;           // bf16 NHWC residual view or null
    const float *om;           // fp32 NHWC [.., 3*kh*kw] offsets / masks (deformable) or null
    long long wgt_img_stride;  // elements between per-image weight sets (0 = shared)
    long long out_img_stride;  // planar mode: floats between images
    long long in_goff, wgt_goff, out_goff;   // per-group element offsets (grouped launch: blockIdx.y = group)
    unsigned in_bytes, wgt_bytes, res_bytes;
    int in_cs, N, H, W, Cin, log2Cin;
    int Cout, Cout_pad, K, KT, kh, kw, stride, pad;
    int Ho, Wo, HoWo, M;
    int out_cs, res_cs, om_cs, ss_goff;
    int out_mode;              // 0 bf16 NHWC, 1 fp32 NHWC, 2 fp32 planar [img][c][HoWo]
    int res_mode, act, sigmoid_from;
    int tiles_m, tiles_n;
    int lane_perm;             // halo kernel: 1 = bank-conflict-free lane -> pixel map (0 = identity, for A/B)
    int uniform_k;             // Cin % 64 == 0 (or 1x1 with K % 64 == 0): every K-step lies in one tap, channel offset is wave-uniform
#ifdef BF16_TRACE
    long long *trace;
#endif
};

__device__ __forceinline__ u32x4 buf_load_u32x4(__amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 unpack_bf16(unsigned u)
{
    f32x2 r;
    r[0] = __uint_as_float(u << 16);
    r[1] = __uint_as_float(u & 0xFFFF0000u);
    return r;
}

// Epilogue shared by the conv kernels: D[row = channel][col = pixel]; lane = pixel l31, channels 8g + 4*lh + (0..3) per
// register group g.  Folded BatchNorm / bias, residual, LeakyReLU / sigmoid in fp32; bf16 NHWC (16-byte stores after a
// v_permlane32_swap of the half-waves), fp32 NHWC or fp32 planar output.
template <int TN, int TM>
__device__ __forceinline__ void conv_epilogue(const Bf16Args &a, f32x16 (&acc)[TN][TM], const int (&mpix)[TM], int n0, int wn, int lh,
                                              int grp)
{
    const float *scale = a.scale ? a.scale + grp * a.ss_goff : nullptr;
    const float *shift = a.shift ? a.shift + grp * a.ss_goff : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mpix[i];                                   // this lane's output pixel (linear n*Ho*Wo index) or -1
        const bool mok = m >= 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cb = n0 + wn + j * 32 + 4 * lh;          // + 8g
            f32x4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = cb + 8 * g;
                f32x4 x = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                // scale / shift hold Cout floats: the last vector of a channel count that is not a multiple of 4 is read by element
                if (c0 + 3 < a.Cout) {
                    if (scale) sc = *reinterpret_cast<const f32x4 *>(scale + c0);
                    if (shift) sh = *reinterpret_cast<const f32x4 *>(shift + c0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c0 + e < a.Cout) {
                            if (scale) sc[e] = scale[c0 + e];
                            if (shift) sh[e] = shift[c0 + e];
                        }
                }
                f32x4 rs = {0.f, 0.f, 0.f, 0.f};
                if (a.res && mok && c0 < a.Cout) {
                    const u32x2 rr = *reinterpret_cast<const u32x2 *>((const __bf16 *)a.res + (size_t)m * a.res_cs + c0);
                    const f32x2 r01 = unpack_bf16(rr[0]), r23 = unpack_bf16(rr[1]);
                    rs[0] = r01[0]; rs[1] = r01[1]; rs[2] = r23[0]; rs[3] = r23[1];
                }
                if (a.res_mode == 1) x = (x + rs) * sc + sh;
                else x = x * sc + sh + rs;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (a.sigmoid_from >= 0 && c0 + e >= a.sigmoid_from) x[e] = sigmoidf_(x[e]);
                    else if (a.act) x[e] = leaky(x[e]);
                }
                v[g] = x;
            }
            if (a.out_mode == 0) {
                // bf16 NHWC: pack 4 channels per group, then v_permlane32_swap pairs lane (pixel, lh = 0) with (pixel, lh = 1)
                // so that each lane ends up with 8 CONSECUTIVE channels of its pixel -> 16-byte stores
                unsigned pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pk[g][0] = pack_bf16(v[g][0], v[g][1]);
                    pk[g][1] = pack_bf16(v[g][2], v[g][3]);
                }
                // before: lanes lh=0 hold channels 8g + 0..3, lanes lh=1 hold 8g + 4..7.  Swap the upper half of group g (g even)
                // with the lower half of group g + 1: lh=0 lanes then hold channels 8g + 0..7 in (pk[g], pk[g+1]); lh=1 lanes hold
                // 8(g+1) + 0..7.
#pragma unroll
                for (int g = 0; g < 4; g += 2)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(pk[g][e], pk[g + 1][e], false, false);
                        pk[g][e] = r[0]; pk[g + 1][e] = r[1];
                    }
                if (mok) {
                    __bf16 *op = (__bf16 *)a.out + grp * a.out_goff + (size_t)m * a.out_cs;
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        const int c0 = n0 + wn + j * 32 + 8 * (g + lh);
                        const u32x4 o = {pk[g][0], pk[g][1], pk[g + 1][0], pk[g + 1][1]};
                        if (c0 + 7 < a.Cout) *reinterpret_cast<u32x4 *>(op + c0) = o;
                        else {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (c0 + e < a.Cout)
                                    reinterpret_cast<unsigned short *>(op)[c0 + e] = (unsigned short)(o[e >> 1] >> ((e & 1) * 16));
                        }
                    }
                }
            } else if (a.out_mode == 1) {
                if (mok) {
                    float *op = (float *)a.out + grp * a.out_goff + (size_t)m * a.out_cs;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = cb + 8 * g;
                        if (c0 + 3 < a.Cout) *reinterpret_cast<f32x4 *>(op + c0) = v[g];
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c0 + e < a.Cout) op[c0 + e] = v[g][e];
                        }
                    }
                }
            } else {
                if (mok) {
                    const int img = m / a.HoWo, p = m - img * a.HoWo;
                    float *op = (float *)a.out + grp * a.out_goff + (size_t)img * a.out_img_stride + p;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int c = cb + 8 * g + e;
                            if (c < a.Cout) op[(size_t)c * a.HoWo] = v[g][e];
                        }
                }
            }
        }
    }
}

