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
    const unsigned *gate;      // deformable implicit-GEMM kernel as the FALLBACK of the LDS-patch kernel (bf16_dcn_patch.hip): one word per
    int gate_th, gate_tpx, gate_tpy;   // patch tile (gate_th x 16 pixels, gate_tpx x gate_tpy tiles per image), non-zero = "my window did
                               // not fit": this kernel works on the 128-pixel tiles that touch such a patch tile and on nothing else
#ifdef BF16_TRACE
    long long *trace;
#endif
};

// ---- largest |offset| of a patch tile (bf16_dcn_patch.hip) ---------------------------------------------------------------------
// Returns the radius R = ceil(max |offset|) the sampling window of the workgroup's pixels has to cover beyond the 3x3 taps, uniform
// over the workgroup; a NaN / inf offset gives 2^20 (never fits).  `scratch`: >= 32 bytes of LDS, free again after the NEXT barrier
// of the caller.  nt = threads of the workgroup (>= 256).
// mx = largest |offset| bit pattern among this thread's pixels (0 for none).  -> the radius shared by the workgroup.
__device__ __forceinline__ int dcn_tile_radius(unsigned mx, unsigned *scratch, int tid, int nt)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, s, 64));
    if ((tid & 63) == 0) scratch[tid >> 6] = mx;
    __syncthreads();
    unsigned m = 0u;
    for (int w = 0; w < nt / 64; ++w) m = max(m, scratch[w]);
    return m >= 0x7f800000u ? (1 << 20) : (int)ceilf(__uint_as_float(m));
}
long long dcn_patch_ws_bytes(const m3d_conv_bf16_desc *d, int variant);
int dcn_patch_variant(const m3d_conv_bf16_desc *d);
struct Bf16Args;
int launch_dcn_patch(const Bf16Args &a0, const m3d_conv_bf16_desc *d, int variant, hipStream_t st);
int conv_wide_applicable(const m3d_conv_bf16_desc *d);                                  // bf16_conv_wide.hip
int launch_conv_wide(const Bf16Args &a, const m3d_conv_bf16_desc *d, hipStream_t st);
int conv_c64_applicable(const m3d_conv_bf16_desc *d);                                   // bf16_conv_c64.hip
int launch_conv_c64(const Bf16Args &a, const m3d_conv_bf16_desc *d, hipStream_t st);
int dcn1x1_applicable(const m3d_conv_bf16_desc *d);                                     // bf16_dcn1x1.hip
int launch_dcn1x1(const Bf16Args &a, const m3d_conv_bf16_desc *d, hipStream_t st);

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
// `ssl`: LDS copy of the tile's affine parameters, [scale of channels n0 .. n0 + BN) | shift ...] with `ssl_bn` = BN (1 / 0 past
// Cout), or null: read from global memory.  The in-kernel timeline of a 1x1 layer (tools/bf16_conv_trace.py) showed the epilogue
// taking as long as the whole K loop (16500 of 32800 cycles): the per-group scale / shift loads and the residual loads sat between
// stores they may alias, one memory round trip each.  Now the residuals of ALL the lane's groups are fetched before the first store
// and the affine parameters come from LDS.
template <int TN, int TM>
__device__ __forceinline__ void conv_epilogue(const Bf16Args &a, f32x16 (&acc)[TN][TM], const int (&mpix)[TM], int n0, int wn, int lh,
                                              int grp, const float *ssl = nullptr, int ssl_bn = 0, unsigned char *otile = nullptr,
                                              const int *lrow = nullptr)
{
    const float *scale = a.scale ? a.scale + grp * a.ss_goff : nullptr;
    const float *shift = a.shift ? a.shift + grp * a.ss_goff : nullptr;
    const bool has_sig = a.sigmoid_from >= 0;
    const float slope = a.act ? M3D_LEAKY_SLOPE : 1.f;
    u32x2 rres[TM][TN][4];
    if (a.res) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = n0 + wn + j * 32 + 4 * lh + 8 * g;
                    rres[i][j][g] = u32x2{0u, 0u};
                    if (mpix[i] >= 0 && c0 < a.Cout)
                        rres[i][j][g] = *reinterpret_cast<const u32x2 *>((const __bf16 *)a.res + (size_t)mpix[i] * a.res_cs + c0);
                }
    }
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
                if (ssl) {
                    sc = *reinterpret_cast<const f32x4 *>(ssl + (c0 - n0));
                    sh = *reinterpret_cast<const f32x4 *>(ssl + ssl_bn + (c0 - n0));
                } else if (c0 + 3 < a.Cout) {
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
                if (a.res) {
                    const f32x2 r01 = unpack_bf16(rres[i][j][g][0]), r23 = unpack_bf16(rres[i][j][g][1]);
                    rs[0] = r01[0]; rs[1] = r01[1]; rs[2] = r23[0]; rs[3] = r23[1];
                }
                if (a.res_mode == 1) x = (x + rs) * sc + sh;
                else x = x * sc + sh + rs;
                // LeakyReLU without a branch (slope 1 = identity); the per-element sigmoid test only in launches that have one.
                // Written per element with both tests inside, this compiled to ~1500 branches per kernel and the epilogue took
                // as long as the K loop of a 1x1 layer (11000 of 30000 cycles, tools/bf16_conv_trace.py).
                if (!has_sig || c0 + 3 < a.sigmoid_from) {      // groups below the first sigmoid channel never evaluate an exponential
                    x = __builtin_elementwise_max(x, x * slope);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = c0 + e >= a.sigmoid_from ? sigmoidf_(x[e]) : fmaxf(x[e], x[e] * slope);
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
                if (otile) {
                    // bf16 tile [row = pixel of the workgroup tile][ssl_bn channels] in LDS, 16-byte chunks XOR-swizzled by the row;
                    // store_otile() writes it out in whole 128-byte lines (the direct form scatters 16 bytes per lane over 64 rows)
                    const int row = lrow[i], rb = ssl_bn * 2, sm = (ssl_bn >= 64 ? 7 : ssl_bn / 8 - 1);
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        const int chunk = (wn + j * 32 + 8 * (g + lh)) >> 3;
                        *reinterpret_cast<u32x4 *>(otile + row * rb + ((chunk ^ (row & sm)) << 4)) =
                            u32x4{pk[g][0], pk[g][1], pk[g + 1][0], pk[g + 1][1]};
                    }
                } else if (mok) {
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

// The common case of conv_epilogue -- bf16 NHWC output through the LDS tile, no sigmoid channels -- as straight-line code: the
// general routine carries every output mode, the sigmoid test and the ragged-Cout paths per element, and its compiled form spent
// 6000-11000 cycles per tile on a path that needs ~100 VALU instructions per 32 x 32 block (in-kernel timeline of a 1x1 layer).
// Same arithmetic, same order: (acc [+ res]) * scale + shift [+ res], LeakyReLU as max(x, slope * x), one rounding to bf16.
template <int TN, int TM>
__device__ __forceinline__ void conv_epilogue_fast(const Bf16Args &a, f32x16 (&acc)[TN][TM], const int (&mpix)[TM], const int (&lrow)[TM],
                                                   int n0, int wn, int lh, const float *ssl, int bn, unsigned char *otile)
{
    const float slope = a.act ? M3D_LEAKY_SLOPE : 1.f;
    const bool hres = a.res != nullptr, rm1 = a.res_mode == 1;
    u32x2 rres[TM][TN][4];
    if (hres) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = n0 + wn + j * 32 + 4 * lh + 8 * g;
                    const bool ok = mpix[i] >= 0 && c0 < a.Cout;            // masked lanes read the first residual element: no branch
                    const u32x2 r = *reinterpret_cast<const u32x2 *>((const __bf16 *)a.res + (ok ? (size_t)mpix[i] * a.res_cs + c0 : 0));
                    rres[i][j][g] = ok ? r : u32x2{0u, 0u};
                }
    }
    const int rb = bn * 2, sm = bn >= 64 ? 7 : bn / 8 - 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = lrow[i];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn + j * 32 + 4 * lh + 8 * g;                 // channel inside the tile
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(ssl + cl), sh = *reinterpret_cast<const f32x4 *>(ssl + bn + cl);
                f32x4 x = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                if (hres) {
                    const f32x2 r01 = unpack_bf16(rres[i][j][g][0]), r23 = unpack_bf16(rres[i][j][g][1]);
                    const f32x4 rs = {r01[0], r01[1], r23[0], r23[1]};
                    if (rm1) x = (x + rs) * sc + sh;
                    else x = x * sc + sh + rs;
                } else {
                    x = x * sc + sh;
                }
                x = __builtin_elementwise_max(x, x * slope);
                pk[g][0] = pack_bf16(x[0], x[1]);
                pk[g][1] = pack_bf16(x[2], x[3]);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const auto r = __builtin_amdgcn_permlane32_swap(pk[g][e], pk[g + 1][e], false, false);
                    pk[g][e] = r[0]; pk[g + 1][e] = r[1];
                }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const int chunk = (wn + j * 32 + 8 * (g + lh)) >> 3;
                *reinterpret_cast<u32x4 *>(otile + row * rb + ((chunk ^ (row & sm)) << 4)) = u32x4{pk[g][0], pk[g][1], pk[g + 1][0], pk[g + 1][1]};
            }
        }
    }
}

// Second half of the bf16 NHWC epilogue: the workgroup's [BM][BN] bf16 tile in LDS (written by conv_epilogue with `otile`) goes to
// global memory with 8 consecutive lanes per pixel row segment (128 bytes), like the staging loads.  rowpix(row) -> linear output
// pixel of tile row `row`, or -1.
template <int BN, int BM, int NT, typename F>
__device__ __forceinline__ void store_otile(const Bf16Args &a, const unsigned char *otile, int n0, int grp, int tid, F rowpix)
{
    constexpr int CPR = BN / 8, SM = BN >= 64 ? 7 : BN / 8 - 1;        // 16-byte chunks per row
    constexpr int RPP = NT / CPR;                                       // rows per pass
    const int ch = tid % CPR, r0 = tid / CPR;
    const int c0 = n0 + ch * 8;
    if (c0 >= a.Cout) return;
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
        const int row = r0 + p * RPP;
        const int m = rowpix(row);
        if (m < 0) continue;
        const u32x4 o = *reinterpret_cast<const u32x4 *>(otile + row * (BN * 2) + ((ch ^ (row & SM)) << 4));
        __bf16 *op = (__bf16 *)a.out + grp * a.out_goff + (size_t)m * a.out_cs;
        if (c0 + 7 < a.Cout) *reinterpret_cast<u32x4 *>(op + c0) = o;
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c0 + e < a.Cout) reinterpret_cast<unsigned short *>(op)[c0 + e] = (unsigned short)(o[e >> 1] >> ((e & 1) * 16));
        }
    }
}
