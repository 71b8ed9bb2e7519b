// Fused front end of the bf16 path: DLA base_layer (7x7, 3 -> 16) -> level0 (3x3, 16 -> 16) -> level1 (3x3 stride 2,
// 16 -> 32), each with its folded BatchNorm + LeakyReLU (model/pose_dla_dcn.py:336-345,391-397), in ONE launch that reads
// the image once and writes only the 32-channel half-resolution map.  Unfused these three layers move 16-channel
// full-resolution tensors through HBM four times (at bs = 64: 4 GB) and, at 16 channels, cannot feed a 128-wide GEMM tile.
//
//   Workgroup (256 threads, 4 waves) = one 8 x 16 tile of level1 outputs.  Working backwards it needs 17 x 33 level0 pixels,
//   19 x 35 stem pixels and a 25 x 41 image patch: all three live in LDS (68 KB: TWO workgroups per CU, so that the load phase
//   of one runs under the compute phases of the other -- the 8 x 32 tile of the first version filled the CU alone and ran
//   1.35 ms against 1.25; bf16; pixel stride 48 bytes for the 16-channel
//   tiles so that 16 consecutive pixels hit 16 different 16-byte bank groups).  Each stage is an implicit GEMM on
//   v_mfma_f32_16x16x32_bf16 with rows = output channels (A operand = weights, held in registers for the whole stage) and
//   columns = 16 consecutive pixels of the flattened region (B operand gathered from the LDS tile):
//     stem   : K = 7 tap rows x 32 (7 taps x 4 channel slots, RGB + 1 zero; the 8th tap slot has zero weights) -> 7 MFMAs / 16 px
//     level0 : K = 9 taps x 16 channels padded to 160                                                            -> 5 MFMAs / 16 px
//     level1 : same K, 32 output channels                                                                        -> 10 MFMAs / 16 px
//   Pixels of an intermediate region that fall outside the image are written as ZERO (they are the zero padding of the next
//   convolution, not the stem of a padded image).  The test-time Preprocess of the reference (uint8 BGR frames, /255, -mean,
//   /std, BGR->RGB; lib/augmentations.py:44-57,472-501) can be applied in the image load, as in m3d_stem_conv7x7_bf16.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define FE_T1H 8
#define FE_T1W 16
#define FE_L0H (2 * FE_T1H + 1)     // 17
#define FE_L0W (2 * FE_T1W + 1)     // 65
#define FE_S0H (FE_L0H + 2)         // 19
#define FE_S0W (FE_L0W + 2)         // 67
#define FE_IMH (FE_S0H + 6)         // 25
#define FE_IMW (FE_S0W + 6)         // 73
#define FE_IMS ((FE_IMW + 3 + 3) / 4 * 4)   // image tile row stride in pixels (8 bytes each): room for the 8-wide tap window
#define FE_NT 256                   // threads per workgroup
#define FE_NW (FE_NT / 64)
#define FE_PS 48                    // bytes per pixel of the level0 tile (read with pixel stride 2 by level1: 16 lanes x 16 B cover the 64 banks once)
#define FE_PS0 32                   // bytes per pixel of the stem tile (read with pixel stride 1 by level0: dense rows are the conflict-free ones;
                                    // at 48 the two 16-byte halves of neighbouring pixels collide 2-way: SQ_LDS_BANK_CONFLICT 136 M cycles per launch)

struct FrontArgs {
    const void *img;                // fp32 [N][3][H][W] or uint8 [N][img_h][img_w][3] (BGR)
    const void *w_stem, *w_l0, *w_l1;   // bf16 [16][224], [16][160], [32][160]
    const float *s_stem, *t_stem, *s_l0, *t_l0, *s_l1, *t_l1;
    void *out;                      // bf16 [N][H/2][W/2][out_cs]
    float mean[3], stds[3];
    int is_u8, img_h, img_w;
    int H, W, out_cs, tiles_x, tiles_y;
};

__device__ __forceinline__ unsigned fpack(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// Epilogue of one MFMA column group: 4 channels of a pixel -> folded BatchNorm + LeakyReLU on the packed-fp32 pipe (2 v_pk_fma +
// 2 v_pk_mul + 4 v_max instead of 12 scalar ops), two bf16 pairs, zeroed through `msk` (0 / ~0) where the pixel lies outside
// the image.  The kernel is VALU-bound (SQ_INSTS_VALU 320 M per launch, MFMA 17 % busy): instructions here are time.
__device__ __forceinline__ u32x2 fe_epilogue(const f32x4 acc, const f32x2 sc01, const f32x2 sc23, const f32x2 sh01, const f32x2 sh23,
                                             unsigned msk)
{
    const f32x2 sl = {M3D_LEAKY_SLOPE, M3D_LEAKY_SLOPE};
    const f32x2 y01 = __builtin_elementwise_fma(f32x2{acc[0], acc[1]}, sc01, sh01);
    const f32x2 y23 = __builtin_elementwise_fma(f32x2{acc[2], acc[3]}, sc23, sh23);
    const f32x2 z01 = y01 * sl, z23 = y23 * sl;
    return u32x2{fpack(fmaxf(y01[0], z01[0]), fmaxf(y01[1], z01[1])) & msk, fpack(fmaxf(y23[0], z23[0]), fmaxf(y23[1], z23[1])) & msk};
}

// amdgpu_waves_per_eu(2): two workgroups per CU is what the LDS tiles allow anyway, and with at most 256 registers per wave hipcc
// keeps the MFMA accumulators in VGPRs (no v_accvgpr_read before every epilogue).
__global__ __launch_bounds__(FE_NT) __attribute__((amdgpu_waves_per_eu(2))) void bf16_frontend_kernel(const FrontArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[FE_IMH * FE_IMS * 8 + FE_S0H * FE_S0W * FE_PS0 + FE_L0H * FE_L0W * FE_PS + 64];
    unsigned char *imt = lds;                                   // [25][76][4 bf16]
    unsigned char *s0t = lds + FE_IMH * FE_IMS * 8;              // [19*67][48 B]
    unsigned char *l0t = s0t + FE_S0H * FE_S0W * FE_PS0;          // [17*65][48 B]  (+64: the k-padding read of the last pixel)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int n = blockIdx.z, ty = blockIdx.y, tx = blockIdx.x;
    const int y1 = ty * FE_T1H, x1 = tx * FE_T1W;               // level1 tile origin (half resolution)
    const int Y0 = 2 * y1 - 1, X0 = 2 * x1 - 1;                 // level0 region origin (full resolution)
    const int YS = Y0 - 1, XS = X0 - 1;                         // stem region origin
    const int YI = YS - 3, XI = XS - 3;                         // image patch origin
    const int H = a.H, W = a.W;

    // ---- image patch -> LDS [y][x][R, G, B, 0] bf16; outside the image: 0 (the stem's zero padding) ---------------------------
    {
        const float *im = static_cast<const float *>(a.img) + (size_t)n * 3 * H * W;
        const unsigned char *frame = static_cast<const unsigned char *>(a.img) + (size_t)n * a.img_h * a.img_w * 3;
        // all loads of the thread are issued before the first conversion: the workgroup is alone on its CU (129 KB of LDS), so
        // a load -> convert -> store loop would pay one memory round trip per iteration with nothing to hide it
        constexpr int NI = (FE_IMH * FE_IMS + FE_NT - 1) / FE_NT;
        float v[NI][3];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * FE_NT;
            const int r = i / FE_IMS, q = i - r * FE_IMS;
            const int h = YI + r, w = XI + q;
            v[it][0] = v[it][1] = v[it][2] = 0.f;
            if (i < FE_IMH * FE_IMS && q < FE_IMW && h >= 0 && h < H && w >= 0 && w < W) {
                if (a.is_u8) {
                    if (h < a.img_h && w < a.img_w) {
                        const unsigned char *px = frame + ((size_t)h * a.img_w + w) * 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) v[it][c] = (float)px[2 - c];   // plane c of the RGB tensor = BGR channel 2 - c
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[it][c] = im[((size_t)c * H + h) * W + w];
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int i = tid + it * FE_NT;
            const int r = i / FE_IMS, q = i - r * FE_IMS;
            const int h = YI + r, w = XI + q;
            if (a.is_u8 && i < FE_IMH * FE_IMS && q < FE_IMW && h >= 0 && h < H && w >= 0 && w < W) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int cb = 2 - c;
                    float x = v[it][c] / 255.0f;
                    x = x - a.mean[cb];
                    v[it][c] = x / a.stds[cb];
                }
            }
            if (i < FE_IMH * FE_IMS) *reinterpret_cast<u32x2 *>(imt + (size_t)i * 8) = u32x2{fpack(v[it][0], v[it][1]), fpack(v[it][2], 0.f)};
        }
    }
    __syncthreads();

    // ---- stem: 19 x 67 pixels, K = 7 tap rows x 32 ------------------------------------------------------------------------
    {
        bf16x8 wf[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) wf[i] = *reinterpret_cast<const bf16x8 *>((const __bf16 *)a.w_stem + l15 * 224 + i * 32 + kg * 8);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.s_stem + 4 * kg), sh = *reinterpret_cast<const f32x4 *>(a.t_stem + 4 * kg);
        const f32x2 sc01 = {sc[0], sc[1]}, sc23 = {sc[2], sc[3]}, sh01 = {sh[0], sh[1]}, sh23 = {sh[2], sh[3]};
        constexpr int NP = FE_S0H * FE_S0W, NG = (NP + 15) / 16;
        // pixel p = g*16 + l15 of the flattened region, g = wave, wave + 4, ...: (row, column) advance by 64 pixels per iteration
        // without a division (the groups past NP read one tile row further -- still inside `lds` -- and are not stored)
        constexpr int QS = (16 * FE_NW) / FE_S0W, RS = (16 * FE_NW) % FE_S0W;
        int p = wave * 16 + l15;
        int ry = p / FE_S0W, rx = p - ry * FE_S0W;
        for (int g = wave; g < NG; g += FE_NW) {
            // tap row i, taps j = 2*kg, 2*kg + 1 of pixel (ry, rx): image tile pixels (ry + i, rx + 2*kg + {0, 1})
            const unsigned char *src = imt + ((size_t)ry * FE_IMS + rx + 2 * kg) * 8;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const u32x2 lo = *reinterpret_cast<const u32x2 *>(src + i * FE_IMS * 8);
                const u32x2 hi = *reinterpret_cast<const u32x2 *>(src + i * FE_IMS * 8 + 8);
                const u32x4 b = {lo[0], lo[1], hi[0], hi[1]};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
            }
            const int h = YS + ry, w = XS + rx;
            const unsigned msk = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ~0u : 0u;
            if (p < NP) *reinterpret_cast<u32x2 *>(s0t + (size_t)p * FE_PS0 + kg * 8) = fe_epilogue(acc, sc01, sc23, sh01, sh23, msk);
            p += 16 * FE_NW;
            rx += RS; ry += QS;
            if (rx >= FE_S0W) { rx -= FE_S0W; ry += 1; }
        }
    }
    __syncthreads();

    // ---- level0: 17 x 65 pixels, K = 9 taps x 16 channels (+ 16 zero) -------------------------------------------------------
    {
        bf16x8 wf[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) wf[t] = *reinterpret_cast<const bf16x8 *>((const __bf16 *)a.w_l0 + l15 * 160 + t * 32 + kg * 8);
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.s_l0 + 4 * kg), sh = *reinterpret_cast<const f32x4 *>(a.t_l0 + 4 * kg);
        const f32x2 sc01 = {sc[0], sc[1]}, sc23 = {sc[2], sc[3]}, sh01 = {sh[0], sh[1]}, sh23 = {sh[2], sh[3]};
        constexpr int NP = FE_L0H * FE_L0W, NG = (NP + 15) / 16;
        // k-group kg of K-step t reads tap 2t + (kg >> 1), channel half kg & 1; tap 9 (t = 4, kg >= 2) has zero weights: it
        // re-reads tap 8 so that the operand stays finite
        int toff[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            int tap = 2 * t + (kg >> 1);
            tap = tap > 8 ? 8 : tap;
            toff[t] = ((tap / 3) * FE_S0W + (tap % 3)) * FE_PS0 + (kg & 1) * 16;
        }
        constexpr int QL = (16 * FE_NW) / FE_L0W, RL = (16 * FE_NW) % FE_L0W;
        int p = wave * 16 + l15;
        int ry = p / FE_L0W, rx = p - ry * FE_L0W;
        for (int g = wave; g < NG; g += FE_NW) {
            const unsigned char *src = s0t + ((size_t)ry * FE_S0W + rx) * FE_PS0;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 5; ++t)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], *reinterpret_cast<const bf16x8 *>(src + toff[t]), acc, 0, 0, 0);
            const int h = Y0 + ry, w = X0 + rx;
            const unsigned msk = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ~0u : 0u;
            if (p < NP) *reinterpret_cast<u32x2 *>(l0t + (size_t)p * FE_PS + kg * 8) = fe_epilogue(acc, sc01, sc23, sh01, sh23, msk);
            p += 16 * FE_NW;
            rx += RL; ry += QL;
            if (rx >= FE_L0W) { rx -= FE_L0W; ry += 1; }
        }
    }
    __syncthreads();

    // ---- level1: 8 x 32 outputs, stride 2, 32 channels -----------------------------------------------------------------------
    {
        bf16x8 wf[2][5];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int t = 0; t < 5; ++t)
                wf[hh][t] = *reinterpret_cast<const bf16x8 *>((const __bf16 *)a.w_l1 + (hh * 16 + l15) * 160 + t * 32 + kg * 8);
        int toff[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            int tap = 2 * t + (kg >> 1);
            tap = tap > 8 ? 8 : tap;
            toff[t] = ((tap / 3) * FE_L0W + (tap % 3)) * FE_PS + (kg & 1) * 16;
        }
        const int Ho = H / 2, Wo = W / 2;
        for (int g = wave; g < (FE_T1H * FE_T1W) / 16; g += FE_NW) {
            const int p = g * 16 + l15;
            const int oy = p / FE_T1W, ox = p - oy * FE_T1W;
            const unsigned char *src = l0t + ((size_t)(2 * oy) * FE_L0W + 2 * ox) * FE_PS;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(src + toff[t]);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][t], b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][t], b, acc1, 0, 0, 0);
            }
            const int h = y1 + oy, w = x1 + ox;
            if (h < Ho && w < Wo) {
                __bf16 *op = (__bf16 *)a.out + ((size_t)(n * Ho + h) * Wo + w) * a.out_cs;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 acc = hh ? acc1 : acc0;
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.s_l1 + hh * 16 + 4 * kg);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(a.t_l1 + hh * 16 + 4 * kg);
                    *reinterpret_cast<u32x2 *>(op + hh * 16 + 4 * kg) =
                        fe_epilogue(acc, f32x2{sc[0], sc[1]}, f32x2{sc[2], sc[3]}, f32x2{sh[0], sh[1]}, f32x2{sh[2], sh[3]}, ~0u);
                }
            }
        }
    }
}

extern "C" int m3d_frontend_bf16_forward(const void *img, int is_u8, int img_h, int img_w, const float *mean3, const float *stds3,
                                         const void *w_stem, const float *s_stem, const float *t_stem, const void *w_l0,
                                         const float *s_l0, const float *t_l0, const void *w_l1, const float *s_l1, const float *t_l1,
                                         void *out, int out_cs, int N, int H, int W, m3d_stream_t stream)
{
    M3D_REQUIRE(img && w_stem && w_l0 && w_l1 && s_stem && t_stem && s_l0 && t_l0 && s_l1 && t_l1 && out, "frontend_bf16: null pointer");
    M3D_REQUIRE(H % 2 == 0 && W % 2 == 0 && out_cs % 8 == 0 && out_cs >= 32, "frontend_bf16: even H, W; out_cs %% 8 == 0, >= 32");
    FrontArgs a = {};
    a.img = img; a.w_stem = w_stem; a.w_l0 = w_l0; a.w_l1 = w_l1; a.s_stem = s_stem; a.t_stem = t_stem; a.s_l0 = s_l0; a.t_l0 = t_l0;
    a.s_l1 = s_l1; a.t_l1 = t_l1; a.out = out; a.is_u8 = is_u8 ? 1 : 0; a.H = H; a.W = W; a.out_cs = out_cs;
    if (is_u8) {
        M3D_REQUIRE(mean3 && stds3 && img_h >= 1 && img_w >= 1 && img_h <= H && img_w <= W, "frontend_bf16: frame / normalisation arguments");
        for (int c = 0; c < 3; ++c) {
            M3D_REQUIRE(stds3[c] != 0.f, "frontend_bf16: zero std");
            a.mean[c] = mean3[c];
            a.stds[c] = stds3[c];
        }
        a.img_h = img_h; a.img_w = img_w;
    }
    a.tiles_x = cdiv(W / 2, FE_T1W); a.tiles_y = cdiv(H / 2, FE_T1H);
    hipLaunchKernelGGL(bf16_frontend_kernel, dim3(a.tiles_x, a.tiles_y, N), dim3(FE_NT), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
