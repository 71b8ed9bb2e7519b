// 3x3 / stride 1 / pad 1 convolution 64 -> 64 channels of the bf16 path (DLA level2: model/pose_dla_dcn.py:107-121 at 96x320, three
// launches per step) as PERSISTENT workgroups with the whole weight tensor resident in LDS.
//
// Why (round 6): on the halo-tile kernel (bf16_conv.hip, 8 x 32 pixels x 64 channels per 8-wave workgroup) these launches took
// 0.19-0.23 ms each at bs 64 = 0.26 of the matrix peak AND 0.38 of the HBM roof.  With Cin = 64 the K loop of a tile is ONE channel
// chunk: 72 MFMAs per wave (2 304 cycles) behind a halo load that nothing covers (HBM round trip >= 4 000 cycles), nine weight
// stagings with a barrier each, and an epilogue -- a latency chain per tile, the same shape the DCNv2 ablations of this round
// showed.  The op is HBM-bound: per 8 x 32 tile 43.5 KB of halo + 16-32 KB of output / residual against 4 608
// MFMA cycles per SIMD pair.  Here:
//   * one workgroup per CU (8 waves, 152 KB of LDS) walks ~30 tiles; the 9 x 64 x 64 weights (72 KB) are staged ONCE;
//   * the halo of tile t + 1 is in flight (6 x 16 bytes per thread) while tile t computes, the residual of tile t + 1 while tile t's
//     output leaves; two barriers per tile (patch free / output tile ready), none per tap;
//   * tiles are walked in an XCD-contiguous order (the 8 x 32 tiles of one image region share their halo rows in one L2).
// Every 64 -> 64 3x3 layer whose map tiles 8 x 32 runs here whatever the batch size (batch invariance: one kernel, one fp32
// summation order -- (tap column, K-step, tap row), not the halo-tile kernel's (tap, K-step)).
#include "bf16_tile.h"

#define C6_PS 144                         // bytes per halo pixel (128 + 16 pad), as the halo-tile kernel
#define C6_TH 8
#define C6_TW 32
#define C6_HW (C6_TW + 2)
#define C6_HPIX ((C6_TH + 2) * C6_HW)     // 340
#define C6_NT 512
#define C6_WALL 0                         // [9 taps][64 rows][128 B], 16-byte pieces XOR (row >> 1) & 7
#define C6_HS (9 * 64 * 128)              // halo patch
#define C6_OT (C6_HS + C6_HPIX * C6_PS)   // output tile [256 pixels][128 B]
#define C6_SS (C6_OT + 256 * 128)         // scale | shift, 64 floats each
#define C6_LDS (C6_SS + 2 * 64 * 4)

template <bool HAS_RES>
__global__ __launch_bounds__(C6_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void bf16_conv3x3_c64_kernel(const Bf16Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[C6_LDS];
    unsigned char *Hs = lds + C6_HS, *Ot = lds + C6_OT;
    float *ssl = reinterpret_cast<float *>(lds + C6_SS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 32;          // 4 x 2 waves: 64 pixels x 32 channels each
    const int l31 = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rwgt = make_rsrc(a.wgt, a.wgt_bytes);

    // ---- tiles of this workgroup: XCD x (= blockIdx & 7) owns the contiguous range [x * per, (x + 1) * per), its workgroups
    // take every (gridDim / 8)-th tile of it ---------------------------------------------------------------------------------------
    const int tpx = (a.Wo + C6_TW - 1) / C6_TW, tpy = (a.Ho + C6_TH - 1) / C6_TH;
    const int ntiles = a.N * tpx * tpy, per = (ntiles + 7) >> 3;
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, stride = gridDim.x >> 3;
    auto tile_of = [&](int it) -> int {
        const int k = it * stride + local;
        const int t = xcd * per + k;
        return (k < per && t < ntiles) ? t : -1;
    };

    // ---- weights: 9 x 64 rows x 8 pieces = 4 608 pieces of 16 bytes, 9 per thread, once -----------------------------------------
    {
        u32x4 rw[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = tid + C6_NT * i, tap = q >> 9, row = (q >> 3) & 63, pc = q & 7;
            rw[i] = buf_load_u32x4(rwgt, ((unsigned)row * (unsigned)(a.KT * 64) + (unsigned)(tap * 64 + pc * 8)) * 2u, 0);
        }
        if (tid < 64) {
            const bool ok = tid < a.Cout;
            ssl[tid] = (ok && a.scale) ? a.scale[tid] : (ok ? 1.f : 0.f);
            ssl[64 + tid] = (ok && a.shift) ? a.shift[tid] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = tid + C6_NT * i, tap = q >> 9, row = (q >> 3) & 63, pc = q & 7;
            *reinterpret_cast<u32x4 *>(lds + C6_WALL + tap * 8192 + row * 128 + ((pc ^ ((row >> 1) & 7)) << 4)) = rw[i];
        }
    }

    // ---- halo staging map: piece q = tid + 512 p -> halo pixel (tid >> 3) + 64 p, 16-byte chunk tid & 7 (6 pieces, 340 pixels) ----
    const int chunk = tid & 7, rsub = tid >> 3;
    constexpr int HP = (C6_HPIX * 8 + C6_NT - 1) / C6_NT;          // 6
    u32x4 rh[HP];
    auto load_halo = [&](int t) __attribute__((always_inline)) {
        const int tv = t < 0 ? 0 : t;
        const int img = tv / (tpx * tpy), trem = tv - img * tpx * tpy;
        const int y0 = (trem / tpx) * C6_TH, x0 = (trem % tpx) * C6_TW;
#pragma unroll
        for (int p = 0; p < HP; ++p) {
            const int hp = rsub + 64 * p, hy = hp / C6_HW, hx = hp - hy * C6_HW;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = t >= 0 && hp < C6_HPIX && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = ok ? ((unsigned)((img * a.H + y) * a.W + x) * (unsigned)a.in_cs + (unsigned)chunk * 8u) * 2u : M3D_BUF_OOB;
            rh[p] = buf_load_u32x4(rin, off, 0);
        }
    };
    auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < HP; ++p)
            if (rsub + 64 * p < C6_HPIX) *reinterpret_cast<u32x4 *>(Hs + (rsub + 64 * p) * C6_PS + chunk * 16) = rh[p];
    };

    // lane -> pixel of a 32-pixel MFMA tile: each ds_read_b128 lane group gets 16 consecutive pixels of one patch row (bf16_conv.hip)
    int lpos;
    if (l31 < 4) lpos = l31;
    else if (l31 < 12) lpos = 16 + (l31 - 4);
    else if (l31 < 16) lpos = 4 + (l31 - 12);
    else if (l31 < 20) lpos = 24 + (l31 - 16);
    else if (l31 < 28) lpos = 8 + (l31 - 20);
    else lpos = 28 + (l31 - 28);
    int pbase[2], prow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        prow[i] = wm + i * 32 + lpos;                                  // pixel of the 8 x 32 tile
        pbase[i] = ((prow[i] >> 5) * C6_HW + (prow[i] & 31)) * C6_PS + lh * 16;
    }
    const int swk = (l31 >> 1) & 7;
    const unsigned char *Wb = lds + C6_WALL + (wn + l31) * 128;

    // residual of a tile: 2 pixels x 4 register groups x 8 bytes per lane (channels wn + 8 g + 4 lh .. + 3)
    u32x2 rres[2][4];
    auto load_res = [&](int t) __attribute__((always_inline)) {
        if constexpr (HAS_RES) {
            const int tv = t < 0 ? 0 : t;
            const int img = tv / (tpx * tpy), trem = tv - img * tpx * tpy;
            const int y0 = (trem / tpx) * C6_TH, x0 = (trem % tpx) * C6_TW;
            const __amdgpu_buffer_rsrc_t rres_ = make_rsrc(a.res, a.res_bytes);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int y = y0 + (prow[i] >> 5), x = x0 + (prow[i] & 31);
                const bool pok = t >= 0 && y < a.Ho && x < a.Wo;
                const unsigned pb = (unsigned)((img * a.Ho + y) * a.Wo + x) * (unsigned)a.res_cs * 2u;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = wn + 4 * lh + 8 * g;
                    rres[i][g] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rres_, (pok && c0 < a.Cout) ? pb + (unsigned)c0 * 2u : M3D_BUF_OOB, 0, 0));
                }
            }
        }
    };

    int it = 0;
    int tile = tile_of(0);
    load_halo(tile);
    load_res(tile);
    __syncthreads();                                               // weights + affine parameters staged
    store_halo();
    int tnext = tile_of(1);
    load_halo(tnext);
    __syncthreads();
    const float slope = a.act ? M3D_LEAKY_SLOPE : 1.f;
    const bool rm1 = a.res_mode == 1;
    while (tile >= 0) {
        // ---- compute: 9 taps x 4 K-steps, weights and pixels from LDS ------------------------------------------------------------
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // K order (tap column dx, K-step s, tap row dy): the wave's two pixel blocks are two consecutive tile rows, so block 1 at tap
        // row dy reads the fragment block 0 reads at dy + 1 -- FOUR patch rows per (dx, s) serve the six (block, dy) pairs: 48 pixel
        // fragment reads per tile and wave instead of 72 (the compute phase is LDS-bound: 1.5 KB of ds_read_b128 per MFMA before).
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int co = ((2 * s + lh) ^ swk) << 4;
                bf16x8 fr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) fr[r] = *reinterpret_cast<const bf16x8 *>(Hs + pbase[0] + (r * C6_HW + dx) * C6_PS + s * 32);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const bf16x8 fw = *reinterpret_cast<const bf16x8 *>(Wb + (dy * 3 + dx) * 8192 + co);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fr[dy], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fr[dy + 1], acc[1], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                           // A: every wave is done with the patch (and with the previous output tile)
        store_halo();                                              // tile t + 1 (zeros past the last tile)
        const int tnn = tile_of(it + 2);
        load_halo(tnn);                                            // tile t + 2 travels under the epilogue, the stores and the next compute
        // ---- epilogue: (acc [+ res]) * scale + shift [+ res], LeakyReLU, one rounding to bf16 -> the output tile in LDS -------------
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = prow[i];
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn + 4 * lh + 8 * g;
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(ssl + cl), sh = *reinterpret_cast<const f32x4 *>(ssl + 64 + cl);
                f32x4 x = {acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
                if constexpr (HAS_RES) {
                    const f32x2 r01 = unpack_bf16(rres[i][g][0]), r23 = unpack_bf16(rres[i][g][1]);
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
                const int ch = (wn + 8 * (g + lh)) >> 3;
                *reinterpret_cast<u32x4 *>(Ot + row * 128 + ((ch ^ (row & 7)) << 4)) = u32x4{pk[g][0], pk[g][1], pk[g + 1][0], pk[g + 1][1]};
            }
        }
        load_res(tnext);                                           // residual of tile t + 1
        __syncthreads();                                           // B: output tile complete, patch of tile t + 1 in place
        {
            const int img = tile / (tpx * tpy), trem = tile - img * tpx * tpy;
            const int y0 = (trem / tpx) * C6_TH, x0 = (trem % tpx) * C6_TW;
            const int Ho = a.Ho, Wo = a.Wo;
            store_otile<64, 256, C6_NT>(a, Ot, 0, 0, tid, [&](int row) {
                const int y = y0 + (row >> 5), x = x0 + (row & 31);
                return (y < Ho && x < Wo) ? (img * Ho + y) * Wo + x : -1;
            });
        }
        tile = tnext;
        tnext = tnn;
        ++it;
    }
}

// 1 if the kernel serves the descriptor (M3D_BF16_C64=0: the halo-tile kernel, A/B)
int conv_c64_applicable(const m3d_conv_bf16_desc *d)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("M3D_BF16_C64"); on = e ? atoi(e) : 1; }
    if (!on || d->dcn_offmask || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->groups != 1 || d->wgt_img_stride) return 0;
    if (d->Cin != 64 || d->Cout_pad != 64 || d->Kpad != 576 || d->out_mode != 0 || d->sigmoid_from >= 0) return 0;
    return d->W % 32 == 0 && d->H % 8 == 0;
}

int launch_conv_c64(const Bf16Args &a0, const m3d_conv_bf16_desc *d, hipStream_t st)
{
    Bf16Args a = a0;
    a.res_bytes = d->res ? (unsigned)((long long)d->N * d->Ho * d->Wo * d->res_cs * 2) : 0u;
    static int wgs = -1;
    if (wgs < 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipGetLastError();
        wgs = (cus / 8) * 8;                                       // one persistent workgroup per CU, a multiple of the 8 XCDs
        if (wgs < 8) wgs = 8;
    }
    if (d->res) hipLaunchKernelGGL(bf16_conv3x3_c64_kernel<true>, dim3(wgs), dim3(C6_NT), 0, st, a);
    else hipLaunchKernelGGL(bf16_conv3x3_c64_kernel<false>, dim3(wgs), dim3(C6_NT), 0, st, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
