// Fused front end of the bf16 path, second form (round 5): DLA base_layer (7x7, 3 -> 16) -> level0 (3x3, 16 -> 16) -> level1
// (3x3 stride 2, 16 -> 32), each with its folded BatchNorm + LeakyReLU (model/pose_dla_dcn.py:336-345,391-397), in ONE launch
// that reads the image once and writes only the 32-channel half-resolution bf16 map.  Same tiling as bf16_frontend.hip (a
// workgroup = 256 threads = one 8 x 16 tile of level1 outputs; 17 x 33 level0, 19 x 35 stem pixels and a 25 x 41 image patch
// in LDS) -- what changed is what bound that kernel (round 4: 1.25 ms at bs 64, MFMA 17 % busy, LDS array busy with b64 operand
// reads, ~35 VALU instructions per 16-pixel MFMA group, 1.84x the algorithmic HBM bytes):
//
//   * Inside the kernel everything is fp16 (11-bit significand: finer than bf16; the tiles never leave LDS): the folded
//     BatchNorm SCALE is multiplied into the fp16 weights on the host, the SHIFT is the C operand of the first MFMA of a
//     chain, and what is left of the epilogue runs on the packed-fp16 pipe: cvt_pk + pk_mul + pk_max per pair = 1.5
//     instructions per element instead of 2.75 in fp32.
//   * Stem on v_mfma_f32_32x32x16_f16 with TWO horizontally adjacent output pixels per MFMA column: rows = 16 channels x 2
//     pixel shifts, a column = the 8-pixel-wide window both pixels share (K = 8 columns x 4 channel slots per tap row, the
//     second pixel's weights shifted by one column).  The window starts at an even pixel, so the B operand is ONE 16-byte
//     aligned ds_read_b128 per lane and half the LDS bytes per output pixel of the 16x16x32 form (which read its 8-byte
//     aligned windows as two b64).  MFMA time is unchanged (7/8 of K useful in both forms).
//   * level0 runs over the LINEAR domain of the stem tile (17 rows x 36 columns, 3 garbage columns per row): source and
//     destination addresses are immediates of the fully unrolled group loop -- no per-group row / column arithmetic.
//   * Pixels outside the image have to be ZERO in the intermediate tiles (they are the next convolution's padding): only the
//     tiles on the image border carry that mask (template parameter, workgroup-uniform branch).
//   * PERSISTENT workgroups (three per CU) walk the tiles in an XCD-contiguous order (8 images per XCD at bs 64): the 2x halo
//     re-reads of the image patch hit the XCD's L2 instead of the fabric, the stem's weights are fetched once per workgroup, and
//     the global loads of the NEXT tile's image patch are in flight (15 registers per thread) under level0 / level1 of the current
//     one -- the load phase of a fresh workgroup was 6 600 of its 14 900 cycles (tools/front2_trace.py).
//   fp16 range: |x| <= 65504.  The image is normalised (O(1)), the stem / level0 activations are BatchNorm outputs (O(1-100)).
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));

#define F2_T1H 8
#define F2_T1W 16
#define F2_L0H (2 * F2_T1H + 1)      // 17 level0 rows
#define F2_S0H (F2_L0H + 2)          // 19 stem rows
#define F2_NPAIR 18                  // stem pixel pairs per row (36 columns, 35 needed)
#define F2_RS 36                     // row stride (pixels) of the stem tile AND of the level0 domain / tile
#define F2_IMH (F2_S0H + 6)          // 25 image rows
#define F2_IMW 41                    // image columns loaded (35 + 6)
#define F2_IMS 44                    // image tile row stride (pixels of 8 bytes); >= 2 * 17 + 8
#define F2_IMROWS 26                 // + 1 row that only the garbage windows of the last stem group read
#define F2_PS0 32                    // bytes per stem-tile pixel (16 fp16): dense rows are the conflict-free ones for level0's b128 reads
#define F2_PS 48                     // bytes per level0-tile pixel: level1 reads it with pixel stride 2
#define F2_NT 256
#define F2_SG 11                     // stem groups of 32 pairs: 19 * 18 = 342 pairs -> 352
#define F2_S0ROWS 20                 // stem tile rows incl. the garbage row the 11th group spills into
#define F2_LG 40                     // level0 groups of 16 pixels over the linear domain 17 * 36 = 612 -> 640 (10 per wave)
#define F2_LDS_S0 (F2_S0ROWS * F2_RS * F2_PS0)                  // 23040
#define F2_LDS_IM (F2_IMROWS * F2_IMS * 8)                      //  9152  (dead once the stem is done: shares its space with the level0 tile)
#define F2_LDS_L0 (F2_LG * 16 * F2_PS)                          // 30720
#define F2_LDS (F2_LDS_S0 + (F2_LDS_L0 > F2_LDS_IM ? F2_LDS_L0 : F2_LDS_IM))

#ifdef BF16_TRACE
static long long *g_front2_trace = nullptr;
extern "C" void m3d_front2_set_trace(void *buf) { g_front2_trace = (long long *)buf; }
#define F2TRACE() do { if (trp && threadIdx.x == 0 && iter == 1 && tri < 8) trp[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define F2TRACE()
#endif

struct Front2Args {
    const void *img;                 // fp32 [N][3][H][W] or uint8 [N][img_h][img_w][3] (BGR)
    const void *w_stem;              // fp16 fragments [7 tap rows][2 K-steps][64 lanes][8]   (BatchNorm scale folded in)
    const void *w_l0, *w_l1;         // fp16 [16][160], [32][160], k = tap * 16 + c           (BatchNorm scale folded in)
    const float *t_stem, *t_l0, *t_l1;   // folded BatchNorm shifts [16] (unused: folded into w_stem's 4th channel slot), [16], [32]
    void *out;                       // bf16 [N][H/2][W/2][out_cs]
    float mean[3], stds[3];
    int is_u8, img_h, img_w;
    int H, W, out_cs, tiles_x, tiles_y, tiles_per_xcd, total;
#ifdef BF16_TRACE
    long long *trace;
#endif
};

__device__ __forceinline__ unsigned f2_pack_h(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
// LeakyReLU on a packed fp16 pair: max(y, slope * y)
__device__ __forceinline__ unsigned f2_leaky_h(unsigned u)
{
    const f16x2 y = __builtin_bit_cast(f16x2, u);
    const f16x2 sl = {(_Float16)M3D_LEAKY_SLOPE, (_Float16)M3D_LEAKY_SLOPE};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(y, y * sl));
}
__device__ __forceinline__ unsigned f2_pack_bf(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}

__device__ __forceinline__ int f2_opaque(int x)       // the same value, as a fresh definition the optimiser cannot hoist or merge
{
    asm volatile("" : "+v"(x));
    return x;
}

constexpr int F2_NPX = F2_IMH * F2_IMS;
constexpr int F2_NI = (F2_NPX + F2_NT - 1) / F2_NT;

struct F2Tile {
    int n, ty, tx;
    bool interior;       // image patch, stem and level0 regions and the level1 tile lie inside the image (uint8: inside the FRAME)
};

__device__ __forceinline__ F2Tile f2_tile(const Front2Args &a, int t)
{
    F2Tile tl;
    const int per_img = a.tiles_x * a.tiles_y;
    tl.n = t / per_img;
    t -= tl.n * per_img;
    tl.ty = t / a.tiles_x;
    tl.tx = t - tl.ty * a.tiles_x;
    const int limH = a.is_u8 ? a.img_h : a.H, limW = a.is_u8 ? a.img_w : a.W;
    tl.interior = tl.ty > 0 && tl.tx > 0 && (2 * tl.ty * F2_T1H - 5 + F2_IMH) <= limH && (2 * tl.tx * F2_T1W - 5 + F2_IMW + 1) <= limW
                  && (tl.ty + 1) * F2_T1H <= a.H / 2 && (tl.tx + 1) * F2_T1W <= a.W / 2;
    return tl;
}

// Global loads of a tile's image patch into registers (fp32 planes or uint8 BGR bytes as floats); nothing is waited for here.
// Buffer addressing: base (per image) + voffset + soffset.  `voff[it]` = the byte offset of the thread's patch pixel `it` relative
// to the patch origin -- the same for every tile, computed once per workgroup (M3D_BUF_OOB for the slots past the patch: they read
// 0); an interior tile adds its origin as the SGPR offset: NO per-tile vector arithmetic.  A border tile re-checks every pixel
// against the image (the range check of the buffer does not see the SGPR offset) and masks through the voffset.  `kill`: every lane
// masked -- the last tile of a workgroup still ISSUES its 15 loads (they touch no memory), because hipcc's wait-count bookkeeping
// merges paths with different numbers of outstanding loads into `s_waitcnt vmcnt(0)`: with a conditional prefetch the wait for
// level1's weights in front of the third barrier became a wait for the whole image patch (level0 5 700 instead of 3 000 cycles).
template <bool CHECK, bool U8>
__device__ __forceinline__ void f2_load(const Front2Args &a, const F2Tile &tl, int tid, const unsigned (&voff)[F2_NI], float (&v)[F2_NI][3],
                                        bool kill = false)
{
    const int H = a.H, W = a.W;
    const int YI = 2 * tl.ty * F2_T1H - 5, XI = 2 * tl.tx * F2_T1W - 5;
    if (!U8) {
        const unsigned plane = (unsigned)H * W * 4;
        const __amdgpu_buffer_rsrc_t r = make_rsrc(static_cast<const float *>(a.img) + (size_t)tl.n * 3 * H * W, 3 * plane);
        const int org = (YI * W + XI) * 4;                       // (>= 0 for an interior tile)
#pragma unroll
        for (int it = 0; it < F2_NI; ++it) {
            unsigned vo = voff[it];
            unsigned so = (unsigned)org;
            if (CHECK) {
                const int i = f2_opaque(tid) + it * F2_NT;        // (recomputed per border tile: hoisted out of the tile loop the
                const int rr = i / F2_IMS, q = i - rr * F2_IMS;   // rows / columns of the five pixels would live in 10 registers)
                const int h = YI + rr, w = XI + q;
                const bool ok = !kill && vo != M3D_BUF_OOB && h >= 0 && h < H && w >= 0 && w < W;
                vo = ok ? (unsigned)((h * W + w) * 4) : M3D_BUF_OOB;
                so = 0;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[it][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so + c * plane, 0));
        }
    } else {
        // uint8 BGR frames: ONE (unaligned) dword per pixel = its three bytes + the next byte; the bytes are picked apart when the
        // tile is written (v_cvt_f32_ubyteN).  The frame's very last pixel is read as the dword that ENDS with it (shift 8): no byte
        // past the frame is touched.  (A conversion here would wait for the load.)
        const unsigned bytes = (unsigned)a.img_h * a.img_w * 3;
        const __amdgpu_buffer_rsrc_t r = make_rsrc(static_cast<const unsigned char *>(a.img) + (size_t)tl.n * bytes, bytes);
        const int org = (YI * a.img_w + XI) * 3;
#pragma unroll
        for (int it = 0; it < F2_NI; ++it) {
            unsigned vo = voff[it];
            unsigned so = (unsigned)org;
            unsigned sh = 0;
            if (CHECK) {
                const int i = f2_opaque(tid) + it * F2_NT;
                const int rr = i / F2_IMS, q = i - rr * F2_IMS;
                const int h = YI + rr, w = XI + q;
                const bool ok = !kill && vo != M3D_BUF_OOB && h >= 0 && h < a.img_h && w >= 0 && w < a.img_w;
                vo = ok ? (unsigned)((h * a.img_w + w) * 3) : M3D_BUF_OOB;
                if (ok && vo + 4 > bytes) { vo -= 1; sh = 8; }
                so = 0;
            }
            v[it][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, so, 0));
            v[it][1] = __uint_as_float(sh);
        }
    }
}

// Registers -> image tile in LDS [y][x][R, G, B, 1] fp16 (slot 3 = 1.0 carries the stem's shift); outside the image: 0 (the stem's
// zero padding).  uint8 frames: the reference pads the FRAME with zeros up to the crop size and normalises afterwards
// (lib/augmentations.py:472-501): pixels between the frame and H x W are (0 - mean) / std.
template <bool CHECK, bool U8>
__device__ __forceinline__ void f2_store(const Front2Args &a, const F2Tile &tl, int tid, float (&v)[F2_NI][3], unsigned char *imt)
{
    const int H = a.H, W = a.W;
    const int YI = 2 * tl.ty * F2_T1H - 5, XI = 2 * tl.tx * F2_T1W - 5;
#pragma unroll
    for (int it = 0; it < F2_NI; ++it) {
        const int i = tid + it * F2_NT;
        float x[3];
        if (U8) {
            bool ok = true;
            if (CHECK) {
                const int j = f2_opaque(tid) + it * F2_NT;
                const int r = j / F2_IMS, q = j - r * F2_IMS;
                const int h = YI + r, w = XI + q;
                ok = h >= 0 && w >= 0 && h < H && w < W;
            }
            const unsigned word = __float_as_uint(v[it][0]) >> (CHECK ? __float_as_uint(v[it][1]) : 0u);   // bytes B, G, R
#pragma unroll
            for (int c = 0; c < 3; ++c) {                        // plane c of the RGB tensor = BGR channel 2 - c
                const int cb = 2 - c;
                float y = (float)((word >> (8 * cb)) & 255u) / 255.0f;
                y = y - a.mean[cb];
                y = y / a.stds[cb];
                x[c] = ok ? y : 0.f;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) x[c] = v[it][c];
        }
        if (i < F2_NPX) *reinterpret_cast<u32x2_ *>(imt + (size_t)i * 8) = u32x2_{f2_pack_h(x[0], x[1]), f2_pack_h(x[2], 1.f)};
    }
}

// ---- stem: 19 rows x 18 pixel pairs, 32 pairs per MFMA group; K = 7 tap rows x (8 window columns x 4 channel slots) -------------
// Every wave runs groups wave, wave + 4 as two interleaved MFMA chains (a chain of dependent 32x32x16 MFMAs alone leaves the pipe
// idle for the latency of each) and group wave + 8 (waves 0-2) as two half chains (even / odd K-steps) that are added at the end.
// The folded BatchNorm shift rides in the 4th channel slot of K (1.0 in the tile, the shift as the weight of tap (0, 0)).
template <bool BORDER>
__device__ __forceinline__ void f2_stem(const Front2Args &a, const F2Tile &tl, const unsigned char *imt, unsigned char *s0t, int lane,
                                        int wave, const f16x8 (&wa)[14], f16x8 (&wf0)[5], f32x4 &csh0)
{
    const int H = a.H, W = a.W;
    const int YS = 2 * tl.ty * F2_T1H - 2, XS = 2 * tl.tx * F2_T1W - 2;      // stem region origin
    const int n32 = lane & 31, half = lane >> 5;
    const int q0 = wave * 32 + n32;                               // pair index of this lane in the flattened (row, pair) domain
    int row[3], m[3];
    row[0] = q0 / F2_NPAIR; m[0] = q0 - row[0] * F2_NPAIR;
#pragma unroll
    for (int k = 1; k < 3; ++k) {                                 // next group of this wave: 128 pairs further = 7 rows + 2 pairs
        row[k] = row[k - 1] + 7; m[k] = m[k - 1] + 2;
        if (m[k] >= F2_NPAIR) { m[k] -= F2_NPAIR; row[k] += 1; }
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto src_of = [&](int k) { return imt + ((size_t)row[k] * F2_IMS + 2 * m[k]) * 8 + half * 16; };
    auto frag = [&](const unsigned char *src, int f) {           // fragment f = tap row f / 2, K-step f % 2
        return *reinterpret_cast<const f16x8 *>(src + (f >> 1) * (F2_IMS * 8) + (f & 1) * 32);
    };
    auto finish = [&](const f32x16 &acc, int k) {
        unsigned msk = ~0u;
        if (BORDER) {
            const int h = YS + row[k], w = XS + 2 * m[k] + half;
            msk = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ~0u : 0u;
        }
        u32x4 o0, o1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o0[e] = f2_leaky_h(f2_pack_h(acc[2 * e], acc[2 * e + 1])) & msk;
            o1[e] = f2_leaky_h(f2_pack_h(acc[8 + 2 * e], acc[8 + 2 * e + 1])) & msk;
        }
        unsigned char *dst = s0t + ((size_t)row[k] * F2_RS + 2 * m[k] + half) * F2_PS0;
        *reinterpret_cast<u32x4 *>(dst) = o0;
        *reinterpret_cast<u32x4 *>(dst + 16) = o1;
    };
    {
        // B fragments of tap row i + 1 are requested before the four MFMAs of row i (sched_barrier: hipcc would otherwise sink
        // each ds_read to its use and expose the LDS latency in every step)
        const unsigned char *sa = src_of(0), *sb = src_of(1);
        f16x8 bA[2][2], bB[2][2];
        bA[0][0] = frag(sa, 0); bA[0][1] = frag(sa, 1); bB[0][0] = frag(sb, 0); bB[0][1] = frag(sb, 1);
        f32x16 accA = zero16, accB = zero16;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            if (i < 6) {
                bA[(i + 1) & 1][0] = frag(sa, 2 * i + 2); bA[(i + 1) & 1][1] = frag(sa, 2 * i + 3);
                bB[(i + 1) & 1][0] = frag(sb, 2 * i + 2); bB[(i + 1) & 1][1] = frag(sb, 2 * i + 3);
            }
            __builtin_amdgcn_sched_barrier(0);
            accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[2 * i], bA[i & 1][0], accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[2 * i], bB[i & 1][0], accB, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[2 * i + 1], bA[i & 1][1], accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[2 * i + 1], bB[i & 1][1], accB, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        finish(accA, 0);
        finish(accB, 1);
    }
    {                                                            // level0 weights: in flight under the third group
        const f16x8 *wl0p = reinterpret_cast<const f16x8 *>((const _Float16 *)a.w_l0 + (lane & 15) * 160 + (lane >> 4) * 8);
#pragma unroll
        for (int t = 0; t < 5; ++t) wf0[t] = wl0p[t * 4];        // (t * 32 halves = 4 f16x8)
        csh0 = *reinterpret_cast<const f32x4 *>(a.t_l0 + 4 * (lane >> 4));   // and its shift (the C operand of its chains)
    }
    if (wave + 8 < F2_SG) {
        const unsigned char *sc = src_of(2);
        f32x16 accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[0], frag(sc, 0), zero16, 0, 0, 0);
        f32x16 accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[1], frag(sc, 1), zero16, 0, 0, 0);
#pragma unroll
        for (int f = 2; f < 14; f += 2) {
            accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[f], frag(sc, f), accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[f + 1], frag(sc, f + 1), accB, 0, 0, 0);
        }
        finish(accA + accB, 2);
    }
}

// ---- level0 over the linear domain of the stem tile: pixel p = row * 36 + column, 16 pixels per MFMA group, 10 groups per wave
// (40 groups: the last 28 pixels are garbage past the 17 x 36 domain, inside both tiles), two groups = two chains at a time
template <bool BORDER>
__device__ __forceinline__ void f2_level0(const Front2Args &a, const F2Tile &tl, const unsigned char *s0t, unsigned char *l0t, int lane,
                                          int wave, const f16x8 (&wf0)[5], const f32x4 csh)
{
    const int H = a.H, W = a.W;
    const int Y0 = 2 * tl.ty * F2_T1H - 1, X0 = 2 * tl.tx * F2_T1W - 1;      // level0 region origin
    const int l15 = lane & 15, kg = lane >> 4;
    // k-group kg of K-step t reads tap 2t + (kg >> 1), channel half kg & 1; tap 9 (t = 4, kg >= 2) has zero weights: it
    // re-reads tap 8 so that the operand stays finite
    const unsigned char *src[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        int tap = 2 * t + (kg >> 1);
        tap = tap > 8 ? 8 : tap;
        src[t] = s0t + (size_t)(wave * 16 + l15) * F2_PS0 + ((tap / 3) * F2_RS + (tap % 3)) * F2_PS0 + (kg & 1) * 16;
    }
    unsigned char *dst = l0t + (size_t)(wave * 16 + l15) * F2_PS + kg * 8;
    const int p = wave * 16 + l15;
    auto finish = [&](const f32x4 &acc, int it) {
        unsigned msk = ~0u;
        if (BORDER) {
            const int pp = p + it * 64;
            const int r = pp / F2_RS, c = pp - r * F2_RS;
            const int h = Y0 + r, w = X0 + c;
            msk = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? ~0u : 0u;
        }
        const u32x2_ o = {f2_leaky_h(f2_pack_h(acc[0], acc[1])) & msk, f2_leaky_h(f2_pack_h(acc[2], acc[3])) & msk};
        *reinterpret_cast<u32x2_ *>(dst + it * (64 * F2_PS)) = o;
    };
    // 5 pairs of groups x 5 K-steps = 25 steps of two MFMAs (two independent chains); the B fragments of step s + 3 are
    // requested before the MFMAs of step s (ring of 4 fragment pairs)
    constexpr int NS = (F2_LG / 8) * 5, DEPTH = 3;
    f16x8 qa[DEPTH + 1], qb[DEPTH + 1];
    auto ldstep = [&](int st) {
        const int pr = st / 5, t = st - pr * 5;
        qa[st & DEPTH] = *reinterpret_cast<const f16x8 *>(src[t] + (2 * pr) * (64 * F2_PS0));
        qb[st & DEPTH] = *reinterpret_cast<const f16x8 *>(src[t] + (2 * pr + 1) * (64 * F2_PS0));
    };
#pragma unroll
    for (int st = 0; st < DEPTH; ++st) ldstep(st);
    f32x4 accA = csh, accB = csh, doneA = csh, doneB = csh;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int pr = st / 5, t = st - pr * 5;
        if (st + DEPTH < NS) ldstep(st + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
        accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[t], qa[st & DEPTH], t == 0 ? csh : accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[t], qb[st & DEPTH], t == 0 ? csh : accB, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // the epilogue of a pair runs behind the FIRST K-step of the next pair: its MFMA results have landed by then and the
        // conversions issue in the shadow of the new chains instead of in front of them
        if (t == 0 && pr > 0) {
            finish(doneA, 2 * pr - 2);
            finish(doneB, 2 * pr - 1);
        }
        if (t == 4) { doneA = accA; doneB = accB; }
    }
    finish(doneA, F2_LG / 4 - 2);
    finish(doneB, F2_LG / 4 - 1);
}

// ---- level1: 8 x 16 outputs, stride 2, 32 channels; a wave = two rows of the tile ---------------------------------------------
template <bool BORDER>
__device__ __forceinline__ void f2_level1(const Front2Args &a, const F2Tile &tl, const unsigned char *l0t, int lane, int wave,
                                          const f16x8 (&wf1)[2][5], const f32x4 csh0, const f32x4 csh1)
{
    const int l15 = lane & 15, kg = lane >> 4;
    const int y1 = tl.ty * F2_T1H, x1 = tl.tx * F2_T1W;
    int toff[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        int tap = 2 * t + (kg >> 1);
        tap = tap > 8 ? 8 : tap;
        toff[t] = ((tap / 3) * F2_RS + (tap % 3)) * F2_PS + (kg & 1) * 16;
    }
    const int Ho = a.H / 2, Wo = a.W / 2;
    const unsigned char *src0 = l0t + ((size_t)(4 * wave) * F2_RS + 2 * l15) * F2_PS;
    f16x8 b[2][5];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int t = 0; t < 5; ++t) b[it][t] = *reinterpret_cast<const f16x8 *>(src0 + it * (2 * F2_RS * F2_PS) + toff[t]);
    f32x4 acc[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        acc[it][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[0][0], b[it][0], csh0, 0, 0, 0);
        acc[it][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[1][0], b[it][0], csh1, 0, 0, 0);
    }
#pragma unroll
    for (int t = 1; t < 5; ++t)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            acc[it][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[0][t], b[it][t], acc[it][0], 0, 0, 0);
            acc[it][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[1][t], b[it][t], acc[it][1], 0, 0, 0);
        }
    // stores: base (per image) + voffset (the lane's pixel of the tile and its 4 channels: the same for every tile) + soffset (the
    // tile's origin); a border tile masks the pixels past the map through the voffset
    const __amdgpu_buffer_rsrc_t ro = make_rsrc((const __bf16 *)a.out + (size_t)tl.n * Ho * Wo * a.out_cs, (unsigned)Ho * Wo * a.out_cs * 2);
    const unsigned so = (unsigned)((y1 * Wo + x1) * a.out_cs * 2);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int oy = wave * 2 + it;                             // group = one row of the 8 x 16 tile
        unsigned vo = (unsigned)(((oy * Wo + l15) * a.out_cs + 4 * kg) * 2);
        if (BORDER && !(y1 + oy < Ho && x1 + l15 < Wo)) vo = M3D_BUF_OOB;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x4 y = acc[it][hh];
            const f32x4 z = y * M3D_LEAKY_SLOPE;
            const u32x2_ o = {f2_pack_bf(fmaxf(y[0], z[0]), fmaxf(y[1], z[1])), f2_pack_bf(fmaxf(y[2], z[2]), fmaxf(y[3], z[3]))};
            __builtin_amdgcn_raw_buffer_store_b64(o, ro, vo, so + hh * 32, 0);
        }
    }
}

template <int OCC, bool U8>
__global__ __launch_bounds__(F2_NT) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void bf16_frontend2_kernel(const Front2Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[F2_LDS];
    unsigned char *s0t = lds;                                    // [20][36][16 fp16]
    unsigned char *imt = lds + F2_LDS_S0;                        // [26][44][R, G, B, 1 fp16]
    unsigned char *l0t = lds + F2_LDS_S0;                        // [640][48 B]   (over the image tile, which is dead by then)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous tile order: workgroup L runs on XCD L % 8 (round-robin dispatch); XCD x owns tiles [x * per, (x + 1) * per)
    // and its workgroups (slots 0 .. gridDim.x / 8 - 1) walk them in row-major order with stride = number of slots, so that the
    // workgroups resident on one XCD are neighbours at any time and share image rows in its L2
    const int L = blockIdx.x, slots = gridDim.x >> 3;
    const int lo = (L & 7) * a.tiles_per_xcd;
    const int hi = min(lo + a.tiles_per_xcd, a.total);
    int t = lo + (L >> 3);
    if (t >= hi) return;
#ifdef BF16_TRACE
    long long *trp = a.trace ? a.trace + (size_t)blockIdx.x * 8 : nullptr;
    int tri = 0;
#endif
    int iter = 0;
    (void)iter;
    unsigned voff[F2_NI];                                        // byte offset of the thread's patch pixels from the patch origin
#pragma unroll
    for (int it = 0; it < F2_NI; ++it) {
        const int i = tid + it * F2_NT;
        const int r = i / F2_IMS, q = i - r * F2_IMS;
        voff[it] = (i < F2_NPX && q < F2_IMW) ? (unsigned)(U8 ? (r * a.img_w + q) * 3 : (r * a.W + q) * 4) : M3D_BUF_OOB;
    }
    F2Tile cur = f2_tile(a, t);
    float v[F2_NI][3];
    if (cur.interior) f2_load<false, U8>(a, cur, tid, voff, v);
    else f2_load<true, U8>(a, cur, tid, voff, v);
    // The stem's 14 weight fragments are requested at the top of a tile and pinned (opaque asm) in front of its first barrier:
    // left to itself hipcc fetches each fragment right in front of the MFMA that uses it -- a global-memory round trip inside the
    // chain.  level0's weights are requested under the stem and level1's under level0.  All three sets are re-read per tile from
    // L1 / L2 (14 + 5 + 10 16-byte loads per lane): resident they would cost the third workgroup per CU (56 + 20 + 40 registers).
    for (;;) {
        F2TRACE();
        f16x8 wa[14];
        {
            const f16x8 *wp = reinterpret_cast<const f16x8 *>(a.w_stem) + lane;
#pragma unroll
            for (int f = 0; f < 14; ++f) wa[f] = wp[f * 64];
        }
        if (cur.interior) f2_store<false, U8>(a, cur, tid, v, imt);
        else f2_store<true, U8>(a, cur, tid, v, imt);
#pragma unroll
        for (int f = 0; f < 14; ++f) asm volatile("" : "+v"(wa[f]));
        __syncthreads();
        F2TRACE();
        f16x8 wf0[5], wf1[2][5];
        f32x4 csh0, csh1[2];
        if (cur.interior) f2_stem<false>(a, cur, imt, s0t, lane, wave, wa, wf0, csh0);
        else f2_stem<true>(a, cur, imt, s0t, lane, wave, wa, wf0, csh0);
#pragma unroll
        for (int k = 0; k < 5; ++k) asm volatile("" : "+v"(wf0[k]));
        asm volatile("" : "+v"(csh0));
        __syncthreads();
        F2TRACE();
        // level1's weights and shifts FIRST, as asm loads with a hand-counted wait in front of the third barrier: the vector-memory
        // counter is in order, so `vmcnt(loads of the image patch)` = "everything older than the patch loads has landed".  hipcc's
        // own bookkeeping turns the wait for these registers into vmcnt(0) at the join of the interior / border paths -- a wait for
        // the whole image patch, HBM latency, inside level0 (5 700 instead of 3 000 cycles; tools/front2_trace.py).  Asm loads are
        // invisible to the register allocator: the kernel must not spill (tools/check_isa_hazards.py lists it as hand-counted).
        {
            const _Float16 *wl1 = (const _Float16 *)a.w_l1 + (lane & 15) * 160 + (lane >> 4) * 8;
            const _Float16 *wl1b = wl1 + 16 * 160;
            const float *t1 = a.t_l1 + 4 * (lane >> 4);
#define F2_ALOAD(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr) : "memory")
            F2_ALOAD(wf1[0][0], wl1, 0); F2_ALOAD(wf1[0][1], wl1, 64); F2_ALOAD(wf1[0][2], wl1, 128); F2_ALOAD(wf1[0][3], wl1, 192);
            F2_ALOAD(wf1[0][4], wl1, 256);
            F2_ALOAD(wf1[1][0], wl1b, 0); F2_ALOAD(wf1[1][1], wl1b, 64); F2_ALOAD(wf1[1][2], wl1b, 128); F2_ALOAD(wf1[1][3], wl1b, 192);
            F2_ALOAD(wf1[1][4], wl1b, 256);
            F2_ALOAD(csh1[0], t1, 0); F2_ALOAD(csh1[1], t1, 64);
#undef F2_ALOAD
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next tile's image patch: requested now, converted and written to LDS at the top of the next iteration
        const int tn = t + slots;
        const bool has_next = tn < hi;
        const F2Tile nxt = has_next ? f2_tile(a, tn) : cur;
        if (has_next && nxt.interior) f2_load<false, U8>(a, nxt, tid, voff, v);
        else f2_load<true, U8>(a, nxt, tid, voff, v, !has_next);
        if (cur.interior) f2_level0<false>(a, cur, s0t, l0t, lane, wave, wf0, csh0);
        else f2_level0<true>(a, cur, s0t, l0t, lane, wave, wf0, csh0);
        // exactly the image-patch loads of the next tile were issued behind level1's weights (15 dwords, or 5 for uint8 frames; the
        // last tile issues them masked): everything older has landed when at most that many are outstanding
        if (U8) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        __syncthreads();
        F2TRACE();
        if (cur.interior) f2_level1<false>(a, cur, l0t, lane, wave, wf1, csh1[0], csh1[1]);
        else f2_level1<true>(a, cur, l0t, lane, wave, wf1, csh1[0], csh1[1]);
        F2TRACE();
        if (!has_next) break;
        __syncthreads();                                         // the level0 tile (= the image tile's space) has been read
        cur = nxt;
        t = tn;
        ++iter;
    }
}

static int front2_occ() { return 2; }   // workgroups per CU (LDS would allow 3; at 168 registers the kernel spills, which the asm loads forbid)

extern "C" int m3d_frontend2_bf16_forward(const void *img, int is_u8, int img_h, int img_w, const float *mean3, const float *stds3,
                                          const void *w_stem_frag, const float *t_stem, const void *w_l0, const float *t_l0,
                                          const void *w_l1, const float *t_l1, void *out, int out_cs, int N, int H, int W,
                                          m3d_stream_t stream)
{
    M3D_REQUIRE(img && w_stem_frag && w_l0 && w_l1 && t_stem && t_l0 && t_l1 && out, "frontend2_bf16: null pointer");
    M3D_REQUIRE(H % 2 == 0 && W % 2 == 0 && out_cs % 8 == 0 && out_cs >= 32, "frontend2_bf16: even H, W; out_cs %% 8 == 0, >= 32");
    M3D_REQUIRE(N >= 1 && H >= 2 && W >= 2, "frontend2_bf16: empty input");
    Front2Args a = {};
    a.img = img; a.w_stem = w_stem_frag; a.w_l0 = w_l0; a.w_l1 = w_l1; a.t_stem = t_stem; a.t_l0 = t_l0; a.t_l1 = t_l1;
    a.out = out; a.is_u8 = is_u8 ? 1 : 0; a.H = H; a.W = W; a.out_cs = out_cs;
    if (is_u8) {
        M3D_REQUIRE(mean3 && stds3 && img_h >= 1 && img_w >= 1 && img_h <= H && img_w <= W, "frontend2_bf16: frame / normalisation arguments");
        for (int c = 0; c < 3; ++c) {
            M3D_REQUIRE(stds3[c] != 0.f, "frontend2_bf16: zero std");
            a.mean[c] = mean3[c];
            a.stds[c] = stds3[c];
        }
        a.img_h = img_h; a.img_w = img_w;
    }
    a.tiles_x = cdiv(W / 2, F2_T1W); a.tiles_y = cdiv(H / 2, F2_T1H);
    const long long total = (long long)a.tiles_x * a.tiles_y * N;
    M3D_REQUIRE(total < (1ll << 30), "frontend2_bf16: too many tiles");
    a.total = (int)total;
    a.tiles_per_xcd = cdiv(total, 8);
#ifdef BF16_TRACE
    a.trace = g_front2_trace;
#endif
    // persistent workgroups: occupancy x CUs of them (a multiple of the 8 XCDs), fewer when the launch has fewer tiles
    int dev = 0, ncu = 0;
    M3D_HIP(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    const int occ = front2_occ();
    const int slots = imin(a.tiles_per_xcd, (occ * ncu / 8 > 0 ? occ * ncu / 8 : 1));
    const int grid = slots * 8;
    // (a spill would be restored over the data of an in-flight asm load: refuse a build that uses scratch memory)
    static int scratch = -1;
    if (scratch < 0) {
        hipFuncAttributes fa0, fa1;
        M3D_HIP(hipFuncGetAttributes(&fa0, reinterpret_cast<const void *>(&bf16_frontend2_kernel<2, false>)));
        M3D_HIP(hipFuncGetAttributes(&fa1, reinterpret_cast<const void *>(&bf16_frontend2_kernel<2, true>)));
        scratch = (int)(fa0.localSizeBytes + fa1.localSizeBytes);
    }
    M3D_REQUIRE(scratch == 0, "frontend2_bf16: the kernel was built with register spills (%d bytes of scratch)", scratch);
    if (is_u8) hipLaunchKernelGGL((bf16_frontend2_kernel<2, true>), dim3(grid), dim3(F2_NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((bf16_frontend2_kernel<2, false>), dim3(grid), dim3(F2_NT), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
