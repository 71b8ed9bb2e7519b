// DCNv2 3x3 (stride 1, pad 1) of the bf16 path with the sampling window resident in LDS ("patch" kernel).
//
// Why: the implicit-GEMM tile of bf16_conv.hip gathers the four bilinear corners of every (pixel, tap, 64-channel K-step)
// from global memory -- 80 KB per workgroup K-step through a vector L1 that delivers 64 B/clk/CU (1280 cycles against 512 of
// MFMA) -- and combines them in fp32 (52 VALU instructions per 16-byte piece, 32 of them unpacking bf16 pairs: another ~1000
// cycles).  On 128 -> 128 @ 48x160, bs 64 that is 0.385 ms = 377 TFLOP/s, 0.15 of the bf16 MFMA peak, with the L1 path alone
// putting a floor of ~0.19 ms under ANY global-gather design (4 corners x 9 taps x the input through 64 B/clk/CU).
//
// Here a workgroup owns a TH x 16 patch of output pixels and a 32-channel chunk at a time:
//   * the input window the 9 taps can reach -- the patch grown by 1 (tap) + R (largest |offset| of the launch, rounded up) + 1
//     (high bilinear corner) on every side -- is read ONCE per chunk with coalesced 16-byte loads, converted bf16 -> fp16
//     (exact: a bf16 value has 8 significant bits, fp16 keeps 11; values beyond +-65504 saturate) and parked in
//     LDS with a pixel stride of 80 bytes (odd multiple of 16: consecutive pixels fall into different bank groups); positions
//     outside the image are stored as zeros, which IS the reference's rule (dcn_v2_im2col_cuda.cu:18-47,165: a corner outside
//     the image contributes nothing, a sample at h <= -1 or h >= H has no corner inside);
//   * a lane = (pixel, K half) of the MFMA B operand reads its four corners straight from the patch (ds_read_b128, immediate
//     offsets for the corner to the right) and combines them with 16 v_pk_fma_f16 per fragment -- fp16 pairs, no unpacking --
//     the result IS the B fragment of v_mfma_f32_32x32x16_f16: no staged sample tile, no second LDS round trip;
//   * the sampling state of a (pixel, tap) -- one LDS offset, four fp16 corner weights with the modulation mask folded in --
//     is built once per workgroup and kept in 27 registers;
//   * weights (fp16 copy of the packed bf16 weights, exact) are staged per (tap, chunk) as in the halo-tile kernel.
// The offsets decide whether the window fits, PER TILE (round 5; before: one decision per launch from a |offset| pre-pass): the
// workgroup reduces max |offset| of its own pixels (it has fetched their offsets anyway), sizes its window to that radius, and when
// the radius exceeds RMAX it only raises the tile's flag word in dcn_ws; the implicit-GEMM kernel launched behind this one recomputes
// the 128-pixel tiles that touch a flagged patch tile and nothing else.  No pre-pass, no host round trip, and a few outlier pixels
// no longer send a whole launch to the slow kernel.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#include "bf16_tile.h"

#ifndef DP_ABL
#define DP_ABL 0                         // diagnostic builds: operand ablations of the K loop (timing only)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#ifdef BF16_TRACE
// diagnostic build (make trace): s_memtime stamps of thread 0 of every workgroup, 160 slots per workgroup (tools/bf16_dcn_trace.py)
#define PTRACE_INIT() long long *trp = a.trace ? a.trace + (size_t)blockIdx.x * 160 : nullptr; int tri = 0
#define PTRACE() do { if (trp && tid == 0 && tri < 160) trp[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define PTRACE_INIT()
#define PTRACE()
#endif

#define DP_PS 80                         // bytes per patch pixel: 32 fp16 channels + 16 bytes pad
#define DP_WKB 64                        // bytes per weight row and step (32 fp16 channels)

// TPS = taps per weight stage (one barrier per stage): 1 for the 4-wave tile (two workgroups per CU cover each other's
// barriers), 3 for the 8-wave tile -- one workgroup per CU whose waves all stop at the same barrier: with a barrier per tap
// (256 MFMA cycles per wave) the 16 x 16 tile spent 2500 cycles per (tap, chunk) on 512 cycles of MFMA per SIMD.
template <int TH, int RMAX, int TPS>
__global__ __launch_bounds__(TH * 32) __attribute__((amdgpu_waves_per_eu(2, 2)))
void bf16_dcn_patch_kernel(const Bf16Args a, const void *__restrict__ wgt16, unsigned *__restrict__ flags)
{
    constexpr int TW = 16, BM = TH * TW, BN = 128, NT = TH * 32, WAVES = TH / 2;
    constexpr int PHMAX = TH + 3 + 2 * RMAX, PWMAX = TW + 3 + 2 * RMAX;
    constexpr int HBYTES = PHMAX * PWMAX * DP_PS;
    constexpr int HP = (PHMAX * PWMAX * 4 + NT - 1) / NT;        // 16-byte patch pieces per thread and chunk, at most
    constexpr int PB = BN * (DP_WKB / 16) / NT;                  // weight pieces per thread and tap
    constexpr int NS = 9 / TPS, WSB = TPS * BN * DP_WKB;         // weight stages per chunk, bytes per stage
    static_assert(PB >= 1 && BM * BN * 2 <= HBYTES && NS * TPS == 9 && (TPS == 1 || PB == 1), "tile shape");
    __shared__ __attribute__((aligned(16))) unsigned char lds[HBYTES + 2 * WSB];
    unsigned char *Hs = lds, *Ws = lds + HBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    PTRACE_INIT();
    PTRACE();
    const int ntiles = a.tiles_m * a.tiles_n;
    int tile = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = tile / a.tiles_n, tile_n = tile - tile_m * a.tiles_n;
    const int n0 = tile_n * BN;
    const int tpx = a.Wo / TW, tpy = (a.Ho + TH - 1) / TH;
    const int img = tile_m / (tpx * tpy), trem = tile_m - img * tpx * tpy;
    const int y0 = (trem / tpx) * TH, x0 = (trem % tpx) * TW;

    // lane -> pixel inside the wave's 2 x 16 pixels: ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} (+32); each group gets the 16 pixels of ONE patch row, so that a smooth offset field reads 16
    // consecutive window pixels per group (pixel stride 80 bytes: 16 different bank groups) -- as in the halo-tile kernel
    int lpos;
    if (l31 < 4) lpos = l31;
    else if (l31 < 12) lpos = 16 + (l31 - 4);
    else if (l31 < 16) lpos = 4 + (l31 - 12);
    else if (l31 < 20) lpos = 24 + (l31 - 16);
    else if (l31 < 28) lpos = 8 + (l31 - 20);
    else lpos = 28 + (l31 - 28);
    const int p = wave * 32 + lpos, ty = p / TW, tx = p - ty * TW;
    const int oy = y0 + ty, ox = x0 + tx;
    const bool pvalid = oy < a.Ho && ox < a.Wo;
    const int mpx = pvalid ? (img * a.Ho + oy) * a.Wo + ox : -1;
    // everything that does not depend on the window radius is fetched before the radius is known
    f32x4 omv[7];
    {
        const f32x4 *omp = reinterpret_cast<const f32x4 *>(a.om + (size_t)(pvalid ? mpx : 0) * a.om_cs);
#pragma unroll
        for (int q = 0; q < 7; ++q) omv[q] = omp[q];
    }
    float ssc = 1.f, ssh = 0.f;              // epilogue scale / shift of channel n0 + tid, fetched now, used after the K loop
    if (tid < BN && n0 + tid < a.Cout) {
        if (a.scale) ssc = a.scale[n0 + tid];
        if (a.shift) ssh = a.shift[n0 + tid];
    }

    // ---- does the window of THIS tile fit?  R = the radius its own pixels need (smaller windows where the offsets are small); a
    // tile that does not fit raises its flag and leaves: the implicit-GEMM kernel launched behind this one recomputes it ------------
    unsigned omx = 0u;
    if (pvalid) {
#pragma unroll
        for (int q = 0; q < 18; ++q) omx = max(omx, __float_as_uint(omv[q >> 2][q & 3]) & 0x7fffffffu);
    }
    const int R = dcn_tile_radius(omx, reinterpret_cast<unsigned *>(lds), tid, NT);
    if (tid == 0) flags[tile_m] = R > RMAX ? 1u : 0u;           // (the channel blocks of a pixel tile write the same word)
    if (R > RMAX) return;
    __syncthreads();                                            // the scratch words are patch bytes from here on
    PTRACE();
    const int PH = TH + 3 + 2 * R, PW = TW + 3 + 2 * R, PWB = PW * DP_PS;
    const int py0 = y0 - 1 - R, px0 = x0 - 1 - R;               // image position of patch pixel (0, 0)

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rwgt = make_rsrc(wgt16, a.wgt_bytes);

    // ---- sampling state of this lane's pixel for the 9 taps (both K halves of a pixel build the same state) -----------------
    int soff[9];
    unsigned swa[9], swb[9];
    auto build_states = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float dh = omv[(2 * t) >> 2][(2 * t) & 3], dw = omv[(2 * t + 1) >> 2][(2 * t + 1) & 3];
            const float mk = pvalid ? omv[(18 + t) >> 2][(18 + t) & 3] : 0.f;
            const float h_im = (float)(oy - 1 + t / 3) + dh, w_im = (float)(ox - 1 + t % 3) + dw;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const float lhh = h_im - hf, lww = w_im - wf, uh = 1.f - lhh, uw = 1.f - lww;
            // |dh|, |dw| <= R puts (hf, wf) inside [py0, py0 + PH - 2] x [px0, px0 + PW - 2]; the clamp only keeps a lane without a
            // pixel (or a launch whose bound was wrong) inside the LDS allocation
            const int r = min(max((int)hf - py0, 0), PH - 2), cc = min(max((int)wf - px0, 0), PW - 2);
            soff[t] = (r * PW + cc) * DP_PS + lh * 16;
            const f32x2 wa = {uh * uw * mk, uh * lww * mk}, wb = {lhh * uw * mk, lhh * lww * mk};
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            swa[t] = __builtin_bit_cast(unsigned, __builtin_convertvector(wa, f16x2));       // round to nearest even
            swb[t] = __builtin_bit_cast(unsigned, __builtin_convertvector(wb, f16x2));
        }
    };

    // ---- patch staging map: piece q = tid + NT*i -> patch pixel q >> 2, 16-byte piece q & 3 ------------------------------------
    const int npieces = PH * PW * 4;
    const int piece = tid & 3, pix0 = tid >> 2;
    const int hy0 = pix0 / PW, hx0 = pix0 - hy0 * PW;
    const int dq = (NT / 4) / PW, dr = (NT / 4) - dq * PW;       // pixel advance per pass, as (rows, columns)
    const int hdst0 = pix0 * DP_PS + piece * 16;
    // weight staging map (as the halo-tile kernel at 32-channel steps): thread = (row tid >> 2, piece tid & 3), piece XOR (row >> 2) & 3
    const int wrsub = tid >> 2;
    constexpr int WRPP = NT / 4;
    const unsigned woff0 = ((unsigned)(n0 + wrsub) * (unsigned)(a.KT * 64) + (unsigned)piece * 8u) * 2u;
    const unsigned wrow_step = (unsigned)WRPP * (unsigned)(a.KT * 64) * 2u;
    const int wdst0 = wrsub * DP_WKB + ((piece ^ ((wrsub >> 2) & 3)) << 4);
    u32x4 rh[HP];
    const int NC = a.Cin >> 5;                                   // 32-channel chunks

    auto load_patch = [&](int c) __attribute__((always_inline)) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(c) * 64u;
        int hy = hy0, hx = hx0;
#pragma unroll
        for (int i = 0; i < HP; ++i) {
            const int y = py0 + hy, x = px0 + hx;
            const bool ok = tid + NT * i < npieces && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = ok ? ((unsigned)((img * a.H + y) * a.W + x) * (unsigned)a.in_cs + (unsigned)piece * 8u) * 2u : M3D_BUF_OOB;
            rh[i] = buf_load_u32x4(rin, off, so);
            hx += dr; hy += dq;
            if (hx >= PW) { hx -= PW; ++hy; }
        }
    };
    auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < HP; ++i)
            if (tid + NT * i < npieces) {
                u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {                   // bf16 pair -> fp16 pair (exact inside the fp16 range)
                    const unsigned d = rh[i][e];
                    v[e] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)));
                }
                *reinterpret_cast<u32x4 *>(Hs + hdst0 + i * (NT / 4) * DP_PS) = v;
            }
    };
    f32x16 acc[4][1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][0][r] = 0.f;
    const int swk = (l31 >> 2) & 3;                             // swizzle term of this lane's weight rows

    // one tap of one chunk: 2 K-steps of 16 channels; Wt = this tap's [128][32] fp16 weight tile in LDS
    auto compute_tap = [&](int u, const unsigned char *Wt) __attribute__((always_inline)) {
        const unsigned char *Wb = Wt + l31 * DP_WKB;
        const unsigned char *P0 = Hs + soff[u], *P1 = P0 + PWB;
        const unsigned wa = swa[u], wb = swb[u];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 c00 = *reinterpret_cast<const u32x4 *>(P0 + s * 32), c01 = *reinterpret_cast<const u32x4 *>(P0 + s * 32 + DP_PS);
#if DP_ABL & 2                     // diagnostic: two of the four corners not read
            const u32x4 c10 = c00, c11 = c01;
#else
            const u32x4 c10 = *reinterpret_cast<const u32x4 *>(P1 + s * 32), c11 = *reinterpret_cast<const u32x4 *>(P1 + s * 32 + DP_PS);
#endif
            const int co = ((2 * s + lh) ^ swk) << 4;
            f16x8 fw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#if DP_ABL & 1                     // diagnostic: half of the weight fragments not read (wrong results, timing only)
                if (j >= 2) { fw[j] = fw[j - 2]; continue; }
#endif
                fw[j] = *reinterpret_cast<const f16x8 *>(Wb + j * 32 * DP_WKB + co);
            }
            // (1-lh)(1-lw) v1 + (1-lh) lw v2 + lh (1-lw) v3 + lh lw v4 (dcn_v2_im2col_cuda.cu:44-46), mask folded into the
            // weights, on fp16 pairs (v_pk_mul_f16 / v_pk_fma_f16).  Plain vector code, not inline asm: the fragment feeds the
            // MFMA right behind it, and the wait states a VALU result needs before a matrix instruction reads it are only
            // inserted for instructions the compiler can see (an asm version of these 16 instructions produced wrong fragments
            // in some waves of the 8-wave tile, depending on the code around it).
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 w2a = __builtin_bit_cast(h2, wa), w2b = __builtin_bit_cast(h2, wb);
            const h2 w00 = __builtin_shufflevector(w2a, w2a, 0, 0), w01 = __builtin_shufflevector(w2a, w2a, 1, 1);
            const h2 w10 = __builtin_shufflevector(w2b, w2b, 0, 0), w11 = __builtin_shufflevector(w2b, w2b, 1, 1);
            u32x4 fb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned d00 = c00[e], d01 = c01[e], d10 = c10[e], d11 = c11[e];
                h2 r = __builtin_bit_cast(h2, d00) * w00;
                r = __builtin_elementwise_fma(__builtin_bit_cast(h2, d01), w01, r);
                r = __builtin_elementwise_fma(__builtin_bit_cast(h2, d10), w10, r);
                r = __builtin_elementwise_fma(__builtin_bit_cast(h2, d11), w11, r);
                fb[e] = __builtin_bit_cast(unsigned, r);
#if DP_ABL & 4                     // diagnostic: no bilinear combine (the first corner is the fragment)
                fb[e] = d00;
#endif
            }
            const f16x8 fp = __builtin_bit_cast(f16x8, fb);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fp, acc[j][0], 0, 0, 0);
        }
    };

    if constexpr (TPS == 1) {
        // ---- a barrier per tap; weight loads run two taps ahead through three register sets (as the halo-tile kernel) ---------
        u32x4 rw[3][PB];
        auto load_w = [&](int c, int u, auto rtag) __attribute__((always_inline)) {
            constexpr int RS = decltype(rtag)::value;
            if (u >= 9) { u -= 9; ++c; }
            if (c >= NC) return;
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(u * a.Cin + c * 32) * 2u;
#pragma unroll
            for (int i = 0; i < PB; ++i) rw[RS][i] = buf_load_u32x4(rwgt, woff0, so + (unsigned)i * wrow_step);
        };
        auto store_w = [&](int buf, auto rtag) __attribute__((always_inline)) {
            constexpr int RS = decltype(rtag)::value;
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<u32x4 *>(Ws + buf * WSB + wdst0 + i * WRPP * DP_WKB) = rw[RS][i];
        };
#define RT(n) std::integral_constant<int, (n) % 3>{}
        load_patch(0);
        load_w(0, 0, RT(0));
        load_w(0, 1, RT(1));
        build_states();                                         // under the latency of the first window / weight fetches
        __syncthreads();                                        // the bound reduction used the start of the LDS
        store_patch();
        store_w(0, RT(0));
        __syncthreads();
        int t = 0;                                              // step index; weight buffer t & 1
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int u = 0; u < 9; ++u, ++t) {
                const bool last = (c == NC - 1) && u == 8;
                if (u % 3 == 0) load_w(c, u + 2, RT(2)); else if (u % 3 == 1) load_w(c, u + 2, RT(0)); else load_w(c, u + 2, RT(1));
                if (u == 5 && c + 1 < NC) load_patch(c + 1);    // the next chunk's window travels under the last taps of this one
                __builtin_amdgcn_sched_barrier(0);
                compute_tap(u, Ws + (t & 1) * WSB);
                if (!last) {
                    if (u % 3 == 0) store_w((t + 1) & 1, RT(1)); else if (u % 3 == 1) store_w((t + 1) & 1, RT(2)); else store_w((t + 1) & 1, RT(0));
                }
                if (u == 8 && c + 1 < NC) {
                    __syncthreads();                            // every wave is done reading the window
                    store_patch();
                }
                __syncthreads();
            }
        }
#undef RT
    } else {
        // ---- a barrier per stage of TPS taps: the weights of the next stage are fetched at the top of a stage and parked in
        // the other LDS buffer at its end (one register set: a stage is long enough to cover the fetch) --------------------------
        u32x4 rw[TPS];
        auto load_w = [&](int c, int st) __attribute__((always_inline)) {
            if (st >= NS) { st -= NS; ++c; }
            if (c >= NC) return;
#pragma unroll
            for (int i = 0; i < TPS; ++i) {
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((st * TPS + i) * a.Cin + c * 32) * 2u;
                rw[i] = buf_load_u32x4(rwgt, woff0, so);
            }
        };
        auto store_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TPS; ++i) *reinterpret_cast<u32x4 *>(Ws + buf * WSB + i * BN * DP_WKB + wdst0) = rw[i];
        };
        load_patch(0);
        load_w(0, 0);
        PTRACE();
        build_states();                                         // under the latency of the first window / weight fetches
        PTRACE();
        __syncthreads();                                        // the bound reduction used the start of the LDS
        store_patch();
        store_w(0);
        __syncthreads();
        PTRACE();
        int t = 0;                                              // stage index; weight buffer t & 1
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int st = 0; st < NS; ++st, ++t) {
                const bool last = (c == NC - 1) && st == NS - 1;
                PTRACE();
                load_w(c, st + 1);
                if (st == NS - 1 && c + 1 < NC) load_patch(c + 1);   // the next chunk's window travels under this chunk's last stage
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TPS; ++i) compute_tap(st * TPS + i, Ws + (t & 1) * WSB + i * BN * DP_WKB);
                PTRACE();
                if (!last) store_w((t + 1) & 1);
                PTRACE();
                if (st == NS - 1 && c + 1 < NC) {
                    __syncthreads();                            // every wave is done reading the window
                    store_patch();
                }
                __syncthreads();
                PTRACE();
            }
        }
    }

    int mpix[1] = {mpx}, lrow[1] = {p};
    float *ssl = reinterpret_cast<float *>(lds + sizeof(lds) - 2 * BN * 4);      // the staging areas are free: the loop ended on a barrier
    if (tid < BN) { ssl[tid] = ssc; ssl[BN + tid] = ssh; }
    __syncthreads();
    PTRACE();
    if (a.out_mode == 0) {
        if (a.sigmoid_from < 0) conv_epilogue_fast<4, 1>(a, acc, mpix, lrow, n0, 0, lh, ssl, BN, lds);
        else conv_epilogue<4, 1>(a, acc, mpix, n0, 0, lh, 0, ssl, BN, lds, lrow);
        PTRACE();
        __syncthreads();
        store_otile<BN, BM, NT>(a, lds, n0, 0, tid, [&](int row) {
            const int y = y0 + row / TW, x = x0 + row % TW;
            return (y < a.Ho && x < a.Wo) ? (img * a.Ho + y) * a.Wo + x : -1;
        });
    } else {
        conv_epilogue<4, 1>(a, acc, mpix, n0, 0, lh, 0, ssl, BN);
    }
    PTRACE();
}

// Which patch kernel serves a descriptor: 0 = none (the implicit-GEMM kernel only), 16 / 8 = rows of the pixel patch.
int dcn_patch_variant(const m3d_conv_bf16_desc *d)
{
    static int on = -1;                   // M3D_BF16_DCN_PATCH=0: implicit-GEMM kernel everywhere (A/B)
    if (on < 0) { const char *e = getenv("M3D_BF16_DCN_PATCH"); on = e ? atoi(e) : 1; }
    if (!on || !d->dcn_offmask || !d->wgt_f16 || !d->dcn_ws) return 0;
    if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->groups != 1 || d->wgt_img_stride != 0) return 0;
    if (d->Cin % 32 != 0 || d->Cout_pad % 128 != 0 || d->W % 16 != 0 || d->dcn_om_cs < 28 || d->dcn_om_cs % 4 != 0) return 0;
    const int v = d->H % 16 == 0 ? 16 : d->H % 8 == 0 ? 8 : 0;
    return (v && d->dcn_ws_bytes >= dcn_patch_ws_bytes(d, v)) ? v : 0;      // (one flag word per pixel tile: m3d_conv_bf16_dcn_ws_bytes)
}

long long dcn_patch_ws_bytes(const m3d_conv_bf16_desc *d, int variant)
{
    return variant ? 4ll * d->N * (d->Ho / variant) * (d->Wo / 16) : 0;
}

int launch_dcn_patch(const Bf16Args &a0, const m3d_conv_bf16_desc *d, int variant, hipStream_t st)
{
    Bf16Args a = a0;
    a.tiles_n = d->Cout_pad / 128;
    unsigned *bound = (unsigned *)d->dcn_ws;                      // one flag per pixel tile
    if (variant == 16) {
        a.tiles_m = d->N * (d->Ho / 16) * (d->Wo / 16);
        hipLaunchKernelGGL((bf16_dcn_patch_kernel<16, 9, 3>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, st, a, d->wgt_f16, bound);
    } else {
        a.tiles_m = d->N * (d->Ho / 8) * (d->Wo / 16);
        hipLaunchKernelGGL((bf16_dcn_patch_kernel<8, 6, 1>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0, st, a, d->wgt_f16, bound);
    }
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" long long m3d_conv_bf16_dcn_ws_bytes(int N, int Ho, int Wo)
{
    const int v = Ho % 16 == 0 ? 16 : Ho % 8 == 0 ? 8 : 0;
    return (v && Wo % 16 == 0) ? 4ll * N * (Ho / v) * (Wo / 16) : 0;
}
