// bf16 implicit-GEMM convolution / DCNv2 for the bs=64 configuration (BASELINE.json configs[2]): bf16 activations and
// weights in HBM, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 epilogue (folded BatchNorm + bias, residual,
// LeakyReLU / sigmoid) and bf16 stores -- every activation crosses HBM once as bf16 (in bf16 the network is
// HBM-bound unless the element-wise work is fused into the producing kernel).
//
//   GEMM view: rows of D = output channels (MFMA "A" operand = weight tile), columns of D = pixels ("B" operand = the
//   im2col / modulated bilinear gather of the input, built on the fly).  With this orientation a lane of the 32x32
//   accumulator holds ONE pixel and 4 consecutive channels per register group, so the epilogue works on channel-contiguous
//   vectors (f32x4 scale / shift / residual, packed bf16 stores) with no LDS transpose.
//   Workgroup = 256 threads = 4 waves; tile = 128 pixels x BN channels (BN in {32, 64, 128}), K-step 64 (one 128-byte line
//   per pixel / weight row); LDS tiles [rows][64 bf16] with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 so that
//   both the staging ds_write_b128 and the fragment ds_read_b128 are bank-conflict free; register-staged double buffering
//   (global loads of K-step k+1 are in flight under the MFMAs of K-step k).
//   Out-of-image taps, rows past M and the zero padding of K use buffer loads whose masked lanes read 0.
//   blockIdx -> tile mapping is XCD-aware: each XCD owns a contiguous range of tiles, and the channel tiles of one pixel tile
//   are adjacent (they re-read the same pixels from that XCD's L2).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#include "bf16_tile.h"

#ifdef BF16_TRACE
// Diagnostic build (make trace): s_memtime stamps of wave 0 of every workgroup -- per K-step: loop top, loads issued, MFMAs
// issued, staging writes done (vmcnt waits), barrier passed.  tools/bf16_conv_trace.py prints the phase averages.
static long long *g_bf16_trace = nullptr;
extern "C" void m3d_bf16_conv_set_trace(void *buf) { g_bf16_trace = (long long *)buf; }
#define BTRACE_INIT() long long *trp = a.trace ? a.trace + (size_t)blockIdx.x * 160 : nullptr; int tri = 0
#define BTRACE() do { if (trp && tid == 0 && tri < 158) trp[tri++] = __builtin_readcyclecounter(); } while (0)
// wall-clock pair (100 MHz s_memrealtime next to s_memtime) in slots 156..159: the shader clock the workgroup actually ran at
#define BTRACE_REAL(k) do { if (trp && tid == 0) { trp[156 + 2 * (k)] = __builtin_readcyclecounter(); trp[157 + 2 * (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define BTRACE_REAL(k)
#define BTRACE_INIT()
#define BTRACE()
#endif

template <int BN, bool DEFORM>      // two workgroups per CU (LDS allows it): the gather of one overlaps the MFMA section of the other
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void bf16_conv_kernel(const Bf16Args a)
{
    constexpr int BM = 128, BK = 64;
    constexpr int WN = BN >= 64 ? 2 : 1;          // waves along the channel dim
    constexpr int WM = 4 / WN;                    // waves along the pixel dim
    constexpr int TN = BN / (32 * WN);            // 32-channel MFMA tiles per wave
    constexpr int TM = BM / (32 * WM);            // 32-pixel MFMA tiles per wave
    constexpr int PB = BN / 32;                   // weight rows per thread and K-step (32 rows per pass)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * BK * 2];
    unsigned char *Ps = lds;                      // pixel tiles  [2][BM][128 B]
    unsigned char *Ws = lds + 2 * BM * BK * 2;    // weight tiles [2][BN][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WN) * (TM * 32), wn = (wave % WN) * (TN * 32);
    if constexpr (DEFORM) {
        // fallback role (bf16_dcn_patch.hip): the LDS-patch kernel has done this launch's work when the sampling window fits
        // the patch kernel left one word per patch tile: non-zero = its window did not fit.  This workgroup's 128 pixels (linear in
        // image, row, column) recompute whatever such tiles they touch -- the whole 128-pixel tile, the values of the neighbours
        // from fitting tiles included (same convolution, rounding of the other kernel) -- and nothing else runs.
        if (a.gate) {
            int need = 0;
            if (tid < BM) {
                const int tl = blockIdx.x;                          // (tile order: see below)
                const int ntl = a.tiles_m * a.tiles_n;
                const int q = ntl >> 3, r = ntl & 7, xcd = tl & 7, idx = tl >> 3;
                const int t2 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
                const int m = (t2 / a.tiles_n) * BM + tid;
                if (m < a.M) {
                    const int img = m / a.HoWo, rem = m - img * a.HoWo;
                    const int y = rem / a.Wo, x = rem - y * a.Wo;
                    need = a.gate[(img * a.gate_tpy + y / a.gate_th) * a.gate_tpx + (x >> 4)] != 0u;
                }
            }
            if (!__syncthreads_or(need)) return;
        }
    }
    BTRACE_INIT();
    BTRACE();
    BTRACE_REAL(0);

    // XCD-aware tile mapping (consecutive workgroup ids round-robin over the 8 XCDs)
    const int ntiles = a.tiles_m * a.tiles_n;
    int tile = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = tile / a.tiles_n, tile_n = tile - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int grp = blockIdx.y;

    const __bf16 *inp = (const __bf16 *)a.in + grp * a.in_goff;
    const __bf16 *wgt = (const __bf16 *)a.wgt + grp * a.wgt_goff;
    if (a.wgt_img_stride) wgt += (long long)(m0 / a.HoWo) * a.wgt_img_stride;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(inp, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rwgt = make_rsrc(wgt, a.wgt_bytes);
    float ssc = 1.f, ssh = 0.f;              // epilogue scale / shift of channel n0 + tid, fetched now, used after the K loop
    if (tid < BN && n0 + tid < a.Cout) {
        if (a.scale) ssc = a.scale[grp * a.ss_goff + n0 + tid];
        if (a.shift) ssh = a.shift[grp * a.ss_goff + n0 + tid];
    }

    // ---- staging map: thread = (row within a 32-row pass, 16-byte chunk of the 128-byte K line) ----------------------
    const int chunk = tid & 7, rsub = tid >> 3;
    int pix_base[4], hi0[4], wi0[4];
    bool rvalid[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = m0 + p * 32 + rsub;
        rvalid[p] = m < a.M;
        const int mm = rvalid[p] ? m : 0;
        const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        pix_base[p] = n * a.H * a.W;
        hi0[p] = ho * a.stride - a.pad;
        wi0[p] = wo * a.stride - a.pad;
    }
    // Deformable mode, uniform K: the sampling state of a (pixel, tap) -- 4 corner weights with the mask folded in, 4 corner
    // offsets -- is needed by the 8 lanes that stage the pixel's 128-byte line.  Lane `chunk` of the row builds the state of
    // piece chunk & 3 and the lanes trade states with ds_bpermute; the offsets / mask of the NEXT tap are fetched a tap ahead.
    // (Each lane building all 4 of its pieces from offsets read at the tap change: ~4800 cycles on those K-steps against
    // ~1650 on the others, tools/bf16_conv_trace.py; now ~2800.)
    const int sp = chunk & 3;                                       // piece this lane produces the state of
    const int spix = m0 + sp * 32 + rsub;
    const bool sprod = DEFORM && spix < a.M;
    const int sinv = sign_smear(a.M - 1 - spix);                    // all ones: no such pixel, the state is "nothing"
    int sh0 = 0, sw0 = 0, sbase = 0;                                // that pixel's tap-(0, 0) input position / image base
    if (sprod) {
        const int n = spix / a.HoWo, rem = spix - n * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        sbase = n * a.H * a.W; sh0 = ho * a.stride - a.pad; sw0 = wo * a.stride - a.pad;
    }
    float omn[3] = {0.f, 0.f, 0.f};                                 // (dh, dw, mask) of the tap whose state is built next
    auto fetch_om = [&](int tap) __attribute__((always_inline)) {
        if constexpr (DEFORM) {
            const int KK = a.kh * a.kw;
            if (sprod && tap < KK) {
                const float *omp = a.om + (size_t)spix * a.om_cs;
                omn[0] = omp[2 * tap]; omn[1] = omp[2 * tap + 1]; omn[2] = omp[2 * KK + tap];
            }
        }
    };
    if (DEFORM && a.uniform_k) fetch_om(0);
    unsigned woff[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p)
        woff[p] = ((unsigned)min(n0 + p * 32 + rsub, a.Cout_pad - 1) * (unsigned)(a.KT * BK) + (unsigned)chunk * 8u) * 2u;

    u32x4 rp[4][DEFORM ? 4 : 1];
    unsigned poff[4] = {0u, 0u, 0u, 0u};
    u32x4 rw[PB];
    float bw[DEFORM ? 4 : 1][4];
    unsigned doff[DEFORM ? 4 : 1][4];
    int samp_tap = -1;

    auto load_tile = [&](int kt) {
        // Fast path (Cin % 64 == 0): the whole K-step lies in ONE tap and the channel offset is wave-uniform -> it rides in the
        // SGPR offset of the buffer loads; per-lane byte offsets / sampling state change only when the tap does (every Cin/64
        // K-steps).  A VALU instruction issued next to a SIMD's MFMA stream costs that stream ~12 cycles on this part
        // (tools/ubench/mfma_side_cost.hip): address math must not run per K-step.
        const int kb = kt * BK;
        const bool uni = a.uniform_k != 0;
        int tap, c;                      // uniform path: scalars; general path: per lane (a K-step may span several taps)
        {
            const int k0 = uni ? kb : kb + chunk * 8;
            if (a.kh * a.kw == 1) { tap = 0; c = k0; }
            else { tap = k0 >> a.log2Cin; c = k0 & (a.Cin - 1); }
        }
        const bool kvalid = uni || (kb + chunk * 8 < a.K);
        const bool fresh = tap != samp_tap;
        int ti = 0, tj = 0;
        if (!uni || fresh) {
            ti = a.kw == 1 ? tap : (a.kw == 3 ? (tap * 11) >> 5 : tap / a.kw);
            tj = tap - ti * a.kw;
        }
        if constexpr (!DEFORM) {
            if (uni) {
                if (fresh) {
                    samp_tap = tap;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int hi = hi0[p] + ti, wi = wi0[p] + tj;
                        const bool ok = rvalid[p] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                        poff[p] = ok ? ((unsigned)(pix_base[p] + hi * a.W + wi) * (unsigned)a.in_cs + (unsigned)chunk * 8u) * 2u
                                     : M3D_BUF_OOB;
                    }
                }
                const unsigned so = (unsigned)c * 2u;
#pragma unroll
                for (int p = 0; p < 4; ++p) rp[p][0] = buf_load_u32x4(rin, poff[p], so);
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int hi = hi0[p] + ti, wi = wi0[p] + tj;
                    const bool ok = kvalid && rvalid[p] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                    const unsigned off = ok ? ((unsigned)(pix_base[p] + hi * a.W + wi) * (unsigned)a.in_cs + (unsigned)c) * 2u
                                            : M3D_BUF_OOB;
                    rp[p][0] = buf_load_u32x4(rin, off, 0);
                }
            }
        } else {
            // sampling state of a pixel for a tap (dcn_v2_im2col_cuda.cu:18-47,150-178): corner weights (mask folded in) and
            // corner pixel offsets inside the image
            auto sample = [&](int h0, int w0, float dh, float dw, float mk, int off, float (&w)[4], int (&o)[4]) __attribute__((always_inline)) {
                int drop[4];
                dcn_corners((float)h0 + dh, (float)w0 + dw, a.H, a.W, off, w, o, drop);     // no SGPR lane masks: common.h
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w[q] *= mk;
                    o[q] &= ~drop[q];                                                       // dropped corner: pixel 0, weight 0
                }
            };
            if (fresh && uni) {
                samp_tap = tap;
                {
                    float w[4] = {0.f, 0.f, 0.f, 0.f};
                    int o[4] = {0, 0, 0, 0};
                    sample(sh0 + ti, sw0 + tj, omn[0], omn[1], omn[2], sinv, w, o);
                    u32x4 sw_, so_;
#pragma unroll
                    for (int q = 0; q < 4; ++q) sw_[q] = __float_as_uint(w[q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) so_[q] = (unsigned)(sbase + o[q]) * (unsigned)a.in_cs * 2u;
                    fetch_om(tap + 1);
                    const unsigned lane_c = (unsigned)chunk * 16u;
                    const int src0 = (lane & ~7) * 4;
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bw[p][q] = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(src0 + p * 4, (int)sw_[q]));
                            doff[p][q] = (unsigned)__builtin_amdgcn_ds_bpermute(src0 + p * 4, (int)so_[q]) + lane_c;
                        }
                }
            } else if (fresh) {      // general K: the tap is per lane, every lane builds the state of its 4 pieces
                samp_tap = tap;
                const int KK = a.kh * a.kw;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float w[4] = {0.f, 0.f, 0.f, 0.f};
                    int o[4] = {0, 0, 0, 0};
                    if (rvalid[p] && kvalid) {
                        const float *omp = a.om + (size_t)(m0 + p * 32 + rsub) * a.om_cs;
                        sample(hi0[p] + ti, wi0[p] + tj, omp[2 * tap], omp[2 * tap + 1], omp[2 * KK + tap], 0, w, o);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bw[p][q] = w[q];
                        doff[p][q] = (unsigned)(pix_base[p] + o[q]) * (unsigned)a.in_cs * 2u;
                    }
                }
            }
            if (uni) {
                const unsigned so = (unsigned)c * 2u;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) rp[p][q] = buf_load_u32x4(rin, doff[p][q], so);
            } else {
                const unsigned cb = (unsigned)c * 2u;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) rp[p][q] = buf_load_u32x4(rin, doff[p][q] + cb, 0);
            }
        }
        const unsigned wso = (unsigned)kt * (BK * 2u);
#pragma unroll
        for (int p = 0; p < PB; ++p) rw[p] = buf_load_u32x4(rwgt, woff[p], wso);
    };

    auto store_tile = [&](int buf) {
        unsigned char *Pb = Ps + buf * BM * 128, *Wb = Ws + buf * BN * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = p * 32 + rsub;
            u32x4 v;
            if constexpr (!DEFORM) {
                v = rp[p][0];
            } else {
                // (w1*v1 + w2*v2 + w3*v3 + w4*v4) with the modulation mask folded into the corner weights, fp32, then one
                // rounding to bf16 (dcn_v2_im2col_cuda.cu:44-46,174)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 v1 = unpack_bf16(rp[p][0][e]), v2 = unpack_bf16(rp[p][1][e]);
                    const f32x2 v3 = unpack_bf16(rp[p][2][e]), v4 = unpack_bf16(rp[p][3][e]);
                    const f32x2 r = v1 * bw[p][0] + v2 * bw[p][1] + v3 * bw[p][2] + v4 * bw[p][3];
                    v[e] = pack_bf16(r[0], r[1]);
                }
            }
            *reinterpret_cast<u32x4 *>(Pb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = v;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int row = p * 32 + rsub;
            *reinterpret_cast<u32x4 *>(Wb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)) = rw[p];
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int l31 = lane & 31, lh = lane >> 5;
    const int sw = (l31 >> 1) & 7;                 // fragment rows are (multiple of 32) + l31: the swizzle term is per lane
    for (int kt = 0; kt < a.KT; ++kt) {
        const int buf = kt & 1;
        BTRACE();
        if (kt + 1 < a.KT) load_tile(kt + 1);
        BTRACE();
        const unsigned char *Pb = Ps + buf * BM * 128 + (wm + l31) * 128;
        const unsigned char *Wb = Ws + buf * BN * 128 + (wn + l31) * 128;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int co = ((2 * s + lh) ^ sw) << 4;
            bf16x8 fw[TN], fp[TM];
#pragma unroll
            for (int j = 0; j < TN; ++j) fw[j] = *reinterpret_cast<const bf16x8 *>(Wb + j * 32 * 128 + co);
#pragma unroll
            for (int i = 0; i < TM; ++i) fp[i] = *reinterpret_cast<const bf16x8 *>(Pb + i * 32 * 128 + co);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fp[i], acc[j][i], 0, 0, 0);
        }
        BTRACE();
        if (kt + 1 < a.KT) store_tile(buf ^ 1);
        BTRACE();
        __syncthreads();
    }
    BTRACE();
    BTRACE_REAL(1);

    int mpix[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm + i * 32 + l31;
        mpix[i] = m < a.M ? m : -1;
    }
    // affine parameters of the tile's channels -> LDS (the staging area is free: the loop ended on a barrier), then the bf16
    // output tile through LDS into 128-byte row segments
    float *ssl = reinterpret_cast<float *>(lds + sizeof(lds) - 2 * BN * 4);
    if (tid < BN) { ssl[tid] = ssc; ssl[BN + tid] = ssh; }
    __syncthreads();
    BTRACE();
    if (a.out_mode == 0) {
        int lrow[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) lrow[i] = wm + i * 32 + l31;
        if (a.sigmoid_from < 0) conv_epilogue_fast<TN, TM>(a, acc, mpix, lrow, n0, wn, lh, ssl, BN, lds);
        else conv_epilogue<TN, TM>(a, acc, mpix, n0, wn, lh, grp, ssl, BN, lds, lrow);
        BTRACE();
        __syncthreads();
        BTRACE();
        store_otile<BN, BM, 256>(a, lds, n0, grp, tid, [&](int row) { return m0 + row < a.M ? m0 + row : -1; });
    } else {
        conv_epilogue<TN, TM>(a, acc, mpix, n0, wn, lh, grp, ssl, BN);
    }
    BTRACE();
}

// =====================================================================================================================
// 3x3 / stride 1 / pad 1 convolution with the input patch resident in LDS ("halo tile").
//
// Why (in-kernel timeline of the generic kernel, tools/bf16_conv_trace.py, 256 -> 256 @ 24x80, bs 64): a K-step takes 2360
// cycles of which the wave spends 764 ISSUING its 8 global loads and 448 waiting for / writing them to LDS, against 820 in the
// MFMA section.  The generic tile re-stages its 128 pixels for each of the 9 taps: per workgroup K-step that is 32
// ds_write_b128 wave-instructions (13 LDS-path cycles each, MI355X_MICROARCH.md section LDS) + 64 ds_read_b128 (4 each) = 672
// LDS cycles against 512 MFMA cycles per SIMD -- the LDS store path, not the matrix core, sets the pace.  Here a workgroup
// owns an 8 x TW patch of output pixels and stages the 10 x (TW + 2) input patch of a 64-channel chunk ONCE; the 9 taps read
// their B fragments from it at shifted addresses (pixel rows of 144 bytes: 16 consecutive pixels of a patch row cover all 64
// banks exactly once, and the tap shift is an IMMEDIATE offset -- no swizzle arithmetic); only the weights are staged per
// (tap, chunk).  K order is (chunk, tap) instead of (tap, chunk): fp32 accumulation order changes, nothing else.
//
// Lane -> pixel map inside a 32-pixel MFMA tile: ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+32); each group is given 16 CONSECUTIVE pixels of one patch row, so every fragment read is
// conflict-free (the identity map puts 8 + 8 pixels of two rows in a group: 2-way conflicts on every read).
#define HT_PS 144                        // bytes per halo pixel (128 + 16 pad)
// WK = channels of a weight K-step: 64, or 32 (two half-steps per tap: half the weight staging buffers, so that the 8 x 16 tile
// fits three workgroups per CU -- at two, the issue / staging / barrier phases of a K-step (800 cycles) are not covered by the
// other workgroup's MFMA section (740), tools/bf16_conv_trace.py).
template <int BN, int TW, int BM, int WAVES, int WK = 64>   // LDS allows 2-3 workgroups per CU: hold the register file to that
__global__ __launch_bounds__(WAVES * 64)
__attribute__((amdgpu_waves_per_eu(WK == 32 ? 3 : WAVES / 2, (WK == 32 ? 3 : WAVES / 2 + (BN * BM <= 64 * 128) + (BN * BM <= 32 * 128)))))
void bf16_conv3x3_halo_kernel(const Bf16Args a)
{
    constexpr int WKB = WK * 2;                        // bytes per weight row per step
    constexpr int NH = 64 / WK;                        // weight half-steps per (tap, 64-channel chunk)
    constexpr int TH = BM / TW, HW = TW + 2, HPIX = (TH + 2) * HW;
    constexpr int NT = WAVES * 64, RPP = NT / 8;       // threads; pixels (or weight rows) staged per pass
    constexpr int WN = BN >= 64 ? 2 : 1, WM = WAVES / WN;   // BN 32 (offset / mask convs: HBM-bound): every wave takes all channels
    constexpr int TN = BN / (32 * WN), TM = BM / (32 * WM);
    constexpr int PB = BN * (WKB / 16) / NT;           // weight pieces (16 B) per thread per K-step
    static_assert(PB >= 1 && TM >= 1 && TN >= 1, "tile shape");
    constexpr int HP = (HPIX * 8 + NT - 1) / NT;       // halo pieces per thread per chunk
    constexpr int HBYTES = HPIX * HT_PS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[HBYTES + 2 * BN * WKB];
    unsigned char *Hs = lds, *Ws = lds + HBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WN) * (TM * 32), wn = (wave % WN) * (TN * 32);
    const int l31 = lane & 31, lh = lane >> 5;
    BTRACE_INIT();
    BTRACE();
    BTRACE_REAL(0);
    const int ntiles = a.tiles_m * a.tiles_n;
    int tile = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = tile / a.tiles_n, tile_n = tile - tile_m * a.tiles_n;
    const int n0 = tile_n * BN;
    const int tpx = (a.Wo + TW - 1) / TW, tpy = (a.Ho + TH - 1) / TH;
    const int img = tile_m / (tpx * tpy), trem = tile_m - img * tpx * tpy;
    const int y0 = (trem / tpx) * TH, x0 = (trem % tpx) * TW;

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rwgt = make_rsrc(a.wgt, a.wgt_bytes);
    float ssc = 1.f, ssh = 0.f;              // epilogue scale / shift of channel n0 + tid, fetched now, used after the K loop
    if (tid < BN && n0 + tid < a.Cout) {
        if (a.scale) ssc = a.scale[n0 + tid];
        if (a.shift) ssh = a.shift[n0 + tid];
    }

    // ---- halo staging map: piece q = tid + NT*p -> halo pixel (tid >> 3) + RPP*p, 16-byte chunk tid & 7 ------------------
    const int chunk = tid & 7, rsub = tid >> 3;
    // Byte offsets are rebuilt per chunk from (hy0, hx0) with a few VALU ops per piece -- once per 9 K-steps -- instead of
    // living in HP registers: the 256-pixel tiles need every VGPR for accumulators and the in-flight patch.
    const int hy0 = rsub / HW, hx0 = rsub - hy0 * HW;
    const unsigned hbase = ((unsigned)((img * a.H + y0 - 1) * a.W + x0 - 1) * (unsigned)a.in_cs + (unsigned)chunk * 8u) * 2u;
    const int hdst0 = rsub * HT_PS + chunk * 16;                                        // + p * RPP * HT_PS
    // weight staging map: WK 64: thread = (row tid >> 3, piece tid & 7) of 128-byte rows, piece XOR (row >> 1) & 7;
    //                     WK 32: thread = (row tid >> 2, piece tid & 3) of  64-byte rows, piece XOR (row >> 2) & 3
    constexpr int WPR = WKB / 16, WRPP = NT / WPR;                                       // pieces per row, rows per pass
    const int wchunk = tid % WPR, wrsub = tid / WPR;
    const unsigned woff0 = ((unsigned)(n0 + wrsub) * (unsigned)(a.KT * 64) + (unsigned)wchunk * 8u) * 2u;
    const unsigned wrow_step = (unsigned)WRPP * (unsigned)(a.KT * 64) * 2u;            // bytes between passes (scalar)
    const int wdst0 = wrsub * WKB + ((wchunk ^ (WK == 64 ? (wrsub >> 1) & 7 : (wrsub >> 2) & 3)) << 4);   // + p * WRPP * WKB
    u32x4 rh[HP], rw[3][PB];                                    // weights: three register sets, loads run two K-steps ahead
    const int NC = a.Cin >> 6;                                  // 64-channel chunks
    auto load_halo = [&](int c) __attribute__((always_inline)) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(c) * 128u;
        int hy = hy0, hx = hx0;
#pragma unroll
        for (int p = 0; p < HP; ++p) {
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = hy < TH + 2 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = ok ? hbase + (unsigned)(hy * a.W + hx) * (unsigned)(a.in_cs * 2) : M3D_BUF_OOB;
            rh[p] = buf_load_u32x4(rin, off, so);
            hx += RPP % HW; hy += RPP / HW;
            if (hx >= HW) { hx -= HW; ++hy; }
        }
    };
    auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < HP; ++p)
            if ((p + 1) * RPP <= HPIX || rsub + RPP * p < HPIX)
                *reinterpret_cast<u32x4 *>(Hs + hdst0 + p * RPP * HT_PS) = rh[p];
    };
    // weights of K-step u (= tap * NH + half) of chunk c -> register set R (compile-time: the steps of a chunk are unrolled and
    // their number is a multiple of 3)
    constexpr int SPC = 9 * NH;                                 // K-steps per 64-channel chunk
    auto load_w = [&](int c, int u, auto rtag) __attribute__((always_inline)) {
        constexpr int R = decltype(rtag)::value;
        if (u >= SPC) { u -= SPC; ++c; }
        if (c >= NC) return;
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((u / NH) * a.Cin + c * 64 + (u % NH) * WK) * 2u;
#pragma unroll
        for (int p = 0; p < PB; ++p) rw[R][p] = buf_load_u32x4(rwgt, woff0, so + (unsigned)p * wrow_step);
    };
    auto store_w = [&](int buf, auto rtag) __attribute__((always_inline)) {
        constexpr int R = decltype(rtag)::value;
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<u32x4 *>(Ws + buf * BN * WKB + wdst0 + p * WRPP * WKB) = rw[R][p];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    // position of this lane's pixel inside its 32-pixel MFMA tile (see the header): ds_read_b128 lane group -> one patch row
    int lpos;
    if (a.lane_perm == 0) lpos = l31;
    else if (l31 < 4) lpos = l31;
    else if (l31 < 12) lpos = 16 + (l31 - 4);
    else if (l31 < 16) lpos = 4 + (l31 - 12);
    else if (l31 < 20) lpos = 24 + (l31 - 16);
    else if (l31 < 28) lpos = 8 + (l31 - 20);
    else lpos = 28 + (l31 - 28);
    int pbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = wm + i * 32 + lpos;
        pbase[i] = ((p / TW) * HW + (p % TW)) * HT_PS + lh * 16;
    }

#define RT(n) std::integral_constant<int, (n) % 3>{}
    load_halo(0);
    load_w(0, 0, RT(0));
    load_w(0, 1, RT(1));
    store_halo();
    store_w(0, RT(0));
    __syncthreads();
    const int swk = WK == 64 ? (l31 >> 1) & 7 : (l31 >> 2) & 3;     // swizzle term of this lane's weight rows
    int t = 0;                                                  // step index; weight buffer t & 1
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int u = 0; u < SPC; ++u, ++t) {
            constexpr int dummy = 0; (void)dummy;
            const int tap = u / NH, half = u % NH;
            const bool last = (c == NC - 1) && u == SPC - 1;
            BTRACE();
            // step t: loads of step t + 2 go out, step t + 1's weights (fetched one step ago) are staged under this step's
            // MFMAs: a fetch issued and consumed inside ONE step (256-512 MFMA cycles) waits out most of its memory round trip
            if (u % 3 == 0) load_w(c, u + 2, RT(2)); else if (u % 3 == 1) load_w(c, u + 2, RT(0)); else load_w(c, u + 2, RT(1));
            if (tap == 6 && half == 0 && c + 1 < NC) load_halo(c + 1);   // the next chunk's patch travels under the last taps of this one
            __builtin_amdgcn_sched_barrier(0);
            BTRACE();
            const unsigned char *Wb = Ws + (t & 1) * BN * WKB + (wn + l31) * WKB;
            const int toff = ((tap / 3) * HW + (tap % 3)) * HT_PS + half * WKB;  // compile-time per unrolled step
#pragma unroll
            for (int s = 0; s < WK / 16; ++s) {
                const int co = ((2 * s + lh) ^ swk) << 4;
                bf16x8 fw[TN], fp[TM];
#pragma unroll
                for (int j = 0; j < TN; ++j) fw[j] = *reinterpret_cast<const bf16x8 *>(Wb + j * 32 * WKB + co);
#pragma unroll
                for (int i = 0; i < TM; ++i) fp[i] = *reinterpret_cast<const bf16x8 *>(Hs + pbase[i] + toff + s * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fp[i], acc[j][i], 0, 0, 0);
            }
            BTRACE();
            if (!last) {
                if (u % 3 == 0) store_w((t + 1) & 1, RT(1)); else if (u % 3 == 1) store_w((t + 1) & 1, RT(2)); else store_w((t + 1) & 1, RT(0));
            }
            if (u == SPC - 1 && c + 1 < NC) {
                __syncthreads();                                // every wave is done reading the patch
                store_halo();
            }
            BTRACE();
            __syncthreads();
        }
    }
#undef RT
    BTRACE();
    BTRACE_REAL(1);
    int mpix[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int p = wm + i * 32 + lpos;
        const int y = y0 + p / TW, x = x0 + p % TW;
        mpix[i] = (y < a.Ho && x < a.Wo) ? (img * a.Ho + y) * a.Wo + x : -1;
    }
    float *ssl = reinterpret_cast<float *>(lds + sizeof(lds) - 2 * BN * 4);      // the staging areas are free: the loop ended on a barrier
    if (tid < BN) { ssl[tid] = ssc; ssl[BN + tid] = ssh; }
    __syncthreads();
    if (a.out_mode == 0) {
        int lrow[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) lrow[i] = wm + i * 32 + lpos;
        if (a.sigmoid_from < 0) conv_epilogue_fast<TN, TM>(a, acc, mpix, lrow, n0, wn, lh, ssl, BN, lds);
        else conv_epilogue<TN, TM>(a, acc, mpix, n0, wn, lh, 0, ssl, BN, lds, lrow);
        __syncthreads();
        store_otile<BN, BM, NT>(a, lds, n0, 0, tid, [&](int row) {
            const int y = y0 + row / TW, x = x0 + row % TW;
            return (y < a.Ho && x < a.Wo) ? (img * a.Ho + y) * a.Wo + x : -1;
        });
    } else {
        conv_epilogue<TN, TM>(a, acc, mpix, n0, wn, lh, 0, ssl, BN);
    }
}

static int ilog2_exact(int v)
{
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// Which kernel m3d_conv_bf16_forward runs for a descriptor (m3d_conv_bf16_variant: 3 / 4 = LDS-patch DCNv2 8 / 16 rows, 5 = wave tile,
// 6 = the 1x1 DCNv2 kernel, 8 = the persistent 64 -> 64 kernel): 0 = generic implicit-GEMM tile, 1 = halo tile 8 x 16 pixels / 4
// waves, 2 = halo tile 8 x 32 pixels / 8 waves.  3x3 / stride 1 / pad 1 on a 64-multiple of channels goes to the halo-tile
// kernel when its patches tile the map well.
// M3D_BF16_HALO: 0 = generic kernel everywhere, 1 = default choice, 2 = 128-pixel patches only, +16 = identity lane map
// (A/B of the LDS bank-conflict fix).  tools/bf16_conv_bench.py, TFLOP/s generic -> 8x16 patch / 4 waves -> 8x32 patch /
// 8 waves: 64->64 @96x320 403 -> 599 -> 664; 128->128 @48x160 543 -> 712 -> 792; 256->256 @24x80 634 -> 845 -> 854
// (8x32 / 4 waves with 128x64 wave tiles: 819, spills).
static int halo_env()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("M3D_BF16_HALO"); v = e ? atoi(e) : 1; }
    return v;
}
static int conv_bf16_variant(const m3d_conv_bf16_desc *d, long long *tiles)
{
    const int halo_on = halo_env() & 15;
    const int bn = d->Cout_pad % 128 == 0 ? 128 : (d->Cout_pad % 64 == 0 ? 64 : 32);
    if (!halo_on || d->dcn_offmask || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->Cin % 64 != 0 ||
        d->groups != 1 || d->wgt_img_stride != 0)
        return 0;
    const int ho = d->H, wo = d->W;
    const long long M = (long long)d->N * ho * wo;
    const long long t16 = (long long)cdiv(wo, 16) * cdiv(ho, 8) * d->N, t32 = (long long)cdiv(wo, 32) * cdiv(ho, 8) * d->N;
    const double e16 = (double)M / (double)(t16 * 128), e32 = (double)M / (double)(t32 * 256);
    // the 256-pixel patch halves the weight staging per MFMA; it needs >= 2 workgroups per CU in flight to pay
    // 128-channel tiles: the 8 x 16 patch with 32-channel weight half-steps runs three workgroups per CU and beats the 8 x 32 /
    // 8-wave tile (128 -> 128 @ 48x160: 904 vs 792 TFLOP/s); 64-channel tiles: 8 x 32 / 8 waves (708 vs 668)
    const bool big = halo_on != 2 && bn == 64 && e32 >= 0.9 * e16 && t32 * (d->Cout_pad / bn) >= 1024;
    if ((big ? e32 : e16) < 0.8) return 0;
    if (tiles) *tiles = big ? t32 : t16;
    return big ? 2 : 1;
}
extern "C" int m3d_conv_bf16_variant(const m3d_conv_bf16_desc *d)
{
    if (!d) return -1;
    if (d->dcn_offmask) {
        if (dcn1x1_applicable(d)) return 6;
        const int pv = dcn_patch_variant(d);
        return pv == 16 ? 4 : (pv == 8 ? 3 : 0);
    }
    if (conv_wide_applicable(d)) return 5;
    if (conv_c64_applicable(d)) return 8;
    return conv_bf16_variant(d, nullptr);
}

extern "C" int m3d_conv_bf16_forward(const m3d_conv_bf16_desc *d, m3d_stream_t stream)
{
    M3D_REQUIRE(d && d->in && d->wgt && d->out, "conv_bf16: null pointer");
    M3D_REQUIRE(d->Cin % 8 == 0 && d->in_cs % 8 == 0, "conv_bf16: Cin (%d) and in_cs (%d) must be multiples of 8", d->Cin, d->in_cs);
    M3D_REQUIRE(d->Cout_pad % 32 == 0 && d->Cout <= d->Cout_pad, "conv_bf16: Cout_pad %% 32");
    M3D_REQUIRE(d->groups >= 1, "conv_bf16: groups >= 1");
    const int taps = d->kh * d->kw;
    const int lg = ilog2_exact(d->Cin);
    M3D_REQUIRE(taps == 1 || lg >= 3, "conv_bf16: a %dx%d kernel needs a power-of-two Cin (got %d)", d->kh, d->kw, d->Cin);
    const int ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1, wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    M3D_REQUIRE(ho == d->Ho && wo == d->Wo, "conv_bf16: Ho/Wo (%d, %d) do not match the geometry (%d, %d)", d->Ho, d->Wo, ho, wo);
    const int K = taps * d->Cin;
    M3D_REQUIRE(d->Kpad % 64 == 0 && d->Kpad >= K, "conv_bf16: Kpad (%d) must be a multiple of 64 >= K (%d)", d->Kpad, K);
    const long long M = (long long)d->N * ho * wo;
    const long long in_bytes = (long long)d->N * d->H * d->W * d->in_cs * 2;
    M3D_REQUIRE(in_bytes < 0x7FFFFFFFLL && M < 0x7FFFFFFFLL, "conv_bf16: input view must be < 2 GiB");
    if (d->out_mode == 0) M3D_REQUIRE(d->out_cs % 8 == 0 && ((uintptr_t)d->out & 15) == 0, "conv_bf16: bf16 output needs out_cs %% 8 == 0 and 16-byte alignment");
    if (d->out_mode == 1) M3D_REQUIRE(d->out_cs % 4 == 0 && ((uintptr_t)d->out & 15) == 0, "conv_bf16: fp32 NHWC output needs out_cs %% 4 == 0");
    if (d->res) M3D_REQUIRE(d->res_cs % 4 == 0 && ((uintptr_t)d->res & 7) == 0, "conv_bf16: residual view alignment");
    if (d->wgt_img_stride) M3D_REQUIRE((ho * wo) % 128 == 0, "conv_bf16: per-image weights need Ho*Wo %% 128 == 0");
    if (d->dcn_offmask) M3D_REQUIRE(d->stride == 1 && d->groups == 1 && taps <= 9, "conv_bf16: deformable mode is stride 1, ungrouped, <= 9 taps");

    Bf16Args a;
    a.in = d->in; a.wgt = d->wgt; a.out = d->out; a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.om = d->dcn_offmask;
    a.wgt_img_stride = d->wgt_img_stride; a.out_img_stride = d->out_img_stride;
    a.in_goff = d->in_group_off; a.wgt_goff = d->wgt_group_off; a.out_goff = d->out_group_off; a.ss_goff = d->ss_group_off;
    a.in_bytes = (unsigned)in_bytes;
    const long long wrows = d->wgt_img_stride ? d->wgt_img_stride : (long long)d->Cout_pad * d->Kpad;
    a.wgt_bytes = (unsigned)(wrows * 2);
    a.res_bytes = 0;
    a.in_cs = d->in_cs; a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.log2Cin = lg < 0 ? 0 : lg;
    a.Cout = d->Cout; a.Cout_pad = d->Cout_pad; a.K = K; a.KT = d->Kpad / 64; a.kh = d->kh; a.kw = d->kw; a.stride = d->stride;
    a.pad = d->pad; a.Ho = ho; a.Wo = wo; a.HoWo = ho * wo; a.M = (int)M;
    a.out_cs = d->out_cs; a.res_cs = d->res_cs; a.om_cs = d->dcn_om_cs;
    a.out_mode = d->out_mode; a.res_mode = d->res_mode; a.act = d->act; a.sigmoid_from = d->sigmoid_from;
    const int bn = d->Cout_pad % 128 == 0 ? 128 : (d->Cout_pad % 64 == 0 ? 64 : 32);
    a.tiles_m = cdiv(M, 128); a.tiles_n = d->Cout_pad / bn;
    a.uniform_k = (d->Cin % 64 == 0 && K % 64 == 0) ? 1 : 0;
    a.gate = nullptr; a.gate_th = a.gate_tpx = a.gate_tpy = 0;
#ifdef BF16_TRACE
    a.trace = g_bf16_trace;
#endif
    const dim3 grid(a.tiles_m * a.tiles_n, d->groups), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool deform = d->dcn_offmask != nullptr;
    if (conv_wide_applicable(d)) return launch_conv_wide(a, d, st);      // 128 x 128 wave tiles (bf16_conv_wide.hip)
    if (deform && dcn1x1_applicable(d)) return launch_dcn1x1(a, d, st);  // 1x1 DCNv2 128 -> 128 (center_align; bf16_dcn1x1.hip)
    if (conv_c64_applicable(d)) return launch_conv_c64(a, d, st);        // 3x3 64 -> 64, persistent workgroups (level2; bf16_conv_c64.hip)
    long long htiles = 0;
    const int variant = conv_bf16_variant(d, &htiles);
    a.lane_perm = (halo_env() & 16) ? 0 : 1;
    if (variant) {
        a.tiles_m = (int)htiles;
        const dim3 hgrid(a.tiles_m * a.tiles_n);
#define HLAUNCH(BN_, TW_, BM_, WV_, WK_) hipLaunchKernelGGL((bf16_conv3x3_halo_kernel<BN_, TW_, BM_, WV_, WK_>), hgrid, dim3(WV_ * 64), 0, st, a)
        static int wk32 = -1;                 // M3D_BF16_HALO_WK=64: 64-channel weight steps for the 8 x 16 tile too (A/B)
        if (wk32 < 0) { const char *e = getenv("M3D_BF16_HALO_WK"); wk32 = (e && atoi(e) == 64) ? 0 : 1; }
        if (variant == 2) HLAUNCH(64, 32, 256, 8, 64);
        else if (bn == 128) { if (wk32) HLAUNCH(128, 16, 128, 4, 32); else HLAUNCH(128, 16, 128, 4, 64); }
        else if (bn == 64) HLAUNCH(64, 16, 128, 4, 64);
        else HLAUNCH(32, 16, 128, 4, 64);
#undef HLAUNCH
        M3D_LAUNCH_CHECK();
        return M3D_OK;
    }
    // DCNv2 3x3 with an fp16 weight copy and a flag workspace: the LDS-patch kernel, one word per patch tile ("my window did not fit"),
    // and this file's implicit-GEMM kernel behind it for the tiles that carry the flag (bf16_dcn_patch.hip)
    const int pvar = deform ? dcn_patch_variant(d) : 0;
    if (pvar) {
        int rc;
        // a fallback workgroup recomputes its WHOLE 128-pixel tile when any patch tile it touches carries the flag: pixels the patch
        // kernel already wrote are written a second time (with this kernel's rounding), which is idempotent only while the
        // residual does not alias the output -- an in-place residual would be added twice
        M3D_REQUIRE(d->res == nullptr || d->res != d->out, "m3d_conv_bf16_forward: the LDS-patch DCNv2 path needs res != out");
        if ((rc = launch_dcn_patch(a, d, pvar, st))) return rc;
        a.gate = (const unsigned *)d->dcn_ws; a.gate_th = pvar; a.gate_tpx = d->Wo / 16; a.gate_tpy = d->Ho / pvar;
    }
#define LAUNCH(BN_, DF_) hipLaunchKernelGGL((bf16_conv_kernel<BN_, DF_>), grid, block, 0, st, a)
    if (bn == 128) { if (deform) LAUNCH(128, true); else LAUNCH(128, false); }
    else if (bn == 64) { if (deform) LAUNCH(64, true); else LAUNCH(64, false); }
    else { if (deform) LAUNCH(32, true); else LAUNCH(32, false); }
#undef LAUNCH
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
