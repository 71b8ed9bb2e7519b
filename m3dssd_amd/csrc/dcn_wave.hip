// Wave-granular implicit-GEMM convolution / deformable convolution for gfx950: ONE WAVE = 32 output pixels x 128 (or 64) output
// channels, 64-thread workgroups, no workgroup barriers; the accumulators (64 AGPRs) and operands leave room for two waves
// per SIMD, which cover each other's load / LDS latency.
//
// Design notes (measured on MI355X, tools/ubench/):
//   * mfma_side_cost: VALU instructions of any wave of a SIMD delay its MFMA stream by ~11 cycles each, LDS and vector-
//     memory instructions do not -> the per-channel work is kept to the 8 packed ops of the bilinear combine; padding
//     and out-of-image corners cost nothing (buffer loads return 0.0f past num_records).
//   * gather_throughput: a 16-byte load whose 64 lanes touch 32 different 128-byte lines (the MFMA A-operand map: lane =
//     pixel) costs 64 TA cycles per wave, the map "8 lanes per line" costs 20.  So the gather uses the line map -- lane
//     (p = lane >> 3, c = lane & 7) fetches chunk c of the 32-channel line of pixels 8g + p, g = 0..3 -- the bilinear
//     combine (model/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:18-47,174, mask folded into the corner weights once per tap)
//     runs in that layout, and the wave's private 4 KB LDS tile transposes the result into the A-operand map
//     (XOR-swizzled 16-byte slots: conflict-free ds_write_b128 / ds_read_b128, no padding).
//   * the sampling state of a (pixel, tap) is computed once by the lane that owns the pixel in the MFMA map and handed to
//     the 8 gather lanes of that pixel through LDS (it would otherwise be recomputed 8 times on the VALU).
//   * B operands: [Cout_pad, kh*kw*Cin] tap-major weights in MFMA-fragment order [Cout_pad/32][K/8][h=2][r=32][t=4]
//     (m3dssd_amd.engine.pack_frag): one coalesced 1 KB load per column tile and k-group, two register sets.
// A step covers the 32 channels (one 128-byte line) of one tap: 4 k-groups x 16 MFMAs.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

struct ConvWaveArgs {
    const float *in, *wfrag, *scale, *shift, *res, *om;
    float *out;
    int in_cs, out_cs, res_cs, om_cs;
    int H, W, Cin, Ho, Wo, HoWo, Cout;
    int kh, kw, stride, pad, dil;
    int M, KG, tiles_n;
    int act, res_mode, sigmoid_from;
    int vec_out;                   // out / res / scale / shift views allow 16-byte accesses: epilogue through the LDS transpose
    long long w_img_stride;        // floats between the per-image weight sets (0: shared weights)
    unsigned in_bytes, out_bytes, res_bytes, w_bytes;
    // split-K across waves (layers with too few 32 x 128 tiles): grid = splits x tiles, split s covers steps
    // [s*ss_per, (s+1)*ss_per) and stores raw partial sums to ws[s][M][Cout_pad]; m3d_launch_splitk_reduce finishes
    float *ws;
    int splits, ss_per, base_waves;
    unsigned ws_bytes;
#ifdef CONV_TRACE
    long long *trace;
#endif
};


#ifdef CONV_TRACE
static long long *g_conv_trace = nullptr;
extern "C" void m3d_conv_wave_set_trace(void *buf) { g_conv_trace = (long long *)buf; }
static int g_conv_variant = 0;
extern "C" void m3d_conv_wave_set_variant(int v) { g_conv_variant = v; }
#define TRACE_INIT() long long *trp = a.trace ? a.trace + (size_t)blockIdx.x * 128 : nullptr; int tri = 0
#define TRACE() do { if (trp && lane == 0 && tri < 128) trp[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define TRACE_INIT()
#define TRACE()
#endif

// NT = column tiles of 32 output channels per wave: 4 (128 channels) or, for 64-channel layers, 2
// NCQ / WPE: experiment knobs (diagnostic library only, m3d_conv_wave_experiment): corners gathered per (pixel, tap) and the
// register bound in waves per SIMD.  The product instantiates the defaults.
template <bool DEFORM, int NT, int NCQ = 4, int WPE = (DEFORM ? 2 : 3)>      // register bound: 3 waves per SIMD for the plain kernel, 2 for the deformable one (as its K loop needs)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void conv_wave_kernel(const ConvWaveArgs a)
{
    __shared__ __attribute__((aligned(16))) float tileA[32 * 32];     // [pixel][8 slots of 4 channels], swizzled
    // [tap][pixel][4 corner offsets (as bits), 4 weights]: DEFORM keeps the sampling state of all (<= 9) taps here, built once in
    // the prologue, because its K loop runs CHUNK-major (32 input channels outer, taps inner): the input lines a wave gathers for
    // one channel chunk are re-used by the 9 taps x 4 corners within 9 steps and the chunk's share of the in-flight pixels
    // (8192 px x 128 B per XCD) fits the 4 MB L2 -- tap-major, every (tap, chunk) visit found its lines evicted: 178 MB of fabric
    // traffic per full-size launch against 84 MB algorithmic (PMC, profiles/r03r, r04a)
    __shared__ __attribute__((aligned(16))) float tapst[(DEFORM ? 9 : 1) * 32 * 8];
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int gp = lane >> 3, gc = lane & 7;                         // gather map: pixel-in-group, chunk
    TRACE_INIT();
    TRACE();
    int blk;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int split = 0;
    if (a.splits > 1) {
        split = blk / a.base_waves;
        blk -= split * a.base_waves;
    }
    const int bm = blk / a.tiles_n, bn = blk - bm * a.tiles_n;
    const int m0 = bm * 32;
    const int KK = a.kh * a.kw, C32 = a.Cin / 32;
    const int ss0 = split * a.ss_per, ss1 = min(KK * C32, ss0 + a.ss_per);        // this wave's steps

    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    // per-image weights (the ANAB logits / P.V GEMMs): the 32 pixels of a wave belong to one image (HoWo % 32 == 0)
    const __amdgpu_buffer_rsrc_t rw =
        make_rsrc(a.wfrag + (size_t)(m0 / a.HoWo) * a.w_img_stride + (size_t)bn * NT * a.KG * 256, a.w_bytes);
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wstride = (unsigned)a.KG * 1024u;              // bytes between column tiles

    // ---- the pixel this lane owns in the MFMA map (DEFORM: it computes that pixel's sampling state) ----------------
    int o_pix = 0, o_hi0 = 0, o_wi0 = 0;
    bool o_valid = false;
    int o_inv = -1;                                    // all ones: this lane owns no pixel (its sampling state is "nothing")
    const float *o_om = a.om;
    // ---- the 4 pixels this lane gathers for (plain mode keeps their coordinates) --------------------------------------
    int g_pix[4], g_hi0[4], g_wi0[4];
    bool g_valid[4];
    {
        const int m = m0 + l31;
        o_valid = m < a.M;
        o_inv = sign_smear(a.M - 1 - m);
        const int mm = o_valid ? m : 0;
        const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        o_pix = n * a.H * a.W;
        o_hi0 = ho * a.stride - a.pad;
        o_wi0 = wo * a.stride - a.pad;
        if (DEFORM) o_om = a.om + (size_t)mm * a.om_cs;
    }
    if constexpr (!DEFORM) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = m0 + 8 * g + gp;
            g_valid[g] = m < a.M;
            const int mm = g_valid[g] ? m : 0;
            const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
            const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
            g_pix[g] = n * a.H * a.W;
            g_hi0[g] = ho * a.stride - a.pad;
            g_wi0[g] = wo * a.stride - a.pad;
        }
    }

    unsigned doff[4][DEFORM ? 4 : 1];          // [pixel group][corner]: byte offset incl. the chunk sub-offset 16*gc
    float bw[DEFORM ? 4 : 1][4];
    // DEFORM prologue: sampling state of every tap (offsets / mask of all taps fetched in one go: 27 loads in flight per lane)
    if constexpr (DEFORM) {
        float raws[9][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tp = min(t, KK - 1);
            raws[t][0] = o_om[2 * tp];
            raws[t][1] = o_om[2 * tp + 1];
            raws[t][2] = o_om[2 * KK + tp];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t < KK) {                          // wave-uniform
                const int ti = t / a.kw, tj = t - ti * a.kw;
                const float mk = raws[t][2];
                const float h_im = (float)(o_hi0 + ti * a.dil) + raws[t][0];
                const float w_im = (float)(o_wi0 + tj * a.dil) + raws[t][1];
                float wq[4];
                int oq[4], drop[4];
                dcn_corners(h_im, w_im, a.H, a.W, o_inv, wq, oq, drop);      // no SGPR lane masks in here: see common.h
                u32x4 ob;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned off = ((unsigned)o_pix + (unsigned)oq[q]) * (unsigned)a.in_cs * 4u;    // garbage where dropped: masked next
                    ob[q] = (off & ~(unsigned)drop[q]) | (M3D_BUF_OOB & (unsigned)drop[q]);     // dropped corner: the load reads 0
                }
                if (h == 0) {
                    *reinterpret_cast<u32x4 *>(&tapst[t * 256 + l31 * 8]) = ob;
                    *reinterpret_cast<f32x4 *>(&tapst[t * 256 + l31 * 8 + 4]) = f32x4{wq[0] * mk, wq[1] * mk, wq[2] * mk, wq[3] * mk};
                }
            }
        }
        // single wave: the LDS writes above are ordered before the reads below by the wave's own lgkmcnt
    }
    auto setup_tap = [&](int tap) {
        if constexpr (DEFORM) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 ov = *reinterpret_cast<const u32x4 *>(&tapst[tap * 256 + (8 * g + gp) * 8]);
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(&tapst[tap * 256 + (8 * g + gp) * 8 + 4]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    doff[g][q] = ov[q] + (unsigned)gc * 16u;          // OOB marker + < 128 stays out of range
                    bw[g][q] = wv[q];
                }
            }
        } else {
            const int ti = tap / a.kw, tj = tap - ti * a.kw;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int hi = g_hi0[g] + ti * a.dil, wi = g_wi0[g] + tj * a.dil;
                const bool ok = g_valid[g] && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                doff[g][0] = ok ? ((unsigned)(g_pix[g] + hi * a.W + wi) * (unsigned)a.in_cs + (unsigned)gc * 4u) * 4u
                                : M3D_BUF_OOB;
            }
        }
    };

    constexpr int NC = DEFORM ? NCQ : 1;
    f32x4 cr[4][NC];                           // gathered chunks of the step in flight: [pixel group][corner]
    f32x4 bfA[NT], bfB[NT];
    auto issue_gather = [&](int c32) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < NC; ++q) cr[g][q] = buf_load_f32x4(rin, doff[g][q], (unsigned)c32 * 128u);
    };
    auto issue_b = [&](int kg, f32x4 (&bf)[NT]) {
        const unsigned bsoff = (unsigned)min(kg, a.KG - 1) * 1024u;      // unconditional, clamped (see wino_kernel)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = buf_load_f32x4(rw, wlane, bsoff + nt * wstride);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // step s of the K loop: plain = tap-major (s = tap * C32 + c32), DEFORM = chunk-major (s = c32 * KK + tap); the weights are
    // packed tap-major either way: k-group of a step = (tap * C32 + c32) * 4
    int tap, c32;
    // K order of the deformable kernel: chunk-major, at COMPILE time.  As a runtime flag (round 3: an A/B switch) the two loop tails
    // defined the 32 sampling registers (doff / bw) on different paths and the compiler shuttled them through a second register
    // set at the join: 64 v_mov per step next to 64 MFMAs, each VALU instruction ~11 cycles of matrix-pipe delay
    // (tools/ubench/mfma_side_cost) -- a sixth of the K loop (round 6, found in the ISA; profiles/r6c_dcn_wave_isa.txt).
    constexpr bool cmaj = DEFORM;
    if constexpr (cmaj) { c32 = ss0 / KK; tap = ss0 - c32 * KK; }
    else { tap = ss0 / C32; c32 = ss0 - tap * C32; }
    setup_tap(tap);
    issue_gather(c32);
    issue_b((tap * C32 + c32) * 4, bfA);

    // LDS slots: pixel row r, 16-byte slot s lives at r*128 + ((s ^ ((r >> 1) & 7)) * 16) bytes
    unsigned wr_off[4], rd_off[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 8 * g + gp;
        wr_off[g] = (unsigned)(r * 32 + ((gc ^ ((r >> 1) & 7)) * 4));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) rd_off[j] = (unsigned)(l31 * 32 + (((2 * j + h) ^ ((l31 >> 1) & 7)) * 4));

    TRACE();
    for (int st = ss0; st < ss1; ++st) {
        TRACE();
        const int kg0 = (tap * C32 + c32) * 4;
        // ---- combine (gather layout) and transpose through LDS into the A-operand layout ---------------------------
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
            if constexpr (DEFORM && NCQ == 4)
                v = pk_fma_s(bw[g][0], cr[g][0], pk_fma_s(bw[g][1], cr[g][1], pk_fma_s(bw[g][2], cr[g][2], pk_mul_s(bw[g][3], cr[g][3]))));
            else if constexpr (DEFORM && NCQ == 2)      // (experiment: wrong results, timing only)
                v = pk_fma_s(bw[g][0], cr[g][0], pk_mul_s(bw[g][1], cr[g][1]));
            else if constexpr (DEFORM)
                v = pk_mul_s(bw[g][0], cr[g][0]);
            else
                v = cr[g][0];
            *reinterpret_cast<f32x4 *>(&tileA[wr_off[g]]) = v;
        }
        // ---- next step ----------------------------------------------------------------------------------------------------
        if constexpr (cmaj) {
            if (++tap == KK) { tap = 0; ++c32; }        // wave-uniform; past the last step: clamped, a redundant in-range reload
            setup_tap(tap);                              // 8 x ds_read_b128: the state of every tap was built in the prologue
            issue_gather(min(c32, C32 - 1));
        } else {
            if (++c32 == C32) {                          // wave-uniform
                c32 = 0;
                ++tap;
                setup_tap(min(tap, KK - 1));
            }
            issue_gather(c32);                           // after the last step: a redundant in-range reload, never used
        }
        const int kgn = (min(tap, KK - 1) * C32 + min(c32, C32 - 1)) * 4;       // first k-group of the next step
        TRACE();
        f32x4 A[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) A[j] = *reinterpret_cast<const f32x4 *>(&tileA[rd_off[j]]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- 4 k-groups x 16 MFMAs; the B fragments of the next group are in flight meanwhile --------------------------
        auto group = [&](int j, int kg_next, f32x4 (&bf)[NT], f32x4 (&bfn)[NT]) {
            issue_b(kg_next, bfn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j][t], bf[nt][t], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        group(0, kg0 + 1, bfA, bfB);
        group(1, kg0 + 2, bfB, bfA);
        group(2, kg0 + 3, bfA, bfB);
        group(3, kgn, bfB, bfA);
    }

    TRACE();
    // ---- epilogue: lane = channel n0 + nt*32 + l31, rows = pixels m0 + (r&3) + 8*(r>>2) + 4h ------------------------------
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res ? a.res : a.out, a.res ? a.res_bytes : 0u);
    const int n0 = bn * 32 * NT;
    const int mb = m0 + 4 * h;
    if (a.splits > 1) {                        // raw partial sums; the reduce launch owns the epilogue
        const __amdgpu_buffer_rsrc_t rws = make_rsrc(a.ws, a.ws_bytes);
        const unsigned sbase = (unsigned)split * (unsigned)a.M;
        const unsigned cpad = (unsigned)(a.tiles_n * 32 * NT);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned co = (unsigned)(n0 + nt * 32 + l31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                const unsigned oo = m < a.M ? ((sbase + (unsigned)m) * cpad + co) * 4u : M3D_BUF_OOB;
                const float pv = acc[nt][r];   // (bit-casting the vector element expression directly stores element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv), rws, oo, 0, 0);
            }
        }
        return;
    }
    if (a.vec_out) {
        // 16-byte epilogue: every 32 x 32 accumulator tile (lane = channel) is turned through the wave's LDS tile so that a lane
        // owns 4 consecutive channels of a pixel: 16 float4 stores (and residual loads) per wave instead of 64 4-byte ones --
        // the 4-byte form took ~10000 cycles per wave, up to a fifth of the wave on the 1x1 layers (tools/conv_wave_trace.py).
        const float slope = a.act == 1 ? M3D_LEAKY_SLOPE : 1.f;
        const int pr = lane >> 3, c4 = lane & 7;                     // read map: pixel pr + 8q of the tile, channels 4*c4 .. +3
        unsigned oq[4], rq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + pr + 8 * q;
            oq[q] = m < a.M ? (unsigned)m * (unsigned)a.out_cs * 4u : M3D_BUF_OOB;
            rq[q] = m < a.M ? (unsigned)m * (unsigned)a.res_cs * 4u : M3D_BUF_OOB;
        }
        // affine parameters of all channel tiles in ONE round trip, before the first store (fetched per tile, after stores they
        // may alias, they cost a memory round trip each: 4 x ~2000 cycles); residuals per tile (16 registers: the plain kernel
        // keeps its three waves per SIMD)
        f32x4 scv[NT], shv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = n0 + nt * 32 + 4 * c4;
            scv[nt] = f32x4{1.f, 1.f, 1.f, 1.f};
            shv[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c + 3 < a.Cout) {
                if (a.scale) scv[nt] = *reinterpret_cast<const f32x4 *>(a.scale + c);
                if (a.shift) shv[nt] = *reinterpret_cast<const f32x4 *>(a.shift + c);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < a.Cout) {
                        if (a.scale) scv[nt][e] = a.scale[c + e];
                        if (a.shift) shv[nt][e] = a.shift[c + e];
                    }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = n0 + nt * 32 + 4 * c4;
            f32x4 rv[4];
            if (a.res) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (c + 3 < a.Cout) {
                        rv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, rq[q] + (unsigned)c * 4u, 0, 0));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            rv[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                rres, c + e < a.Cout ? rq[q] + (unsigned)(c + e) * 4u : M3D_BUF_OOB, 0, 0));
                    }
                }
            }
            // accumulator element r of lane (channel l31, half h) is pixel (r & 3) + 8 * (r >> 2) + 4 * h of the tile
#pragma unroll
            for (int r = 0; r < 16; ++r) tileA[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[nt][r];
            const bool tile_sig = a.sigmoid_from >= 0 && n0 + nt * 32 + 31 >= a.sigmoid_from;      // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(&tileA[(pr + 8 * q) * 32 + 4 * c4]);   // the wave's own writes: in order
                if (a.res) v = a.res_mode ? (v + rv[q]) * scv[nt] + shv[nt] : v * scv[nt] + shv[nt] + rv[q];
                else v = v * scv[nt] + shv[nt];
                if (tile_sig) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = c + e >= a.sigmoid_from ? sigmoidf_(v[e]) : fmaxf(v[e], v[e] * slope);
                } else {
                    v = __builtin_elementwise_max(v, v * slope);
                }
                if (c + 3 < a.Cout) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rout, oq[q] + (unsigned)c * 4u, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rout,
                                                              c + e < a.Cout ? oq[q] + (unsigned)(c + e) * 4u : M3D_BUF_OOB, 0, 0);
                }
            }
        }
        TRACE();
        return;
    }
    // 4-byte form (views that do not allow 16-byte accesses)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + nt * 32 + l31;
        const bool cok = co < a.Cout;
        const float sc = (cok && a.scale) ? a.scale[co] : 1.f;
        const float sh = (cok && a.shift) ? a.shift[co] : 0.f;
        float rv[16];
        if (a.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                const unsigned ro = (cok && m < a.M) ? ((unsigned)m * (unsigned)a.res_cs + (unsigned)co) * 4u : M3D_BUF_OOB;
                rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, ro, 0, 0));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            float v = acc[nt][r];
            if (a.res) v = a.res_mode ? (v + rv[r]) * sc + sh : v * sc + sh + rv[r];
            else v = v * sc + sh;
            if (a.sigmoid_from >= 0 && co >= a.sigmoid_from) v = sigmoidf_(v);
            else if (a.act == 1) v = fmaxf(v, v * M3D_LEAKY_SLOPE);
            const unsigned oo = (cok && m < a.M) ? ((unsigned)m * (unsigned)a.out_cs + (unsigned)co) * 4u : M3D_BUF_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, oo, 0, 0);
        }
    }
    TRACE();
}

// Launch plan: *splits (split-K factor, 1 = none) and the total number of waves; 0 = not applicable (hard constraints) or,
// with enforce_min, still too few waves.  Measured (profiles/): below one wave per SIMD the LDS-tiled kernel wins and the
// deformable gather wants two waves per SIMD, so thin layers are split along K until ~1800 waves exist (at least 4 steps of
// 32 channels per split).  Tuning knobs (experiments only): M3D_CONV_WAVE_MIN / M3D_DCN_WAVE_MIN / M3D_CONV_WAVE_SPLITK=0.
static int conv_wave_plan(const m3d_conv_desc *d, bool enforce_min, int *splits, int *ss_per)
{
    *splits = 1;
    *ss_per = d->kh * d->kw * (d->Cin / 32);
    if (d->out_nchw) return 0;
    if (d->dcn_offmask && d->kh * d->kw > 9) return 0;        // the sampling state of every tap lives in LDS (9 KB per wave)
    if (d->Cin % 32 != 0 || d->Cout_pad % 64 != 0) return 0;
    const int cw = d->Cout_pad % 128 == 0 ? 128 : 64;      // channels per wave
    if (d->wgt_img_stride && (d->Ho * d->Wo) % 32 != 0) return 0;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    const long long base = ((M + 31) / 32) * (d->Cout_pad / cw);
    if (base >= (1ll << 28)) return 0;
    static int wave_min = -1, dcn_min = -1, splitk = -1;
    if (wave_min < 0) { const char *e = getenv("M3D_CONV_WAVE_MIN"); wave_min = e ? atoi(e) : 900; }
    if (dcn_min < 0) { const char *e = getenv("M3D_DCN_WAVE_MIN"); dcn_min = e ? atoi(e) : 1500; }
    if (splitk < 0) { const char *e = getenv("M3D_CONV_WAVE_SPLITK"); splitk = e ? atoi(e) : 1; }
    const long long need = d->dcn_offmask ? dcn_min : wave_min;
    const int nss = *ss_per;
    if (base < need && splitk && d->splitk_ws) {
        int s = (int)((1800 + base - 1) / base);
        if (s > nss / 4) s = nss / 4;
        if (s > 8) s = 8;
        if (s >= 2) {
            *ss_per = (nss + s - 1) / s;
            *splits = (nss + *ss_per - 1) / *ss_per;
        }
    }
    if (enforce_min && base * *splits < need) { *splits = 1; *ss_per = nss; return 0; }
    return (int)(base * *splits);
}

extern "C" int m3d_conv_wave_applicable(const m3d_conv_desc *d)
{
    int s, p;
    return d ? conv_wave_plan(d, true, &s, &p) : 0;
}

// Split-K plan of m3d_conv_wave_forward: the caller that wants thin layers on this path passes a scratch buffer of
// *ws_bytes through splitk_ws (the plan is computed AS IF one were given).
extern "C" int m3d_conv_wave_splitk_plan(const m3d_conv_desc *d, int *splits, long long *ws_bytes)
{
    M3D_REQUIRE(d && splits && ws_bytes, "conv_wave_splitk_plan: null pointer");
    m3d_conv_desc t = *d;
    static float dummy;
    t.splitk_ws = &dummy;
    int p;
    const int waves = conv_wave_plan(&t, true, splits, &p);
    if (waves == 0) *splits = 1;
    *ws_bytes = *splits > 1 ? (long long)*splits * d->N * d->Ho * d->Wo * d->Cout_pad * 4 : 0;
    return M3D_OK;
}

extern "C" int m3d_conv_wave_forward(const m3d_conv_desc *d, m3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    M3D_REQUIRE(d && d->in && d->wgt && d->out, "conv_wave: null pointer");
    int splits = 1, ss_per = 0;
    const int waves = conv_wave_plan(d, false, &splits, &ss_per);       // the fill heuristic is advisory here
    M3D_REQUIRE(waves > 0, "conv_wave: needs Cin %% 32 == 0, Cout_pad %% 64 == 0, NHWC output, Ho*Wo %% 32 == 0 with per-image weights");
    const int ho = (d->H + 2 * d->pad - (d->dil * (d->kh - 1) + 1)) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - (d->dil * (d->kw - 1) + 1)) / d->stride + 1;
    M3D_REQUIRE(ho == d->Ho && wo == d->Wo, "conv_wave: Ho/Wo mismatch");
    M3D_REQUIRE(d->in_cs % 32 == 0 && d->in_cs >= d->Cin && ((uintptr_t)d->in & 127) == 0 && ((uintptr_t)d->wgt & 15) == 0,
                "conv_wave: the input view must be 128-byte aligned with in_cs %% 32 == 0");
    const long long M = (long long)d->N * d->Ho * d->Wo;
    M3D_REQUIRE(!d->wgt_img_stride || (d->wgt_img_stride % 4 == 0 && d->stride == 1 && d->Ho == d->H && d->Wo == d->W),
                "conv_wave: per-image weights need an image-aligned 16-byte stride and a same-size output");
    M3D_REQUIRE((long long)d->N * d->H * d->W * d->in_cs * 4 < (1ll << 31) && M * d->out_cs * 4 < (1ll << 31) &&
                M * d->res_cs * 4 < (1ll << 31) && (long long)d->Cout_pad * d->kh * d->kw * d->Cin * 4 < (1ll << 31),
                "conv_wave: views must be < 2 GiB");
    ConvWaveArgs a;
    a.in = d->in; a.wfrag = d->wgt; a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.om = d->dcn_offmask;
    a.out = d->out;
    a.in_cs = d->in_cs; a.out_cs = d->out_cs; a.res_cs = d->res_cs; a.om_cs = d->dcn_om_cs;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.HoWo = d->Ho * d->Wo; a.Cout = d->Cout;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    const int cw = d->Cout_pad % 128 == 0 ? 128 : 64;
    a.M = (int)M; a.KG = d->kh * d->kw * d->Cin / 8; a.tiles_n = d->Cout_pad / cw;
    a.act = d->act; a.res_mode = d->res_mode; a.sigmoid_from = d->sigmoid_from; a.w_img_stride = d->wgt_img_stride;
    static int vec_epi = -1;       // M3D_WAVE_VEC_EPILOGUE=0: 4-byte stores straight from the accumulators (A/B)
    if (vec_epi < 0) { const char *e = getenv("M3D_WAVE_VEC_EPILOGUE"); vec_epi = e ? atoi(e) : 1; }
    a.vec_out = vec_epi && d->out_cs % 4 == 0 && ((uintptr_t)d->out & 15) == 0 &&
                (!d->res || (d->res_cs % 4 == 0 && ((uintptr_t)d->res & 15) == 0)) &&
                (!d->scale || ((uintptr_t)d->scale & 15) == 0) && (!d->shift || ((uintptr_t)d->shift & 15) == 0);
    a.in_bytes = (unsigned)((long long)d->N * d->H * d->W * d->in_cs * 4);
    a.out_bytes = (unsigned)(M * d->out_cs * 4);
    a.res_bytes = (unsigned)(M * d->res_cs * 4);
    a.w_bytes = (unsigned)((long long)cw * d->kh * d->kw * d->Cin * 4);
#ifdef CONV_TRACE
    a.trace = g_conv_trace;
#endif
    a.ws = nullptr; a.splits = 1; a.ss_per = ss_per; a.base_waves = waves; a.ws_bytes = 0;
    if (splits > 1) {
        const long long need = (long long)splits * M * d->Cout_pad * 4;
        M3D_REQUIRE(need <= d->splitk_ws_bytes && need < (1ll << 31) && ((uintptr_t)d->splitk_ws & 15) == 0,
                    "conv_wave: split-K workspace too small (%lld bytes, see m3d_conv_wave_splitk_plan), >= 2 GiB or misaligned",
                    d->splitk_ws_bytes);
        a.ws = d->splitk_ws; a.splits = splits; a.base_waves = waves / splits; a.ws_bytes = (unsigned)need;
    }
#ifdef CONV_TRACE
    // diagnostic library only (tools/conv_wave_third_wave.py): would a third wave per SIMD pay if the gathered
    // corners (64 of the kernel's registers) lived elsewhere?  Variants gather fewer corners (WRONG results, timing only) at a
    // register bound of two or three waves per SIMD.
    if (g_conv_variant && d->dcn_offmask && cw == 128) {
        switch (g_conv_variant) {
        case 1: hipLaunchKernelGGL((conv_wave_kernel<true, 4, 1, 2>), dim3(waves), dim3(64), 0, stream, a); break;
        case 2: hipLaunchKernelGGL((conv_wave_kernel<true, 4, 1, 3>), dim3(waves), dim3(64), 0, stream, a); break;
        case 3: hipLaunchKernelGGL((conv_wave_kernel<true, 4, 2, 2>), dim3(waves), dim3(64), 0, stream, a); break;
        case 4: hipLaunchKernelGGL((conv_wave_kernel<true, 4, 2, 3>), dim3(waves), dim3(64), 0, stream, a); break;
        default: hipLaunchKernelGGL((conv_wave_kernel<true, 4, 4, 3>), dim3(waves), dim3(64), 0, stream, a); break;   // 5: as built but bound to 3 waves (spills)
        }
        M3D_LAUNCH_CHECK();
        return M3D_OK;
    }
#endif
    if (cw == 128) {
        if (d->dcn_offmask) hipLaunchKernelGGL((conv_wave_kernel<true, 4>), dim3(waves), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((conv_wave_kernel<false, 4>), dim3(waves), dim3(64), 0, stream, a);
    } else {
        if (d->dcn_offmask) hipLaunchKernelGGL((conv_wave_kernel<true, 2>), dim3(waves), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((conv_wave_kernel<false, 2>), dim3(waves), dim3(64), 0, stream, a);
    }
    M3D_LAUNCH_CHECK();
    if (splits > 1) {
        SplitkReduceArgs r;
        r.ws = a.ws; r.scale = d->scale; r.shift = d->shift; r.res = d->res; r.out = d->out;
        r.M = (int)M; r.Cout = d->Cout; r.Cout_pad = d->Cout_pad; r.splits = splits; r.out_cs = d->out_cs; r.res_cs = d->res_cs;
        r.res_mode = d->res_mode; r.act = d->act; r.sigmoid_from = d->sigmoid_from;
        return m3d_launch_splitk_reduce(r, stream);
    }
    return M3D_OK;
}
