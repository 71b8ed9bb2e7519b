// Implicit-GEMM convolution / deformable convolution for gfx950 on the fp32 MFMA pipe.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]      m = output pixel (N*Ho*Wo), n = output channel,
//                                           k = (tap i*kw+j) * Cin + c      (NHWC: c contiguous)
//
// * A is never materialised (no im2col `columns` buffer, cf. the reference's 35 MB scratch per
//   128-channel DCN call, model/DCNv2/src/dcn_v2_cuda.c:54): each k-tile (BK channels of ONE tap)
//   is gathered straight from the NHWC activation into LDS with 16-byte loads.  In deformable
//   mode the gather is the modulated bilinear sample of DCNv2
//   (model/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:18-47,129-178): the 4 corner weights / offsets of
//   a (pixel, tap) are computed once per tap in registers and reused for all Cin/BK k-tiles.
// * Math is v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD =
//   157 TFLOP/s chip peak (MI355X_MICROARCH.md).  One wave owns TM x TN tiles of 32x32.
//   The k order inside an 8-wide group is permuted (lane half h, step t -> k = 8g + 4h + t) so that
//   every lane fetches its 4 A and 4 B values of a group with ONE ds_read_b128 each.
// * LDS rows are padded to BK+4 floats: conflict-free for the b128 fragment reads.
// * Global->register prefetch of tile kt+1 overlaps the MFMAs of tile kt; one barrier per k-tile.
// * Epilogue fuses per-channel affine (folded BatchNorm + conv bias), residual add, LeakyReLU or
//   sigmoid, and writes NHWC (lanes along channels: 128 B per half-wave) or, in SWAP mode, planar
//   NCHW (operands swapped so lanes run along pixels) -- the layout the RPN outputs need.
// * blockIdx is remapped so each XCD (private L2) works on a contiguous range of tiles.
#include <stdlib.h>

#include "common.h"

// Phase ablations (M3D_ABLATE, the probe tools) exist in the DIAGNOSTIC library only (make trace): as runtime flags they put uniform
// branches into the K loops of the product kernels.
#ifdef IGEMM_TRACE
#define IG_ABL(bit) (a.ablate & (bit))
#else
#define IG_ABL(bit) false
#endif

struct IgemmArgs {
    const float *in;
    const float *wgt;
    float *out;
    const float *scale;
    const float *shift;
    const float *res;
    const float *om;
    const float *zero;        // 16 bytes of zeros in device memory
    long long wgt_img_stride;
    long long out_img_stride;
    int in_cs, out_cs, res_cs, om_cs;
    int H, W, Cin, Ho, Wo, HoWo;
    int Cout, Cout_pad;
    int kh, kw, stride, pad, dil;
    int M, Ktot, KT;
    int tiles_m, tiles_n;
    int act, sigmoid_from, res_mode;
    int ablate;   // diagnostics only (M3D_ABLATE): 1 = no steady-state global loads, 2 = no MFMA, 4 = no LDS refill
    // split-K (small-M layers that cannot fill 256 CUs): grid = splits x tiles, split s accumulates k-tiles
    // [s*kt_per, (s+1)*kt_per) and stores raw partial sums to ws[s][M][Cout_pad]; splitk_reduce_kernel adds them in
    // split order (deterministic) and applies the epilogue.
    float *ws;
    int splits, kt_per;
    unsigned in_bytes, wgt_bytes;   // extents for the buffer-addressed loads (range check: masked lanes read 0.0f)
#ifdef IGEMM_TRACE
    long long *trace;   // [block][wave][128] s_memtime stamps (diagnostic build only, tools/igemm_trace.py)
#endif
};

#ifdef IGEMM_TRACE
#define TRACE_INIT() long long *trp = a.trace ? a.trace + ((size_t)blockIdx.x * 4 + wave) * 128 : nullptr; int tri = 0
#define TRACE() do { if (trp && lane == 0 && tri < 128) trp[tri++] = __builtin_readcyclecounter(); } while (0)
static long long *g_igemm_trace = nullptr;
extern "C" void m3d_igemm_set_trace(void *buf) { g_igemm_trace = (long long *)buf; }
#else
#define TRACE_INIT()
#define TRACE()
#endif

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool DEFORM, bool SWAP>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs a)
{
    constexpr int TM = BM / (32 * WAVES_M);
    constexpr int TN = BN / (32 * WAVES_N);
    constexpr int LDK = BK + 4;
    constexpr int TPR = BK / 4;      // threads covering one row of a k-tile (float4 each)
    constexpr int RPP = 256 / TPR;   // rows per pass
    constexpr int PA = BM / RPP;     // A passes per thread
    constexpr int PB = (BN + RPP - 1) / RPP;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    static_assert(TM >= 1 && TN >= 1 && PA >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                      // [2][BM][LDK]
    float *Bs = smem + 2 * BM * LDK;       // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = (wave / WAVES_N) * (TM * 32);
    const int wn = (wave % WAVES_N) * (TN * 32);
    TRACE_INIT();
    TRACE();

    // XCD-aware tile id: block b runs on XCD b%8; give each XCD a contiguous tile range.
    int tile;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int split = 0;
    if (a.splits > 1) {
        const int ntiles = a.tiles_m * a.tiles_n;
        split = tile / ntiles;
        tile -= split * ntiles;
    }
    const int kt_begin = split * a.kt_per, kt_end = min(a.KT, kt_begin + a.kt_per);
    const int tile_m = tile / a.tiles_n, tile_n = tile - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const float *__restrict__ wgt = a.wgt;
    if (a.wgt_img_stride) wgt += (long long)(m0 / a.HoWo) * a.wgt_img_stride;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rwgt = make_rsrc(wgt, a.wgt_bytes);

    // ---- per-thread A rows -------------------------------------------------------------
    const int rsub = tid / TPR;            // row within a pass
    const int csub = (tid % TPR) * 4;      // channel offset within the k-tile
    int pix_base[PA], hi0[PA], wi0[PA];
    bool rvalid[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + p * RPP + rsub;
        rvalid[p] = m < a.M;
        const int mm = rvalid[p] ? m : 0;
        const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        pix_base[p] = n * a.H * a.W;
        hi0[p] = ho * a.stride - a.pad;
        wi0[p] = wo * a.stride - a.pad;
    }
    // deformable: bilinear state of the current tap
    float bw[DEFORM ? PA : 1][4];
    unsigned doff[DEFORM ? PA : 1][4];   // byte offsets of the 4 corners (pixel + channel sub-offset)

    f32x4 ra[PA][DEFORM ? 4 : 1];
    f32x4 rb[PB];

    // k-tile cursor, advanced incrementally (no integer division in the loop): tiles are loaded in order 0,1,2,...
    // (a split may start in the middle of a tap: the first load refreshes the per-tap state regardless of cur_c)
    int cur_tap = 0, cur_c = 0, cur_ti = 0, cur_tj = 0;
    bool tap_fresh = true;
    if (kt_begin) {
        const int k0 = kt_begin * BK;
        cur_tap = k0 / a.Cin;
        cur_c = k0 - cur_tap * a.Cin;
        cur_ti = cur_tap / a.kw;
        cur_tj = cur_tap - cur_ti * a.kw;
    }
    unsigned aoff[PA];
    unsigned boff[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p)   // rows past Cout_pad are clamped (their products land in never-stored channels)
        boff[p] = ((unsigned)min(n0 + p * RPP + rsub, a.Cout_pad - 1) * (unsigned)a.Ktot + (unsigned)csub) * 4u;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        const int tap = cur_tap;
        const int c0 = cur_c + csub;
        const int ti = cur_ti, tj = cur_tj;
        const bool first_of_tap = cur_c == 0 || tap_fresh;
        tap_fresh = false;
        cur_c += BK;
        if (cur_c >= a.Cin) {
            cur_c = 0;
            ++cur_tap;
            if (++cur_tj == a.kw) { cur_tj = 0; ++cur_ti; }
        }
        if constexpr (!DEFORM) {
            // buffer addressing: resource (SGPR x4) + per-thread 32-bit byte offset that only changes with the tap + SGPR
            // channel offset -> no per-load VALU address math; padding taps carry the out-of-range offset and read 0.0f
            if (first_of_tap) {
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    const int hi = hi0[p] + ti * a.dil, wi = wi0[p] + tj * a.dil;
                    const bool ok = rvalid[p] && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                    aoff[p] = ok ? ((unsigned)(pix_base[p] + hi * a.W + wi) * (unsigned)a.in_cs + (unsigned)csub) * 4u
                                 : M3D_BUF_OOB;
                }
            }
            const unsigned asoff = (unsigned)(c0 - csub) * 4u;
#pragma unroll
            for (int p = 0; p < PA; ++p) ra[p][0] = buf_load_f32x4(rin, aoff[p], asoff);
        } else {
            if (first_of_tap) {   // first k-tile of a tap: refresh the sampling state
                const int KK = a.kh * a.kw;
#pragma unroll
                for (int p = 0; p < PA; ++p) {
                    // rows past M read row M - 1 and are switched off through `drop_all`; no lane mask in an SGPR (common.h)
                    const int row = m0 + p * RPP + rsub;
                    const float *omp = a.om + (size_t)min(row, a.M - 1) * a.om_cs;
                    const float dh = omp[2 * tap], dw = omp[2 * tap + 1], mk = omp[2 * KK + tap];
                    float wq[4];
                    int oq[4], drop[4];
                    dcn_corners((float)(hi0[p] + ti * a.dil) + dh, (float)(wi0[p] + tj * a.dil) + dw, a.H, a.W,
                                sign_smear(a.M - 1 - row), wq, oq, drop);
                    const float w1 = wq[0], w2 = wq[1], w3 = wq[2], w4 = wq[3];
                    const int o1 = oq[0] & ~drop[0], o2 = oq[1] & ~drop[1], o3 = oq[2] & ~drop[2], o4 = oq[3] & ~drop[3];
                    // the modulation mask is folded into the corner weights once per tap (dcn_v2_im2col_cuda.cu:174)
                    bw[p][0] = w1 * mk; bw[p][1] = w2 * mk; bw[p][2] = w3 * mk; bw[p][3] = w4 * mk;
                    doff[p][0] = ((unsigned)(pix_base[p] + o1) * (unsigned)a.in_cs + (unsigned)csub) * 4u;
                    doff[p][1] = ((unsigned)(pix_base[p] + o2) * (unsigned)a.in_cs + (unsigned)csub) * 4u;
                    doff[p][2] = ((unsigned)(pix_base[p] + o3) * (unsigned)a.in_cs + (unsigned)csub) * 4u;
                    doff[p][3] = ((unsigned)(pix_base[p] + o4) * (unsigned)a.in_cs + (unsigned)csub) * 4u;
                }
            }
            const unsigned dsoff = (unsigned)(c0 - csub) * 4u;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ra[p][q] = buf_load_f32x4(rin, doff[p][q], dsoff);
            }
        }
        const unsigned bsoff = (unsigned)k0 * 4u;
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[p] = buf_load_f32x4(rwgt, boff[p], bsoff);
    };

    auto store_tile = [&](int buf) {
        float *Ab = As + buf * BM * LDK;
        float *Bb = Bs + buf * BN * LDK;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            f32x4 v;
            if constexpr (!DEFORM) {
                v = ra[p][0];
            } else {
                // (w1*v1 + w2*v2 + w3*v3 + w4*v4) * mask   -- dcn_v2_im2col_cuda.cu:44-46,174 -- as 8 packed-fp32 VALU ops
                v = pk_fma_s(bw[p][0], ra[p][0], pk_fma_s(bw[p][1], ra[p][1], pk_fma_s(bw[p][2], ra[p][2], pk_mul_s(bw[p][3], ra[p][3]))));
            }
            *reinterpret_cast<f32x4 *>(Ab + (p * RPP + rsub) * LDK + csub) = v;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int r = p * RPP + rsub;
            if (r < BN) *reinterpret_cast<f32x4 *>(Bb + r * LDK + csub) = rb[p];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(kt_begin);
    store_tile(0);
    TRACE();
    __syncthreads();
    TRACE();

    const int l31 = lane & 31, lh4 = (lane >> 5) * 4;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end && !IG_ABL(1)) load_tile(kt + 1);
        TRACE();
        const float *Ab = As + buf * BM * LDK + (wm + l31) * LDK + lh4;
        const float *Bb = Bs + buf * BN * LDK + (wn + l31) * LDK + lh4;
        if (!IG_ABL(2)) {
            // fragment reads are software-pipelined one k-group ahead of the MFMAs that consume them, so the LDS
            // latency hides behind 16*TM*TN/4 MFMAs instead of stalling the (in-order) wave twice per k-tile
            f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * LDK);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * LDK);
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                const int cur = g & 1, nxt = cur ^ 1;
                if (g + 1 < BK / 8) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[nxt][i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * LDK + (g + 1) * 8);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[nxt][j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * LDK + (g + 1) * 8);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if constexpr (SWAP)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[cur][j][t], fa[cur][i][t], acc[i][j], 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][t], fb[cur][j][t], acc[i][j], 0, 0, 0);
                        }
                // pin the issue order: the reads of group g+1 first, then the MFMAs of group g that cover their latency
                if (g + 1 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * TM * TN, 0);
            }
        }
        TRACE();
        if (kt + 1 < kt_end && !IG_ABL(4)) store_tile(buf ^ 1);
        TRACE();
        if (!IG_ABL(4)) __syncthreads();
        TRACE();
    }

    // ---- epilogue ----------------------------------------------------------------------
    // D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    const int hrow = 4 * (lane >> 5);
    if constexpr (!SWAP) {
        if (a.splits > 1) {               // raw partial sums; the reduce kernel owns the epilogue
            float *wsp = a.ws + (size_t)split * a.M * a.Cout_pad;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + wn + j * 32 + l31;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + hrow;
                        if (co < a.Cout_pad && m < a.M) wsp[(size_t)m * a.Cout_pad + co] = acc[i][j][r];
                    }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn + j * 32 + l31;
            const bool cok = co < a.Cout;
            const float sc = (cok && a.scale) ? a.scale[co] : 1.f;
            const float sh = (cok && a.shift) ? a.shift[co] : 0.f;
            const bool sg = a.sigmoid_from >= 0 && co >= a.sigmoid_from;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + hrow;
                    if (cok && m < a.M) {
                        float v = acc[i][j][r];
                        if (a.res) {
                            const float rv = a.res[(size_t)m * a.res_cs + co];
                            v = a.res_mode ? (v + rv) * sc + sh : v * sc + sh + rv;
                        } else {
                            v = v * sc + sh;
                        }
                        if (sg) v = sigmoidf_(v);
                        else if (a.act == 1) v = leaky(v);
                        a.out[(size_t)m * a.out_cs + co] = v;
                    }
                }
            }
        }
    } else {
        // D[row = channel][col = pixel]
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm + i * 32 + l31;
            const bool mok = m < a.M;
            const int mm = mok ? m : 0;
            const int n = mm / a.HoWo, pix = mm - n * a.HoWo;
            float *ob = a.out + (long long)n * a.out_img_stride + pix;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = n0 + wn + j * 32 + (r & 3) + 8 * (r >> 2) + hrow;
                    if (mok && co < a.Cout) {
                        const float sc = a.scale ? a.scale[co] : 1.f;
                        const float sh = a.shift ? a.shift[co] : 0.f;
                        float v = acc[i][j][r] * sc + sh;
                        if (a.sigmoid_from >= 0 && co >= a.sigmoid_from) v = sigmoidf_(v);
                        else if (a.act == 1) v = leaky(v);
                        ob[(size_t)co * a.HoWo] = v;
                    }
                }
            }
        }
    }
}

// Sum the split-K partials in split order and apply the igemm epilogue; one thread per 4 channels of a pixel.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SplitkReduceArgs a, int vec_out)
{
    const int c4n = a.Cout_pad >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.M * c4n) return;
    const int m = (int)(i / c4n), co0 = (int)(i - (long long)m * c4n) * 4;
    if (co0 >= a.Cout) return;
    const size_t stride = (size_t)a.M * a.Cout_pad;
    const float *p = a.ws + (size_t)m * a.Cout_pad + co0;
    f32x4 v = *reinterpret_cast<const f32x4 *>(p);
    for (int s = 1; s < a.splits; ++s) v = v + *reinterpret_cast<const f32x4 *>(p + s * stride);
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = co0 + q;
        float x = v[q];
        if (co < a.Cout) {
            const float sc = a.scale ? a.scale[co] : 1.f;
            const float sh = a.shift ? a.shift[co] : 0.f;
            if (a.res) {
                const float rv = a.res[(size_t)m * a.res_cs + co];
                x = a.res_mode ? (x + rv) * sc + sh : x * sc + sh + rv;
            } else {
                x = x * sc + sh;
            }
            if (a.sigmoid_from >= 0 && co >= a.sigmoid_from) x = sigmoidf_(x);
            else if (a.act == 1) x = leaky(x);
        }
        o[q] = x;
    }
    float *op = a.out + (size_t)m * a.out_cs + co0;
    if (vec_out && co0 + 3 < a.Cout) {
        *reinterpret_cast<f32x4 *>(op) = o;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (co0 + q < a.Cout) op[q] = o[q];
    }
}

int m3d_launch_splitk_reduce(const SplitkReduceArgs &a, hipStream_t stream)
{
    const long long n4 = (long long)a.M * (a.Cout_pad >> 2);
    const int vec_out = (a.out_cs % 4 == 0) && (((uintptr_t)a.out & 15) == 0);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a, vec_out);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool DEFORM, bool SWAP>
static int launch_igemm(const IgemmArgs &a, hipStream_t stream)
{
    constexpr size_t smem = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    auto kern = igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, DEFORM, SWAP>;
    static bool attr_set = false;
    if (!attr_set) {
        M3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    IgemmArgs b = a;
    b.tiles_m = cdiv(a.M, BM);
    b.tiles_n = cdiv(a.Cout_pad, BN);
    hipLaunchKernelGGL(kern, dim3(b.tiles_m * b.tiles_n * b.splits), dim3(256), smem, stream, b);
    M3D_LAUNCH_CHECK();
    if (b.splits > 1) {
        SplitkReduceArgs r;
        r.ws = b.ws; r.scale = b.scale; r.shift = b.shift; r.res = b.res; r.out = b.out;
        r.M = b.M; r.Cout = b.Cout; r.Cout_pad = b.Cout_pad; r.splits = b.splits; r.out_cs = b.out_cs; r.res_cs = b.res_cs;
        r.res_mode = b.res_mode; r.act = b.act; r.sigmoid_from = b.sigmoid_from;
        return m3d_launch_splitk_reduce(r, stream);
    }
    return M3D_OK;
}

struct TileChoice { int bm, bn, bk; };

// Split-K factor for a layer whose tile count cannot give every CU a workgroup: as many splits as keep the
// whole grid co-resident (<= 2 workgroups per CU), at least 8 k-tiles per split.
static int choose_split(const m3d_conv_desc *d, const TileChoice &t, int *kt_per)
{
    const long long M = (long long)d->N * d->Ho * d->Wo;
    const int KT = d->kh * d->kw * d->Cin / t.bk;
    *kt_per = KT;
    static int enabled = -1;                     // tuning knob (experiments only): M3D_SPLITK=0 disables
    if (enabled < 0) { const char *e = getenv("M3D_SPLITK"); enabled = e ? atoi(e) : 1; }
    if (!enabled || d->out_nchw || d->wgt_img_stride) return 1;
    const long long blocks = ((M + t.bm - 1) / t.bm) * ((d->Cout_pad + t.bn - 1) / t.bn);
    if (blocks > 256) return 1;
    int s = (int)(512 / blocks);
    if (s > KT / 8) s = KT / 8;
    if (s > 16) s = 16;
    if (s < 2) return 1;
    *kt_per = (KT + s - 1) / s;
    return (KT + *kt_per - 1) / *kt_per;
}

static int choose_tile(const m3d_conv_desc *d, TileChoice *t)
{
    const long long M = (long long)d->N * d->Ho * d->Wo;
    t->bk = (d->Cin % 32 == 0) ? 32 : 16;
    if (d->Cout_pad <= 32) t->bn = 32;
    else if (d->Cout_pad <= 64 || (d->Cout_pad % 128 != 0 && d->Cout_pad % 64 == 0 && d->Cout_pad < 256)) t->bn = 64;
    else t->bn = 128;
    if (t->bk == 16) t->bn = 32;
    t->bm = 128;
    if (t->bn >= 64) {
        static int thr = -1;                     // tuning knob (experiments only): M3D_BM_THRESHOLD
        if (thr < 0) { const char *e = getenv("M3D_BM_THRESHOLD"); thr = e ? atoi(e) : 400; }
        const long long blocks128 = ((M + 127) / 128) * ((d->Cout_pad + t->bn - 1) / t->bn);
        if (blocks128 < thr) t->bm = 64;         // under two blocks per CU: smaller tiles fill the chip
    }
    {
        static int fbn = -1;                     // tuning knob (experiments only): M3D_FORCE_BN
        if (fbn < 0) { const char *e = getenv("M3D_FORCE_BN"); fbn = e ? atoi(e) : 0; }
        if (fbn && t->bk == 32 && t->bn > fbn && !d->out_nchw) t->bn = fbn;
    }
    if (d->wgt_img_stride && (d->Ho * d->Wo) % t->bm != 0) {
        t->bm = 64;
        if ((d->Ho * d->Wo) % 64 != 0) return -1;
    }
    return 0;
}

extern "C" int m3d_conv2d_splitk_plan(const m3d_conv_desc *d, int *splits, long long *ws_bytes)
{
    TileChoice t;
    M3D_REQUIRE(d && splits && ws_bytes, "conv2d_splitk_plan: null pointer");
    M3D_REQUIRE(choose_tile(d, &t) == 0, "per-image weights need Ho*Wo %% 64 == 0");
    int kt_per;
    *splits = choose_split(d, t, &kt_per);
    *ws_bytes = *splits > 1 ? (long long)*splits * d->N * d->Ho * d->Wo * d->Cout_pad * 4 : 0;
    return M3D_OK;
}

extern "C" int m3d_conv2d_tile(const m3d_conv_desc *d, int *bm, int *bn, int *bk, int *grid)
{
    TileChoice t;
    M3D_REQUIRE(choose_tile(d, &t) == 0, "per-image weights need Ho*Wo %% 64 == 0");
    const long long M = (long long)d->N * d->Ho * d->Wo;
    *bm = t.bm; *bn = t.bn; *bk = t.bk;
    *grid = cdiv(M, t.bm) * cdiv(d->Cout_pad, t.bn);
    return M3D_OK;
}

// 256 zero bytes per device: padding taps of the igemm load from here (hipMalloc'ed once, never freed).
static const float *zero_page()
{
    static const float *pages[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!pages[dev]) {
        void *p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
        pages[dev] = (const float *)p;
    }
    return pages[dev];
}

extern "C" int m3d_conv2d_forward(const m3d_conv_desc *d, m3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    M3D_REQUIRE(d && d->in && d->wgt && d->out, "conv2d: null pointer");
    M3D_REQUIRE(d->Cin % 16 == 0, "conv2d: Cin (%d) must be a multiple of 16", d->Cin);
    M3D_REQUIRE(d->Cout_pad % 32 == 0 && d->Cout <= d->Cout_pad, "conv2d: bad Cout_pad %d", d->Cout_pad);
    M3D_REQUIRE(d->in_cs % 4 == 0 && d->in_cs >= d->Cin, "conv2d: in_cs must be a multiple of 4 and >= Cin");
    M3D_REQUIRE(((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->wgt & 15) == 0, "conv2d: 16-byte alignment");
    const int ho = (d->H + 2 * d->pad - (d->dil * (d->kh - 1) + 1)) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - (d->dil * (d->kw - 1) + 1)) / d->stride + 1;
    M3D_REQUIRE(ho == d->Ho && wo == d->Wo, "conv2d: Ho/Wo mismatch (%d,%d) vs (%d,%d)", d->Ho, d->Wo, ho, wo);
    const long long M = (long long)d->N * d->Ho * d->Wo;
    M3D_REQUIRE(M > 0 && M < (1ll << 31) / 4, "conv2d: M out of range");
    M3D_REQUIRE((long long)d->N * d->H * d->W * d->in_cs * 4 < (1ll << 31), "conv2d: input view must be < 2 GiB (buffer offsets)");
    M3D_REQUIRE((long long)d->Cout_pad * d->kh * d->kw * d->Cin * 4 < (1ll << 31), "conv2d: weights must be < 2 GiB");
    if (d->dcn_offmask) M3D_REQUIRE(!d->out_nchw && d->Cout_pad % 64 == 0, "deformable conv: NHWC out, Cout_pad %% 64");
    if (d->out_nchw) M3D_REQUIRE(!d->res, "planar output does not take a residual");

    TileChoice t;
    M3D_REQUIRE(choose_tile(d, &t) == 0, "per-image weights need Ho*Wo %% 64 == 0");

    IgemmArgs a;
    a.in = d->in; a.wgt = d->wgt; a.out = d->out; a.scale = d->scale; a.shift = d->shift; a.res = d->res;
    a.om = d->dcn_offmask; a.zero = zero_page();
    M3D_REQUIRE(a.zero != nullptr, "conv2d: could not allocate the zero page");
    a.wgt_img_stride = d->wgt_img_stride; a.out_img_stride = d->out_img_stride;
    a.in_cs = d->in_cs; a.out_cs = d->out_cs; a.res_cs = d->res_cs; a.om_cs = d->dcn_om_cs;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.HoWo = d->Ho * d->Wo;
    a.Cout = d->Cout; a.Cout_pad = d->Cout_pad;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.M = (int)M; a.Ktot = d->kh * d->kw * d->Cin; a.KT = a.Ktot / t.bk;
    a.tiles_m = a.tiles_n = 0;
    a.act = d->act; a.sigmoid_from = d->sigmoid_from; a.res_mode = d->res_mode;
    {
        static int abl = -1;
        if (abl < 0) { const char *e = getenv("M3D_ABLATE"); abl = e ? atoi(e) : 0; }
        a.ablate = abl;
    }

    a.ws = nullptr; a.splits = 1; a.kt_per = a.KT;
#ifdef IGEMM_TRACE
    a.trace = g_igemm_trace;
#endif
    a.in_bytes = (unsigned)((long long)d->N * d->H * d->W * d->in_cs * 4);
    a.wgt_bytes = (unsigned)((long long)d->Cout_pad * d->kh * d->kw * d->Cin * 4);
    if (d->splitk_ws) {
        int kt_per;
        const int s = choose_split(d, t, &kt_per);
        if (s > 1) {
            M3D_REQUIRE((long long)s * M * d->Cout_pad * 4 <= d->splitk_ws_bytes && ((uintptr_t)d->splitk_ws & 15) == 0,
                        "conv2d: split-K workspace too small (%lld bytes, see m3d_conv2d_splitk_plan) or misaligned",
                        d->splitk_ws_bytes);
            a.ws = d->splitk_ws; a.splits = s; a.kt_per = kt_per;
        }
    }

    const bool deform = d->dcn_offmask != nullptr, swap = d->out_nchw != 0;
    if (deform) {
        if (t.bn == 128 && t.bm == 128) return launch_igemm<128, 128, 32, 2, 2, true, false>(a, stream);
        if (t.bn == 128 && t.bm == 64) return launch_igemm<64, 128, 32, 2, 2, true, false>(a, stream);
        if (t.bn == 64 && t.bm == 128) return launch_igemm<128, 64, 32, 2, 2, true, false>(a, stream);
        if (t.bn == 64 && t.bm == 64) return launch_igemm<64, 64, 32, 2, 2, true, false>(a, stream);
        M3D_REQUIRE(false, "deformable conv: unsupported tile (%d,%d,%d)", t.bm, t.bn, t.bk);
    }
    if (swap) {
        M3D_REQUIRE(t.bk == 32, "planar output needs Cin %% 32 == 0");
        if (t.bn == 32) return launch_igemm<128, 32, 32, 4, 1, false, true>(a, stream);
        if (t.bn == 64 && t.bm == 128) return launch_igemm<128, 64, 32, 2, 2, false, true>(a, stream);
        if (t.bn == 64 && t.bm == 64) return launch_igemm<64, 64, 32, 2, 2, false, true>(a, stream);
        if (t.bn == 128 && t.bm == 128) return launch_igemm<128, 128, 32, 2, 2, false, true>(a, stream);
        if (t.bn == 128 && t.bm == 64) return launch_igemm<64, 128, 32, 2, 2, false, true>(a, stream);
    }
    if (t.bk == 16) return launch_igemm<128, 32, 16, 4, 1, false, false>(a, stream);
    if (t.bn == 32) return launch_igemm<128, 32, 32, 4, 1, false, false>(a, stream);
    if (t.bn == 64 && t.bm == 128) return launch_igemm<128, 64, 32, 2, 2, false, false>(a, stream);
    if (t.bn == 64 && t.bm == 64) return launch_igemm<64, 64, 32, 2, 2, false, false>(a, stream);
    if (t.bn == 128 && t.bm == 128) return launch_igemm<128, 128, 32, 2, 2, false, false>(a, stream);
    if (t.bn == 128 && t.bm == 64) return launch_igemm<64, 128, 32, 2, 2, false, false>(a, stream);
    M3D_REQUIRE(false, "conv2d: no kernel for tile (%d,%d,%d)", t.bm, t.bn, t.bk);
}
