// 3x3 stride-1 convolution of the bf16 path by Winograd F(2x2, 3x3) on fp16 MFMA (model/pose_dla_dcn.py:107-121 BasicBlock convs of DLA
// level3 / level4, model/M3d_inference_align.py:66-75 cls.0):   out = act(conv3x3(x) * scale + shift (+ res))
//
// STATUS: measured alternative, NOT used by the plan (round 5).  Correct (tests/test_gpu_bf16.py: test_wino2_bf16_matches_torch) and
// slower than the direct wave-tile kernel it was meant to replace: 0.204 / 0.159 ms against 0.150 / 0.117 ms (128 -> 128 @ 48x160,
// 256 -> 256 @ 24x80, bs 64; tools/wino2_bench.py).  F(2x2, 3x3) needs 2.25x fewer MFMA passes, but with all 16 positions of an output
// block in one wave's registers (256 accumulators = a 32 x 32 block per position) no operand fragment is re-used: every MFMA wants a
// fresh 1 KB A fragment (transformed weights) and a fresh 1 KB B fragment (transformed input), where the direct kernel's 4 x 4 blocks
// need 0.25 KB each.  Timeline (tools/wino2_trace.py, diagnostic build; cycles per 32-channel chunk = 32 MFMAs per wave):
//   as built 2 750 | without the weight stream 2 528 | without the input transform 2 072 | without the B reads 2 660 | MFMAs only 660
// -- the weight stream alone (128 KB per CU and chunk through the vector memory path: 64 B / clk) costs 2 050, the LDS traffic (B
// fragments 128 KB + transform 92 KB per chunk) 2 500; splitting the positions over the waves to re-use fragments (4 positions x 2 x 2
// blocks) halves the weight stream but doubles the transform work per output (a V tile then serves 64 instead of 128 output
// channels) and leaves LDS at 2 400+.  At fp16 MFMA rates F(2x2) is operand-bound on this chip; it pays in the fp32 path
// (csrc/wino44_conv.hip), whose MFMAs are 16x slower.  Kept with its test and tools as the record of that measurement.
//
// Arithmetic: bf16 -> fp16 is exact, the input transform B^T d B adds four 8-bit significands (exact in 11 bits unless the
// exponents are far apart), the weights G g G^T are formed in fp32 (host, BatchNorm scale folded in) and rounded ONCE to fp16 (3 more
// bits than the bf16 weights of the direct kernels), products accumulate in fp32, the output transform A^T M A runs in fp32.
//
//   Workgroup = 256 threads = 4 waves, ONE per CU (a wave keeps all 16 Winograd positions of its output block in registers: 16 x 16 =
//   256 accumulators).  It owns an 8 x 16 block of output pixels = 4 x 8 = 32 tiles (the N of v_mfma_f32_32x32x16_f16) x 128 output
//   channels; wave w owns channels [32 w, 32 w + 32) and all 32 tiles.  Per 32-channel chunk of Cin:
//     * the 10 x 18 halo patch is staged in LDS as fp16 (`raw`, one conversion per pixel), 16-byte pieces ordered
//       [channel group][row][column parity][column / 2]: the stride-2 window reads of eight neighbouring tiles hit eight banks;
//     * every thread transforms (tile, 8 channels, half of the 16 positions): 12 ds_read_b128, 64 v_pk_add_f16, 8 ds_write_b128 into
//       V[position][tile][32 channels] (80-byte rows: the B-fragment reads of 8 neighbouring tiles are conflict-free);
//     * every wave multiplies: per position and 16-channel K step one MFMA, B fragment from V, A fragment = transformed weights
//       streamed global -> register in fragment order (m3dssd_amd/engine_bf16.py: pack_wino2; a ring of 16 fragments in flight).
//   V and raw are double-buffered: the transform of chunk c + 1 is interleaved with the MFMAs of chunk c, one barrier per chunk.
//   Output transform lane-local (a lane holds the 16 positions of 16 channels of one tile): 4 pixels x 16 consecutive channels per
//   lane, folded shift, residual, LeakyReLU, bf16, 16-byte stores.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifndef W2_ABL
#define W2_ABL 0                      // diagnostic builds: phase ablations of the K loop (tools/wino2_trace.py)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void w2_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w2_static_for<I + 1, N>(f);
    }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 w2_bf16x2 __attribute__((ext_vector_type(2)));

#define W2_PH 10                       // halo rows of an 8 x 16 output block
#define W2_PW 18                       // halo columns
#define W2_NPX (W2_PH * W2_PW)         // 180
#define W2_RAW_G (W2_NPX * 16)         // bytes of one channel group of the raw patch
#define W2_RAW (4 * W2_RAW_G)          // 11 520 bytes per buffer
#define W2_VROW 80                     // bytes per (position, tile) row of V: 32 fp16 + 16 pad
#define W2_VPOS (32 * W2_VROW)         // 2 560
#define W2_V (16 * W2_VPOS)            // 40 960 bytes per buffer
#define W2_LDS (2 * W2_V + 2 * W2_RAW)

#ifdef BF16_TRACE
// diagnostic build (make trace): s_memtime stamps of lane 0 of every wave, 16 slots per wave (tools/wino2_trace.py)
static long long *g_w2_trace = nullptr;
extern "C" void m3d_wino2_set_trace(void *buf) { g_w2_trace = (long long *)buf; }
#define W2T(i) do { if (trp && lane == 0) trp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define W2T(i)
#endif

struct Wino2Args {
    const void *in, *wfrag, *res;
    const float *shift;
    void *out;
    int in_cs, res_cs, out_cs;
    int N, H, W, Cin, Cout, bx, by, nchunks, cblocks, act;
    unsigned in_bytes, res_bytes, out_bytes, w_bytes;
#ifdef BF16_TRACE
    long long *trace;
#endif
};

__device__ __forceinline__ unsigned w2_bf16pair_to_f16(unsigned d)
{
    const f32x2 v = {__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ unsigned w2_pack_bf16(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, w2_bf16x2));
}

template <bool HAS_RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bf16_wino2_kernel(const Wino2Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const vbuf = lds, *const rawbuf = lds + 2 * W2_V;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef BF16_TRACE
    long long *trp = a.trace ? a.trace + ((size_t)blockIdx.x * 4 + wave) * 16 : nullptr;
#endif
    W2T(0);
    // workgroup -> (block, channel block); XCD-contiguous order (workgroup L runs on XCD L % 8), channel blocks of a block adjacent
    int L = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = L & 7, loc = L >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int cblk = L % a.cblocks;
    int blk = L / a.cblocks;
    const int per_img = a.bx * a.by;
    const int n = blk / per_img;
    blk -= n * per_img;
    const int byi = blk / a.bx, bxi = blk - byi * a.bx;
    const int oy0 = byi * 8, ox0 = bxi * 16;

    // ---- roles ------------------------------------------------------------------------------------------------------------------
    // staging: items (pixel, channel group) tid, tid + 256, tid + 512 of the 720 of a chunk
    const __amdgpu_buffer_rsrc_t rin = make_rsrc((const __bf16 *)a.in + (size_t)n * a.H * a.W * a.in_cs, (unsigned)a.H * a.W * a.in_cs * 2);
    unsigned st_voff[3], st_lds[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int id = tid + 256 * k;
        const int px = id >> 2, g = id & 3;
        const int row = px / W2_PW, col = px - row * W2_PW;
        const int y = oy0 - 1 + row, x = ox0 - 1 + col;
        const bool ok = id < 4 * W2_NPX && y >= 0 && y < a.H && x >= 0 && x < a.W;
        st_voff[k] = ok ? (unsigned)(((y * a.W + x) * a.in_cs + g * 8) * 2) : M3D_BUF_OOB;
        st_lds[k] = id < 4 * W2_NPX ? (unsigned)(g * W2_RAW_G + (row * W2_PW + (col & 1) * 9 + (col >> 1)) * 16) : 0xffffffffu;
    }
    // transform: tile = lane % 32, (channel group, half) from the thread's 32-lane group
    const int tile = lane & 31, tq = tid >> 5;
    const int tg = tq & 3, th = tq >> 2;                                  // th = 0: position rows 0, 1; th = 1: rows 2, 3
    const int tty = tile >> 3, ttx = tile & 7;
    // window rows X0, X1, X2 of this half (see the row stage below): th 0 -> 0, 1, 2; th 1 -> 2, 3, 1
    unsigned tr_rd[3];
    {
        const int xr[3] = {2 * th, 1 + 2 * th, 2 - th};
#pragma unroll
        for (int k = 0; k < 3; ++k) tr_rd[k] = (unsigned)(tg * W2_RAW_G + ((2 * tty + xr[k]) * W2_PW + ttx) * 16);
    }
    const unsigned tr_wr = (unsigned)((8 * th) * W2_VPOS + tile * W2_VROW + tg * 16);
    const _Float16 sg = th ? (_Float16)-1.f : (_Float16)1.f;
    const f16x8 sgn = {sg, sg, sg, sg, sg, sg, sg, sg};
    // multiply: B fragment of (position p, K step s) at V + p * VPOS + s * 32 + this
    const unsigned mb_off = (unsigned)(tile * W2_VROW + (lane >> 5) * 16);
    const int ws = cblk * 4 + wave;                                        // 32-channel slice of Cout
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wfrag, a.w_bytes);
    const unsigned w_base = (unsigned)ws * (unsigned)a.nchunks * (32u * 1024u) + (unsigned)lane * 16u;

    auto stage_load = [&](int c, u32x4 (&v)[3]) __attribute__((always_inline)) {
        const unsigned add = c < a.nchunks ? (unsigned)c * 64u : M3D_BUF_OOB;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rin, st_voff[k] | (add & M3D_BUF_OOB), add & ~M3D_BUF_OOB, 0);
    };
    auto stage_store = [&](int buf, const u32x4 (&v)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            u32x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = w2_bf16pair_to_f16(v[k][e]);
            if (st_lds[k] != 0xffffffffu) *reinterpret_cast<u32x4 *>(rawbuf + buf * W2_RAW + st_lds[k]) = h;
        }
    };
    // the whole transform of one chunk for this thread (prologue; the K loop runs the same steps interleaved with its MFMAs)
    auto tr_read = [&](int buf, int k, int cc) __attribute__((always_inline)) {
        return *reinterpret_cast<const f16x8 *>(rawbuf + buf * W2_RAW + tr_rd[k] + ((cc & 1) * 9 + (cc >> 1)) * 16);
    };
    auto tr_cols = [&](const f16x8 (&t)[4], int buf, int ii) __attribute__((always_inline)) {
        unsigned char *dst = vbuf + buf * W2_V + tr_wr + (4 * ii) * W2_VPOS;
        *reinterpret_cast<f16x8 *>(dst) = t[0] - t[2];
        *reinterpret_cast<f16x8 *>(dst + W2_VPOS) = t[1] + t[2];
        *reinterpret_cast<f16x8 *>(dst + 2 * W2_VPOS) = t[2] - t[1];
        *reinterpret_cast<f16x8 *>(dst + 3 * W2_VPOS) = t[1] - t[3];
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // ---- prologue: raw(0), raw(1) -> LDS, V(0) ------------------------------------------------------------------------------
    f16x8 wa[16];                                                          // ring of A fragments: fragment f lives in wa[f % 16]
#pragma unroll
    for (int f = 0; f < 16; ++f)
        wa[f] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, w_base + (unsigned)f * 1024u, 0, 0));
    {
        u32x4 v0[3], v1[3];
        stage_load(0, v0);
        stage_load(1, v1);
        stage_store(0, v0);
        stage_store(1, v1);
    }
    W2T(1);
    __syncthreads();
    W2T(2);
    {
        f16x8 d[3][4];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) d[k][cc] = tr_read(0, k, cc);
        f16x8 ta[4], tb[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { ta[cc] = d[0][cc] - d[2][cc]; tb[cc] = d[2][cc] + sgn * d[1][cc]; }
        tr_cols(ta, 0, 0);
        tr_cols(tb, 0, 1);
    }
    __syncthreads();
    W2T(3);

    // ---- K loop -----------------------------------------------------------------------------------------------------------------
    for (int c = 0; c < a.nchunks; ++c) {
        const int vb = c & 1, nb = vb ^ 1;
        u32x4 nx[3];
        stage_load(c + 2, nx);                                             // raw(c + 2) -> registers (masked past the last chunk)
        const unsigned char *vsrc = vbuf + vb * W2_V + mb_off;
        const unsigned wnext = w_base + (unsigned)(c * 32 + 16) * 1024u;
        f16x8 d[3][4], ta[4], tb[4];
        f16x8 q[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) q[i] = *reinterpret_cast<const f16x8 *>(vsrc + (i >> 1) * W2_VPOS + (i & 1) * 32);
        w2_static_for<0, 32>([&](auto fc) {
            constexpr int f = decltype(fc)::value;                         // fragment = (position f / 2, K step f % 2)
            constexpr int p = f >> 1;
#if !(W2_ABL & 4)
            if constexpr (f + 3 < 32) q[(f + 3) & 3] = *reinterpret_cast<const f16x8 *>(vsrc + ((f + 3) >> 1) * W2_VPOS + ((f + 3) & 1) * 32);
#endif
            // transform of chunk c + 1, one step per MFMA slot
#if !(W2_ABL & 2)
            if constexpr (f < 12) d[f >> 2][f & 3] = tr_read(nb, f >> 2, f & 3);
            else if constexpr (f < 16) { constexpr int cc = f - 12; ta[cc] = d[0][cc] - d[2][cc]; tb[cc] = d[2][cc] + sgn * d[1][cc]; }
            else if constexpr (f == 18) tr_cols(ta, nb, 0);
            else if constexpr (f == 22) tr_cols(tb, nb, 1);
            else if constexpr (f == 26) stage_store(vb, nx);               // raw(c + 2) -> the raw buffer chunk c read (free since the last barrier)
#endif
            __builtin_amdgcn_sched_barrier(0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[f & 15], q[f & 3], acc[p], 0, 0, 0);
#if !(W2_ABL & 1)
            wa[f & 15] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, wnext + (unsigned)f * 1024u, 0, 0));
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
#ifdef BF16_TRACE
        if (c < 4) W2T(4 + 2 * c);
#endif
        __syncthreads();
#ifdef BF16_TRACE
        if (c < 4) W2T(5 + 2 * c);
#endif
    }
    W2T(12);

    // ---- output transform + epilogue ----------------------------------------------------------------------------------------------
    // lane: tile -> output pixels (oy0 + 2 tty' + i, ox0 + 2 ttx' + j); channels ch0 + r, r = 0 .. 15
    const int oty = tile >> 3, otx = tile & 7;
    const int ch0 = ws * 32 + 16 * (lane >> 5);
    const __amdgpu_buffer_rsrc_t rout = make_rsrc((__bf16 *)a.out + (size_t)n * a.H * a.W * a.out_cs, (unsigned)a.H * a.W * a.out_cs * 2);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(HAS_RES ? (const __bf16 *)a.res + (size_t)n * a.H * a.W * a.res_cs : (const __bf16 *)a.out,
                                                  HAS_RES ? (unsigned)a.H * a.W * a.res_cs * 2 : 0u);
    unsigned pix[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pix[k] = (unsigned)((oy0 + 2 * oty + (k >> 1)) * a.W + ox0 + 2 * otx + (k & 1));
    const float slope = a.act ? M3D_LEAKY_SLOPE : 1.f;
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {                                       // 8 channels per pass: one 16-byte store per pixel
        u32x4 rv[4];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rv[k] = __builtin_amdgcn_raw_buffer_load_b128(rres, (pix[k] * (unsigned)a.res_cs + (unsigned)(ch0 + 8 * h8)) * 2u, 0, 0);
        }
        const f32x4 s0 = *reinterpret_cast<const f32x4 *>(a.shift + ch0 + 8 * h8), s1 = *reinterpret_cast<const f32x4 *>(a.shift + ch0 + 8 * h8 + 4);
        float y[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 8 * h8 + e;
            float t0[4], t1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t0[i] = acc[4 * i][r] + acc[4 * i + 1][r] + acc[4 * i + 2][r];
                t1[i] = acc[4 * i + 1][r] - acc[4 * i + 2][r] - acc[4 * i + 3][r];
            }
            const float sh = e < 4 ? s0[e & 3] : s1[e & 3];
            y[0][e] = t0[0] + t0[1] + t0[2] + sh;
            y[1][e] = t1[0] + t1[1] + t1[2] + sh;
            y[2][e] = t0[1] - t0[2] - t0[3] + sh;
            y[3][e] = t1[1] - t1[2] - t1[3] + sh;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            u32x4 o;
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                float lo = y[k][2 * e2], hi = y[k][2 * e2 + 1];
                if constexpr (HAS_RES) {
                    lo += __uint_as_float(rv[k][e2] << 16);
                    hi += __uint_as_float(rv[k][e2] & 0xffff0000u);
                }
                lo = fmaxf(lo, lo * slope);
                hi = fmaxf(hi, hi * slope);
                o[e2] = w2_pack_bf16(lo, hi);
            }
            buf_store_f32x4_nop(__builtin_bit_cast(f32x4, o), rout, (pix[k] * (unsigned)a.out_cs + (unsigned)(ch0 + 8 * h8)) * 2u, 0);
        }
    }
    W2T(13);
}

extern "C" int m3d_wino2_bf16_applicable(const m3d_wino2_bf16_desc *d)
{
    if (!d) return 0;
    if (d->Cin % 32 != 0 || d->Cin < 64 || d->Cout % 128 != 0 || d->H % 8 != 0 || d->W % 16 != 0) return 0;
    if (d->in_cs % 8 != 0 || d->out_cs % 8 != 0 || (d->res && d->res_cs % 8 != 0)) return 0;
    if ((long long)d->H * d->W * d->in_cs * 2 >= (1ll << 31) || (long long)d->H * d->W * d->out_cs * 2 >= (1ll << 31)) return 0;
    if (d->res && (long long)d->H * d->W * d->res_cs * 2 >= (1ll << 31)) return 0;
    if ((long long)d->Cout * d->Cin * 16 * 2 >= (1ll << 31)) return 0;
    return 1;
}

extern "C" int m3d_wino2_bf16_forward(const m3d_wino2_bf16_desc *d, m3d_stream_t stream)
{
    M3D_REQUIRE(d && d->in && d->wfrag && d->shift && d->out, "wino2_bf16: null pointer");
    M3D_REQUIRE(m3d_wino2_bf16_applicable(d), "wino2_bf16: needs Cin %% 32 == 0 (>= 64), Cout %% 128 == 0, H %% 8 == 0, W %% 16 == 0, pixel strides %% 8 == 0, views < 2 GiB per image");
    M3D_REQUIRE((((uintptr_t)d->in | (uintptr_t)d->out | (uintptr_t)d->res | (uintptr_t)d->wfrag | (uintptr_t)d->shift) & 15) == 0, "wino2_bf16: 16-byte aligned views");
    M3D_REQUIRE(d->act == 0 || d->act == 1, "wino2_bf16: act 0 / 1");
    Wino2Args a;
    a.in = d->in; a.wfrag = d->wfrag; a.res = d->res; a.shift = d->shift; a.out = d->out;
    a.in_cs = d->in_cs; a.res_cs = d->res_cs; a.out_cs = d->out_cs;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.bx = d->W / 16; a.by = d->H / 8;
    a.nchunks = d->Cin / 32; a.cblocks = d->Cout / 128; a.act = d->act;
    a.w_bytes = (unsigned)((long long)d->Cout * d->Cin * 16 * 2);
    a.in_bytes = a.res_bytes = a.out_bytes = 0;
#ifdef BF16_TRACE
    a.trace = g_w2_trace;
#endif
    const long long grid = (long long)a.bx * a.by * d->N * a.cblocks;
    M3D_REQUIRE(grid < (1ll << 31), "wino2_bf16: too many workgroups");
    static const int lds_ok = []() {
        int ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_wino2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS) == hipSuccess &&
                 hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_wino2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        return ok;
    }();
    M3D_REQUIRE(lds_ok, "wino2_bf16: cannot reserve %d bytes of LDS", (int)W2_LDS);
    if (d->res) hipLaunchKernelGGL(bf16_wino2_kernel<true>, dim3((unsigned)grid), dim3(256), W2_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(bf16_wino2_kernel<false>, dim3((unsigned)grid), dim3(256), W2_LDS, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
