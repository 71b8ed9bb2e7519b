// Fused 3-layer RPN head on bf16 MFMA (model/M3d_inference_align.py:77-210): per 128-pixel tile
//   [1x1 Cin(128) -> 256 + affine + LeakyReLU] -> [1x1 256 -> 256 + affine + LeakyReLU] -> [1x1 256 -> Cout + affine]
// in ONE launch; the two 256-channel hidden activations never leave the CU (unfused they are 2 x 252 MB per head at bs = 64,
// written and read back: 12 GB per step over the 12 heads).  Heads that read the same feature map share a launch
// (blockIdx.y = head).
//
//   Workgroup = 512 threads = 8 waves = 2 (pixel halves of 64) x 4 (channel quarters of 64); D rows = channels (MFMA A operand
//   = weights), D columns = pixels, as in bf16_conv.hip.  LDS: one 64 KB region holds the input tile [128 px][128 ch] during
//   layer 1 and then the hidden tile [128 px][256 ch] (layer 2 overwrites it in place after a barrier: every wave has
//   finished reading it when the accumulators are complete); weights stream through two 32 KB staging buffers in K-chunks of
//   64 ([rows][64 k], the chunk sequence runs across the three layers: 2 + 4 + 4 chunks), global loads of chunk i + 1 in
//   and i + 2 in flight under the MFMAs of chunk i (two register sets: an L2 round trip is longer than one chunk of MFMAs).  16-byte LDS chunks are XOR-swizzled by the row so that staging writes and fragment
//   reads are bank-conflict free.  Output: planar fp32 out[img][c][HW] (what m3d_anchor_select / m3d_align_offsets /
//   m3d_bundle_outputs consume).
#include <algorithm>
#include <type_traits>

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifdef BF16_TRACE
static long long *g_head_trace = nullptr;
extern "C" void m3d_bf16_head_set_trace(void *buf) { g_head_trace = (long long *)buf; }
#define HTRACE() do { if (trp && tid == 0 && tri < 32) trp[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define HTRACE()
#endif

struct HeadArgs {
    const void *in;                 // bf16 [M][in_cs]
    const void *w1, *w2, *w3;       // bf16 [G][256][Cin], [G][256][256], [G][Cout_pad][256]
    const float *s1, *t1, *s2, *t2, *s3, *t3;   // [G][256], [G][256], [G][Cout]
    float *out;                     // planar: out + g*out_goff + img*out_img_stride + c*HW + p
    long long out_goff, out_img_stride;
    int in_cs, M, HW, Cout, Cout_pad, tiles_m;
#ifdef BF16_TRACE
    long long *trace;
#endif
};

__device__ __forceinline__ unsigned hpack(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// byte offset of 16-byte chunk c of row r in the activation region (row = pixel, rb = bytes per row: 256 or 512)
__device__ __forceinline__ int act_off(int r, int c, int rb) { return r * rb + ((((c & 15) ^ (r & 15)) | (c & 16)) << 4); }
// byte offset of chunk c (0..7) of row r in a weight staging buffer (128-byte rows)
__device__ __forceinline__ int wst_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

__global__ __launch_bounds__(512) void bf16_head_mlp_kernel(const HeadArgs a)
{
    constexpr int CIN = 128, HID = 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536 + 2 * 32768 + 1152 * 4];
    unsigned char *act = lds, *wst = lds + 65536;
    float *aff = reinterpret_cast<float *>(lds + 65536 + 2 * 32768);      // s1 t1 s2 t2 [256 each], s3 t3 [64 each] of this head

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef BF16_TRACE
    long long *trp = a.trace ? a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 : nullptr;
    int tri = 0;
#endif
    HTRACE();
    // layers 1, 2: wave = 64 px x 64 ch (2 x 2 tiles); layer 3 (64 channels): wave = 32 px x 32 ch
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 64;
    const int wm3 = (wave >> 1) * 32, wn3 = (wave & 1) * 32;
    const int l31 = lane & 31, lh = lane >> 5;
    const int g = blockIdx.y;
    const __bf16 *w1 = (const __bf16 *)a.w1 + (size_t)g * HID * CIN;
    const __bf16 *w2 = (const __bf16 *)a.w2 + (size_t)g * HID * HID;
    const __bf16 *w3 = (const __bf16 *)a.w3 + (size_t)g * a.Cout_pad * HID;

    // ---- weight chunk pipeline: chunk i of the sequence [L1: 0,1] [L2: 2..5] [L3: 6..9], TWO chunks in flight ---------------
    const int chunk = tid & 7, rsub = tid >> 3;          // thread = (row within a 64-row pass, 16-byte piece of the 128-byte line)
    const int wsto = rsub * 128 + ((chunk ^ ((rsub >> 1) & 7)) << 4);     // staging offset of pass 0 (pass p: + p * 8192)
    u32x4 rw[2][4];
    auto load_chunk = [&](auto itag, auto rtag) {
        constexpr int i = decltype(itag)::value, R = decltype(rtag)::value;
        const __bf16 *w = i < 2 ? w1 : (i < 6 ? w2 : w3);
        constexpr int K = i < 2 ? CIN : HID, kc = i < 2 ? i : (i < 6 ? i - 2 : i - 6), passes = i < 6 ? 4 : 1;
        const __bf16 *src = w + (size_t)rsub * K + kc * 64 + chunk * 8;
#pragma unroll
        for (int p = 0; p < passes; ++p) rw[R][p] = *reinterpret_cast<const u32x4 *>(src + (size_t)p * 64 * K);
    };
    auto store_chunk = [&](auto itag, auto rtag, auto btag) {
        constexpr int i = decltype(itag)::value, R = decltype(rtag)::value, buf = decltype(btag)::value;
        constexpr int passes = i < 6 ? 4 : 1;
#pragma unroll
        for (int p = 0; p < passes; ++p) *reinterpret_cast<u32x4 *>(wst + buf * 32768 + p * 8192 + wsto) = rw[R][p];
    };
#define IC(n) std::integral_constant<int, n>{}

    // ---- affine parameters of the head -> LDS once (read from global memory inside write_hidden they cost a memory round trip
    // per layer with the whole CU waiting: the workgroup is alone on its CU) ---------------------------------------------------
    if (tid < HID) {
        aff[tid] = a.s1[g * HID + tid]; aff[HID + tid] = a.t1[g * HID + tid];
        aff[2 * HID + tid] = a.s2[g * HID + tid]; aff[3 * HID + tid] = a.t2[g * HID + tid];
    }
    if (tid < 64) {
        aff[4 * HID + tid] = tid < a.Cout ? a.s3[g * a.Cout + tid] : 0.f;
        aff[4 * HID + 64 + tid] = tid < a.Cout ? a.t3[g * a.Cout + tid] : 0.f;
    }
    // The workgroup walks over pixel tiles (grid.x ~ CUs / heads): the next tile's input and the first two weight chunks are
    // fetched while the current tile computes (in-kernel trace of the one-tile-per-workgroup form: 4400 of 28000 cycles waiting
    // for the input tile, 5100 in the 4-byte output stores, 2 x 3400 in the hidden-tile writes).
    const int c16 = tid & 15, r0 = tid >> 4;             // input staging: 16 pieces per row, 32 rows per pass
    u32x4 vin[4];
    auto load_input = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = tile * 128 + p * 32 + r0;
            vin[p] = u32x4{0u, 0u, 0u, 0u};
            if (tile < a.tiles_m && m < a.M) vin[p] = *reinterpret_cast<const u32x4 *>((const __bf16 *)a.in + (size_t)m * a.in_cs + c16 * 8);
        }
    };
    load_input(blockIdx.x);
    load_chunk(IC(0), IC(0));
    load_chunk(IC(1), IC(1));

    // ---- fragment addresses, hoisted: every term that depends on the lane is computed ONCE (a VALU instruction issued next
    // to a SIMD's MFMA stream costs it ~12 cycles on this part); what varies inside the loops is an immediate offset --------
    //   activation chunk index c = 8*kc + 2*s + lh; swizzled piece = (c & 16) | ((c & 15) ^ (row & 15))
    //   = 16*(kc >> 1)  |  8*((kc & 1) ^ ((row >> 3) & 1))  |  ((2*s + lh) ^ (row & 7))           (rows = x*32 + l31)
    int pre[4], prew[4];
#pragma unroll
    for (int sI = 0; sI < 4; ++sI) {
        pre[sI] = ((2 * sI + lh) ^ (l31 & 7)) << 4;
        prew[sI] = ((2 * sI + lh) ^ ((l31 >> 1) & 7)) << 4;
    }
    const int t3 = (l31 >> 3) & 1;
    const int apar[2] = {t3 << 7, (t3 ^ 1) << 7};

    f32x16 acc[2][2];                                     // [channel tile][pixel tile]
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    };
    // one K-chunk of 64 of layers 1 / 2: wave = 64 px x 64 ch; RB = bytes per activation row
    auto mma_chunk = [&](auto btag, auto kctag, auto rbtag) {
        constexpr int buf = decltype(btag)::value, kc = decltype(kctag)::value, RB = decltype(rbtag)::value;
        const unsigned char *wb = wst + buf * 32768 + (wn + l31) * 128;
        const unsigned char *ab = act + (wm + l31) * RB + apar[kc & 1] + ((kc >> 1) << 8);
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
            bf16x8 fw[2], fp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[j] = *reinterpret_cast<const bf16x8 *>(wb + j * 4096 + prew[sI]);
#pragma unroll
            for (int i = 0; i < 2; ++i) fp[i] = *reinterpret_cast<const bf16x8 *>(ab + i * 32 * RB + pre[sI]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fp[i], acc[j][i], 0, 0, 0);
        }
    };
    // affine + LeakyReLU on the accumulators, bf16, into the hidden tile [128 px][256 ch] (512-byte rows)
    auto write_hidden = [&](const float *sc, const float *sh) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cb = wn + j * 32 + 4 * lh;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned pk[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4 *>(sc + cb + 8 * q), t4 = *reinterpret_cast<const f32x4 *>(sh + cb + 8 * q);
                    // packed fp32 (v_pk_fma_f32 / v_pk_mul_f32): the hidden-tile writes are VALU-bound (3300 cycles per layer)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const f32x2 x = {acc[j][i][4 * q + e], acc[j][i][4 * q + e + 1]};
                        const f32x2 r = x * f32x2{s4[e], s4[e + 1]} + f32x2{t4[e], t4[e + 1]};
                        const f32x2 l = r * M3D_LEAKY_SLOPE;
                        pk[q][e >> 1] = hpack(fmaxf(r[0], l[0]), fmaxf(r[1], l[1]));
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q += 2)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(pk[q][e], pk[q + 1][e], false, false);
                        pk[q][e] = r[0]; pk[q + 1][e] = r[1];
                    }
                const int px = wm + i * 32 + l31;
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    const int c0 = wn + j * 32 + 8 * (q + lh);                  // 8 consecutive channels of this pixel
                    *reinterpret_cast<u32x4 *>(act + act_off(px, c0 >> 3, 512)) = u32x4{pk[q][0], pk[q][1], pk[q + 1][0], pk[q + 1][1]};
                }
            }
        }
    };

    const float *s1 = aff, *t1 = aff + HID, *s2 = aff + 2 * HID, *t2 = aff + 3 * HID, *s3 = aff + 4 * HID, *t3p = aff + 4 * HID + 64;
    for (int tile = blockIdx.x; tile < a.tiles_m; tile += gridDim.x) {
    const int m0 = tile * 128;
    // ---- stage the input tile [128 px][128 ch] (256-byte rows) and weight chunk 0; the next tile's input goes in flight --------
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4 *>(act + act_off(p * 32 + r0, c16, 256)) = vin[p];
    store_chunk(IC(0), IC(0), IC(0));
    load_input(tile + gridDim.x);
    __syncthreads();
    HTRACE();
    // step i: [load chunk i+2 -> register set i&1]  [store chunk i+1 -> buffer (i+1)&1: free since the barrier of step i-1]
    // MFMAs of chunk i (LDS buffer i&1)  barrier -- the staging writes (13 LDS cycles per ds_write_b128) run under the MFMAs
    // ---- layer 1: K = 128 (chunks 0, 1) ------------------------------------------------------------------------------------
    zero_acc();
    load_chunk(IC(2), IC(0));
    store_chunk(IC(1), IC(1), IC(1));
    mma_chunk(IC(0), IC(0), IC(256));
    __syncthreads();
    HTRACE();
    load_chunk(IC(3), IC(1));
    store_chunk(IC(2), IC(0), IC(0));
    mma_chunk(IC(1), IC(1), IC(256));
    __syncthreads();                                      // every wave is done reading the input tile
    write_hidden(s1, t1);
    __syncthreads();
    HTRACE();
    // ---- layer 2: K = 256 (chunks 2..5), hidden tile read from and written back to the same LDS region ----------------------
    zero_acc();
    load_chunk(IC(4), IC(0));
    store_chunk(IC(3), IC(1), IC(1));
    mma_chunk(IC(0), IC(0), IC(512));
    __syncthreads();
    HTRACE();
    load_chunk(IC(5), IC(1));
    store_chunk(IC(4), IC(0), IC(0));
    mma_chunk(IC(1), IC(1), IC(512));
    __syncthreads();
    HTRACE();
    load_chunk(IC(6), IC(0));
    store_chunk(IC(5), IC(1), IC(1));
    mma_chunk(IC(0), IC(2), IC(512));
    __syncthreads();
    HTRACE();
    load_chunk(IC(7), IC(1));
    store_chunk(IC(6), IC(0), IC(0));
    mma_chunk(IC(1), IC(3), IC(512));
    __syncthreads();
    HTRACE();
    write_hidden(s2, t2);
    __syncthreads();
    HTRACE();
    // ---- layer 3: K = 256 (chunks 6..9), 64 output channels: wave = 32 px x 32 ch ---------------------------------------------
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    auto mma3 = [&](auto btag, auto kctag) {
        constexpr int buf = decltype(btag)::value, kc = decltype(kctag)::value;
        const unsigned char *wb = wst + buf * 32768 + (wn3 + l31) * 128;
        const unsigned char *ab = act + (wm3 + l31) * 512 + apar[kc & 1] + ((kc >> 1) << 8);
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
            const bf16x8 fw = *reinterpret_cast<const bf16x8 *>(wb + prew[sI]);
            const bf16x8 fp = *reinterpret_cast<const bf16x8 *>(ab + pre[sI]);
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw, fp, o, 0, 0, 0);
        }
    };
    load_chunk(IC(8), IC(0));
    store_chunk(IC(7), IC(1), IC(1));
    mma3(IC(0), IC(0));
    __syncthreads();
    HTRACE();
    load_chunk(IC(9), IC(1));
    store_chunk(IC(8), IC(0), IC(0));
    mma3(IC(1), IC(1));
    __syncthreads();
    HTRACE();
    store_chunk(IC(9), IC(1), IC(1));
    load_chunk(IC(0), IC(0));                             // both register sets are free: the next tile's first two chunks
    load_chunk(IC(1), IC(1));
    mma3(IC(0), IC(2));
    __syncthreads();
    HTRACE();
    mma3(IC(1), IC(3));
    HTRACE();
    // ---- output: accumulators (lane = pixel, channels wn3 + 8q + 4*lh + e) -> LDS [64 ch][128 px] fp32 -> planar fp32 rows,
    // 16 bytes per lane when the tile lies inside one image (4-byte stores straight from the accumulators: 5100 cycles) --------
    __syncthreads();                                      // every wave is done with the hidden tile
    HTRACE();
    float *ot = reinterpret_cast<float *>(act);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) ot[(wn3 + 8 * q + 4 * lh + e) * 128 + wm3 + l31] = o[4 * q + e];
    __syncthreads();
    if (a.HW % 128 == 0) {
        const int img = m0 / a.HW, p0 = m0 - img * a.HW;
        float *ob = a.out + g * a.out_goff + (size_t)img * a.out_img_stride + p0;
        for (int i = tid; i < a.Cout * 32; i += 512) {
            const int c = i >> 5, p4 = (i & 31) * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(ot + c * 128 + p4);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = v[e] * s3[c] + t3p[c];
            *reinterpret_cast<f32x4 *>(ob + (size_t)c * a.HW + p4) = r;
        }
    } else {
        for (int i = tid; i < a.Cout * 128; i += 512) {
            const int c = i >> 7, m = m0 + (i & 127);
            if (m < a.M) {
                const int img = m / a.HW, pp = m - img * a.HW;
                a.out[g * a.out_goff + (size_t)img * a.out_img_stride + (size_t)c * a.HW + pp] = ot[i] * s3[c] + t3p[c];
            }
        }
    }
    __syncthreads();                                      // the next tile's input overwrites the region
    HTRACE();
    }
#undef IC
}

extern "C" int m3d_head_mlp_bf16_forward(const m3d_head_bf16_desc *d, m3d_stream_t stream)
{
    M3D_REQUIRE(d && d->in && d->w1 && d->w2 && d->w3 && d->out && d->s1 && d->t1 && d->s2 && d->t2 && d->s3 && d->t3,
                "head_mlp_bf16: null pointer");
    M3D_REQUIRE(d->Cin == 128 && d->in_cs % 8 == 0 && ((uintptr_t)d->in & 15) == 0, "head_mlp_bf16: Cin must be 128, 16-byte aligned rows");
    M3D_REQUIRE(d->Cout >= 1 && d->Cout <= 64 && d->Cout_pad == 64, "head_mlp_bf16: Cout <= 64, Cout_pad == 64");
    M3D_REQUIRE(d->groups >= 1 && d->M >= 1 && d->HW >= 1, "head_mlp_bf16: bad sizes");
    HeadArgs a;
    a.in = d->in; a.w1 = d->w1; a.w2 = d->w2; a.w3 = d->w3; a.s1 = d->s1; a.t1 = d->t1; a.s2 = d->s2; a.t2 = d->t2;
    a.s3 = d->s3; a.t3 = d->t3; a.out = d->out; a.out_goff = d->out_group_off; a.out_img_stride = d->out_img_stride;
    a.in_cs = d->in_cs; a.M = (int)d->M; a.HW = d->HW; a.Cout = d->Cout; a.Cout_pad = d->Cout_pad; a.tiles_m = cdiv(d->M, 128);
#ifdef BF16_TRACE
    a.trace = g_head_trace;
#endif
    // one workgroup per CU (132 KB of LDS): the CUs are split between the heads of the launch and every workgroup walks tiles
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    }
    const int nb = std::max(1, std::min(a.tiles_m, ncu / d->groups));
    hipLaunchKernelGGL(bf16_head_mlp_kernel, dim3(nb, d->groups), dim3(512), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
