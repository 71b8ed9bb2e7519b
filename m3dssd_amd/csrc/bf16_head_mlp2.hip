// Fused 3-layer RPN head of the bf16 path, second form (round 5; model/M3d_inference_align.py:77-210): per 128-pixel tile
//   [1x1 128 -> 256 + affine + LeakyReLU] -> [1x1 256 -> 256 + affine + LeakyReLU] -> [1x1 256 -> Cout + affine]
// with the weights of layers 1 and 2 RESIDENT IN REGISTERS for the whole launch.
//
// What bound the first form (bf16_head_mlp.hip, 0.27 of the bf16 MFMA peak, 22 500 cycles per tile of which 7 200 MFMA): every tile
// re-staged all 224 KB of weights through LDS in ten 32 KB chunks, each behind a workgroup barrier (14 barriers per tile), and its
// epilogues (folded BatchNorm + LeakyReLU + bf16 packing of 2 x 32 K hidden values, in fp32) ran with the matrix pipes idle.  Here:
//   * Workgroup = 512 threads = 8 waves = one CU; wave w owns output channels [32w, 32w + 32) of layers 1 and 2 for ALL 128 pixels
//     of a tile: its slice of W1 (8 K-steps) and W2 (16 K-steps) is 24 A-operand fragments = 96 registers, fetched once per
//     launch (the host packs them in fragment order with the BatchNorm SCALE multiplied in: m3dssd_amd/engine_bf16.py:
//     pack_head2).  The B operand (pixels) comes from the shared activation tile in LDS; four pixel blocks = four independent
//     accumulator chains per wave.  No weight staging, five barriers per tile.
//   * The BatchNorm SHIFT is the C operand of the first MFMA of each chain; the hidden activations are fp16 (layers 2 and 3 run
//     on v_mfma_f32_32x32x16_f16; 11-bit significand, BatchNorm-scale values), so that what is left of an epilogue is cvt_pk +
//     pk_mul + pk_max on the packed-fp16 pipe: 1.5 instructions per element instead of 2.75.
//   * MFMA rows are mapped to channels so that a lane owns 16 CONSECUTIVE channels of its pixel (row r -> channel 16 * ((r % 8)
//     / 4) + 4 * (r / 8) + r % 4): the hidden tile is written with two 16-byte stores per pixel block, no lane exchange.
//   * Layer 3 (Cout_pad = 64): wave = (channel half, pixel quarter), W3 (32 KB, scale folded) staged in LDS once per launch.
//   * A tile runs as two halves of 64 pixels per layer (32 accumulator registers instead of 64: the resident weights leave no more),
//     two pixel blocks = two independent MFMA chains at a time.
// LDS: input tile 32 KB (later h2 of the first half tile), hidden tile 64 KB (h1; h2 of the second half and the per-wave output
// transposition move into rows every wave has finished reading), W3 32 KB, shifts 2.3 KB = 130 KB.  fp16 range: |x| <= 65504 for hidden activations and the scaled W2 / W3.
#include <algorithm>

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifdef BF16_TRACE
static long long *g_head2_trace = nullptr;
extern "C" void m3d_bf16_head2_set_trace(void *buf) { g_head2_trace = (long long *)buf; }
#define H2TRACE() do { if (trp && tid == 0 && iter == 1 && tri < 16) trp[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define H2TRACE()
#endif

struct Head2Args {
    const void *in;                 // bf16 [M][in_cs], first 128 channels
    const void *w1f;                // bf16 fragments [G][8 waves][8 K-steps][64 lanes][8]   (scale folded)
    const void *w2f;                // fp16 fragments [G][8 waves][16 K-steps][64 lanes][8]  (scale folded)
    const void *w3;                 // fp16 [G][64][256] row-major (scale folded, rows >= Cout zero)
    const float *t1, *t2, *t3;      // shifts [G][256], [G][256], [G][64]
    float *out;                     // planar: out + g*out_goff + img*out_img_stride + c*HW + p
    long long out_goff, out_img_stride;
    int in_cs, M, HW, Cout, tiles_m;
#ifdef BF16_TRACE
    long long *trace;
#endif
};

#define H2_IN 0                     // [128 px][128 ch] bf16, 256-byte rows; later h2 of pixels 0-63 (512-byte rows)
#define H2_H 32768                  // [128 px][256 ch] fp16, 512-byte rows
#define H2_W3 (32768 + 65536)       // [64 ch][256 k] fp16, 512-byte rows
#define H2_SH (32768 + 65536 + 32768)   // t1 [256], t2 [256], t3 [64] fp32
#define H2_LDS (H2_SH + 576 * 4)

// byte offset of 16-byte chunk c of row r (rb = bytes per row, 256 or 512): chunks XOR-swizzled by the row inside each 256-byte half
__device__ __forceinline__ int h2_off(int r, int c, int rb) { return r * rb + ((((c & 15) ^ (r & 15)) | (c & 16)) << 4); }

// The hidden activations are fp16: a value past +-65504 would convert to +-inf and turn into NaN in the next layer's MFMA
// (inf * 0 weight padding, inf - inf) without any signal.  M3D_F16_SATURATE (default on) clamps the packed pair to the
// finite range -- two packed-fp16 instructions per pair -- so that an out-of-range activation degrades like a saturating
// quantiser instead of poisoning the planar outputs; -DM3D_F16_SATURATE=0 builds the unclamped form.
#ifndef M3D_F16_SATURATE
#define M3D_F16_SATURATE 1
#endif
__device__ __forceinline__ f16x2 f16_sat(f16x2 y)
{
#if M3D_F16_SATURATE
    const f16x2 hi = {(_Float16)65504.0f, (_Float16)65504.0f};
    return __builtin_elementwise_max(__builtin_elementwise_min(y, hi), -hi);
#else
    return y;
#endif
}

__device__ __forceinline__ unsigned h2_leaky_pack(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    const f16x2 y = f16_sat(__builtin_convertvector(v, f16x2));
    const f16x2 sl = {(_Float16)M3D_LEAKY_SLOPE, (_Float16)M3D_LEAKY_SLOPE};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(y, y * sl));
}

__global__ __launch_bounds__(512) void bf16_head2_kernel(const Head2Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[H2_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int g = blockIdx.y;
    int iter = 0;
    (void)iter;
#ifdef BF16_TRACE
    long long *trp = a.trace ? a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 : nullptr;
    int tri = 0;
#endif

    // ---- once per workgroup: this wave's W1 / W2 fragments -> registers, W3 and the shifts -> LDS -------------------------------
    bf16x8 w1[8];
    f16x8 w2[16];
    {
        const bf16x8 *p1 = reinterpret_cast<const bf16x8 *>(a.w1f) + ((size_t)(g * 8 + wave) * 8) * 64 + lane;
        const f16x8 *p2 = reinterpret_cast<const f16x8 *>(a.w2f) + ((size_t)(g * 8 + wave) * 16) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 8; ++s) w1[s] = p1[s * 64];
#pragma unroll
        for (int s = 0; s < 16; ++s) w2[s] = p2[s * 64];
    }
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>((const _Float16 *)a.w3 + (size_t)g * 64 * 256);
#pragma unroll
        for (int p = 0; p < 4; ++p) {                              // 64 rows x 32 chunks = 2048 chunks / 512 threads
            const int i = tid + p * 512, r = i >> 5, c = i & 31;
            *reinterpret_cast<u32x4 *>(lds + H2_W3 + h2_off(r, c, 512)) = src[i];
        }
        float *sh = reinterpret_cast<float *>(lds + H2_SH);
        if (tid < 256) { sh[tid] = a.t1[g * 256 + tid]; sh[256 + tid] = a.t2[g * 256 + tid]; }
        if (tid < 64) sh[512 + tid] = a.t3[g * 64 + tid];
    }
    const float *sh = reinterpret_cast<const float *>(lds + H2_SH);

    // input staging: 16 pieces of 16 bytes per row, 32 rows per pass; the next tile's input is in flight under the current tile
    const int c16 = tid & 15, r0 = tid >> 4;
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

    // B-operand addresses: pixel row cb * 32 + l31, chunk 2 * s + lh of the K dimension.  swizzled piece = (c & 16) | ((c & 15) ^ (row & 15));
    // with c = 2s + lh:  ((2s + lh) & 15) ^ (l31 & 15) -- the (l31 & 15) term is a lane constant, the s term a compile-time constant, and
    // XOR distributes: piece = (lanepart ^ (2s & 15)) with lanepart = lh ^ (l31 & 15)  (bit 0 of 2s is 0, so lh folds into the lane term)
    const int lanepart = (lh ^ (l31 & 15)) << 4;
    const unsigned char *inrow = lds + H2_IN + l31 * 256;
    const unsigned char *hrow = lds + H2_H + l31 * 512;
    auto frag_in = [&](int cb, int s) {                         // layer 1: K = 128 = 8 K-steps, 256-byte rows
        return *reinterpret_cast<const bf16x8 *>(inrow + cb * (32 * 256) + (lanepart ^ ((2 * s) << 4)));
    };
    auto frag_h = [&](int cb, int s) {                          // layers 2 / 3: K = 256 = 16 K-steps, 512-byte rows
        return *reinterpret_cast<const f16x8 *>(hrow + cb * (32 * 512) + ((lanepart ^ (((2 * s) & 15) << 4)) | (((2 * s) & 16) << 4)));
    };
    // epilogue of layers 1 / 2 for one half tile (2 pixel blocks): the lane holds channels 32 * wave + 16 * lh + j (j = 0..15) of pixel
    // 32 * cb + l31; `dst` = first row of the half's 64 rows (512-byte rows)
    auto write_hidden = [&](const f32x16 (&acc)[2], unsigned char *dst) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            u32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o0[e] = h2_leaky_pack(acc[cb][2 * e], acc[cb][2 * e + 1]);
                o1[e] = h2_leaky_pack(acc[cb][8 + 2 * e], acc[cb][8 + 2 * e + 1]);
            }
            const int px = cb * 32 + l31, c0 = 4 * wave + 2 * lh;
            *reinterpret_cast<u32x4 *>(dst + h2_off(px, c0, 512)) = o0;
            *reinterpret_cast<u32x4 *>(dst + h2_off(px, c0 + 1, 512)) = o1;
        }
    };
    auto load_shift = [&](int base) {                           // 16 consecutive floats -> the C operand of a chain's first MFMA
        f32x16 t;
        const f32x4 *p = reinterpret_cast<const f32x4 *>(sh + base);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = p[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[4 * q + e] = v[e];
        }
        return t;
    };
    // one half tile (64 pixels = 2 pixel blocks = 2 independent chains) of layer 1 (bf16, K = 128) / layer 2 (fp16, K = 256): the
    // fragment of step i + 3 is requested before the MFMA of step i (ring of 4)
    auto layer1_half = [&](int half, f32x16 (&acc)[2], const f32x16 &t) {
        bf16x8 q[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) q[i] = frag_in(2 * half + (i & 1), i >> 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cb = i & 1, s = i >> 1;
            if (i + 3 < 16) q[(i + 3) & 3] = frag_in(2 * half + ((i + 3) & 1), (i + 3) >> 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[s], q[i & 3], s == 0 ? t : acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto layer2_half = [&](int half, f32x16 (&acc)[2], const f32x16 &t) {
        f16x8 q[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) q[i] = frag_h(2 * half + (i & 1), i >> 1);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int cb = i & 1, s = i >> 1;
            if (i + 3 < 32) q[(i + 3) & 3] = frag_h(2 * half + ((i + 3) & 1), (i + 3) >> 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s], q[i & 3], s == 0 ? t : acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Where things live during a tile (H = the 64 KB hidden region, rows of 512 bytes; IN = the 32 KB input region):
    //   h1: H rows 0-127.  h2 of pixels 0-63: IN (the input is dead behind barrier B), of pixels 64-127: H rows 0-63 (h1 rows every
    //   wave has read when it passes barrier C).  Output transposition: H rows 64-127 (dead behind barrier D), 4 KB per wave.
    for (int tile = blockIdx.x; tile < a.tiles_m; tile += gridDim.x, ++iter) {
        const int m0 = tile * 128;
        H2TRACE();
        // ---- input tile -> LDS (swizzled), next tile's input in flight ---------------------------------------------------------
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4 *>(lds + H2_IN + h2_off(p * 32 + r0, c16, 256)) = vin[p];
        load_input(tile + gridDim.x);
        __syncthreads();                                        // A: input tile (and, first time, W3 / shifts) complete
        H2TRACE();
        f32x16 acc[2];
        // ---- layer 1 (the hidden region is free: every wave passed barrier E behind layer 3 of the previous tile) ------------------
        {
            const f32x16 t = load_shift(32 * wave + 16 * lh);
            layer1_half(0, acc, t);
            write_hidden(acc, lds + H2_H);
            layer1_half(1, acc, t);
            write_hidden(acc, lds + H2_H + 64 * 512);
        }
        H2TRACE();
        __syncthreads();                                        // B: h1 complete, input tile dead
        H2TRACE();
        // ---- layer 2 ---------------------------------------------------------------------------------------------------------------
        {
            const f32x16 t = load_shift(256 + 32 * wave + 16 * lh);
            layer2_half(0, acc, t);
            write_hidden(acc, lds + H2_IN);                     // h2 of pixels 0-63 -> the input region
            layer2_half(1, acc, t);
            H2TRACE();
            __syncthreads();                                    // C: every wave has read h1 rows 0-63 (and is past its layer-2 MFMAs)
            write_hidden(acc, lds + H2_H);                      // h2 of pixels 64-127 -> H rows 0-63
        }
        __syncthreads();                                        // D: h2 complete, h1 dead
        H2TRACE();
        // ---- layer 3: wave = (channel half rb, pixel quarter cb3), 16 K-steps, W3 from LDS ---------------------------------------
        const int rb = wave & 1, cb3 = wave >> 1;
        f32x16 o;
        {
            // rows of this block: channel 32 * rb + 8 * (j / 4) + 4 * lh + j % 4 -> the shift vector is gathered per register
            f32x16 t;
#pragma unroll
            for (int j = 0; j < 16; ++j) t[j] = sh[512 + 32 * rb + 8 * (j >> 2) + 4 * lh + (j & 3)];
            const unsigned char *w3row = lds + H2_W3 + (32 * rb + l31) * 512;
            const unsigned char *h2row = lds + (cb3 < 2 ? H2_IN : H2_H) + ((cb3 & 1) * 32 + l31) * 512;
            auto frag_w3 = [&](int s) {
                return *reinterpret_cast<const f16x8 *>(w3row + ((lanepart ^ (((2 * s) & 15) << 4)) | (((2 * s) & 16) << 4)));
            };
            auto frag_h2 = [&](int s) {
                return *reinterpret_cast<const f16x8 *>(h2row + ((lanepart ^ (((2 * s) & 15) << 4)) | (((2 * s) & 16) << 4)));
            };
            f16x8 qa[4], qb[4];
#pragma unroll
            for (int s = 0; s < 2; ++s) { qa[s] = frag_w3(s); qb[s] = frag_h2(s); }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 2 < 16) { qa[(s + 2) & 3] = frag_w3(s + 2); qb[(s + 2) & 3] = frag_h2(s + 2); }
                __builtin_amdgcn_sched_barrier(0);
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s & 3], qb[s & 3], s == 0 ? t : o, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        H2TRACE();
        // ---- output: accumulators (lane = pixel, 16 channels) -> this wave's 4 KB of H rows 64-127 [32 ch][32 px] fp32 -> planar fp32
        // rows, 16 bytes per lane when the pixel block lies inside one image
        float *ot = reinterpret_cast<float *>(lds + H2_H + 64 * 512 + wave * 4096);
#pragma unroll
        for (int j = 0; j < 16; ++j) ot[(8 * (j >> 2) + 4 * lh + (j & 3)) * 32 + l31] = o[j];
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): the wave's own LDS writes (no other wave reads them)
        __builtin_amdgcn_wave_barrier();
        const int mq = m0 + 32 * cb3;                           // first pixel of this wave's block
        if (a.HW % 32 == 0 && mq + 32 <= a.M) {
            const int img = mq / a.HW, p0 = mq - img * a.HW;
            float *ob = a.out + g * a.out_goff + (size_t)img * a.out_img_stride + p0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lane + k * 64, c = i >> 3, p4 = (i & 7) * 4;   // 32 channels x 8 pieces of 4 pixels
                const int ch = 32 * rb + c;
                if (ch < a.Cout) *reinterpret_cast<f32x4 *>(ob + (size_t)ch * a.HW + p4) = *reinterpret_cast<const f32x4 *>(ot + c * 32 + p4);
            }
        } else {
            for (int i = lane; i < 32 * 32; i += 64) {
                const int c = i >> 5, m = mq + (i & 31), ch = 32 * rb + c;
                if (m < a.M && ch < a.Cout) {
                    const int img = m / a.HW, pp = m - img * a.HW;
                    a.out[g * a.out_goff + (size_t)img * a.out_img_stride + (size_t)ch * a.HW + pp] = ot[i];
                }
            }
        }
        H2TRACE();
        __syncthreads();                                        // E: the input region (h2 of pixels 0-63) and the hidden region are free
        H2TRACE();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Two-layer tail of the class head (M3d_inference_align.py:66-76: cls.3 = 1x1 256 -> 256 + BatchNorm + LeakyReLU, cls.6 = 1x1 256 ->
// 4 * A + bias; the 3x3 cls.0 in front of them is a convolution launch): the same scheme -- wave w owns rows [32w, 32w + 32) of BOTH
// layers with their 2 x 16 fragments resident in registers (waves whose rows lie past Cout sit layer 2 out), hidden tile fp16 in
// LDS, three barriers per tile, planar fp32 output through a per-wave transposition in the (dead) input region.  Two generic
// implicit-GEMM launches (0.16 ms each at bs 64, 400 / 220 TFLOP/s) become one.
#define T2_IN 0                     // [128 px][256 ch] bf16, 512-byte rows; later 8 x 8 KB output transposition [32 ch][64 px] fp32
#define T2_H 65536                  // [128 px][256 ch] fp16
#define T2_SH (65536 + 65536)       // t1 [256] fp16 (512 bytes), then t2 [256] fp32
#define T2_LDS (T2_SH + 512 + 1024)

struct Tail2Args {
    const void *in;                 // bf16 [M][in_cs], first 256 channels
    const void *waf;                // bf16 fragments [8 waves][16 K-steps][64 lanes][8]   (layer 1, scale folded)
    const void *wbf;                // fp16 fragments [8 waves][16 K-steps][64 lanes][8]   (layer 2, scale folded, rows >= Cout zero)
    const float *t1, *t2;           // shifts [256], [256]
    float *out;                     // planar: out + img*out_img_stride + c*HW + p
    long long out_img_stride;
    int in_cs, M, HW, Cout, tiles_m;
};

// packed fp16 pair: convert, add the shift pair, LeakyReLU
__device__ __forceinline__ unsigned t2_shift_leaky_pack(float lo, float hi, unsigned shift_pair)
{
    const f32x2 v = {lo, hi};
    const f16x2 y = f16_sat(__builtin_convertvector(v, f16x2) + __builtin_bit_cast(f16x2, shift_pair));
    const f16x2 sl = {(_Float16)M3D_LEAKY_SLOPE, (_Float16)M3D_LEAKY_SLOPE};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(y, y * sl));
}

__global__ __launch_bounds__(512) void bf16_tail2_kernel(const Tail2Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[T2_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    bf16x8 wa[16];
    f16x8 wb[16];
    {
        const bf16x8 *pa = reinterpret_cast<const bf16x8 *>(a.waf) + ((size_t)wave * 16) * 64 + lane;
        const f16x8 *pb = reinterpret_cast<const f16x8 *>(a.wbf) + ((size_t)wave * 16) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 16; ++s) wa[s] = pa[s * 64];
#pragma unroll
        for (int s = 0; s < 16; ++s) wb[s] = pb[s * 64];
    }
    // shifts: layer 1's as fp16 (added to the converted pairs in the epilogue), layer 2's as fp32 (added when the transposed rows
    // are stored) -- as C operands of the chains they would cost 16 registers this kernel does not have (2 x 64 of resident weights)
    {
        _Float16 *sh1 = reinterpret_cast<_Float16 *>(lds + T2_SH);
        float *sh2w = reinterpret_cast<float *>(lds + T2_SH + 512);
        if (tid < 256) { sh1[tid] = (_Float16)a.t1[tid]; sh2w[tid] = a.t2[tid]; }
    }
    const float *sh2 = reinterpret_cast<const float *>(lds + T2_SH + 512);
    // input staging: 32 pieces of 16 bytes per row, 16 rows per pass, 8 passes
    const int c32 = tid & 31, r0 = tid >> 5;
    u32x4 vin[8];
    auto load_input = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int m = tile * 128 + p * 16 + r0;
            vin[p] = u32x4{0u, 0u, 0u, 0u};
            if (tile < a.tiles_m && m < a.M) vin[p] = *reinterpret_cast<const u32x4 *>((const __bf16 *)a.in + (size_t)m * a.in_cs + c32 * 8);
        }
    };
    load_input(blockIdx.x);
    const int lanepart = (lh ^ (l31 & 15)) << 4;
    const unsigned char *inrow = lds + T2_IN + l31 * 512;
    const unsigned char *hrow = lds + T2_H + l31 * 512;
    // byte offset of this lane's chunk of K-step s inside its row, computed AT the read (one v_xor next to a 32-cycle MFMA): hoisted
    // out of the loops the 2 x 16 offsets of the two tiles would occupy 32 of the registers the resident weights need
    auto piece = [&](int s) {
        int r;
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(lanepart), "s"((((2 * s) & 15) << 4) | (((2 * s) & 16) << 4)));
        return r;
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool active2 = 32 * wave < a.Cout;                   // (wave-uniform) this wave has output rows in layer 2

    for (int tile = blockIdx.x; tile < a.tiles_m; tile += gridDim.x) {
        const int m0 = tile * 128;
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<u32x4 *>(lds + T2_IN + h2_off(p * 16 + r0, c32, 512)) = vin[p];
        __syncthreads();                                        // A: input tile complete
        f32x16 acc[2];
        // ---- layer 1 (bf16, K = 256): two half tiles of two pixel blocks ---------------------------------------------------------
        {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                bf16x8 q[3];
#pragma unroll
                for (int i = 0; i < 2; ++i) q[i] = *reinterpret_cast<const bf16x8 *>(inrow + (2 * half + (i & 1)) * (32 * 512) + piece(i >> 1));
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int cb = i & 1, s = i >> 1;
                    if (i + 2 < 32)
                        q[(i + 2) % 3] = *reinterpret_cast<const bf16x8 *>(inrow + (2 * half + ((i + 2) & 1)) * (32 * 512) + piece((i + 2) >> 1));
                    __builtin_amdgcn_sched_barrier(0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[s], q[i % 3], s == 0 ? zero16 : acc[cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const u32x4 *shp = reinterpret_cast<const u32x4 *>(lds + T2_SH + (32 * wave + 16 * lh) * 2);
                const u32x4 s0 = shp[0], s1 = shp[1];             // the lane's 16 shifts as 8 fp16 pairs
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    u32x4 o0, o1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = t2_shift_leaky_pack(acc[cb][2 * e], acc[cb][2 * e + 1], s0[e]);
                        o1[e] = t2_shift_leaky_pack(acc[cb][8 + 2 * e], acc[cb][8 + 2 * e + 1], s1[e]);
                    }
                    const int px = half * 64 + cb * 32 + l31, c0 = 4 * wave + 2 * lh;
                    *reinterpret_cast<u32x4 *>(lds + T2_H + h2_off(px, c0, 512)) = o0;
                    *reinterpret_cast<u32x4 *>(lds + T2_H + h2_off(px, c0 + 1, 512)) = o1;
                }
            }
        }
        __syncthreads();                                        // B: hidden tile complete, input tile dead
        load_input(tile + gridDim.x);                           // (the next tile's input: in flight under layer 2 -- its 32 registers are free now)
        // ---- layer 2 (fp16, K = 256, no activation) + output -----------------------------------------------------------------------
        if (active2) {
            float *ot = reinterpret_cast<float *>(lds + T2_IN + wave * 8192);      // [32 ch][64 px] fp32, wave-private
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f16x8 q[3];
#pragma unroll
                for (int i = 0; i < 2; ++i) q[i] = *reinterpret_cast<const f16x8 *>(hrow + (2 * half + (i & 1)) * (32 * 512) + piece(i >> 1));
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const int cb = i & 1, s = i >> 1;
                    if (i + 2 < 32)
                        q[(i + 2) % 3] = *reinterpret_cast<const f16x8 *>(hrow + (2 * half + ((i + 2) & 1)) * (32 * 512) + piece((i + 2) >> 1));
                    __builtin_amdgcn_sched_barrier(0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wb[s], q[i % 3], s == 0 ? zero16 : acc[cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (half) __builtin_amdgcn_wave_barrier();          // (the wave's reads of the previous half's transposition are done: LDS is in order per wave)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int j = 0; j < 16; ++j) ot[(16 * lh + j) * 64 + cb * 32 + l31] = acc[cb][j];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                const int mq = m0 + 64 * half;
                if (a.HW % 64 == 0 && mq + 64 <= a.M) {
                    // buffer stores: base = the image's planes, voffset = the lane's (channel row, 4-pixel piece) -- the same for every
                    // tile -- soffset = the half tile's first pixel + 4 channel rows per step; rows past Cout fall outside the
                    // resource's range and are dropped by the hardware
                    const int img = mq / a.HW, p0 = mq - img * a.HW;
                    const __amdgpu_buffer_rsrc_t ro = make_rsrc(a.out + (size_t)img * a.out_img_stride, (unsigned)a.Cout * a.HW * 4);
                    const unsigned vo = (unsigned)(((32 * wave + (lane >> 4)) * a.HW + (lane & 15) * 4) * 4);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {                 // 32 channels x 16 pieces of 4 pixels = 8 x 64 lanes
                        const int c = (lane >> 4) + 4 * kk;
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(ot + c * 64 + (lane & 15) * 4) + sh2[32 * wave + c];
                        buf_store_f32x4_nop(v, ro, vo, (unsigned)((p0 + kk * 4 * a.HW) * 4));
                    }
                } else {
                    for (int i = lane; i < 32 * 64; i += 64) {
                        const int c = i >> 6, m = mq + (i & 63), ch = 32 * wave + c;
                        if (m < a.M && ch < a.Cout) {
                            const int img = m / a.HW, pp = m - img * a.HW;
                            a.out[(size_t)img * a.out_img_stride + (size_t)ch * a.HW + pp] = ot[i] + sh2[ch];
                        }
                    }
                }
            }
        }
        __syncthreads();                                        // E: input region (transposition) and hidden tile are free
    }
}

extern "C" int m3d_head_tail2_bf16_forward(const m3d_tail2_bf16_desc *d, m3d_stream_t stream)
{
    M3D_REQUIRE(d && d->in && d->waf && d->wbf && d->out && d->t1 && d->t2, "head_tail2_bf16: null pointer");
    M3D_REQUIRE(d->in_cs % 8 == 0 && d->in_cs >= 256 && ((uintptr_t)d->in & 15) == 0, "head_tail2_bf16: 256 input channels, 16-byte aligned rows");
    M3D_REQUIRE(d->Cout >= 1 && d->Cout <= 256 && d->M >= 1 && d->HW >= 1, "head_tail2_bf16: bad sizes (Cout <= 256)");
    M3D_REQUIRE((((uintptr_t)d->waf | (uintptr_t)d->wbf) & 15) == 0, "head_tail2_bf16: 16-byte aligned weights");
    Tail2Args a;
    a.in = d->in; a.waf = d->waf; a.wbf = d->wbf; a.t1 = d->t1; a.t2 = d->t2; a.out = d->out; a.out_img_stride = d->out_img_stride;
    a.in_cs = d->in_cs; a.M = (int)d->M; a.HW = d->HW; a.Cout = d->Cout; a.tiles_m = cdiv(d->M, 128);
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    }
    static int scratch = -1;
    if (scratch < 0) {
        hipFuncAttributes fa;
        M3D_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&bf16_tail2_kernel)));
        scratch = (int)fa.localSizeBytes;
    }
    M3D_REQUIRE(scratch == 0, "head_tail2_bf16: the kernel was built with register spills (%d bytes of scratch)", scratch);
    hipLaunchKernelGGL(bf16_tail2_kernel, dim3(std::max(1, std::min(a.tiles_m, ncu))), dim3(512), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The four bias-free 1x1 projections of ANAB (model/module/attention.py:169-173, 183-200: query 128 -> 168, key 128 -> 168, value
// 128 -> 128, spatial gates 128 -> 4 + sigmoid) as ONE launch over the shared input: the stacked weight matrix [q | k | v | s] is
// padded to 512 rows, wave w owns row blocks w and w + 8 (2 x 8 fragments = 64 registers, resident), a tile of 128 pixels runs as
// two halves, and the epilogue writes every 8-row piece where it belongs: q bf16 [px][q_cs] (rows [168, 192) come out as exact zeros:
// the logits GEMM reads them as K padding), k|v bf16 [px][kv_cs], gates fp32 [px][s_cs] through a sigmoid.  Three implicit-GEMM
// launches (0.31 ms at bs 64, each re-reading the 126 MB input at 11-210 TFLOP/s) become one that is bound by its 0.6 GB of stores.
#define QK_IN 0                     // [128 px][128 ch] bf16, 256-byte rows, swizzled
#define QK_LDS 32768

struct QkvsArgs {
    const void *in;                 // bf16 [M][in_cs], first 128 channels
    const void *wf;                 // bf16 fragments [16 row blocks][8 K-steps][64 lanes][8]
    void *q, *kv;                   // bf16
    float *s;                       // fp32
    int in_cs, q_cs, kv_cs, s_cs, M, tiles_m;
    int q_rows, kv_rows, s_rows;    // q_rows = padded query rows (multiple of 8), kv_rows multiple of 8, s_rows <= 8
};

__global__ __launch_bounds__(512) void bf16_qkvs_kernel(const QkvsArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[QK_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    bf16x8 w[2][8];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const bf16x8 *p = reinterpret_cast<const bf16x8 *>(a.wf) + ((size_t)(wave + 8 * b) * 8) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 8; ++s) w[b][s] = p[s * 64];
    }
    const int c16 = tid & 15, r0 = tid >> 4;
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
    const int lanepart = (lh ^ (l31 & 15)) << 4;
    const unsigned char *inrow = lds + QK_IN + l31 * 256;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int kv0 = a.q_rows, s0 = a.q_rows + a.kv_rows;

    for (int tile = blockIdx.x; tile < a.tiles_m; tile += gridDim.x) {
        const int m0 = tile * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4 *>(lds + QK_IN + h2_off(p * 32 + r0, c16, 256)) = vin[p];
        load_input(tile + gridDim.x);
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 acc[2][2];                                   // [row block][pixel block]
            bf16x8 qf[4];                                       // B fragments: step i = (K-step, pixel block), ring of 4, 3 ahead
#pragma unroll
            for (int i = 0; i < 3; ++i) qf[i] = *reinterpret_cast<const bf16x8 *>(inrow + (2 * half + (i & 1)) * (32 * 256) + (lanepart ^ ((2 * (i >> 1)) << 4)));
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cb = i & 1, s = i >> 1;
                if (i + 3 < 16)
                    qf[(i + 3) & 3] = *reinterpret_cast<const bf16x8 *>(inrow + (2 * half + ((i + 3) & 1)) * (32 * 256) + (lanepart ^ ((2 * ((i + 3) >> 1)) << 4)));
                __builtin_amdgcn_sched_barrier(0);
                acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0][s], qf[i & 3], s == 0 ? zero16 : acc[0][cb], 0, 0, 0);
                acc[1][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[1][s], qf[i & 3], s == 0 ? zero16 : acc[1][cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the lane holds rows 32 * blk + 16 * lh + (0..15) of pixel m0 + 64 * half + 32 * cb + l31: two pieces of 8 rows
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const int m = m0 + 64 * half + 32 * cb + l31;
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const int row = 32 * (wave + 8 * b) + 16 * lh + 8 * pc;      // first row of the piece
                        if (m < a.M) {
                            if (row < s0) {
                                u32x4 o;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const f32x2 v = {acc[b][cb][8 * pc + 2 * e], acc[b][cb][8 * pc + 2 * e + 1]};
                                    o[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                                }
                                __bf16 *dst = row < kv0 ? (__bf16 *)a.q + (size_t)m * a.q_cs + row : (__bf16 *)a.kv + (size_t)m * a.kv_cs + (row - kv0);
                                global_store_u32x4_nop(dst, o);
                            } else if (row == s0) {
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    if (e < a.s_rows) a.s[(size_t)m * a.s_cs + e] = sigmoidf_(acc[b][cb][8 * pc + e]);
                            }
                        }
                    }
                }
        }
        __syncthreads();                                        // the input tile is free
    }
}

extern "C" int m3d_anab_qkvs_bf16_forward(const m3d_qkvs_bf16_desc *d, m3d_stream_t stream)
{
    M3D_REQUIRE(d && d->in && d->wf && d->q && d->kv && d->s, "anab_qkvs_bf16: null pointer");
    M3D_REQUIRE(d->in_cs % 8 == 0 && d->in_cs >= 128 && ((uintptr_t)d->in & 15) == 0, "anab_qkvs_bf16: 128 input channels, 16-byte aligned rows");
    M3D_REQUIRE(d->q_rows % 8 == 0 && d->kv_rows % 8 == 0 && d->s_rows >= 1 && d->s_rows <= 8 && d->q_rows + d->kv_rows + 8 <= 512,
                "anab_qkvs_bf16: q_rows / kv_rows multiples of 8, s_rows <= 8, at most 512 rows in all");
    M3D_REQUIRE(d->q_cs % 8 == 0 && d->q_cs >= d->q_rows && d->kv_cs % 8 == 0 && d->kv_cs >= d->kv_rows && d->s_cs >= d->s_rows &&
                (((uintptr_t)d->q | (uintptr_t)d->kv | (uintptr_t)d->wf) & 15) == 0, "anab_qkvs_bf16: output strides / alignment");
    M3D_REQUIRE(d->M >= 1, "anab_qkvs_bf16: empty input");
    QkvsArgs a;
    a.in = d->in; a.wf = d->wf; a.q = d->q; a.kv = d->kv; a.s = d->s;
    a.in_cs = d->in_cs; a.q_cs = d->q_cs; a.kv_cs = d->kv_cs; a.s_cs = d->s_cs; a.M = (int)d->M; a.tiles_m = cdiv(d->M, 128);
    a.q_rows = d->q_rows; a.kv_rows = d->kv_rows; a.s_rows = d->s_rows;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    }
    hipLaunchKernelGGL(bf16_qkvs_kernel, dim3(std::max(1, std::min(a.tiles_m, ncu))), dim3(512), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_head_mlp2_bf16_forward(const m3d_head2_bf16_desc *d, m3d_stream_t stream)
{
    M3D_REQUIRE(d && d->in && d->w1f && d->w2f && d->w3 && d->out && d->t1 && d->t2 && d->t3, "head_mlp2_bf16: null pointer");
    M3D_REQUIRE(d->in_cs % 8 == 0 && d->in_cs >= 128 && ((uintptr_t)d->in & 15) == 0, "head_mlp2_bf16: 128 input channels, 16-byte aligned rows");
    M3D_REQUIRE(d->Cout >= 1 && d->Cout <= 64, "head_mlp2_bf16: Cout <= 64");
    M3D_REQUIRE(d->groups >= 1 && d->M >= 1 && d->HW >= 1, "head_mlp2_bf16: bad sizes");
    M3D_REQUIRE((((uintptr_t)d->w1f | (uintptr_t)d->w2f | (uintptr_t)d->w3) & 15) == 0, "head_mlp2_bf16: 16-byte aligned weights");
    Head2Args a;
    a.in = d->in; a.w1f = d->w1f; a.w2f = d->w2f; a.w3 = d->w3; a.t1 = d->t1; a.t2 = d->t2; a.t3 = d->t3;
    a.out = d->out; a.out_goff = d->out_group_off; a.out_img_stride = d->out_img_stride;
    a.in_cs = d->in_cs; a.M = (int)d->M; a.HW = d->HW; a.Cout = d->Cout; a.tiles_m = cdiv(d->M, 128);
#ifdef BF16_TRACE
    a.trace = g_head2_trace;
#endif
    // one workgroup per CU (130 KB of LDS, 512 threads): the CUs are split between the heads of the launch, every workgroup walks tiles
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    }
    static int scratch = -1;          // resident weights in registers: a spilled build would re-read them from scratch memory per tile
    if (scratch < 0) {
        hipFuncAttributes fa;
        M3D_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&bf16_head2_kernel)));
        scratch = (int)fa.localSizeBytes;
    }
    M3D_REQUIRE(scratch == 0, "head_mlp2_bf16: the kernel was built with register spills (%d bytes of scratch)", scratch);
    const int nb = std::max(1, std::min(a.tiles_m, ncu / d->groups));
    hipLaunchKernelGGL(bf16_head2_kernel, dim3(nb, d->groups), dim3(512), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
