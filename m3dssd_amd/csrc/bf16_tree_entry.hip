// Entry of a DLA tree in ONE launch (bf16 path; model/pose_dla_dcn.py:314-327 Tree.forward, :107-121 BasicBlock.conv1):
//     bottom   = MaxPool2d(2, 2)(x)                                         (Tree.downsample)
//     residual = BatchNorm(Conv1x1(bottom))                                 (Tree.project)
//     t        = LeakyReLU(BatchNorm(Conv3x3 stride 2 pad 1 (x)))           (tree1.conv1 of the first block)
// Unfused these are three launches that each read x from HBM (max-pool 5.4 TB/s, the 1x1 at 2.2x and the stride-2 3x3 on the
// implicit-GEMM tile at 2x their byte floors: 1.0 ms of the 11.3 ms step at bs 64 over the four tree levels).  The three share
// their input: the 2x2 pooling window of an output pixel is taps (1,1) (1,2) (2,1) (2,2) of its 3x3 stride-2 window.
//
//   Workgroup = 256 threads = an 8 x 16 tile of output pixels x CB = 128 (or 64) output channels.  Per 32-channel chunk of Cin the
//   17 x 33 input halo tile is staged ONCE in LDS as fp16 (bf16 -> fp16 is exact; 80-byte pixel records: the stride-2 reads of a
//   16x16x32 B fragment are bank-conflict free, as in the front end's level1), and every wave walks the 9 taps: one ds_read_b128 per
//   (tap, pixel row) feeds the MFMAs of the wave's two 16-channel row blocks; the four pooling taps are max-reduced on the packed
//   fp16 pipe on the way (v_pk_max_f16) and the result is the B operand of a tenth "tap" with the 1x1 project weights -- and is
//   what the workgroup of channel block 0 writes as `bottom`.  Weights: fp16, BatchNorm scale folded, in fragment order
//   [Cout/32][Cin/32][10][2 blocks][64 lanes][8] (m3dssd_amd/engine_bf16.py: pack_tree_entry), streamed global -> register two
//   taps ahead; the shifts are the C operands of the chains.  MFMA rows are mapped to channels so that a lane ends up with 8
//   consecutive channels of its pixel: one 16-byte store per pixel row and output.
//
// Measured (bs 64, tools/tree_entry_trace.py): level2 0.25 ms (was 0.45 for the three launches), level3 0.17 (0.26), level4 0.12 (0.17),
// level5 0.135 (0.136).  Tried and not kept (round 5): all 20 weight fragments of a chunk up-front on 64-channel blocks (the MFMA phase
// of a tile drops from 5 700 to 2 600 cycles, but two workgroups per CU instead of three and twice the channel blocks: 0.29 / 0.23 /
// 0.16 / 0.19 ms), and persistent workgroups with the next tile's loads in flight under the MFMAs on top of that (0.25 / 0.21 / 0.16
// / 0.23 ms): the per-tile VALU work -- piece addresses, bf16 -> fp16, store addresses: ~600 instructions per wave against 80-160
// MFMAs -- is what bounds the kernel, not the exposed latencies.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

// compile-time loop: `#pragma unroll` on a 72-step body is silently left rolled (the accumulator arrays then live in scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void te_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        te_static_for<I + 1, N>(f);
    }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

#define TE_TH 8
#define TE_TW 16
#define TE_HH (2 * TE_TH + 1)        // 17 halo rows
#define TE_HW (2 * TE_TW + 1)        // 33 halo columns
#define TE_PS 80                     // bytes per halo pixel record (32 fp16 + 16 pad)
#define TE_NPX (TE_HH * TE_HW)       // 561
#define TE_NPIECE (TE_NPX * 4)       // 16-byte pieces of a chunk tile
#define TE_NIT ((TE_NPIECE + 255) / 256)   // 9 staging iterations per thread

struct TreeEntryArgs {
    const void *in;                  // bf16 [N][H][W][in_cs]
    const void *wfrag;               // fp16 [Cout/32][Cin/32][10 taps][2 blocks][64 lanes][8]
    const float *shift1, *shiftp;    // [Cout]
    void *t, *res, *bottom;          // bf16 [N][H/2][W/2][*_cs]; bottom may be null
    int in_cs, t_cs, res_cs, bottom_cs;
    int N, H, W, Cin, Cout, Ho, Wo, tiles_x, tiles_y, nchunks, cblocks;
};

__device__ __forceinline__ unsigned te_bf16pair_to_f16(unsigned d)
{
    const f32x2 v = {__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ unsigned te_f16pair_to_bf16(unsigned d)
{
    const f16x2 h = __builtin_bit_cast(f16x2, d);
    const f32x2 v = {(float)h[0], (float)h[1]};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned te_pack_bf16(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ f16x8 te_max(f16x8 a, f16x8 b) { return __builtin_elementwise_max(a, b); }

// CW = waves along the channel dimension (4: CB = 128 channels per workgroup, a wave owns all 8 pixel rows; 2: CB = 64, a wave owns
// 4 pixel rows); NCB = pixel rows (column blocks of 16 pixels) per wave
template <int CW, int OCC>           // OCC = workgroups per CU: 2 for CB = 128 (256 registers), 3 for CB = 64 (164 registers, 45 KB of LDS each)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void bf16_tree_entry_kernel(const TreeEntryArgs a)
{
    constexpr int NCB = CW == 4 ? 8 : 4;
    __shared__ __attribute__((aligned(16))) unsigned char tile[TE_NPX * TE_PS + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    // workgroup -> (tile, channel block); XCD-contiguous tile order (workgroup L runs on XCD L % 8)
    int L = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = L & 7, loc = L >> 3;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int cblk = L % a.cblocks;
    int tl = L / a.cblocks;
    const int per_img = a.tiles_x * a.tiles_y;
    const int n = tl / per_img;
    tl -= n * per_img;
    const int ty = tl / a.tiles_x, tx = tl - ty * a.tiles_x;
    const int oy0 = ty * TE_TH, ox0 = tx * TE_TW;
    const int Y0 = 2 * oy0 - 1, X0 = 2 * ox0 - 1;                  // halo origin
    const int wc = CW == 4 ? wave : (wave & 1);                     // channel slice of the workgroup's block
    const int cb0 = CW == 4 ? 0 : (wave >> 1) * 4;                  // first pixel row of this wave
    const int ws = cblk * CW + wc;                                  // global 32-channel slice
    const int ch0 = ws * 32 + 8 * kg;                               // the lane's 8 output channels: ch0 .. ch0 + 7

    const __amdgpu_buffer_rsrc_t rin = make_rsrc((const __bf16 *)a.in + (size_t)n * a.H * a.W * a.in_cs, (unsigned)a.H * a.W * a.in_cs * 2);
    const f16x8 *wbase = reinterpret_cast<const f16x8 *>(a.wfrag) + ((size_t)ws * a.nchunks) * (20 * 64) + lane;

    // accumulators: [output][block][pixel row]; D rows of block b -> channels ch0 + 4 b + (0..3)
    f32x4 acc1[2][NCB], acc2[2][NCB];
    {
        const f32x4 s10 = *reinterpret_cast<const f32x4 *>(a.shift1 + ch0), s11 = *reinterpret_cast<const f32x4 *>(a.shift1 + ch0 + 4);
        const f32x4 sp0 = *reinterpret_cast<const f32x4 *>(a.shiftp + ch0), sp1 = *reinterpret_cast<const f32x4 *>(a.shiftp + ch0 + 4);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) { acc1[0][cb] = s10; acc1[1][cb] = s11; acc2[0][cb] = sp0; acc2[1][cb] = sp1; }
    }
    const unsigned char *brow = tile + (2 * l15) * TE_PS + kg * 16;  // B fragment base of this lane: pixel 2 * l15 of halo row 0, its k-group

    for (int c = 0; c < a.nchunks; ++c) {
        // ---- weights of the first two taps of this chunk (in flight under the staging) ------------------------------------------
        const f16x8 *wp = wbase + (size_t)c * (20 * 64);
        f16x8 wa[3][2];                                             // ring of 3 taps x 2 blocks
#pragma unroll
        for (int t = 0; t < 2; ++t) { wa[t][0] = wp[(2 * t) * 64]; wa[t][1] = wp[(2 * t + 1) * 64]; }
        // ---- stage the halo tile of channels [32c, 32c + 32): bf16 -> fp16, zeros outside the image ------------------------------
        if (c > 0) __syncthreads();                                 // every wave is done with the previous chunk's tile
        {
            u32x4 v[TE_NIT];
#pragma unroll
            for (int it = 0; it < TE_NIT; ++it) {
                const int id = tid + it * 256;
                const int hp = id >> 2, pc = id & 3;
                const int r = hp / TE_HW, col = hp - r * TE_HW;
                const int y = Y0 + r, x = X0 + col;
                const bool ok = id < TE_NPIECE && y >= 0 && y < a.H && x >= 0 && x < a.W;
                const unsigned vo = ok ? (unsigned)(((y * a.W + x) * a.in_cs + c * 32 + pc * 8) * 2) : M3D_BUF_OOB;
                v[it] = __builtin_amdgcn_raw_buffer_load_b128(rin, vo, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < TE_NIT; ++it) {
                const int id = tid + it * 256;
                const int hp = id >> 2, pc = id & 3;
                u32x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = te_bf16pair_to_f16(v[it][e]);
                if (id < TE_NPIECE) *reinterpret_cast<u32x4 *>(tile + hp * TE_PS + pc * 16) = h;
            }
        }
        __syncthreads();
        // ---- 9 taps of the stride-2 3x3 (+ the pooling reduce), then the 1x1 on the pooled pixels --------------------------------
        // linear walk over (tap, pixel row): the B fragment of step st + 3 is requested before the MFMAs of step st (ring of 4)
        f16x8 pm[NCB];
        constexpr int NST = 9 * NCB;
        f16x8 q[4];
        auto bfrag = [&](int st) {
            const int tap = st / NCB, cb = st - tap * NCB;
            const int ti = tap / 3, tj = tap - 3 * ti;
            return *reinterpret_cast<const f16x8 *>(brow + ((2 * (cb0 + cb) + ti) * TE_HW + tj) * TE_PS);
        };
#pragma unroll
        for (int st = 0; st < 3; ++st) q[st] = bfrag(st);
        te_static_for<0, NST>([&](auto stc) {
            constexpr int st = decltype(stc)::value;
            constexpr int tap = st / NCB, cb = st - tap * NCB;
            if (cb == 0 && tap + 2 < 10) { wa[(tap + 2) % 3][0] = wp[(2 * (tap + 2)) * 64]; wa[(tap + 2) % 3][1] = wp[(2 * (tap + 2) + 1) * 64]; }
            if (st + 3 < NST) q[(st + 3) & 3] = bfrag(st + 3);
            __builtin_amdgcn_sched_barrier(0);
            acc1[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[tap % 3][0], q[st & 3], acc1[0][cb], 0, 0, 0);
            acc1[1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[tap % 3][1], q[st & 3], acc1[1][cb], 0, 0, 0);
            if (tap == 4) pm[cb] = q[st & 3];
            else if (tap == 5 || tap == 7 || tap == 8) pm[cb] = te_max(pm[cb], q[st & 3]);
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            acc2[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[9 % 3][0], pm[cb], acc2[0][cb], 0, 0, 0);
            acc2[1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[9 % 3][1], pm[cb], acc2[1][cb], 0, 0, 0);
        }
        // ---- bottom = the pooled pixels of this chunk (channel block 0 only; the waves that share a pixel row take turns by chunk) ----
        if (a.bottom && cblk == 0 && wc == (c % CW)) {
            const __amdgpu_buffer_rsrc_t rb = make_rsrc((__bf16 *)a.bottom + (size_t)n * a.Ho * a.Wo * a.bottom_cs, (unsigned)a.Ho * a.Wo * a.bottom_cs * 2);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int oy = oy0 + cb0 + cb, ox = ox0 + l15;
                const u32x4 p = __builtin_bit_cast(u32x4, pm[cb]);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = te_f16pair_to_bf16(p[e]);
                const unsigned vo = (oy < a.Ho && ox < a.Wo) ? (unsigned)(((oy * a.Wo + ox) * a.bottom_cs + c * 32 + kg * 8) * 2) : M3D_BUF_OOB;
                buf_store_f32x4_nop(__builtin_bit_cast(f32x4, o), rb, vo, 0);
            }
        }
    }

    // ---- epilogue: t = LeakyReLU(acc1), res = acc2, 8 consecutive channels per lane and pixel ---------------------------------------
    const __amdgpu_buffer_rsrc_t rt = make_rsrc((__bf16 *)a.t + (size_t)n * a.Ho * a.Wo * a.t_cs, (unsigned)a.Ho * a.Wo * a.t_cs * 2);
    const __amdgpu_buffer_rsrc_t rr = make_rsrc((__bf16 *)a.res + (size_t)n * a.Ho * a.Wo * a.res_cs, (unsigned)a.Ho * a.Wo * a.res_cs * 2);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int oy = oy0 + cb0 + cb, ox = ox0 + l15;
        const bool ok = oy < a.Ho && ox < a.Wo;
        const unsigned pix = (unsigned)(oy * a.Wo + ox);
        u32x4 o;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f32x4 y = acc1[blk][cb];
            const f32x4 z = y * M3D_LEAKY_SLOPE;
            o[2 * blk] = te_pack_bf16(fmaxf(y[0], z[0]), fmaxf(y[1], z[1]));
            o[2 * blk + 1] = te_pack_bf16(fmaxf(y[2], z[2]), fmaxf(y[3], z[3]));
        }
        buf_store_f32x4_nop(__builtin_bit_cast(f32x4, o), rt, ok ? (pix * (unsigned)a.t_cs + (unsigned)ch0) * 2u : M3D_BUF_OOB, 0);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const f32x4 y = acc2[blk][cb];
            o[2 * blk] = te_pack_bf16(y[0], y[1]);
            o[2 * blk + 1] = te_pack_bf16(y[2], y[3]);
        }
        buf_store_f32x4_nop(__builtin_bit_cast(f32x4, o), rr, ok ? (pix * (unsigned)a.res_cs + (unsigned)ch0) * 2u : M3D_BUF_OOB, 0);
    }
}

extern "C" int m3d_tree_entry_bf16_applicable(const m3d_tree_entry_bf16_desc *d)
{
    if (!d) return 0;
    if (d->Cin % 32 != 0 || d->Cout % 64 != 0 || d->H % 2 != 0 || d->W % 2 != 0) return 0;
    if (d->in_cs % 8 != 0 || d->t_cs % 8 != 0 || d->res_cs % 8 != 0 || (d->bottom && d->bottom_cs % 8 != 0)) return 0;
    if ((long long)d->H * d->W * d->in_cs * 2 >= (1ll << 31)) return 0;
    return 1;
}

extern "C" int m3d_tree_entry_bf16_forward(const m3d_tree_entry_bf16_desc *d, m3d_stream_t stream)
{
    M3D_REQUIRE(d && d->in && d->wfrag && d->shift1 && d->shiftp && d->t && d->res, "tree_entry_bf16: null pointer");
    M3D_REQUIRE(m3d_tree_entry_bf16_applicable(d), "tree_entry_bf16: needs Cin %% 32 == 0, Cout %% 64 == 0, even H / W, pixel strides %% 8 == 0");
    M3D_REQUIRE((((uintptr_t)d->in | (uintptr_t)d->t | (uintptr_t)d->res | (uintptr_t)d->bottom | (uintptr_t)d->wfrag) & 15) == 0 &&
                (((uintptr_t)d->shift1 | (uintptr_t)d->shiftp) & 15) == 0, "tree_entry_bf16: 16-byte aligned views");
    TreeEntryArgs a;
    a.in = d->in; a.wfrag = d->wfrag; a.shift1 = d->shift1; a.shiftp = d->shiftp; a.t = d->t; a.res = d->res; a.bottom = d->bottom;
    a.in_cs = d->in_cs; a.t_cs = d->t_cs; a.res_cs = d->res_cs; a.bottom_cs = d->bottom_cs;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Ho = d->H / 2; a.Wo = d->W / 2;
    a.tiles_x = cdiv(a.Wo, TE_TW); a.tiles_y = cdiv(a.Ho, TE_TH); a.nchunks = d->Cin / 32;
    static const int cb64 = []() { const char *e = getenv("M3D_TE_CB64"); return e ? atoi(e) : 0; }();   // experiments: 64-channel blocks everywhere
    const bool wide = d->Cout % 128 == 0 && !cb64;
    a.cblocks = d->Cout / (wide ? 128 : 64);
    const long long grid = (long long)a.tiles_x * a.tiles_y * d->N * a.cblocks;
    M3D_REQUIRE(grid < (1ll << 31), "tree_entry_bf16: too many workgroups");
    if (wide) hipLaunchKernelGGL((bf16_tree_entry_kernel<4, 2>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((bf16_tree_entry_kernel<2, 3>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
