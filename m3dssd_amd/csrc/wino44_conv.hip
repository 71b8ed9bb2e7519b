// Winograd F(4x4,3x3) convolution (3x3, stride 1, pad 1, fp32) on v_mfma_f32_16x16x4_f32.
//
// Why: the F(2x2,3x3) wave kernel (wino_conv.hip) executes 16 multiplies per 4 outputs (direct: 36); F(4x4,3x3) executes 36
// per 16 outputs -- 1.78x fewer MFMA FLOPs again.  Its 36 transform positions do not fit the one-wave 32 x 32-tile design (36 x
// 16 accumulator registers), so the tile is 16 output tiles x 32 output channels per wave on 16x16x4 blocks (36 x 2 x 4 = 288
// accumulator registers of the wave's 512), and the price of the smaller MFMA block -- twice the operand traffic per FLOP --
// is paid by sharing the input side across the workgroup:
//   * workgroup = 4 waves = 16 tiles (16 x 1 strip of 4x4-pixel output tiles) x 128 output channels; a stage = 16 input
//     channels; thread (tile, channel) loads its 6x6 patch (36 dword loads, padding positions read 0 through the buffer range
//     check), runs B^T d B (144 scalar FMA / add: ONE input channel per thread, so the transform costs 1/4 of what it would
//     cost per wave) and writes the 36 values to LDS, V[xi][tile][16 channels];
//   * per transform position xi a wave reads its A operand with ONE ds_read_b128 (lane = (tile, channel quad): the four
//     floats are the four k-steps of the stage) and runs 8 MFMAs (4 k-steps x 2 channel blocks of 16) against B fragments that
//     stream global -> register in fragment order, four positions ahead (hand-counted s_waitcnt, as in wino_conv.hip);
//   * V is double buffered: the transform of stage s + 1 is written while other waves may still read stage s -- one barrier
//     per stage (288 MFMAs per wave);
//   * A^T M A (36 -> 16 values per (tile, channel)), folded BatchNorm / bias, residual and LeakyReLU run in registers.
// fp32 error: the F(4x4,3x3) transforms amplify rounding ~7x relative to F(2x2,3x3) / direct summation (rms 1.5e-6 vs 2.2e-7
// of the output scale at Cin = 128, max 1.5e-5; tools/ and DESIGN.md); U = G g G^T is computed in fp64 and rounded once.
#include <stdlib.h>

#include <mutex>

#include <type_traits>

#include "common.h"

struct Wino44Args {
    const float *in;
    const float *U;          // [Cout_pad/32][Cin/16][36][2][64 lanes][4]
    float *out;
    const float *scale, *shift, *res;
    int in_cs, out_cs, res_cs;
    unsigned in_bytes, out_bytes, res_bytes;
    int N, H, W, Cin, Cout;
    int TH, TW, NT;          // 4x4 output tiles per image (rows, cols), total
    int act, res_mode;
    int nsl;                 // stages (16 input channels each) per K slice: Cin / 16 / gridDim.z
    long long ws_slice;      // split-K: floats between the raw partial outputs of consecutive K slices (a.out = slice 0), else 0
    const char *touch;       // optional: tensor whose 128-byte lines the launch pulls into the memory-side cache (the NEXT
    int touch_lines;         // F(4x4) layer's U), touch_lines of them
#ifdef WINO_TRACE
    long long *trace;
#endif
};

#define W44_LA 4             // transform positions the B fragments run ahead (ring of 4 register sets)

#ifdef WINO_TRACE
// diagnostic build (make trace): s_memtime stamps of lane 0 of waves 0 and 3 of every workgroup, 64 slots each (tools/wino44_trace.py)
static long long *g_w44_trace = nullptr;
extern "C" void m3d_wino44_set_trace(void *p) { g_w44_trace = (long long *)p; }
#define W44_TRACE_INIT() long long *trp = (a.trace && (wave == 0 || wave == 3)) ? a.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wave ? 1 : 0)) * 64 : nullptr; int tri = 0
#define W44_TRACE() do { if (trp && lane == 0 && tri < 64) trp[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define W44_TRACE_INIT()
#define W44_TRACE()
#endif

__device__ __forceinline__ void w44_load_dword(float &dst, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ void w44_load_x4(f32x4 &dst, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
template <int N, int NB>
__device__ __forceinline__ void w44_wait(f32x4 (&u)[2])
{
    // the registers are operands so that no use of them can be scheduled above the wait.  One operand per register set: naming
    // the same variable twice makes the compiler COPY it for the second operand -- a copy of a register whose load is still in
    // flight, written back over the landed value afterwards (this is what the single-block variant first did).
    if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(u[0]), "+v"(u[1]) : "n"(N));
    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(u[0]) : "n"(N));
}

template <int I, int N, typename F>
__device__ __forceinline__ void w44_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        w44_static_for<I + 1, N>(f);
    }
}

// Position of channel c (0..15) inside tile t's 16-float row of V[xi][tile][channel]: the 16-byte quad index is XORed with
// 2 * (t >> 3).  The A operand is read with ds_read_b128 by lane = (tile = lane & 15, quad = lane >> 4); the LDS serves such a read in
// four groups of 16 lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md), each mixing tiles {0-3, 12-15} at one quad with tiles 4-11
// at the next -- in the plain layout (64-byte rows) tiles t and t + 12 / t + 4 of a group share their banks: 2-way conflicts on every
// A read (PMC, r04c: 2.7 M conflict cycles of 6.8 M LDS-array cycles per launch).  With the XOR every group touches 16 distinct
// 16-byte bank groups (checked exhaustively); the transform's 4-byte writes stay a permutation of 64 consecutive floats per wave.
__device__ __forceinline__ int w44_vswz(int t, int c) { return c ^ ((t >> 3) << 3); }

// y = B^T x for F(4x4,3x3): B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void w44_bt(const float x0, const float x1, const float x2, const float x3, const float x4, const float x5,
                                       float &y0, float &y1, float &y2, float &y3, float &y4, float &y5)
{
    y0 = fmaf(4.f, x0, fmaf(-5.f, x2, x4));
    const float a = fmaf(-4.f, x2, x4), b = fmaf(-4.f, x1, x3);
    y1 = a + b;
    y2 = a - b;
    const float c = x4 - x2, e = x3 - x1;
    y3 = fmaf(2.f, e, c);
    y4 = fmaf(-2.f, e, c);
    y5 = fmaf(4.f, x1, fmaf(-5.f, x3, x5));
}
// y = A^T m: A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]; on f32x4 = the four tiles of a lane at once (four
// independent chains per instruction: the epilogue runs with one wave per SIMD and nothing else to hide a dependent add behind)
template <typename VT>
__device__ __forceinline__ void w44_at(const VT m0, const VT m1, const VT m2, const VT m3, const VT m4, const VT m5,
                                       VT &y0, VT &y1, VT &y2, VT &y3)
{
    const VT s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = d2 * 2.f + d1;
    y2 = s2 * 4.f + s1;
    y3 = d2 * 8.f + d1 + m5;
}

// NB = 16-channel blocks per wave: 2 (workgroup = 128 output channels) or 1 (64 output channels: Cout = 64 layers, and layers
// whose 16-tile strips x 128-channel blocks would leave CUs idle -- 256 -> 256 @ 24x80, bs 8: 120 workgroups become 240).
// OCC = waves per SIMD the build is sized for: 1 = the whole 512-register file (NB = 2 needs it: 288 accumulators); 2 (NB = 1 only:
// 144 accumulators, <= 256 registers) = TWO workgroups per CU (2 x 73.8 KB of LDS).  Why two: tools/ubench/mfma16_fillers.hip -- a
// VALU instruction between the v_mfma_f32_16x16x4_f32 of ONE wave costs the stream 12.6 cycles (6-7 each in runs of 4-6), whatever
// the instruction (v_fma / v_add / packed / v_accvgpr_mov alike); with a second wave on the SIMD the same instruction costs 1.4-4
// cycles, because the partner's MFMAs issue into the gap -- and the prologue / epilogue of one workgroup run under the other's MFMAs.
// KS = 2 (NB = 1, OCC = 2): the "K-pair" workgroup for layers whose 16-tile strips x 64-channel blocks leave half the CU slots
// empty (256 -> 256 @ 24x80 at bs 8: 240 workgroups for 512 slots -- one wave per SIMD, every transform instruction at the
// 12.6-cycle price): 512 threads, waves 0-3 run the first half of the slice's input channels, waves 4-7 the second half, each
// half with its own V double buffer (2 x 73.7 KB: what two co-resident workgroups would hold) -- two waves per SIMD again without
// a global workspace; at the end the halves trade half of their accumulators through the (free) V buffers (half 0 keeps tiles
// {0, 1} of every lane's four, half 1 tiles {2, 3}), add, and EACH runs the output transform / epilogue of its 8 tiles.
template <int NB, int OCC, int KS = 1>
__global__ __launch_bounds__(256 * KS) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void wino44_kernel(const Wino44Args a)
{
    static_assert(KS == 1 || (KS == 2 && NB == 1 && OCC == 2), "K-pair form: 64-channel waves, two per SIMD");
    __shared__ __attribute__((aligned(16))) float Vs_all[KS * 2 * 36 * 256];   // V[K half][buf][xi][tile 16][channel 16]
    __shared__ int pixb[16];                                             // first output pixel of each tile (-1: no such tile)
    const int tid8 = threadIdx.x, lane = tid8 & 63;
    const int kh = KS == 2 ? __builtin_amdgcn_readfirstlane(tid8 >> 8) : 0;      // K half of this wave
    const int tid = tid8 & 255;                                          // thread within its half: transform unit, wave = channel block
    float *const Vs = Vs_all + kh * (2 * 36 * 256);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // split-K (grid.z slices of the input channels, small maps): this workgroup runs stages [s0, s0 + NS) and stores its raw
    // A^T M A sums to slice blockIdx.z of the workspace (the launcher clears scale / shift / residual / activation for the
    // launch; m3d_launch_splitk_reduce adds the slices in order and applies the epilogue)
    const int NS = a.nsl;
    // ---- XCD-aware block map for the split-K launches.  Workgroups go round-robin to the 8 XCDs by their linear id; each XCD has
    // its own 4 MB L2.  All strips of one (channel block, K slice) combination read the same 2.4 MB slice of U: with C = 8, 16, ...
    // combinations an XCD works through C / 8 of them one after the other, so that its L2 holds one slice at a time instead of
    // missing on all of them (PMC, 512 -> 512 @ 12x40 split-K 4: 427 MB per launch through the L2s for 115 MB of operands with the
    // plain map; ~3 % per layer).  Pinning the 2 / 4 channel blocks of an unsplit launch to XCD pairs measured slower (the input
    // strips are then read through four L2s instead of two): those keep the plain map. ------------------------------------------
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int X = gridDim.x, C = gridDim.y * gridDim.z;
        if (gridDim.z > 1 && (C & 7) == 0) {
            const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * X + blockIdx.x, xcd = lin & 7, idx = lin >> 3;
            const int combo = xcd * (C >> 3) + idx / X;
            bx = idx % X;
            by = combo % (int)gridDim.y;
            bz = combo / (int)gridDim.y;
        }
    }
    const int s0 = (bz * KS + kh) * a.nsl;             // (a.nsl = stages per wave: Cin / 16 / (gridDim.z * KS))
    W44_TRACE_INIT();
    W44_TRACE();

    // ---- this thread's transform unit: tile ut of the strip, input channel uc of the stage ----------------------------------
    const int ut = tid >> 4, uc = tid & 15;
    unsigned vM[6], v0[6], v5[6];    // byte offsets of the patch rows at the tile's first pixel column: columns 1..4 / column 0
                                     // (base one pixel to the left, out of range at the left image border) / column 5 (+ 4 pixels)
    {
        const int t = bx * 16 + ut;
        const bool tv = t < a.NT;
        const int tt = tv ? t : 0;
        const int n = tt / (a.TH * a.TW), rem = tt - n * a.TH * a.TW;
        const int ty = rem / a.TW, tx = rem - ty * a.TW;
        if (uc == 0 && kh == 0) pixb[ut] = tv ? (n * a.H + 4 * ty) * a.W + 4 * tx : -1;
        const bool c5ok = 4 * tx + 4 < a.W;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int hi = 4 * ty - 1 + r;
            const bool ok = tv && hi >= 0 && hi < a.H;
            const unsigned base = ((unsigned)((n * a.H + hi) * a.W + 4 * tx) * (unsigned)a.in_cs + (unsigned)uc) * 4u;
            vM[r] = ok ? base : M3D_BUF_OOB;
            v0[r] = (ok && tx > 0) ? base - (unsigned)a.in_cs * 4u : M3D_BUF_OOB;
            v5[r] = (ok && c5ok) ? base : M3D_BUF_OOB;
        }
    }
    const unsigned cs4 = (unsigned)a.in_cs * 4u;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in + s0 * 16, a.in_bytes - (unsigned)s0 * 64u);
    // B fragments of this wave's 16 * NB output channels: [32-channel block][stage][xi][j][lane][4]
    const int cb16 = (by * 4 + wave) * NB;                       // first 16-channel block of this wave
    const int cb32 = cb16 >> 1, j0 = cb16 & 1;
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.U + ((size_t)cb32 * (a.Cin >> 4) + s0) * (36 * 2 * 256), (unsigned)NS * (36u * 2u * 1024u));
    const unsigned ulane = (unsigned)lane * 16u + (unsigned)j0 * 1024u;

    float d[36];
    auto load_patch = [&](int s) __attribute__((always_inline)) {
        const unsigned so = (unsigned)s * 64u;                   // 16 channels per stage
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            w44_load_dword(d[r * 6 + 0], rin, v0[r], so);
#pragma unroll
            for (int c = 1; c < 5; ++c) w44_load_dword(d[r * 6 + c], rin, vM[r], so + (unsigned)(c - 1) * cs4);
            w44_load_dword(d[r * 6 + 5], rin, v5[r], so + 4u * cs4);
        }
    };
    auto load_patch_one = [&](int s, auto qtag) __attribute__((always_inline)) {
        constexpr int q = decltype(qtag)::value, r = q / 6, c = q % 6;
        const unsigned so = (unsigned)s * 64u;
        if constexpr (c == 0) w44_load_dword(d[q], rin, v0[r], so);
        else if constexpr (c == 5) w44_load_dword(d[q], rin, v5[r], so + 4u * cs4);
        else w44_load_dword(d[q], rin, vM[r], so + (unsigned)(c - 1) * cs4);
    };
    // B^T d B of the patch in d[] -> V[buf][xi][ut][uc], in 12 pieces (6 columns of the first pass, in place, 6 rows of the
    // second pass + their 6 LDS writes): inside a stage the pieces ride between the MFMAs of positions 5..16 -- one wave per
    // SIMD issues in order, a 16x16x4 MFMA occupies the matrix pipe for 32 cycles and the issue port for a fraction of that, so
    // a dozen scalar FMAs per position fill slots that are otherwise empty (as a block after the 288 MFMAs the transform cost
    // 1900 of every 14600 cycles with the matrix pipe idle, tools/wino44_trace.py)
    auto transform_col = [&](auto ctag) __attribute__((always_inline)) {       // first pass IN PLACE: d[] doubles as the temporary
        constexpr int c = decltype(ctag)::value;
        float t0, t1, t2, t3, t4, t5;
        w44_bt(d[0 * 6 + c], d[1 * 6 + c], d[2 * 6 + c], d[3 * 6 + c], d[4 * 6 + c], d[5 * 6 + c], t0, t1, t2, t3, t4, t5);
        d[0 * 6 + c] = t0; d[1 * 6 + c] = t1; d[2 * 6 + c] = t2; d[3 * 6 + c] = t3; d[4 * 6 + c] = t4; d[5 * 6 + c] = t5;
    };
    auto transform_row = [&](int buf, auto rtag) __attribute__((always_inline)) {
        constexpr int r = decltype(rtag)::value;
        float *vb = Vs + buf * (36 * 256) + ut * 16 + w44_vswz(ut, uc);
        float v[6];
        w44_bt(d[r * 6 + 0], d[r * 6 + 1], d[r * 6 + 2], d[r * 6 + 3], d[r * 6 + 4], d[r * 6 + 5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
        for (int c = 0; c < 6; ++c) vb[(r * 6 + c) * 256] = v[c];
    };
    auto transform_store = [&](int buf) __attribute__((always_inline)) {
        transform_col(std::integral_constant<int, 0>{}); transform_col(std::integral_constant<int, 1>{});
        transform_col(std::integral_constant<int, 2>{}); transform_col(std::integral_constant<int, 3>{});
        transform_col(std::integral_constant<int, 4>{}); transform_col(std::integral_constant<int, 5>{});
        transform_row(buf, std::integral_constant<int, 0>{}); transform_row(buf, std::integral_constant<int, 1>{});
        transform_row(buf, std::integral_constant<int, 2>{}); transform_row(buf, std::integral_constant<int, 3>{});
        transform_row(buf, std::integral_constant<int, 4>{}); transform_row(buf, std::integral_constant<int, 5>{});
    };
    auto tie_patch = [&]() __attribute__((always_inline)) {      // the loads have landed (vmcnt waits below); pin the registers
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]),
                          "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]), "+v"(d[16]), "+v"(d[17]));
        asm volatile("" : "+v"(d[18]), "+v"(d[19]), "+v"(d[20]), "+v"(d[21]), "+v"(d[22]), "+v"(d[23]), "+v"(d[24]), "+v"(d[25]), "+v"(d[26]),
                          "+v"(d[27]), "+v"(d[28]), "+v"(d[29]), "+v"(d[30]), "+v"(d[31]), "+v"(d[32]), "+v"(d[33]), "+v"(d[34]), "+v"(d[35]));
    };

    f32x4 Uf[4][2];              // ring of four register sets: one in use, three in flight
    auto load_u = [&](int s, int xi, auto slot_tag) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_tag)::value;
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((s * 36 + xi) * 2) * 1024u;
        w44_load_x4(Uf[SL][0], ru, ulane, so);
        if constexpr (NB == 2) w44_load_x4(Uf[SL][1], ru, ulane, so + 1024u);
    };

    // ---- cache warm-up for the next F(4x4) layer: its U tensor (2.4-38 MB) is cold in HBM when that layer starts, and the B
    // fragments run only four positions (~1400 cycles: an L2 / MALL round trip, not an HBM one) ahead of the MFMAs -- measured
    // on 256 -> 256 @ 24x80: 0.065 ms with U resident, 0.092 ms cold, 0.087 ms in the network.  Every thread of THIS launch
    // touches a few lines of it before its own first loads: the round trip rides under the patch loads' (same vmcnt(0) below).
    // (Issued at the end of the kernel instead -- closer to the consumer -- the step measured 6.45 instead of 6.39 ms.) ---------
    float touched = 0.f;
    if (a.touch) {
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int i = wg * (256 * KS) + tid8; i < a.touch_lines; i += nwg * (256 * KS))
            asm volatile("global_load_dword %0, %1, off" : "=v"(touched) : "v"(a.touch + (size_t)i * 128) : "memory");
    }
    // ---- prologue: stage 0's patch, transform, first B fragments, stage 1's patch ------------------------------------------
    load_patch(0);
    f32x4 acc[36][NB];
#pragma unroll
    for (int x = 0; x < 36; ++x)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[x][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(touched) : : "memory");      // (the touch loads' target stays allocated until here)
    tie_patch();
    W44_TRACE();
    transform_store(0);
    W44_TRACE();
    // INTERLEAVE (16-channel waves): the patch loads of stage s + 2 ride inside stage s.  With 32-channel waves (288 accumulators of
    // the 512 registers) that variant made the compiler spill, and a spill is a vector memory instruction in the middle of the
    // hand-counted vmcnt sequence (wrong fragments): there the 36 loads stay a block at the end of the stage.
    constexpr bool INTERLEAVE = NB == 1;
    if (INTERLEAVE && NS > 1) load_patch(1);          // older than the first fragments: the steady-state order of a stage entry
    load_u(0, 0, std::integral_constant<int, 0>{});
    load_u(0, 1, std::integral_constant<int, 1>{});
    load_u(0, 2, std::integral_constant<int, 2>{});
    load_u(0, 3, std::integral_constant<int, 3>{});
    if (!INTERLEAVE && NS > 1) load_patch(1);
    __syncthreads();
    W44_TRACE();

    const int atile = lane & 15, aq = lane >> 4;
    // One stage: 36 positions x 8 MFMAs on V[buf].  Outstanding loads on entry, oldest first: the 36 patch loads of the next stage
    // (unless LAST), then the B fragments of positions 0..3 (4 NB loads).  Behind position xi the fragments of xi + 4 go out (the
    // next stage's first four behind positions 32..35), and -- ISSUE: there is a stage s + 2 -- behind positions 17..28 three of the
    // 36 patch loads of stage s + 2 each (d[] is free once the first transform pass has read it, position 10, and the second pass
    // has released its temporaries, position 16 -- earlier the two together spill; issued as a block at the end of the stage the
    // loads cost ~1000 of its ~12700 cycles with the matrix pipe idle).  Loads younger than position xi's fragments when they are
    // waited for: 3 NB fragment loads + 3 per position of [xi-4, xi-1] that lies in [17, 28].
    auto stage = [&](int s, auto last_tag, auto issue_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value, ISSUE = decltype(issue_tag)::value;
        const float *vbuf = Vs + (s & 1) * (36 * 256) + atile * 16 + w44_vswz(atile, aq * 4);
        f32x4 anext = *reinterpret_cast<const f32x4 *>(vbuf);
        // (compile-time positions: a `#pragma unroll` loop of this size was left rolled once the transform pieces were added, which
        // made acc[xi] a dynamically indexed array -- 288 accumulators in scratch memory)
        w44_static_for<0, 36>([&](auto xtag) __attribute__((always_inline)) {
            constexpr int xi = decltype(xtag)::value;
            constexpr int nx = xi + W44_LA;
            // slot xi % 4 holds position xi until its MFMAs are issued; the fragments of xi + 4 go out behind them
            constexpr int plo = xi - 4 > 17 ? xi - 4 : 17, phi = xi - 1 < 28 ? xi - 1 : 28;
            constexpr int pyoung = !INTERLEAVE ? (xi < W44_LA ? 36 : 0)                  // block issue: the next stage's patch sits
                                               : ((ISSUE && phi >= plo) ? 3 * (phi - plo + 1) : 0);   // behind the first fragments
            if constexpr (!LAST) w44_wait<3 * NB + pyoung, NB>(Uf[xi % 4]);
            else if constexpr (xi <= 32) w44_wait<3 * NB, NB>(Uf[xi % 4]);
            else w44_wait<(35 - xi) * NB, NB>(Uf[xi % 4]);
            // A operand: this position's was fetched during the previous position; the next one goes out before this position's
            // MFMAs (left alone, the scheduler sinks the read to the last MFMA of the position)
            const f32x4 av = anext;
            if constexpr (xi + 1 < 36) anext = *reinterpret_cast<const f32x4 *>(vbuf + (xi + 1) * 256);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[xi][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], Uf[xi % 4][j][e], acc[xi][j], 0, 0, 0);
            if constexpr (!LAST) {
                // the next stage's patch has landed by position 4 in either issue order.  The pieces cost
                // MFMA time where they sit (~10 cycles per VALU instruction: nothing hides under a SIMD's MFMA stream on this part,
                // DESIGN.md) -- but they no longer cost a separate phase with its own latencies
                if constexpr (xi == 4) tie_patch();
                if constexpr (xi >= 5 && xi <= 10) transform_col(std::integral_constant<int, xi - 5>{});
                if constexpr (xi >= 11 && xi <= 16) transform_row((s + 1) & 1, std::integral_constant<int, xi - 11>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            // fragments of position xi + 4 (this stage) or of the next stage's position xi + 4 - 36.  (Issued BETWEEN the MFMAs
            // of the position with a lookahead of 3 they measured the same for 32-channel waves and 10 % slower for 16-channel
            // waves: the matrix pipe is not what waits.)
            if constexpr (nx < 36) load_u(s, nx, std::integral_constant<int, nx % 4>{});
            else if constexpr (!LAST) load_u(s + 1, nx - 36, std::integral_constant<int, nx % 4>{});
            if constexpr (INTERLEAVE && ISSUE && xi >= 17 && xi <= 28) {
                load_patch_one(s + 2, std::integral_constant<int, 3 * (xi - 17)>{});
                load_patch_one(s + 2, std::integral_constant<int, 3 * (xi - 17) + 1>{});
                load_patch_one(s + 2, std::integral_constant<int, 3 * (xi - 17) + 2>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        W44_TRACE();
        if constexpr (!LAST) {
            if constexpr (!INTERLEAVE) { if (s + 2 < NS) load_patch(s + 2); }
            __builtin_amdgcn_sched_barrier(0);
            W44_TRACE();
            __syncthreads();
            W44_TRACE();
        }
    };
    if constexpr (INTERLEAVE) {
        for (int s = 0; s + 2 < NS; ++s) stage(s, std::false_type{}, std::true_type{});
        if (NS > 1) stage(NS - 2, std::false_type{}, std::false_type{});
    } else {
        for (int s = 0; s + 1 < NS; ++s) stage(s, std::false_type{}, std::false_type{});
    }
    stage(NS - 1, std::true_type{}, std::false_type{});

    // ---- A^T M A in registers (lane = channel co0 + 16 j + (lane & 15), tiles 4 (lane >> 4) + i), then through the wave's slice
    // of the (now free) V buffers so that the stores are 16 bytes per lane: a lane holds ONE channel of 4 tiles x 16 pixels -- written
    // out directly that is 128 four-byte stores per lane, 64 bytes contiguous per pixel, and took 18000 of the workgroup's 131000
    // cycles (tools/wino44_trace.py).  Per 16-channel block: [16 tiles][16 pixels + pad][16 channels] in LDS (tile stride 272
    // floats: the four tile groups of a store instruction land in different banks), read back as (tile k, pixel lane >> 2,
    // channel quad lane & 3): 1 KB contiguous per read, 16 bytes per lane per store, residual fetched in the same shape.
    __syncthreads();                                  // every wave is done with the last stage's V
    // K-pair form: trade accumulator halves.  A lane's f32x4 accumulator covers tiles 4 aq + {0, 1, 2, 3}; half 0 finishes tiles
    // {0, 1}, half 1 tiles {2, 3}: each writes the pair the OTHER half finishes ([half][xi][thread] f32x2: 2 x 73.7 KB = the V
    // buffers), then adds what it receives to the pair it kept (own + other: the two halves add the same two numbers).
    constexpr int NI = KS == 2 ? 2 : 4;               // tiles of a lane this wave finishes
    constexpr int NTW = KS == 2 ? 8 : 16;             // ... of the strip's 16
    using VT = std::conditional_t<KS == 2, f32x2, f32x4>;
    const int ioff = KS == 2 ? 2 * kh : 0;
    VT accv[36][NB];
    if constexpr (KS == 2) {
        f32x2 *xb = reinterpret_cast<f32x2 *>(Vs_all);
#pragma unroll
        for (int x = 0; x < 36; ++x) {
            const f32x4 v = acc[x][0];
            xb[(kh * 36 + x) * 256 + tid] = kh == 0 ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]};
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 36; ++x) {
            const f32x4 v = acc[x][0];
            const f32x2 o = xb[((kh ^ 1) * 36 + x) * 256 + tid];
            accv[x][0] = (kh == 0 ? f32x2{v[0], v[1]} : f32x2{v[2], v[3]}) + o;
            // (all 36 reads hoisted in front of the adds = 72 more live registers next to the 144 accumulators: spills)
            if (x % 6 == 5) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                              // the traded values are consumed: the LDS takes the output tiles now
    } else {
#pragma unroll
        for (int x = 0; x < 36; ++x)
#pragma unroll
            for (int j = 0; j < NB; ++j) accv[x][j] = __builtin_bit_cast(VT, acc[x][j]);
    }
    // ---- A^T M A in registers (lane = channel co0 + 16 j + (lane & 15), tiles 4 (lane >> 4) + i), then through the wave's slice
    // of the (now free) V buffers so that the stores are 16 bytes per lane: a lane holds ONE channel of 4 tiles x 16 pixels -- written
    // out directly that is 128 four-byte stores per lane, 64 bytes contiguous per pixel, and took 18000 of the workgroup's 131000
    // cycles (tools/wino44_trace.py).  Per 16-channel block: [tiles][16 pixels + pad][16 channels] in LDS (tile stride 272
    // floats: the four tile groups of a store instruction land in different banks), read back as (tile k, pixel lane >> 2,
    // channel quad lane & 3): 1 KB contiguous per read, 16 bytes per lane per store, residual fetched in the same shape.
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out + (size_t)bz * a.ws_slice, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res ? a.res : a.out, a.res ? a.res_bytes : 0u);
    float *ot = Vs_all + (kh * 4 + wave) * (NTW * 272);   // this wave's slice: 17 KB (K-pair: 8.5 KB) of the V buffers
    const int opx = lane >> 2, ocq = lane & 3;        // read-back role: pixel of the 4x4 tile, channel quad
    const unsigned pixoff = (unsigned)((opx >> 2) * a.W + (opx & 3));
    // slot lt of the wave's slice holds tile tile_of(lt) of the strip
    auto tile_of = [&](int lt) __attribute__((always_inline)) { return KS == 2 ? 4 * (lt >> 1) + ioff + (lt & 1) : lt; };
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        // the residual of the whole block is fetched in one go (loads in flight under the second transform pass; fetched inside
        // the store loop every iteration waited out its own memory round trip)
        const int c0 = (cb16 + j) * 16 + ocq * 4;     // first of this lane's four channels
        const bool blockfull = (cb16 + j) * 16 + 16 <= a.Cout;
        const unsigned ovoff = (pixoff * (unsigned)a.out_cs + (unsigned)c0) * 4u, rvoff = (pixoff * (unsigned)a.res_cs + (unsigned)c0) * 4u;
        f32x4 rvv[NTW];
        {
            VT z[4][6];                            // A^T M: rows yy, columns b; the vector runs over the lane's tiles
#pragma unroll
            for (int b = 0; b < 6; ++b)
                w44_at(accv[0 * 6 + b][j], accv[1 * 6 + b][j], accv[2 * 6 + b][j], accv[3 * 6 + b][j], accv[4 * 6 + b][j], accv[5 * 6 + b][j],
                       z[0][b], z[1][b], z[2][b], z[3][b]);
            // the block's accumulators are dead now: their registers take the residual, fetched under the second pass (earlier --
            // before the first pass -- the kernel needed more than its 512 registers, and a compiler-inserted spill is a vector
            // memory instruction that breaks the hand-counted vmcnt of the main loop)
            if (a.res && blockfull) {
#pragma unroll
                for (int k = 0; k < NTW; ++k) {
                    const int pb = __builtin_amdgcn_readfirstlane(pixb[tile_of(k)]);
                    rvv[k] = buf_load_f32x4(rres, rvoff, pb >= 0 ? (unsigned)pb * (unsigned)a.res_cs * 4u : M3D_BUF_OOB);
                }
            }                                      // (no residual: rvv stays unread -- zero-filled, the K-pair build spilled the zeros)
            float *op = ot + (NI * aq) * 272 + atile;
#pragma unroll
            for (int yy = 0; yy < 4; ++yy) {
                VT y[4];
                w44_at(z[yy][0], z[yy][1], z[yy][2], z[yy][3], z[yy][4], z[yy][5], y[0], y[1], y[2], y[3]);
#pragma unroll
                for (int xx = 0; xx < 4; ++xx)
#pragma unroll
                    for (int i = 0; i < NI; ++i) op[i * 272 + (yy * 4 + xx) * 16] = y[xx][i];
            }
        }
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < a.Cout) {
                if (a.scale) sc[e] = a.scale[c0 + e];
                if (a.shift) sh[e] = a.shift[c0 + e];
            }
        // Masking without branches: a tile that does not exist puts the out-of-range marker into the SGPR offset of its loads /
        // stores, channels past Cout into the lane offset (the buffer range check drops the access).  Only a 16-channel block
        // that straddles Cout (wave-uniform test) takes the element-wise path.
        if (blockfull) {
            // LeakyReLU as max(v, slope v) with slope 1 = none
            const float slope = a.act == 1 ? M3D_LEAKY_SLOPE : 1.f;
            auto body = [&](auto mode_tag) __attribute__((always_inline)) {
                constexpr int MODE = decltype(mode_tag)::value;          // 0: no residual, 1: (v * sc + sh) + r, 2: (v + r) * sc + sh
#pragma unroll
                for (int k = 0; k < NTW; ++k) {
                    const int pb = __builtin_amdgcn_readfirstlane(pixb[tile_of(k)]);
                    const unsigned osoff = pb >= 0 ? (unsigned)pb * (unsigned)a.out_cs * 4u : M3D_BUF_OOB;
                    f32x4 v = *reinterpret_cast<const f32x4 *>(ot + k * 272 + lane * 4);
                    if constexpr (MODE == 2) v = (v + rvv[k]) * sc + sh;
                    else if constexpr (MODE == 1) v = v * sc + sh + rvv[k];
                    else v = v * sc + sh;
                    v = __builtin_elementwise_max(v, v * slope);
                    buf_store_f32x4_nop(v, rout, ovoff, osoff);      // (+ the store-data wait states: common.h)
                }
            };
            if (!a.res) body(std::integral_constant<int, 0>{});
            else if (a.res_mode) body(std::integral_constant<int, 2>{});
            else body(std::integral_constant<int, 1>{});
        } else {
            for (int k = 0; k < NTW; ++k) {
                const int pb = __builtin_amdgcn_readfirstlane(pixb[tile_of(k)]);
                const unsigned osoff = pb >= 0 ? (unsigned)pb * (unsigned)a.out_cs * 4u : M3D_BUF_OOB;
                const unsigned rsoff = pb >= 0 ? (unsigned)pb * (unsigned)a.res_cs * 4u : M3D_BUF_OOB;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(ot + k * 272 + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned em = c0 + e < a.Cout ? (unsigned)e * 4u : M3D_BUF_OOB;
                    float v = v0[e];
                    if (a.res) {
                        const float rv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, rvoff + em, rsoff, 0));
                        if (a.res_mode) v = fmaf(v + rv, sc[e], sh[e]);
                        else v = fmaf(v, sc[e], sh[e]) + rv;
                    } else {
                        v = fmaf(v, sc[e], sh[e]);
                    }
                    if (a.act == 1) v = fmaxf(v, v * M3D_LEAKY_SLOPE);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, ovoff + em, osoff, 0);
                }
            }
        }
    }
    W44_TRACE();
}

// ---- F(4x4,3x3) for the 16 -> 16 full-resolution layer (DLA level0, pose_dla_dcn.py:341-342) -------------------------------------
// One stage (16 input channels), 16 output channels = exactly the N of v_mfma_f32_16x16x4_f32: no K loop, three phases through LDS.
//   1. thread (tile, input channel) of a 16-tile strip: 6x6 patch (36 dword loads, borders through the buffer range check), B^T d B,
//      36 values to V[xi][tile][channel];
//   2. wave w takes the transform positions 9w .. 9w + 8: A = V[xi] (one ds_read_b128 per lane), B = U[xi] (9 fragments per lane,
//      held in registers), 4 MFMAs, and M[xi][tile][cout] goes back into the SAME 1 KB of LDS -- V[xi] is read by this wave only;
//   3. thread (tile, output channel): 36 values, A^T m A, folded BatchNorm + LeakyReLU, 16 pixels x 4 bytes (the 16 channels of a
//      pixel are 64 contiguous bytes across the lanes).
// 36 KB of LDS, ~100 registers: four workgroups per CU cover each other's phases.  The direct MFMA kernel (backbone_kernels.hip) is
// bound by its 576 MFMAs per 256 pixels (144 here) and runs 0.18 ms at bs 8; this one 0.12 ms (~450 VALU / LDS / memory
// instructions per thread; HBM floor 252 MB in + 252 MB out = 0.063 ms).
__device__ __forceinline__ void w44_at1(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5,
                                        float &y0, float &y1, float &y2, float &y3)
{
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = fmaf(2.f, d2, d1);
    y2 = fmaf(4.f, s2, s1);
    y3 = fmaf(8.f, d2, d1) + m5;
}

struct Wino44C16Args {
    const float *in, *U, *scale, *shift;
    float *out;
    int in_cs, out_cs;
    unsigned in_bytes, out_bytes;
    int H, W, TH, TW, NT;
};

__global__ __launch_bounds__(256) void wino44_c16_kernel(const Wino44C16Args a)
{
    __shared__ __attribute__((aligned(16))) float Vs[36 * 256];          // V[xi][tile][channel], then M[xi][tile][cout]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ut = tid >> 4, uc = tid & 15;
    // B fragments of this wave's nine positions: U packed [xi][lane = 16 (cin / 4) + cout][cin % 4]
    f32x4 ub[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) ub[k] = *reinterpret_cast<const f32x4 *>(a.U + ((wave * 9 + k) * 64 + lane) * 4);

    // ---- phase 1: patch of (tile ut, channel uc) ----------------------------------------------------------------------------------
    // (an XCD-aware strip order -- contiguous strip ranges per XCD, for L2 hits on the shared halo rows -- measured the same)
    const int t = blockIdx.x * 16 + ut;
    const bool tv = t < a.NT;
    const int tt = tv ? t : 0;
    const int n = tt / (a.TH * a.TW), rem = tt - n * a.TH * a.TW;
    const int ty = rem / a.TW, tx = rem - ty * a.TW;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const unsigned cs4 = (unsigned)a.in_cs * 4u;
    float d[36];
    {
        // lane offsets of the tile's own first pixel (patch row 1, column 1) and of the three border origins (row 0, column 0, the
        // corner: out of range where the image ends -- the range check looks at the lane offset alone, so it must never wrap); the
        // (row, column) part of the address rides in the SGPR offset of the load: no per-load integer multiplies
        const unsigned org = tv ? ((unsigned)((n * a.H + 4 * ty) * a.W + 4 * tx) * (unsigned)a.in_cs + (unsigned)uc) * 4u : M3D_BUF_OOB;
        const bool r0ok = tv && ty > 0, r5ok = 4 * ty + 4 < a.H, c0ok = tv && tx > 0, c5ok = 4 * tx + 4 < a.W;
        const unsigned rowb = (unsigned)a.W * cs4;
        const unsigned orgR = r0ok ? org - rowb : M3D_BUF_OOB, orgC = c0ok ? org - cs4 : M3D_BUF_OOB;
        const unsigned orgRC = (r0ok && c0ok) ? org - rowb - cs4 : M3D_BUF_OOB;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                unsigned vo = (r == 0 && c == 0) ? orgRC : (r == 0 ? orgR : (c == 0 ? orgC : org));
                if (r == 5) vo = r5ok ? vo : M3D_BUF_OOB;
                if (c == 5) vo = c5ok ? vo : M3D_BUF_OOB;
                const int rr = r == 0 ? 0 : r - 1, cc = c == 0 ? 0 : c - 1;
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(rr * a.W + cc) * cs4;
                d[r * 6 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rin, vo, so, 0));
            }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float t0, t1, t2, t3, t4, t5;
        w44_bt(d[0 * 6 + c], d[1 * 6 + c], d[2 * 6 + c], d[3 * 6 + c], d[4 * 6 + c], d[5 * 6 + c], t0, t1, t2, t3, t4, t5);
        d[0 * 6 + c] = t0; d[1 * 6 + c] = t1; d[2 * 6 + c] = t2; d[3 * 6 + c] = t3; d[4 * 6 + c] = t4; d[5 * 6 + c] = t5;
    }
    float *vb = Vs + ut * 16 + w44_vswz(ut, uc);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        float v[6];
        w44_bt(d[r * 6 + 0], d[r * 6 + 1], d[r * 6 + 2], d[r * 6 + 3], d[r * 6 + 4], d[r * 6 + 5], v[0], v[1], v[2], v[3], v[4], v[5]);
#pragma unroll
        for (int c = 0; c < 6; ++c) vb[(r * 6 + c) * 256] = v[c];
    }
    __syncthreads();

    // ---- phase 2: positions 9 wave .. 9 wave + 8: M[xi] = V[xi] (16 tiles x 16 channels) x U[xi] (16 x 16), in place -----------------
    {
        const int atile = lane & 15, aq = lane >> 4;
        float *slot = Vs + wave * 9 * 256;
        f32x4 av[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) av[k] = *reinterpret_cast<const f32x4 *>(slot + k * 256 + atile * 16 + w44_vswz(atile, aq * 4));
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][e], ub[k][e], acc, 0, 0, 0);
            // D: lane = cout (lane & 15), registers = tiles 4 aq + i
#pragma unroll
            for (int i = 0; i < 4; ++i) slot[k * 256 + (4 * aq + i) * 16 + atile] = acc[i];
        }
    }
    __syncthreads();

    // ---- phase 3: A^T m A of (tile ut, cout uc), affine, LeakyReLU, 16 pixels ----------------------------------------------------------
    {
        float m[36];
#pragma unroll
        for (int x = 0; x < 36; ++x) m[x] = Vs[x * 256 + ut * 16 + uc];
        float z[4][6];
#pragma unroll
        for (int b = 0; b < 6; ++b)
            w44_at1(m[0 * 6 + b], m[1 * 6 + b], m[2 * 6 + b], m[3 * 6 + b], m[4 * 6 + b], m[5 * 6 + b], z[0][b], z[1][b], z[2][b], z[3][b]);
        const float sc = a.scale[uc], sh = a.shift[uc];
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, a.out_bytes);
        const unsigned obase = tv ? ((unsigned)((n * a.H + 4 * ty) * a.W + 4 * tx) * (unsigned)a.out_cs + (unsigned)uc) * 4u : M3D_BUF_OOB;
        const unsigned ocs4 = (unsigned)a.out_cs * 4u;
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) {
            float y[4];
            w44_at1(z[yy][0], z[yy][1], z[yy][2], z[yy][3], z[yy][4], z[yy][5], y[0], y[1], y[2], y[3]);
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                float v = fmaf(y[xx], sc, sh);
                v = fmaxf(v, v * M3D_LEAKY_SLOPE);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, obase,
                                                      (unsigned)__builtin_amdgcn_readfirstlane(yy * a.W + xx) * ocs4, 0);
            }
        }
    }
}

// level0 on the F(4x4,3x3) kernel: `U` = G g G^T packed [36][64 lanes = 16 (cin / 4) + cout][cin % 4] (m3dssd_amd/engine.py:pack_wino44_c16)
extern "C" int m3d_conv3x3_c16_wino(const float *in, int in_cs, const float *U, const float *scale, const float *shift, float *out,
                                    int out_cs, int N, int H, int W, m3d_stream_t stream)
{
    M3D_REQUIRE(in && U && scale && shift && out && in_cs >= 16 && out_cs >= 16 && N >= 1, "conv3x3_c16_wino: bad arguments");
    M3D_REQUIRE(H % 4 == 0 && W % 4 == 0 && H >= 4 && W >= 4, "conv3x3_c16_wino: H and W must be multiples of 4");
    M3D_REQUIRE(((uintptr_t)U & 15) == 0, "conv3x3_c16_wino: U must be 16-byte aligned");
    const long long ib = (long long)N * H * W * in_cs * 4, ob = (long long)N * H * W * out_cs * 4;
    M3D_REQUIRE(ib < (1ll << 31) && ob < (1ll << 31), "conv3x3_c16_wino: views must be < 2 GiB");
    Wino44C16Args a;
    a.in = in; a.U = U; a.scale = scale; a.shift = shift; a.out = out; a.in_cs = in_cs; a.out_cs = out_cs;
    a.in_bytes = (unsigned)ib; a.out_bytes = (unsigned)ob;
    a.H = H; a.W = W; a.TH = H / 4; a.TW = W / 4; a.NT = N * a.TH * a.TW;
    hipLaunchKernelGGL(wino44_c16_kernel, dim3(cdiv(a.NT, 16)), dim3(256), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// The main loop counts its outstanding vector-memory instructions by hand: a build in which the compiler spills registers (scratch
// loads / stores are vector-memory instructions too) would wait for the wrong loads.  Probed ONCE per process; such a build reports
// "not applicable" and the engine routes the layers to the F(2x2) / implicit-GEMM kernels instead of failing at launch time (ADVICE
// r3).  No device / probe failure = unknown = served (the launcher keeps its own REQUIRE as the backstop).
static int w44_scratch_bytes()
{
    static std::once_flag once;
    static int scratch = 0;
    std::call_once(once, []() {
#ifndef WINO_TRACE
        hipFuncAttributes f1, f2, f3, f4;
        if (hipFuncGetAttributes(&f1, reinterpret_cast<const void *>(&wino44_kernel<1, 1>)) == hipSuccess &&
            hipFuncGetAttributes(&f2, reinterpret_cast<const void *>(&wino44_kernel<2, 1>)) == hipSuccess &&
            hipFuncGetAttributes(&f3, reinterpret_cast<const void *>(&wino44_kernel<1, 2>)) == hipSuccess &&
            hipFuncGetAttributes(&f4, reinterpret_cast<const void *>(&wino44_kernel<1, 2, 2>)) == hipSuccess)
            scratch = (int)(f1.localSizeBytes + f2.localSizeBytes + f3.localSizeBytes + f4.localSizeBytes);
        else
            (void)hipGetLastError();
#endif
    });
    return scratch;
}
static int w44_build_ok() { return w44_scratch_bytes() == 0 ? 1 : 0; }

// 1 if the F(4x4,3x3) kernel serves the descriptor (geometry only; the caller passes U44-packed weights in d->wgt)
extern "C" int m3d_wino44_applicable(const m3d_conv_desc *d)
{
    if (!d || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1 || d->dil > 1) return 0;
    if (d->dcn_offmask || d->out_nchw || d->wgt_img_stride || d->sigmoid_from >= 0) return 0;
    if (d->H % 4 || d->W % 4 || d->Cin % 16 || d->Cout_pad % 64 || d->in_cs % 4) return 0;
    return w44_build_ok();
}

// Split-K plan of the F(4x4,3x3) kernel: *splits > 1 when the layer's 16-tile strips x 128-channel blocks do not fill the chip
// (256 -> 256 @ 24x80 and 512 -> 512 @ 12x40 at bs 8: 120 / 60 workgroups) and the input channels can be cut into slices of at
// least 64 that do; *ws_bytes = splits * N*H*W * Cout_pad * 4 to pass through splitk_ws.  A layer that needs no split (or that
// the kernel does not serve) reports 1 / 0.
// Form of the split-K launches: M3D_W44_SPLIT_NB = 2 (128-channel workgroups, one per CU) or 1 (64-channel workgroups, two per CU:
// `fill` = 480 workgroups); M3D_W44_SPLIT_FILL overrides the number of workgroups a layer must reach to run unsplit.
// 3 = K-pair workgroups (64-channel, 512 threads, one per CU) inside every global slice: half the slices, half the workspace.
static int w44_split_nb() { static const int v = []() { const char *e = getenv("M3D_W44_SPLIT_NB"); return e ? atoi(e) : 1; }(); return (v == 2 || v == 3) ? v : 1; }
static int w44_split_fill()
{
    static const int v = []() { const char *e = getenv("M3D_W44_SPLIT_FILL"); return e ? atoi(e) : 0; }();
    return v > 0 ? v : (w44_split_nb() == 1 ? 400 : 200);
}

extern "C" int m3d_wino44_splitk_plan(const m3d_conv_desc *d, int *splits, long long *ws_bytes)
{
    M3D_REQUIRE(d && splits && ws_bytes, "wino44_splitk_plan: null pointer");
    *splits = 1; *ws_bytes = 0;
    const int cw = w44_split_nb() == 2 ? 128 : 64;
    if (!m3d_wino44_applicable(d) || d->Cout_pad % cw) return M3D_OK;
    const long long wgs = (long long)cdiv(d->N * (d->H / 4) * (d->W / 4), 16) * (d->Cout_pad / cw);
    const int kp = w44_split_nb() == 3 ? 2 : 1;       // K-pair: every slice is halved again inside its workgroups
    const int ns = d->Cin / 16 / kp, fill = w44_split_fill();
    if (wgs >= fill || (d->Cin / 16) % kp) return M3D_OK;
    for (int sp = 2; sp <= 8; sp *= 2) {
        if (ns % sp || ns / sp < 4) break;
        if (wgs * sp >= fill) {
            const long long need = (long long)sp * d->N * d->H * d->W * d->Cout_pad * 4;
            if (need / sp < (1ll << 31)) { *splits = sp; *ws_bytes = need; }
            break;
        }
    }
    return M3D_OK;
}

// 1 if the unsplit 64-channel launch of the layer runs K-pair workgroups (512 threads, the two halves of the input channels side
// by side): the 64-channel workgroups would leave more than ~40 % of the 512 CU slots empty and each half still runs >= 4 stages
extern "C" int m3d_wino44_kpair(const m3d_conv_desc *d)
{
    if (!d || !m3d_wino44_applicable(d)) return 0;
    static const int kpair_max = []() { const char *e = getenv("M3D_W44_KPAIR_MAX"); return e ? atoi(e) : 300; }();
    static const int occ2 = []() { const char *e = getenv("M3D_W44_OCC2"); return e ? atoi(e) : 1; }();
    const int ns = d->Cin / 16;
    const long long wgs = (long long)cdiv(d->N * (d->H / 4) * (d->W / 4), 16) * (d->Cout_pad / 64);
    return (occ2 && wgs <= kpair_max && ns % 2 == 0 && ns / 2 >= 4) ? 1 : 0;
}

extern "C" int m3d_wino44_conv3x3_forward_touch(const m3d_conv_desc *d, int nb, const void *touch, long long touch_bytes,
                                                m3d_stream_t stream)
{
    M3D_REQUIRE(d && d->in && d->wgt && d->out, "wino44: null pointer");
    M3D_REQUIRE(m3d_wino44_applicable(d), "wino44: 3x3 / stride 1 / pad 1, H %% 4 == W %% 4 == 0, Cin %% 16 == 0, Cout_pad %% 64 == 0 only");
    M3D_REQUIRE(d->Ho == d->H && d->Wo == d->W, "wino44: Ho/Wo mismatch");
    Wino44Args a;
    a.in = d->in; a.U = d->wgt; a.out = d->out; a.scale = d->scale; a.shift = d->shift; a.res = d->res;
    a.in_cs = d->in_cs; a.out_cs = d->out_cs; a.res_cs = d->res_cs;
    const long long pix = (long long)d->N * d->H * d->W;
    M3D_REQUIRE(pix * d->in_cs * 4 < (1ll << 31) && pix * d->out_cs * 4 < (1ll << 31), "wino44: views must be < 2 GiB");
    a.in_bytes = (unsigned)(pix * d->in_cs * 4);
    a.out_bytes = (unsigned)(pix * d->out_cs * 4);
    a.res_bytes = d->res ? (unsigned)(pix * d->res_cs * 4) : 0u;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
    a.TH = d->H / 4; a.TW = d->W / 4; a.NT = d->N * a.TH * a.TW;
    a.act = d->act; a.res_mode = d->res_mode;
    a.nsl = d->Cin / 16; a.ws_slice = 0;
    M3D_REQUIRE(touch_bytes >= 0 && touch_bytes / 128 < (1ll << 31), "wino44: touch range");
    a.touch = touch_bytes >= 128 ? (const char *)touch : nullptr; a.touch_lines = touch ? (int)(touch_bytes / 128) : 0;
#ifdef WINO_TRACE
    a.trace = g_w44_trace;
#endif
    M3D_REQUIRE(w44_scratch_bytes() == 0, "wino44: this build of the kernel uses %d bytes of scratch memory per lane (register spills)",
                w44_scratch_bytes());
    const int strips = cdiv(a.NT, 16);
    M3D_REQUIRE(nb >= 0 && nb <= 2 && !(nb == 2 && d->Cout_pad % 128), "wino44: nb = 0 (automatic), 1 or 2 (needs Cout_pad %% 128 == 0)");
    // split-K when the caller provides the workspace the plan asks for (the form -- 128- or 64-channel workgroups -- is the plan's)
    if (d->splitk_ws && (nb == 0 || nb == (w44_split_nb() == 2 ? 2 : 1))) {
        int splits = 1;
        long long need = 0;
        if (const int rc = m3d_wino44_splitk_plan(d, &splits, &need)) return rc;
        if (splits > 1) {
            M3D_REQUIRE(need <= d->splitk_ws_bytes && ((uintptr_t)d->splitk_ws & 15) == 0,
                        "wino44: split-K workspace too small (%lld bytes, see m3d_wino44_splitk_plan) or misaligned", d->splitk_ws_bytes);
            a.out = d->splitk_ws; a.out_cs = d->Cout_pad; a.out_bytes = (unsigned)(pix * d->Cout_pad * 4);
            a.ws_slice = pix * d->Cout_pad; a.nsl = d->Cin / 16 / splits;
            a.scale = a.shift = a.res = nullptr; a.res_bytes = 0; a.act = 0; a.res_mode = 0; a.Cout = d->Cout_pad;
            if (w44_split_nb() == 1)
                hipLaunchKernelGGL((wino44_kernel<1, 2>), dim3(strips, d->Cout_pad / 64, splits), dim3(256), 0, (hipStream_t)stream, a);
            else if (w44_split_nb() == 3) {       // K-pair workgroups inside every global K slice (half the slices, half the workspace)
                a.nsl = d->Cin / 16 / (splits * 2);
                hipLaunchKernelGGL((wino44_kernel<1, 2, 2>), dim3(strips, d->Cout_pad / 64, splits), dim3(512), 0, (hipStream_t)stream, a);
            } else
                hipLaunchKernelGGL((wino44_kernel<2, 1>), dim3(strips, d->Cout_pad / 128, splits), dim3(256), 0, (hipStream_t)stream, a);
            M3D_LAUNCH_CHECK();
            SplitkReduceArgs r;
            r.ws = d->splitk_ws; r.scale = d->scale; r.shift = d->shift; r.res = d->res; r.out = d->out;
            r.M = (int)pix; r.Cout = d->Cout; r.Cout_pad = d->Cout_pad; r.splits = splits; r.out_cs = d->out_cs;
            r.res_cs = d->res_cs; r.res_mode = d->res_mode; r.act = d->act; r.sigmoid_from = d->sigmoid_from;
            return m3d_launch_splitk_reduce(r, (hipStream_t)stream);
        }
    }
    // 64-channel workgroups, two per CU (round 4: faster than the 128-channel form on every layer measured); nb = 2 asks for the
    // 128-channel form (one per CU, the whole register file) explicitly
    const bool nb2 = d->Cout_pad % 128 == 0 && nb == 2;
    static const int occ2 = []() { const char *e = getenv("M3D_W44_OCC2"); return e ? atoi(e) : 1; }();
    // K-pair workgroups (512 threads, the two halves of the input channels side by side) where the 64-channel workgroups would
    // leave more than ~40 % of the 512 CU slots empty and each half still runs >= 4 stages
    if (!nb2 && m3d_wino44_kpair(d)) {
        a.nsl = d->Cin / 16 / 2;
        hipLaunchKernelGGL((wino44_kernel<1, 2, 2>), dim3(strips, d->Cout_pad / 64), dim3(512), 0, (hipStream_t)stream, a);
        M3D_LAUNCH_CHECK();
        return M3D_OK;
    }
    if (nb2) hipLaunchKernelGGL((wino44_kernel<2, 1>), dim3(strips, d->Cout_pad / 128), dim3(256), 0, (hipStream_t)stream, a);
    else if (occ2) hipLaunchKernelGGL((wino44_kernel<1, 2>), dim3(strips, d->Cout_pad / 64), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((wino44_kernel<1, 1>), dim3(strips, d->Cout_pad / 64), dim3(256), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_wino44_conv3x3_forward_ex(const m3d_conv_desc *d, int nb, m3d_stream_t stream)
{
    return m3d_wino44_conv3x3_forward_touch(d, nb, nullptr, 0, stream);
}

extern "C" int m3d_wino44_conv3x3_forward(const m3d_conv_desc *d, m3d_stream_t stream)
{
    return m3d_wino44_conv3x3_forward_touch(d, 0, nullptr, 0, stream);
}
