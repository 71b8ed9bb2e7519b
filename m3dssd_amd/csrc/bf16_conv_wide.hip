// 3x3 / stride 1 / pad 1 bf16 convolution with 128 x 128 wave tiles: ONE WAVE = an 8 x 16 patch of output pixels x 128 output
// channels, one wave per SIMD, no workgroup barriers.
//
// Why: the halo-tile kernel (bf16_conv.hip) gives a wave 64 pixels x 64 channels = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16; every
// block needs one 1 KB weight fragment and one 1 KB pixel fragment from LDS per 2 MFMAs -- 1 KB of ds_read_b128 per MFMA, 128 bytes
// per clock and CU at the matrix peak, which is what the LDS delivers: the kernel sits at 0.37-0.40 of the bf16 peak with the
// matrix pipe waiting for operands.  A 4 x 4 block tile needs 8 fragments per 16 MFMAs -- 0.5 KB per MFMA --
// and at one wave per SIMD (256 accumulator registers in AGPRs, as in wino44_conv.hip) the weight half of that can come straight
// from global memory in fragment order (4 x 16 bytes per lane and K-step, fully coalesced, the four waves of a CU hit the same
// lines in L1), which leaves the LDS 4 ds_read_b128 per 16 MFMAs.
//   * The wave's input patch (10 x 18 pixels, 32 channels per chunk) lives in its OWN 15 KB LDS buffer, double buffered: no
//     barriers.  Pixel records are 80 bytes (64 + 16 pad), rows 1536 bytes (a multiple of 256): the 16-lane groups of a
//     ds_read_b128 -- {0-3, 12-15, 20-27} = columns 0-3, 12-15 of one patch row and 4-11 of the next -- cover all 64 banks
//     exactly once, and a tap is an immediate offset.
//   * K order: chunk of 32 input channels, tap, 16-channel K-step = 18 positions x 16 MFMAs per chunk.  Weight fragments run six
//     positions (3000 cycles) ahead in a ring of six register sets with hand-counted s_waitcnt vmcnt; the next chunk's patch (12
//     16-byte loads per lane) goes out once per chunk right behind a weight issue and is written to LDS eight positions later.
//   * Epilogue: bf16_tile.h (folded BatchNorm / bias, residual, LeakyReLU), through the wave's LDS region in whole pixel rows.
// Weights: m3dssd_amd.engine_bf16.PackedBf16.wave3x3() -- [Cout_pad/128][Cin/32][9 taps][2 K-steps][4 channel blocks][64 lanes][8].
#include <mutex>
#include <type_traits>

#include "bf16_tile.h"

#define CW_PS 80                          // bytes per patch pixel (32 channels + 16 pad)
#define CW_RS 1536                        // bytes per patch row (18 pixels = 1440, padded to a multiple of 256)
#define CW_HB (10 * CW_RS)                // one patch buffer
#define CW_LA 6                           // weight positions in flight
#define CW_WAVE_LDS (32768 + 1024)        // per wave: two patch buffers, later the 128 x 128 bf16 output tile; + scale / shift

struct WideArgs {
    Bf16Args b;
    const void *wfrag;
    int ppx, ppy, npatch;                 // 8 x 16 patches per row / column / in total
    unsigned wfrag_group_bytes;           // bytes of one 128-channel group of the fragment-ordered weights
    unsigned out_bytes;
};

#ifdef BF16_TRACE
// diagnostic build (make trace): s_memtime stamps of lane 0 of every wave, 64 slots per wave (tools/bf16_wide_trace.py)
#define CW_TRACE_INIT() long long *trp = a.trace ? a.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 64 : nullptr; int tri = 0
#define CW_TRACE() do { if (trp && lane == 0 && tri < 64) trp[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define CW_TRACE_INIT()
#define CW_TRACE()
#endif

template <int I, int N, typename F>
__device__ __forceinline__ void cw_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cw_static_for<I + 1, N>(f);
    }
}

template <int CT>
__device__ __forceinline__ void cw_load_w(u32x4 &dst, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(dst) : "v"(voff), "s"(r), "s"(soff), "n"(CT * 1024) : "memory");
}
__device__ __forceinline__ void cw_load_h(u32x4 &dst, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
// the four fragments of a ring slot are operands of the wait: nothing that uses them can be scheduled above it
template <int N>
__device__ __forceinline__ void cw_wait(u32x4 (&w)[4])
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N));
}

template <bool HAS_RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bf16_conv3x3_wide_kernel(const WideArgs wa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cw_lds[];
    const Bf16Args &a = wa.b;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    unsigned char *wl = cw_lds + wave * CW_WAVE_LDS;
    float *ssl = reinterpret_cast<float *>(wl + 32768);

    CW_TRACE_INIT();
    CW_TRACE();
    const int patch = blockIdx.x * 4 + wave;
    if (patch >= wa.npatch) return;                         // (no barriers in this kernel)
    const int n0 = blockIdx.y * 128;
    const int per = wa.ppx * wa.ppy;
    // (wave-uniform, but integer division runs on the VALU: without the readfirstlane these live in VGPRs across the K loop)
    const int img = __builtin_amdgcn_readfirstlane(patch / per), prem = patch - img * per;
    const int py = __builtin_amdgcn_readfirstlane(prem / wa.ppx), px = prem - py * wa.ppx;
    const int y0 = py * 8, x0 = px * 16;

    // ---- patch staging map (12 pieces of 16 bytes per lane): pieces 0..9 = patch row p, columns 0..15 (lane >> 2), 16-byte part
    // lane & 3; piece 10 = columns 16, 17 of rows 0..7 (lane >> 3), piece 11 (16 lanes) = those of rows 8, 9.  Row validity is
    // wave-uniform (a select against an SGPR mask per piece), column validity is folded into the lane's base offset: an
    // out-of-range marker stays out of range when the row offset is added. ------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const unsigned rowb = (unsigned)a.W * (unsigned)a.in_cs * 2u;            // bytes per image row
    const unsigned part16 = (unsigned)(lane & 3) * 16u;
    unsigned hbA, hbB, hbC, hlA, hlB;
    {
        const int hxA = lane >> 2, hxB = 16 + ((lane >> 2) & 1), hyB = lane >> 3;
        const unsigned org = ((unsigned)((img * a.H + y0 - 1) * a.W + x0 - 1)) * (unsigned)a.in_cs * 2u;     // (wraps for the row above image 0: masked)
        hbA = (x0 + hxA - 1 >= 0) ? org + (unsigned)hxA * (unsigned)a.in_cs * 2u + part16 : M3D_BUF_OOB;
        const bool colB = x0 + hxB - 1 < a.W;
        const unsigned oB = org + (unsigned)hxB * (unsigned)a.in_cs * 2u + part16;
        hbB = (colB && y0 - 1 + hyB >= 0) ? oB + (unsigned)hyB * rowb : M3D_BUF_OOB;
        hbC = (colB && lane < 16 && y0 + 7 + hyB < a.H) ? oB + (unsigned)(8 + hyB) * rowb : M3D_BUF_OOB;
        hlA = (unsigned)(hxA * CW_PS) + part16;
        hlB = (unsigned)(hyB * CW_RS + hxB * CW_PS) + part16;
    }
    const bool row0ok = y0 > 0, row9ok = y0 + 8 < a.H;
    u32x4 hv[12];
    auto load_patch = [&](int c) __attribute__((always_inline)) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(c) * 64u;       // 32 channels
#pragma unroll
        for (int p = 0; p < 10; ++p) {
            unsigned vo = hbA + (unsigned)p * rowb;
            if (p == 0) vo = row0ok ? vo : M3D_BUF_OOB;
            if (p == 9) vo = row9ok ? vo : M3D_BUF_OOB;
            cw_load_h(hv[p], rin, vo, so);
        }
        cw_load_h(hv[10], rin, hbB, so);
        cw_load_h(hv[11], rin, hbC, so);
    };
    auto tie_patch = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]));
        asm volatile("" : "+v"(hv[6]), "+v"(hv[7]), "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]));
    };
    auto store_patch = [&](int buf, auto ptag) __attribute__((always_inline)) {        // pieces 3 P .. 3 P + 2
        constexpr int P = decltype(ptag)::value;
        unsigned char *hb = wl + buf * CW_HB;
#pragma unroll
        for (int p = 3 * P; p < 3 * P + 3; ++p) {
            if (p < 10) *reinterpret_cast<u32x4 *>(hb + hlA + p * CW_RS) = hv[p];
            else if (p == 10) *reinterpret_cast<u32x4 *>(hb + hlB) = hv[10];
            else if (lane < 16) *reinterpret_cast<u32x4 *>(hb + hlB + 8 * CW_RS) = hv[11];
        }
    };

    // ---- weight fragments: [group][chunk][tap][K-step][channel block][lane][8]: 4 KB per position -----------------------------------
    const __amdgpu_buffer_rsrc_t rw = make_rsrc((const char *)wa.wfrag + (size_t)blockIdx.y * wa.wfrag_group_bytes, wa.wfrag_group_bytes);
    const unsigned wlane = (unsigned)lane * 16u;
    u32x4 wf[CW_LA][4];
    auto load_w = [&](int pos, auto slot_tag) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_tag)::value;
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(pos) * 4096u;
        cw_load_w<0>(wf[SL][0], rw, wlane, so);
        cw_load_w<1>(wf[SL][1], rw, wlane, so);
        cw_load_w<2>(wf[SL][2], rw, wlane, so);
        cw_load_w<3>(wf[SL][3], rw, wlane, so);
    };

    const int NCH = a.Cin >> 5, NPOS = NCH * 18;
    f32x16 acc[4][4];                                       // [channel block][pixel block]; the first position multiplies onto a zero
                                                            // operand (256 v_accvgpr_write cost 2 400 cycles in front of the first MFMA)

    // ---- prologue: patch of chunk 0, the first six weight positions ----------------------------------------------------------------
    load_patch(0);
    load_w(0, std::integral_constant<int, 0>{}); load_w(1, std::integral_constant<int, 1>{}); load_w(2, std::integral_constant<int, 2>{});
    load_w(3, std::integral_constant<int, 3>{}); load_w(4, std::integral_constant<int, 4>{}); load_w(5, std::integral_constant<int, 5>{});
    // ---- affine parameters of the 128 channels -> the wave's LDS copy (behind the first loads: its round trip
    // overlaps theirs; the compiler's own waits only make the hand-counted one below trivially true) -----------------------------
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        const bool ok = n0 + c < a.Cout;
        ssl[c] = (ok && a.scale) ? a.scale[n0 + c] : (ok ? 1.f : 0.f);
        ssl[128 + c] = (ok && a.shift) ? a.shift[n0 + c] : 0.f;
    }

    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");       // the 12 patch loads are the oldest
    tie_patch();
    store_patch(0, std::integral_constant<int, 0>{}); store_patch(0, std::integral_constant<int, 1>{});
    store_patch(0, std::integral_constant<int, 2>{}); store_patch(0, std::integral_constant<int, 3>{});

    // pixel fragment of block i at (tap, K-step): row (2 i + dy + (l31 >> 4)), column (l31 & 15) + dx, channels 16 ks + 8 lh
    const unsigned pbase = (unsigned)((l31 >> 4) * CW_RS + (l31 & 15) * CW_PS + lh * 16);
    auto read_pix = [&](bf16x8 (&pf)[4], int buf, auto ptag) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value, tap = P >> 1, ks = P & 1, dy = tap / 3, dx = tap % 3;
        const unsigned char *src = wl + buf * CW_HB + pbase;
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[i] = *reinterpret_cast<const bf16x8 *>(src + (2 * i + dy) * CW_RS + dx * CW_PS + ks * 32);
    };
    bf16x8 pfa[4], pfb[4];
    read_pix(pfa, 0, std::integral_constant<int, 0>{});
    CW_TRACE();

    // One chunk: 18 positions.  Outstanding vector-memory loads when position P waits for its weights (oldest first): the weights of P,
    // [the 12 patch loads of the next chunk if they went out behind position 0 and P <= 6], the weights of P + 1 .. P + 5.
    auto chunk = [&](int c, auto first_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        const int buf = c & 1, pos0 = c * 18;
        cw_static_for<0, 18>([&](auto ptag) __attribute__((always_inline)) {
            constexpr int P = decltype(ptag)::value;
            constexpr int ahead = LAST ? (17 - P < 5 ? 17 - P : 5) : 5;
            constexpr int young = 4 * ahead + ((!LAST && P >= 1 && P <= 6) ? 12 : 0);
            cw_wait<young>(wf[P % CW_LA]);
            bf16x8 (&cur)[4] = (P & 1) ? pfb : pfa;
            bf16x8 (&nxt)[4] = (P & 1) ? pfa : pfb;
            constexpr int SL = P % CW_LA;
            constexpr bool MOREW = !LAST || P + CW_LA < 18;          // the slot's next user: position P + 6 (of the next chunk from P = 12 on)
            const unsigned wso = (unsigned)__builtin_amdgcn_readfirstlane(pos0 + P + CW_LA) * 4096u;
            // Everything that is not an MFMA rides BETWEEN the four MFMA groups (one group = one weight fragment x 4 pixel blocks = 128
            // matrix-pipe cycles): issued as a block behind the 16th MFMA, the 4 loads + 4 LDS reads + the wait left the pipe idle
            // ~65 of every 577 cycles (tools/bf16_wide_trace.py).  A fragment's register set is reloaded as soon as its group is issued.
            auto group = [&](auto jtag) __attribute__((always_inline)) {
                constexpr int j = decltype(jtag)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (FIRST && P == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[SL][j]), cur[i], zero, 0, 0, 0);
                    } else {
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[SL][j]), cur[i], acc[j][i], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MOREW) cw_load_w<j>(wf[SL][j], rw, wlane, wso);
            };
            __builtin_amdgcn_sched_barrier(0);
            group(std::integral_constant<int, 0>{});
            if constexpr (P + 1 < 18) read_pix(nxt, buf, std::integral_constant<int, P + 1>{});
            else if constexpr (!LAST) read_pix(nxt, buf ^ 1, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            group(std::integral_constant<int, 1>{});
            if constexpr (!LAST) {
                if constexpr (P == 7) tie_patch();           // position 7's weights went out behind the patch loads: they have landed
                if constexpr (P >= 8 && P <= 11) store_patch(buf ^ 1, std::integral_constant<int, P - 8>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            group(std::integral_constant<int, 2>{});
            __builtin_amdgcn_sched_barrier(0);
            group(std::integral_constant<int, 3>{});
            if constexpr (!LAST && P == 0) load_patch(c + 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (P == 0 || P == 6 || P == 7 || P == 12 || P == 17) CW_TRACE();
        });
    };
    chunk(0, std::true_type{}, std::false_type{});           // (Cin >= 64: at least two chunks)
    for (int c = 1; c + 1 < NCH; ++c) chunk(c, std::false_type{}, std::false_type{});
    chunk(NCH - 1, std::false_type{}, std::true_type{});
    (void)NPOS;

    // ---- epilogue: acc -> (affine, residual, LeakyReLU) -> bf16 -> the wave's LDS tile [128 pixels][128 channels] -> whole rows.
    // One wave per SIMD: nothing runs under it, every instruction counts (first version: 20 000 of a 4-chunk wave's 78 000 cycles).
    //   * channel block j outermost: its 8 scale / shift vectors are read from LDS once for the 4 pixel blocks; the residual of block
    //     j + 1 (16 eight-byte loads) is in flight under the arithmetic of block j;
    //   * no runtime branches (with `if (residual)` / `if (res_mode)` paths around 256 accumulators the compiler spilled 6-70
    //     registers): the residual is a template parameter, res_mode a factor;
    //   * the lane id is recomputed: derived from the kernel's `lane`, the index arithmetic was computed -- and spilled -- before the
    //     K loop;
    //   * stores: row r0 + 4 p of the LDS tile is pixel (2 (p >> 3) + ((p >> 2) & 1), r0 + 4 (p & 3)) of the patch: the p part of the
    //     address is wave-uniform (SGPR offset of the buffer store), the LDS address an immediate + one of two swizzled lane offsets. ----
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int l31e = lane_e & 31, lhe = lane_e >> 5;
    {
        const float slope = a.act ? M3D_LEAKY_SLOPE : 1.f;
        const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res ? a.res : a.out, a.res ? a.res_bytes : 0u);
        const bool rm1 = a.res_mode == 1;
        unsigned rpix[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rpix[i] = (unsigned)((img * a.H + y0 + 2 * i + (l31e >> 4)) * a.W + x0 + (l31e & 15)) * (unsigned)a.res_cs * 2u;
        u32x2 rr[2][4][4];                                  // [buffer][pixel block][register group]
        auto load_res = [&](int j, u32x2 (&r)[4][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = n0 + j * 32 + 4 * lhe + 8 * g;
                    r[i][g] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, c0 < a.Cout ? rpix[i] + (unsigned)c0 * 2u : M3D_BUF_OOB, 0, 0));
                }
        };
        if constexpr (HAS_RES) load_res(0, rr[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (HAS_RES) { if (j + 1 < 4) load_res(j + 1, rr[(j + 1) & 1]); }
            // res_mode 0: acc * scale + shift + res; res_mode 1: (acc + res) * scale + shift = acc * scale + shift + res * scale:
            // one form, acc * scale + shift + res * k with k = 1 or scale (differs from the two-step form by an fp32 rounding)
            f32x4 sc[4], sh[4], rk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = j * 32 + 4 * lhe + 8 * g;
                sc[g] = *reinterpret_cast<const f32x4 *>(ssl + cl);
                sh[g] = *reinterpret_cast<const f32x4 *>(ssl + 128 + cl);
                rk[g] = rm1 ? sc[g] : f32x4{1.f, 1.f, 1.f, 1.f};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 32 + l31e;
                unsigned pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 x = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                    x = x * sc[g] + sh[g];
                    if constexpr (HAS_RES) {
                        const unsigned r0 = rr[j & 1][i][g][0], r1 = rr[j & 1][i][g][1];
                        const f32x4 rs = __builtin_bit_cast(f32x4, u32x4{r0 << 16, r0 & 0xFFFF0000u, r1 << 16, r1 & 0xFFFF0000u});
                        x = rs * rk[g] + x;
                    }
                    x = __builtin_elementwise_max(x, x * slope);
                    pk[g][0] = pack_bf16(x[0], x[1]);
                    pk[g][1] = pack_bf16(x[2], x[3]);
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(pk[g][e], pk[g + 1][e], false, false);
                        pk[g][e] = r[0]; pk[g + 1][e] = r[1];
                    }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const int ch16 = (j * 32 + 8 * (g + lhe)) >> 3;
                    *reinterpret_cast<u32x4 *>(wl + row * 256 + ((ch16 ^ (row & 7)) << 4)) = u32x4{pk[g][0], pk[g][1], pk[g + 1][0], pk[g + 1][1]};
                }
                __builtin_amdgcn_sched_barrier(0);             // (keeps the scheduler from piling up all 16 blocks' temporaries)
            }
        }
    }
    CW_TRACE();
    if ((a.Cout & 7) == 0) {
        const int ch = lane_e & 15, r0 = lane_e >> 4;
        const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, wa.out_bytes);
        const unsigned ocs2 = (unsigned)a.out_cs * 2u;
        const unsigned voff = n0 + ch * 8 < a.Cout ? (unsigned)((img * a.H + y0) * a.W + x0 + r0) * ocs2 + (unsigned)(n0 + ch * 8) * 2u : M3D_BUF_OOB;
        const unsigned char *l0 = wl + r0 * 256 + ((ch ^ r0) << 4), *l1 = wl + r0 * 256 + ((ch ^ (r0 + 4)) << 4);
#pragma unroll
        for (int p = 0; p < 32; ++p) {
            const u32x4 o = *reinterpret_cast<const u32x4 *>(((p & 1) ? l1 : l0) + p * 1024);
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((2 * (p >> 3) + ((p >> 2) & 1)) * a.W + 4 * (p & 3)) * ocs2;
            __builtin_amdgcn_raw_buffer_store_b128(o, rout, voff, soff, 0);
        }
    } else {
        const int H = a.H, W = a.W;
        store_otile<128, 128, 64>(a, wl, n0, 0, lane_e, [&](int row) {
            const int i = row >> 5, r = row & 31;
            return (img * H + y0 + 2 * i + (r >> 4)) * W + x0 + (r & 15);
        });
    }
    CW_TRACE();
}

// 1 if the kernel serves the descriptor (geometry / modes only; the caller passes fragment-ordered weights in d->wgt_wave)
int conv_wide_applicable(const m3d_conv_bf16_desc *d)
{
    if (!d->wgt_wave || d->dcn_offmask || d->kh != 3 || d->kw != 3 || d->stride != 1 || d->pad != 1) return 0;
    if (d->groups != 1 || d->wgt_img_stride || d->out_mode != 0 || d->sigmoid_from >= 0) return 0;
    if (d->Cin % 32 || d->Cin < 64 || d->Cout_pad % 128 || d->H % 8 || d->W % 16) return 0;
    // hand-counted vmcnt: a build in which the compiler spills (scratch loads / stores inside the K loop shift every wait) must not
    // run -- probed ONCE here, so that the engine routes the layer to the halo-tile kernel instead of failing at launch time
    // (launch_conv_wide keeps its REQUIRE as the backstop).  No device / probe failure = unknown = served.
    static std::once_flag once;
    static int on = 1;
    std::call_once(once, []() {
        const char *e = getenv("M3D_BF16_WIDE");
        on = e ? atoi(e) : 1;
#ifndef BF16_TRACE
        hipFuncAttributes fa, fb;
        if (on && hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&bf16_conv3x3_wide_kernel<true>)) == hipSuccess &&
            hipFuncGetAttributes(&fb, reinterpret_cast<const void *>(&bf16_conv3x3_wide_kernel<false>)) == hipSuccess) {
            if (fa.localSizeBytes + fb.localSizeBytes) on = 0;
        } else {
            (void)hipGetLastError();
        }
#endif
    });
    return on;
}

int launch_conv_wide(const Bf16Args &a, const m3d_conv_bf16_desc *d, hipStream_t st)
{
    WideArgs wa;
    wa.b = a;
    wa.b.res_bytes = d->res ? (unsigned)((long long)d->N * d->H * d->W * d->res_cs * 2) : 0u;
    wa.wfrag = d->wgt_wave;
    wa.ppx = d->W / 16; wa.ppy = d->H / 8; wa.npatch = d->N * wa.ppx * wa.ppy;
    const long long gb = (long long)(d->Cin / 32) * 18 * 4096;
    M3D_REQUIRE(gb < (1ll << 31), "conv_bf16 (wide): weight group too large");
    wa.wfrag_group_bytes = (unsigned)gb;
    const long long ob = (long long)d->N * d->H * d->W * d->out_cs * 2;
    M3D_REQUIRE(ob < (1ll << 31), "conv_bf16 (wide): output view must be < 2 GiB");
    wa.out_bytes = (unsigned)ob;
    static bool attr = false;
    if (!attr) {
        M3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_conv3x3_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * CW_WAVE_LDS));
        M3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bf16_conv3x3_wide_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * CW_WAVE_LDS));
        hipFuncAttributes fa, fb;
        M3D_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&bf16_conv3x3_wide_kernel<true>)));
        M3D_HIP(hipFuncGetAttributes(&fb, reinterpret_cast<const void *>(&bf16_conv3x3_wide_kernel<false>)));
        fa.localSizeBytes += fb.localSizeBytes;
        // hand-counted vmcnt: a spill (scratch load / store) inside the K loop would shift every wait
#ifndef BF16_TRACE
        M3D_REQUIRE(fa.localSizeBytes == 0, "conv_bf16 (wide): this build of the kernel spills %d bytes per lane", (int)fa.localSizeBytes);
#endif
        attr = true;
    }
    const dim3 grid(cdiv(wa.npatch, 4), d->Cout_pad / 128);
    if (d->res) hipLaunchKernelGGL(bf16_conv3x3_wide_kernel<true>, grid, dim3(256), 4 * CW_WAVE_LDS, st, wa);
    else hipLaunchKernelGGL(bf16_conv3x3_wide_kernel<false>, grid, dim3(256), 4 * CW_WAVE_LDS, st, wa);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
