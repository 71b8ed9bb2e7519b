// Winograd F(2x2, 3x3) convolution for gfx950 on the fp32 MFMA pipe: 3x3, stride 1, pad 1, NHWC, fp32 throughout.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile / 4x4 input tile, summed over input channels
//
// 16 multiplies per 4 outputs instead of 36: 2.25x fewer MFMA FLOPs than the direct implicit GEMM (igemm_conv.hip) --
// the same algebraic reformulation cuDNN applies to these layers in the reference (nn.Conv2d 3x3 in
// model/pose_dla_dcn.py:96-103 and the cls head).  The 16 transform positions xi = (a, b) are 16 independent GEMMs
// M[xi] (tiles x Cout) = V[xi] (tiles x Cin) . U[xi] (Cin x Cout).
//
// One workgroup = 64 output tiles (256 output pixels) x 32 output channels, 8 waves, wave-specialised:
//   * waves 4-7: one (tile, channel quad) per thread: 16 x 16-byte loads of the 4x4 input patch (offsets constant over
//     the whole K loop, SGPR channel base advances), B^T d B in registers, V[xi][tile][quad] to the idle LDS buffer;
//   * waves 0-3: each owns 4 of the 16 xi for all 64 tiles x 32 couts (8 accumulators);
//   * U = G g G^T is transformed offline (fp64 -> fp32) and packed in MFMA-fragment order, so every wave reads the B
//     fragments of ITS xi straight global->register (coalesced 1 KB loads, no LDS, no reuse lost);
//   * v_mfma_f32_32x32x2_f32 with the k-permuted fragment order (lane half h, step t -> k = 8g + 4h + t);
//   * the 16 accumulated M[xi] meet in LDS, A^T M A + affine/residual/LeakyReLU/sigmoid are applied per (tile, cout)
//     and written NHWC with lanes along channels.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

// Phase ablations (M3D_ABLATE, the probe tools) exist in the DIAGNOSTIC library only (make trace): as runtime flags they put uniform
// branches into the K loops of the product kernels.
#ifdef WINO_TRACE
#define WN_ABL(bit) (a.ablate & (bit))
#else
#define WN_ABL(bit) false
#endif

struct WinoArgs {
    const float *in;
    const float *U;          // [16][Cout_pad/32][Cin/8][64 lanes][4]
    float *out;
    const float *scale;
    const float *shift;
    const float *res;
    int in_cs, out_cs, res_cs;
    unsigned in_bytes;       // extent of the input view from `in` (buffer range check)
    unsigned out_bytes, res_bytes;
    int N, H, W, Cin, Cout, Cout_pad;
    int TH, TW, NT;          // tiles per image (rows, cols), total tiles
    int tiles_n;             // Cout_pad / 32
    int act, sigmoid_from, res_mode;
    int ablate;   // diagnostics (M3D_ABLATE): 1 = loader skips steady-state loads, 2 = no MFMA, 4 = loader skips transform+store, 8 = no U loads
    // split-K across waves (wave kernel only; layers with too few 32 x 32 tiles): grid = splits x tiles, split s covers k-steps
    // [s*ks_per, (s+1)*ks_per) and stores its share of the UNTRANSFORMED-BACK outputs A^T M A (linear in M) to
    // ws[s][N*H*W][Cout_pad]; m3d_launch_splitk_reduce adds them in split order and applies the epilogue
    float *ws;
    int splits, ks_per, base_waves;
    unsigned ws_bytes;
#ifdef WINO_TRACE
    long long *trace;   // [block][wave][64] s_memtime stamps (diagnostic build only, tools/wino_trace.py)
#endif
};

#ifdef WINO_TRACE
#define TRACE_INIT() long long *trp = a.trace ? a.trace + ((size_t)blockIdx.x * 8 + wave) * 128 : nullptr; int tri = 0
#define TRACE() do { if (trp && lane == 0 && tri < 126) { trp[tri++] = __builtin_readcyclecounter(); trp[tri == 1 ? 126 : 127] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define TRACE_INIT()
#define TRACE()
#endif

#define WINO_T 64            // tiles per workgroup (two 32-row MFMA tiles per compute wave)
#define WINO_BK 16           // input channels per k-step
#define WINO_VBUF (16 * WINO_T * WINO_BK)   // floats per V buffer: V[xi][tile][16], unpadded, XOR-swizzled quads
#define WINO_LDM 33

// LDS read whose completion is tracked BY HAND (hipcc does not count asm memory ops, cdna_hip_programming.md 5.7):
// lets the V fragments of the next MFMA group be in flight while the current group's MFMAs issue; the consumer
// runs `lds_wait()` first.  addr = LDS byte address.
__device__ __forceinline__ f32x4 lds_read_b128_async(unsigned addr)
{
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void lds_wait()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);      // MFMAs must not be hoisted above the wait (rule 18)
}

// Workgroup barrier that only waits for this wave's LDS traffic.  __syncthreads() would also emit s_waitcnt vmcnt(0)
// and drain the global prefetch loads issued just before it, serialising HBM/L2 latency into every k-step.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// 512 threads, wave-specialised: waves 0-3 compute (each owns 4 of the 16 xi, 64 tiles x 32 couts = 8 accumulators,
// U fragments straight from global and reused for both tile groups), waves 4-7 load the 4x4 input patches (one
// (tile, channel quad) per thread, 16 x 16-byte loads with constant offsets), apply B^T d B and fill the other V buffer.
// One barrier per k-step; per MFMA the CU moves ~3.4x fewer bytes through its vector-memory path than a
// 32-tile x 32-cout block where every wave both loads and computes.
__global__ __launch_bounds__(512, 2) void wino_kernel(const WinoArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // V[2][16][64][16]  then  M[16][64][33]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform -> SGPR, scalar branches
    const int l31 = lane & 31, hrow = 4 * (lane >> 5);
#ifdef WINO_LOADERS_FIRST
    const bool is_loader = wave < 4;
    const int cw = wave - 4;                  // compute-wave index
    const int lt0 = 0;
#else
    const bool is_loader = wave >= 4;
    const int cw = wave;
    const int lt0 = 256;
#endif
    TRACE_INIT();
    TRACE();

    int tile_blk;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        tile_blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = tile_blk / a.tiles_n, bn = tile_blk - bm * a.tiles_n;
    const int t0 = bm * WINO_T, n0 = bn * 32;
    const int KS = a.Cin / WINO_BK;

    // V element (xi, tile, quad q) lives at ((xi*64 + tile)*16 + 4*(q ^ ((tile >> 2) & 3))): 64-byte rows without
    // padding; the XOR spreads the 16 tiles of a ds_read_b128 lane group over all 64 banks (conflict-free)
    if (is_loader) {
        // ================================ loader / input-transform waves ======================================
        if (WN_ABL(16)) __builtin_amdgcn_s_setprio(3);
        const int lt = tid - lt0;
        const int ltile = lt >> 2, lq = lt & 3;
        const int wq = (lq ^ ((ltile >> 2) & 3)) * 4;
        // byte offsets of the 4x4 patch; positions outside the image get the out-of-range marker and read as 0.0f
        unsigned poff[16];
        {
            const int t = t0 + ltile;
            const bool tv = t < a.NT;
            const int tt = tv ? t : 0;
            const int n = tt / (a.TH * a.TW), rem = tt - n * a.TH * a.TW;
            const int ty = rem / a.TW, tx = rem - ty * a.TW;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int hi = 2 * ty - 1 + r, wi = 2 * tx - 1 + c;
                    const bool ok = tv && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                    poff[r * 4 + c] = ok ? ((unsigned)((n * a.H + hi) * a.W + wi) * (unsigned)a.in_cs + (unsigned)(lq * 4)) * 4u
                                         : M3D_BUF_OOB;
                }
        }
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
        // two patches in flight (these waves hold no accumulators, registers are plentiful): the loads of step ks+2
        // and ks+3 are outstanding while step ks computes, so HBM/L2 latency never reaches the barrier
        f32x4 dA[16], dB[16];
        auto load_patch = [&](int ks, f32x4 (&d)[16]) {
            const unsigned soff = (unsigned)(ks * WINO_BK) * 4u;
#pragma unroll
            for (int i = 0; i < 16; ++i) d[i] = buf_load_f32x4(rin, poff[i], soff);
        };
        auto transform_store = [&](int buf, f32x4 (&d)[16]) {
#ifdef WINO_TRACE
            if (WN_ABL(64)) {               // diagnostics: LDS writes only (no transform VALU)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *vb = smem + buf * WINO_VBUF + ((r * 4) * WINO_T + ltile) * WINO_BK + wq;
#pragma unroll
                    for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4 *>(vb + c * WINO_T * WINO_BK) = d[r * 4 + c];
                }
                return;
            }
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) {      // V row r: (B^T d)[r] per column, then (.) B along the columns
                f32x4 t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (r == 0) t[c] = pk_sub(d[0 * 4 + c], d[2 * 4 + c]);
                    else if (r == 1) t[c] = pk_add(d[1 * 4 + c], d[2 * 4 + c]);
                    else if (r == 2) t[c] = pk_sub(d[2 * 4 + c], d[1 * 4 + c]);
                    else t[c] = pk_sub(d[1 * 4 + c], d[3 * 4 + c]);
                }
                float *vb = smem + buf * WINO_VBUF + ((r * 4) * WINO_T + ltile) * WINO_BK + wq;
#ifdef WINO_TRACE
                if (WN_ABL(128)) {          // diagnostics: transform VALU only (results kept alive, no LDS writes)
                    asm volatile("" :: "v"(pk_sub(t[0], t[2])), "v"(pk_add(t[1], t[2])), "v"(pk_sub(t[2], t[1])), "v"(pk_sub(t[1], t[3])));
                    continue;
                }
#endif
                *reinterpret_cast<f32x4 *>(vb) = pk_sub(t[0], t[2]);
                *reinterpret_cast<f32x4 *>(vb + 1 * WINO_T * WINO_BK) = pk_add(t[1], t[2]);
                *reinterpret_cast<f32x4 *>(vb + 2 * WINO_T * WINO_BK) = pk_sub(t[2], t[1]);
                *reinterpret_cast<f32x4 *>(vb + 3 * WINO_T * WINO_BK) = pk_sub(t[1], t[3]);
            }
        };
        load_patch(0, dA);
        if (KS > 1) load_patch(1, dB);
        transform_store(0, dA);
        if (KS > 2) load_patch(2, dA);
        TRACE();
        lds_barrier();                                   // V(0) visible
        TRACE();
        for (int ks = 0; ks < KS; ks += 2) {
            // step ks: produce V(ks+1) from dB, refill dB with patch ks+3
            if (ks + 1 < KS) {
                if (!WN_ABL(4)) transform_store((ks + 1) & 1, dB);   // buffer last read in step ks-1 (barrier passed)
                TRACE();
                if (ks + 3 < KS && !WN_ABL(1)) load_patch(ks + 3, dB);
            }
            TRACE();
            lds_barrier();
            TRACE();
            // step ks+1: produce V(ks+2) from dA, refill dA with patch ks+4
            if (ks + 1 < KS) {
                if (ks + 2 < KS) {
                    if (!WN_ABL(4)) transform_store((ks + 2) & 1, dA);
                    TRACE();
                    if (ks + 4 < KS && !WN_ABL(1)) load_patch(ks + 4, dA);
                }
                TRACE();
                lds_barrier();
                TRACE();
            }
        }
    } else {
        // ======================================= compute waves ================================================
        if (WN_ABL(32)) __builtin_amdgcn_s_setprio(3);
        const int kgroups = a.Cin / 8;
        // two U-fragment register sets used alternately (loop unrolled by two, no copies): the loads of step ks+1 are
        // issued before the MFMAs of step ks and only waited for one full step later
        f32x4 fbA[4][2], fbB[4][2];
        // U fragment (xi, k-group G) of this cout block sits at ub + xi_local*xstride + G*256 floats (+ lane*4):
        // a uniform SGPR base plus a constant 32-bit lane offset -> one instruction per load, no per-load VALU math
        const unsigned xstride = (unsigned)(a.tiles_n * kgroups) * 1024u;          // bytes between consecutive xi
        const float *ub = a.U + ((size_t)(cw * 4) * a.tiles_n + bn) * kgroups * 256;
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(ub, 4u * xstride);
        unsigned uoff[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) uoff[x] = (unsigned)lane * 16u + x * xstride;
        auto load_u = [&](int ks, f32x4 (&dst)[4][2]) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int g = 0; g < 2; ++g) dst[x][g] = buf_load_f32x4(ru, uoff[x], (unsigned)(ks * 2 + g) * 1024u);
        };
        f32x16 acc[4][2];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][m][r] = 0.f;
        const int rq0 = ((lane >> 5) ^ ((l31 >> 2) & 3)) * 4;          // swizzled quad of k-group 0 (q = h)
        const int rq1 = ((2 + (lane >> 5)) ^ ((l31 >> 2) & 3)) * 4;    // k-group 1 (q = 2 + h)
        const unsigned lds0 = (unsigned)(uintptr_t)smem;   // LDS byte address of the dynamic segment
        auto compute = [&](int buf, f32x4 (&fb)[4][2]) {
            // 8 groups (xi x, k-group g) of 8 MFMAs; the V fragments of group i+1 are requested from LDS (hand-counted
            // asm reads) before the MFMAs of group i issue and waited for after them: the single compute wave of a
            // SIMD never stalls on LDS latency
            const unsigned vbase = lds0 + (unsigned)(buf * WINO_VBUF + (cw * 4 * WINO_T + l31) * WINO_BK) * 4u;
            f32x4 fa[2][2];                                // [parity][m]
            fa[0][0] = lds_read_b128_async(vbase + rq0 * 4);
            fa[0][1] = lds_read_b128_async(vbase + (32 * WINO_BK + rq0) * 4);
            lds_wait();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int x = i >> 1, g = i & 1, cur = i & 1, nxt = cur ^ 1;
                if (i + 1 < 8) {
                    const int xn = (i + 1) >> 1, gn = (i + 1) & 1;
                    const unsigned vb = vbase + (unsigned)(xn * WINO_T * WINO_BK + (gn ? rq1 : rq0)) * 4u;
                    fa[nxt][0] = lds_read_b128_async(vb);
                    fa[nxt][1] = lds_read_b128_async(vb + 32 * WINO_BK * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[x][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][m][s], fb[x][g][s], acc[x][m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                lds_wait();
            }
        };
        // NOTE: the prefetch loads are UNCONDITIONAL (index clamped on the last step).  With a conditional load the
        // compiler cannot know how many loads are in flight and emits pessimistic s_waitcnt vmcnt(N) that also wait
        // for the loads just issued -- which serialises the full memory latency into every k-step.
        load_u(0, fbA);
        TRACE();
        lds_barrier();                                     // V(0) visible
        TRACE();
        for (int ks = 0; ks < KS; ks += 2) {
            if (!WN_ABL(8)) load_u(min(ks + 1, KS - 1), fbB);
            __builtin_amdgcn_sched_barrier(0);             // keep the loads ahead of the MFMAs (the scheduler sinks them)
            if (!WN_ABL(2)) compute(0, fbA);
            TRACE();
            lds_barrier();
            TRACE();
            if (ks + 1 < KS) {
                if (!WN_ABL(8)) load_u(min(ks + 2, KS - 1), fbA);
                __builtin_amdgcn_sched_barrier(0);
                if (!WN_ABL(2)) compute(1, fbB);
                TRACE();
                lds_barrier();
                TRACE();
            }
        }
        // ---- gather the 16 M[xi] in LDS: M[xi][tile][cout] (the loop ended with a barrier: V is dead) -----------
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float *mb = smem + ((size_t)(cw * 4 + x) * WINO_T + m * 32) * WINO_LDM + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) mb[((r & 3) + 8 * (r >> 2) + hrow) * WINO_LDM] = acc[x][m][r];
            }
    }
    TRACE();
    __syncthreads();
    TRACE();

    // ---- A^T M A + epilogue: thread = (tile, cout), all 512 threads -------------------------------------------
    const int co = n0 + (tid & 31);
    const bool cok = co < a.Cout;
    const float sc = (cok && a.scale) ? a.scale[co] : 1.f;
    const float sh = (cok && a.shift) ? a.shift[co] : 0.f;
    const bool sg = a.sigmoid_from >= 0 && co >= a.sigmoid_from;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tl = (tid >> 5) + 16 * q;
        const int t = t0 + tl;
        if (t >= a.NT || !cok) continue;
        float m[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) m[x] = smem[((size_t)x * WINO_T + tl) * WINO_LDM + (tid & 31)];
        float s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s0[b] = m[0 * 4 + b] + m[1 * 4 + b] + m[2 * 4 + b];
            s1[b] = m[1 * 4 + b] - m[2 * 4 + b] - m[3 * 4 + b];
        }
        float y[4];
        y[0] = s0[0] + s0[1] + s0[2];
        y[1] = s0[1] - s0[2] - s0[3];
        y[2] = s1[0] + s1[1] + s1[2];
        y[3] = s1[1] - s1[2] - s1[3];
        const int n = t / (a.TH * a.TW), rem = t - n * a.TH * a.TW;
        const int ty = rem / a.TW, tx = rem - ty * a.TW;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const size_t pix = (size_t)(n * a.H + 2 * ty + i) * a.W + 2 * tx + j;
                float v = y[i * 2 + j];
                if (a.res) {
                    const float rv = a.res[pix * a.res_cs + co];
                    v = a.res_mode ? (v + rv) * sc + sh : v * sc + sh + rv;
                } else {
                    v = v * sc + sh;
                }
                if (sg) v = sigmoidf_(v);
                else if (a.act == 1) v = leaky(v);
                a.out[pix * a.out_cs + co] = v;
            }
    }
    TRACE();
}

#ifdef WINO_TRACE
static long long *g_wino_trace = nullptr;
extern "C" void m3d_wino_set_trace(void *buf) { g_wino_trace = (long long *)buf; }
#endif


// ======================================================================================================================
// Register-resident variant: ONE WAVE = 32 tiles x 32 couts x all 16 xi, no LDS, no barriers.
//
// Why: on gfx950 nothing a second wave does on a SIMD hides under that SIMD's MFMA stream -- VALU instructions of the
// partner wave cost the stream ~11 cycles each (tools/ubench/mfma_side_cost.hip) and the s_memtime timeline of the
// wave-specialised kernel above (tools/wino_trace.py) shows its loader wave finishing exactly when the MFMA burst ends,
// then a ~900-cycle tail (LDS-write drain, load issue, barrier) per k-step: 6300 cycles per 4096 cycles of MFMA.  With the
// whole 512-register file (256 VGPR + 256 AGPR) one wave holds the 16 accumulators of a 32x32 tile itself, so
//   * lane (tile i = lane&31, half h = lane>>5) loads the 4x4 patch of ITS tile for channels 8s+4h..+3 (16 x 16 B),
//     runs B^T d B on it (64 v_pk_add_f32) and the result IS the A operand of the next 64 MFMAs (k = 8s + 4h + t);
//   * the B operand is the same fragment-packed U as above, one 16-byte load per xi and k-step, re-issued right after the
//     MFMAs that consumed the previous one;
//   * all 16 M[xi] of a (tile, cout) sit in the same lane: A^T M A and the epilogue run in registers.
// Waves are independent units (grid = tile groups x cout groups, 64 threads each): the hardware balances them per SIMD.
// Buffer load whose completion is counted BY HAND: hipcc joins the waits of a loop-carried prefetch into one
// s_waitcnt vmcnt(0) at the loop header, which would expose the latency of the loads issued last.  Issue order per k-step is
// fixed (16 patch loads, then 4 x 4 U loads), so the counts are static: see the s_waitcnt comments in the loop.
__device__ __forceinline__ void buf_load_async(f32x4 &dst, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(r), "s"(soff) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm4(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d)
{
    // the registers are operands so that no use of them can be scheduled above the wait
    if constexpr (N == 28) asm volatile("s_waitcnt vmcnt(28)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// PART = split-K form: the wave covers k-steps [ks0, ks1) and stores raw partial outputs to the workspace
template <bool PART>
__global__ __launch_bounds__(64) void wino_wave_kernel(const WinoArgs a)
{
    __shared__ __attribute__((aligned(16))) int pixb[32];   // first output pixel of each tile (-1: no such tile)
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
#ifdef WINO_TRACE
    const int wave = 0;
#endif
    TRACE_INIT();
    TRACE();
    int blk;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int split = 0;
    if constexpr (PART) {
        split = blk / a.base_waves;
        blk -= split * a.base_waves;
    }
    const int bm = blk / a.tiles_n, bn = blk - bm * a.tiles_n;
    const int t0 = bm * 32, n0 = bn * 32;
    const int KS = a.Cin / 8;
    const int ks0 = PART ? split * a.ks_per : 0, ks1 = PART ? min(KS, ks0 + a.ks_per) : KS;   // this wave's k-steps

    unsigned poff[16];
    {
        const int t = t0 + l31;
        const bool tv = t < a.NT;
        const int tt = tv ? t : 0;
        const int n = tt / (a.TH * a.TW), rem = tt - n * a.TH * a.TW;
        const int ty = rem / a.TW, tx = rem - ty * a.TW;
        if (h == 0) pixb[l31] = tv ? (n * a.H + 2 * ty) * a.W + 2 * tx : -1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int hi = 2 * ty - 1 + r, wi = 2 * tx - 1 + c;
                const bool ok = tv && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                poff[r * 4 + c] = ok ? ((unsigned)((n * a.H + hi) * a.W + wi) * (unsigned)a.in_cs + (unsigned)(h * 4)) * 4u
                                     : M3D_BUF_OOB;
            }
    }
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const unsigned xstride = (unsigned)(a.tiles_n * KS) * 1024u;               // bytes between consecutive xi
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.U + (size_t)bn * KS * 256, 16u * xstride);
    const unsigned ulane = (unsigned)lane * 16u;

    f32x4 d[16], V[16], Uf[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) buf_load_async(d[i], rin, poff[i], (unsigned)ks0 * 32u);
#pragma unroll
    for (int x = 0; x < 16; ++x) buf_load_async(Uf[x], ru, ulane, x * xstride + (unsigned)ks0 * 1024u);
    __builtin_amdgcn_sched_barrier(0);        // the accumulators are cleared while the first loads are in flight
    f32x16 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    TRACE();

    // One k-step (8 input channels).  LAST = no prefetch: the wait counts change, nothing is left in flight at the end.
    auto step = [&](int s, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        TRACE();
        // outstanding, oldest first: patch(s) x16, U(s) x16  ->  the patch has landed when <= 16 remain
        asm volatile("s_waitcnt vmcnt(16)"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]),
                       "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]));
        // ---- B^T d B in registers ------------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x4 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (r == 0) t[c] = pk_sub(d[0 * 4 + c], d[2 * 4 + c]);
                else if (r == 1) t[c] = pk_add(d[1 * 4 + c], d[2 * 4 + c]);
                else if (r == 2) t[c] = pk_sub(d[2 * 4 + c], d[1 * 4 + c]);
                else t[c] = pk_sub(d[1 * 4 + c], d[3 * 4 + c]);
            }
            V[r * 4 + 0] = pk_sub(t[0], t[2]);
            V[r * 4 + 1] = pk_add(t[1], t[2]);
            V[r * 4 + 2] = pk_sub(t[2], t[1]);
            V[r * 4 + 3] = pk_sub(t[1], t[3]);
        }
        if constexpr (!LAST) {
            const unsigned soff = (unsigned)(s + 1) * 32u;
#pragma unroll
            for (int i = 0; i < 16; ++i) buf_load_async(d[i], rin, poff[i], soff);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- 64 MFMAs; the U fragment of each xi is re-loaded for the next step as soon as its MFMAs are issued ------
        const unsigned usn = (unsigned)(s + 1) * 1024u;
        auto group = [&](auto gtag) {
            constexpr int g = decltype(gtag)::value;
            // outstanding: U(s) groups g..3, then (if not LAST) patch(s+1) x16 and U(s+1) groups 0..g-1  = 32
            if constexpr (!LAST) wait_vm4<28>(Uf[4 * g], Uf[4 * g + 1], Uf[4 * g + 2], Uf[4 * g + 3]);
            else wait_vm4<12 - 4 * g>(Uf[4 * g], Uf[4 * g + 1], Uf[4 * g + 2], Uf[4 * g + 3]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int x = 4 * g; x < 4 * g + 4; ++x)
                    acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[x][t], Uf[x][t], acc[x], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!LAST) {
#pragma unroll
                for (int x = 4 * g; x < 4 * g + 4; ++x) buf_load_async(Uf[x], ru, ulane, x * xstride + usn);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
    };
    for (int s = ks0; s + 1 < ks1; ++s) step(s, std::false_type{});
    step(ks1 - 1, std::true_type{});
    TRACE();

    // ---- A^T M A + epilogue in registers: lane = cout n0 + l31, tiles t0 + 4h + {0..3, 8..11, 16..19, 24..27} -------------
    const int co = n0 + l31;
    const bool cok = co < a.Cout;
    const float sc = (cok && a.scale) ? a.scale[co] : 1.f;
    const float sh = (cok && a.shift) ? a.shift[co] : 0.f;
    const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res ? a.res : a.out, a.res ? a.res_bytes : 0u);
    // byte offsets of the four output pixels of a tile relative to its first one ride in the SGPR offset of the store
    const unsigned ocs4 = (unsigned)a.out_cs * 4u, rcs4 = (unsigned)a.res_cs * 4u;
    // split-K form: the same stores go to the partial-sum workspace [split][pixel][Cout_pad] instead (every padded channel is
    // written, the reduce launch skips co >= Cout); one offset array serves both so that the register budget is unchanged
    constexpr bool part = PART;
    const __amdgpu_buffer_rsrc_t rdst = part ? make_rsrc(a.ws, a.ws_bytes) : rout;
    const unsigned dcs = part ? (unsigned)a.Cout_pad : (unsigned)a.out_cs;
    const unsigned dcs4 = dcs * 4u, sbase = part ? (unsigned)split * (unsigned)(a.N * a.H * a.W) : 0u;
    unsigned obase[16];
    float rv[16][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int4 pb = *reinterpret_cast<const int4 *>(&pixb[4 * h + 8 * q]);
        const int pbv[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = cok && pbv[k] >= 0;
            obase[4 * q + k] = ((part || cok) && pbv[k] >= 0) ? ((sbase + (unsigned)pbv[k]) * dcs + (unsigned)co) * 4u : M3D_BUF_OOB;
            if (!PART && a.res) {              // all residual loads are in flight before the arithmetic starts
                const unsigned rb = ok ? ((unsigned)pbv[k] * (unsigned)a.res_cs + (unsigned)co) * 4u : M3D_BUF_OOB;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        rv[4 * q + k][i * 2 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            rres, rb, (unsigned)(i * a.W + j) * rcs4, 0));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {          // two tiles at a time on the packed-fp32 VALU
        f32x2 m[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) m[x] = f32x2{acc[x][r], acc[x][r + 1]};
        f32x2 s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s0[b] = m[0 * 4 + b] + m[1 * 4 + b] + m[2 * 4 + b];
            s1[b] = pk_sub2(pk_sub2(m[1 * 4 + b], m[2 * 4 + b]), m[3 * 4 + b]);
        }
        f32x2 y[4];
        y[0] = s0[0] + s0[1] + s0[2];
        y[1] = pk_sub2(pk_sub2(s0[1], s0[2]), s0[3]);
        y[2] = s1[0] + s1[1] + s1[2];
        y[3] = pk_sub2(pk_sub2(s1[1], s1[2]), s1[3]);
        if constexpr (PART) {                  // raw partial outputs; the reduce launch owns residual / affine / activation
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float pv = y[k][e];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv), rdst, obase[r + e],
                                                          (unsigned)((k >> 1) * a.W + (k & 1)) * dcs4, 0);
                }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x2 v = y[k];
            if (a.res) {
                const f32x2 r2 = {rv[r][k], rv[r + 1][k]};
                if (a.res_mode) v = __builtin_elementwise_fma(v + r2, sc2, sh2);
                else v = __builtin_elementwise_fma(v, sc2, sh2) + r2;
            } else {
                v = __builtin_elementwise_fma(v, sc2, sh2);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float o = v[e];
                if (a.act == 1) o = fmaxf(o, o * M3D_LEAKY_SLOPE);    // == leaky(o) for 0 < slope < 1
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rout, obase[r + e],
                                                      (unsigned)((k >> 1) * a.W + (k & 1)) * ocs4, 0);
            }
        }
    }
    TRACE();
}

// Kernel choice and split-K plan.  Returns 1 for the register-resident wave kernel (with *splits >= 1), 0 for the LDS kernel.
// The wave kernel needs enough waves for the 1024 SIMDs: thin layers (the 12x40 maps, the 27-channel offset/mask convs) are
// split along K across waves when the caller provides a workspace; the sigmoid epilogue only exists in the split path (the
// reduce launch applies it).  Tuning knobs (experiments only): M3D_WINO_VARIANT=0 forces the LDS kernel, M3D_WINO_WAVE_MIN the
// wave threshold, M3D_WINO_SPLITK=1 recommends the split form (off: it measured slower than the LDS kernel).
static int wino_plan(const m3d_conv_desc *d, bool have_ws, int *splits, int *ks_per)
{
    static int wave_min = -1, variant = -1, splitk = -1;
    if (wave_min < 0) { const char *e = getenv("M3D_WINO_WAVE_MIN"); wave_min = e ? atoi(e) : 800; }
    if (variant < 0) { const char *e = getenv("M3D_WINO_VARIANT"); variant = e ? atoi(e) : 1; }
    if (splitk < 0) { const char *e = getenv("M3D_WINO_SPLITK"); splitk = e ? atoi(e) : 0; }
    const int KS = d->Cin / 8;
    *splits = 1;
    *ks_per = KS;
    if (variant != 1) return 0;
    const long long nt = (long long)d->N * (d->H / 2) * (d->W / 2);
    const long long base = ((nt + 31) / 32) * (d->Cout_pad / 32);
    if (base < wave_min && have_ws) {
        int s = (int)((1600 + base - 1) / base);
        if (s > KS / 8) s = KS / 8;             // at least 8 k-steps (64 channels) per split
        if (s > 4) s = 4;
        if (s >= 2) {
            *ks_per = (KS + s - 1) / s;
            *splits = (KS + *ks_per - 1) / *ks_per;
        }
    }
    // (splits / ks_per stay as computed: a caller that forces the wave kernel uses them even below the fill threshold)
    // measured (bs=8): the split form loses to the LDS kernel on the layers it would apply to (level5 0.108 vs 0.096 ms, the
    // 27-channel offset/mask convs 0.045 vs 0.035 ms), so it is only recommended when M3D_WINO_SPLITK=1
    if (*splits > 1 && !splitk) return 0;
    return !(base * *splits < wave_min || (d->sigmoid_from >= 0 && *splits <= 1));
}

extern "C" int m3d_wino_conv3x3_variant(const m3d_conv_desc *d)
{
    int s, k;
    return d ? wino_plan(d, d->splitk_ws != nullptr, &s, &k) : 0;
}

// Split-K plan of the wave kernel computed AS IF a workspace were given: *splits and the bytes to pass through splitk_ws.
extern "C" int m3d_wino_conv3x3_splitk_plan(const m3d_conv_desc *d, int *splits, long long *ws_bytes)
{
    M3D_REQUIRE(d && splits && ws_bytes, "wino_splitk_plan: null pointer");
    int k;
    if (!wino_plan(d, true, splits, &k)) *splits = 1;            // not recommended: no split, the LDS kernel runs
    *ws_bytes = *splits > 1 ? (long long)*splits * d->N * d->H * d->W * d->Cout_pad * 4 : 0;
    return M3D_OK;
}

extern "C" int m3d_wino_conv3x3_forward_ex(const m3d_conv_desc *d, int variant, m3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    M3D_REQUIRE(d && d->in && d->wgt && d->out, "wino: null pointer");
    M3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1, "wino: 3x3 stride 1 pad 1 only");
    M3D_REQUIRE(d->Cin % 16 == 0 && d->Cout_pad % 32 == 0 && d->Cout <= d->Cout_pad, "wino: Cin %% 16, Cout_pad %% 32");
    M3D_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0 && d->Ho == d->H && d->Wo == d->W, "wino: even H and W");
    M3D_REQUIRE(!d->out_nchw && !d->dcn_offmask && !d->wgt_img_stride, "wino: NHWC output, plain conv, shared weights");
    M3D_REQUIRE(d->in_cs % 4 == 0 && ((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->wgt & 15) == 0, "wino: alignment");
    M3D_REQUIRE((long long)d->N * d->H * d->W * d->in_cs * 4 < (1ll << 31), "wino: input view must be < 2 GiB");
    M3D_REQUIRE((long long)16 * d->Cout_pad * d->Cin * 4 < (1ll << 31), "wino: transformed weights must be < 2 GiB");
    WinoArgs a;
    a.in = d->in; a.U = d->wgt; a.out = d->out; a.scale = d->scale; a.shift = d->shift; a.res = d->res;
    a.in_cs = d->in_cs; a.out_cs = d->out_cs; a.res_cs = d->res_cs;
    a.in_bytes = (unsigned)((long long)d->N * d->H * d->W * d->in_cs * 4);
    a.out_bytes = (unsigned)((long long)d->N * d->H * d->W * d->out_cs * 4);
    a.res_bytes = (unsigned)((long long)d->N * d->H * d->W * d->res_cs * 4);
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cout_pad = d->Cout_pad;
    a.TH = d->H / 2; a.TW = d->W / 2; a.NT = d->N * a.TH * a.TW; a.tiles_n = d->Cout_pad / 32;
    a.act = d->act; a.sigmoid_from = d->sigmoid_from; a.res_mode = d->res_mode;
    {
        static int abl = -1;
        if (abl < 0) { const char *e = getenv("M3D_ABLATE"); abl = e ? atoi(e) : 0; }
        a.ablate = abl;
    }
#ifdef WINO_TRACE
    a.trace = g_wino_trace;
#endif
    constexpr size_t smem = (size_t)16 * WINO_T * WINO_LDM * sizeof(float);   // 135,168 B (>= 2 V buffers: 131,072 B)
    static bool attr_set = false;
    if (!attr_set) {
        M3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
        attr_set = true;
    }
    M3D_REQUIRE(variant >= -1 && variant <= 1, "wino: variant must be -1 (auto), 0 (LDS kernel) or 1 (wave kernel)");
    int splits = 1, ks_per = d->Cin / 8;
    const int planned = wino_plan(d, d->splitk_ws != nullptr, &splits, &ks_per);
    if (variant == 0 || (variant < 0 && !planned)) { splits = 1; ks_per = d->Cin / 8; }
    M3D_REQUIRE(variant != 1 || d->sigmoid_from < 0 || splits > 1,
                "wino: the wave kernel applies the sigmoid epilogue only in its split-K form (needs a workspace)");
    a.ws = nullptr; a.splits = 1; a.ks_per = ks_per; a.base_waves = cdiv(a.NT, 32) * a.tiles_n; a.ws_bytes = 0;
    if (variant == 1 || (variant < 0 && planned)) {
        if (splits > 1) {
            const long long need = (long long)splits * d->N * d->H * d->W * d->Cout_pad * 4;
            M3D_REQUIRE(need <= d->splitk_ws_bytes && need < (1ll << 31) && ((uintptr_t)d->splitk_ws & 15) == 0,
                        "wino: split-K workspace too small (%lld bytes, see m3d_wino_conv3x3_splitk_plan), >= 2 GiB or misaligned",
                        d->splitk_ws_bytes);
            a.ws = d->splitk_ws; a.splits = splits; a.ws_bytes = (unsigned)need;
        }
        M3D_REQUIRE((long long)d->N * d->H * d->W * d->out_cs * 4 < (1ll << 31) &&
                    (long long)d->N * d->H * d->W * d->res_cs * 4 < (1ll << 31), "wino: output / residual views must be < 2 GiB");
        M3D_REQUIRE(d->Cin % 8 == 0, "wino: Cin %% 8");
        if (a.splits > 1) hipLaunchKernelGGL(wino_wave_kernel<true>, dim3(a.base_waves * a.splits), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL(wino_wave_kernel<false>, dim3(a.base_waves), dim3(64), 0, stream, a);
        M3D_LAUNCH_CHECK();
        if (a.splits > 1) {
            SplitkReduceArgs r;
            r.ws = a.ws; r.scale = d->scale; r.shift = d->shift; r.res = d->res; r.out = d->out;
            r.M = d->N * d->H * d->W; r.Cout = d->Cout; r.Cout_pad = d->Cout_pad; r.splits = a.splits; r.out_cs = d->out_cs;
            r.res_cs = d->res_cs; r.res_mode = d->res_mode; r.act = d->act; r.sigmoid_from = d->sigmoid_from;
            return m3d_launch_splitk_reduce(r, stream);
        }
        return M3D_OK;
    }
    const int grid = cdiv(a.NT, WINO_T) * a.tiles_n;
    hipLaunchKernelGGL(wino_kernel, dim3(grid), dim3(512), smem, stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_wino_conv3x3_forward(const m3d_conv_desc *d, m3d_stream_t stream)
{
    return m3d_wino_conv3x3_forward_ex(d, -1, stream);
}
