// Fused RPN head: the per-pixel MLP  1x1 Cin->256 (+BN+LeakyReLU) -> 1x1 256->256 (+BN+LeakyReLU) -> 1x1 256->Cout
// of model/M3d_inference_align.py:77-210 as ONE kernel per head (the reference runs 3 convs + 2 BN + 2 LeakyReLU
// launches per head and round-trips two 256-channel intermediates, 7.9 MB each per image, through HBM).
//
// One workgroup owns 64 pixels and ALL 256 hidden channels, so the hidden activations never leave the CU:
//   * the activation tile lives in LDS as act[64][256+4] (fp32, 65 KB -> two workgroups per CU) and is rewritten
//     in place between layers (layer output sits in the MFMA accumulators until every wave has finished reading);
//   * each wave owns its own 64 output channels, so a weight tile has NO reuse inside the workgroup: weights are
//     pre-packed in MFMA-fragment order ([row-tile][k-group][lane][4], see Engine/pack_frag) and every wave
//     loads its B fragments straight global->register -- one fully coalesced 1 KB load per (row-tile, k-group) --
//     double-buffered one k-tile ahead across layer boundaries.  No weight staging in LDS, no barrier inside a layer;
//   * math: v_mfma_f32_32x32x2_f32 with the k-permuted fragment order of igemm_conv.hip (lane half h, step t ->
//     k = 8g + 4h + t); the last layer swaps operands so lanes run along pixels and writes the planar [Cout][HW]
//     layout the RPN outputs need (lib/rpn_util.py:892-901 row order).
// Arithmetic per output element is the same k-ordered fp32 fma chain as the unfused igemm path.
#include <type_traits>

#include "common.h"

// Phase ablations (M3D_ABLATE_MLP, tools/head_probe.py) exist in the DIAGNOSTIC library only (make trace: -DHEAD_TRACE).  As a runtime
// flag in the product kernel they put a conditional branch behind every MFMA of the output layer and around every LDS store of the
// hidden-layer epilogues (round 6, tools/loop_isa_mix.py: 50 single-MFMA basic blocks with 16 v_accvgpr_write each).
#ifdef HEAD_TRACE
#define MLP_ABL(bit) (a.ablate & (bit))
#else
#define MLP_ABL(bit) false
#endif

struct MlpArgs {
    const float *in;
    const float *w[3];       // fragment-packed; w[0] may be null (2-layer form: input is the first hidden)
    const float *scale[3];
    const float *shift[3];
    float *out;
    long long out_img_stride;
    int in_cs, M, HW, Cin, Cout, Cout_pad;
    int ablate;   // diagnostics (M3D_ABLATE): 1 = no weight loads, 2 = no MFMA, 4 = no input staging, 8 = no LDS epilogue writes
};

#define MLP_BM 64
#define MLP_H 256
#define MLP_LDA (MLP_H + 4)
#define MLP_BK 32

#define MLP_MAX_HEADS 16
#ifdef HEAD_TRACE
static long long *g_head_trace = nullptr;
extern "C" void m3d_head_set_trace(void *buf) { g_head_trace = (long long *)buf; }
#define TRACE_INIT() long long *trp = batch.trace ? batch.trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 64 : nullptr; int tri = 0
#define TRACE() do { if (trp && lane == 0 && tri < 64) trp[tri++] = __builtin_readcyclecounter(); } while (0)
#else
#define TRACE_INIT()
#define TRACE()
#endif

struct MlpBatch {
#ifdef HEAD_TRACE
    long long *trace;
#endif
    MlpArgs head[MLP_MAX_HEADS];                      // blockIdx.y selects the head (same M for all of them)
};

template <bool HAS_L1, int N3>
__global__ __launch_bounds__(256) void head_mlp_kernel(const MlpBatch batch)
{
    const MlpArgs &a = batch.head[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) float act[];   // [64][260]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh4 = (lane >> 5) * 4, hrow = 4 * (lane >> 5);
    const int m0 = blockIdx.x * MLP_BM;
    TRACE_INIT();
    TRACE();
    const int kt1 = HAS_L1 ? a.Cin / MLP_BK : 0, kt2 = MLP_H / MLP_BK, kt3 = MLP_H / MLP_BK;
    const int n_tiles = kt1 + kt2 + kt3;
    // output-layer tiling: N3 = 64 -> waves 2 (px) x 2 (ch) of 32x32;  N3 = 256 -> like the hidden layers
    constexpr int TNo = (N3 == 64) ? 1 : 2;

    // ---- B-fragment stream: tile t of the concatenated layers -> 2 row-tiles x 4 k-groups per wave --------
    // packed weight element W[J*32 + l31][G*8 + h*4 + 0..3] sits at ((J*(K/8) + G)*64 + lane)*4
    // two register sets used alternately (k loops unrolled by two, every layer starts on an even stream position).
    // The prefetch is UNCONDITIONAL (stream position clamped at the end): with a conditional load or a set-to-set copy
    // hipcc emits s_waitcnt vmcnt(0) for the loads just issued, serialising the memory latency into every k-tile.
    // The loads are asm with HAND-COUNTED waits (as in wino_conv.hip): hipcc's own accounting of this loop-carried prefetch
    // put `s_waitcnt vmcnt(0)` in front of the MFMAs of every second hidden-layer tile and of every output-layer tile -- i.e.
    // behind the loads issued a few instructions earlier: one exposed memory round trip per tile.  Every call issues exactly 8
    // loads (the 64-channel output layer re-reads its row tile for the second slot), so "the previous set has landed" is
    // always vmcnt(8).
    f32x4 fbA[2][4], fbB[2][4];
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto load_frags = [&](int t_req, f32x4 (&dst)[2][4]) {
        const int t = min(t_req, n_tiles - 1);
        const float *w;
        int kgroups, kt, j0, nj;
        if (HAS_L1 && t < kt1) { w = a.w[0]; kgroups = a.Cin / 8; kt = t; j0 = swave * 2; nj = 2; }
        else if (t < kt1 + kt2) { w = a.w[1]; kgroups = MLP_H / 8; kt = t - kt1; j0 = swave * 2; nj = 2; }
        else { w = a.w[2]; kgroups = MLP_H / 8; kt = t - kt1 - kt2; j0 = (N3 == 64) ? (swave & 1) : swave * 2; nj = TNo; }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // wave-uniform base in SGPRs, lane term in one VGPR, k-group as the immediate offset: no per-load VALU
            const float *base = w + (size_t)((j0 + min(j, nj - 1)) * kgroups + kt * 4) * 256;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst[j][g]) : "v"(lane16), "s"(base), "n"(g * 1024));
        }
    };
    // the set `fb` (issued one tile ago) has landed when at most the 8 loads issued after it are outstanding
    auto wait_frags = [&](f32x4 (&fb)[2][4]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(8)"
                     : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]),
                       "+v"(fb[1][3]));
    };

    load_frags(0, fbA);
    // ---- stage the input tile: act[row][0..Cin) ----------------------------------------------------------
    // As few instructions as possible: while the co-resident workgroup streams MFMAs this wave issues roughly one instruction
    // per MFMA slot (tools/head_probe.py HEAD_TRACE=1 showed 20-30k cycles for the old index arithmetic), so: one
    // per-thread offset, buffer loads that differ only in the SGPR offset (rows past M read 0.0f), immediate LDS offsets.
    {
        constexpr int CIN = HAS_L1 ? 128 : 256;       // enforced by m3d_head_mlp_forward*
        constexpr int C4N = CIN / 4, RPP = 256 / C4N, NP = MLP_BM / RPP;
        const int row0 = tid / C4N, c4 = tid % C4N;
        const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, (unsigned)a.M * (unsigned)a.in_cs * 4u);
        const unsigned voff = ((unsigned)(m0 + row0) * (unsigned)a.in_cs + (unsigned)c4 * 4u) * 4u;
        const unsigned pstep = (unsigned)(RPP * a.in_cs) * 4u;
        f32x4 v[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) v[k] = MLP_ABL(4) ? f32x4{0.f, 0.f, 0.f, 0.f} : buf_load_f32x4(rin, voff, k * pstep);
        float *dst = act + row0 * MLP_LDA + c4 * 4;
#pragma unroll
        for (int k = 0; k < NP; ++k) *reinterpret_cast<f32x4 *>(dst + k * RPP * MLP_LDA) = v[k];
    }
    TRACE();
    __syncthreads();
    TRACE();

    int t = 0;   // position in the weight stream
    // ---- hidden layers: each wave computes 64 px x 64 ch -------------------------------------------------
    auto hidden_layer = [&](int layer, int KT) {
        // the accumulators are never cleared: the very first MFMA of a layer takes a zero literal as C (clearing 64 AGPRs per
        // layer would be 64 VALU instructions, each of which stalls the SIMD's MFMA stream -- tools/ubench/mfma_side_cost.hip)
        f32x16 acc[2][2];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int wn = wave * 64;
        auto tile = [&](int kt, f32x4 (&fb)[2][4], auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const float *Ab = act + l31 * MLP_LDA + kt * MLP_BK + lh4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 fa[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * MLP_LDA + g * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][g][s],
                                                                             (FIRST && g == 0 && s == 0) ? zero16 : acc[i][j], 0, 0, 0);
            }
        };
        // this lane's affine parameters: requested before the k loop, used after it (latency off the layer boundary)
        float sj[2], bj[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            sj[j] = a.scale[layer][wn + j * 32 + l31];
            bj[j] = a.shift[layer][wn + j * 32 + l31];
        }
        // KT is even (Cin and 256 are multiples of 64); the first pair is peeled for the zero-C start
        load_frags(t + 1, fbB);
        __builtin_amdgcn_sched_barrier(0);            // keep the prefetch ahead of the MFMAs
        wait_frags(fbA);
        tile(0, fbA, std::true_type{});
        TRACE();
        ++t;
        load_frags(t + 1, fbA);
        __builtin_amdgcn_sched_barrier(0);
        wait_frags(fbB);
        tile(1, fbB, std::false_type{});
        TRACE();
        ++t;
        for (int kt = 2; kt < KT; kt += 2) {
            load_frags(t + 1, fbB);
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(fbA);
            tile(kt, fbA, std::false_type{});
            TRACE();
            ++t;
            load_frags(t + 1, fbA);
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(fbB);
            tile(kt + 1, fbB, std::false_type{});
            TRACE();
            ++t;
        }
        // affine + LeakyReLU in registers, then overwrite the activation tile in place
        __syncthreads();                              // every wave has finished reading the old tile
        TRACE();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = wn + j * 32 + l31;
            const float s = sj[j], b = bj[j];
            const f32x2 s2 = {s, s}, b2 = {b, b}, k2 = {M3D_LEAKY_SLOPE, M3D_LEAKY_SLOPE};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {     // two rows per packed op: fma, slope multiply; max per element
                    const f32x2 v = __builtin_elementwise_fma(f32x2{acc[i][j][r], acc[i][j][r + 1]}, s2, b2);
                    f32x2 lo;                             // spelled in asm: hipcc splits the <2 x float> multiply whose lanes are used separately
                    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(v), "v"(k2));
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int row = i * 32 + ((r + e) & 3) + 8 * ((r + e) >> 2) + hrow;
                        if (!MLP_ABL(8)) act[row * MLP_LDA + co] = fmaxf(v[e], lo[e]);   // == leaky(v): 0 < slope < 1
                    }
                }
        }
        TRACE();
        __syncthreads();
        TRACE();
    };
    if (HAS_L1) hidden_layer(0, kt1);
    hidden_layer(1, kt2);

    // ---- output layer: 64 px x N3 channels, operands swapped -> D[channel][pixel] ------------------------
    {
        constexpr int TMo = (N3 == 64) ? 1 : 2;
        const int wm = (N3 == 64) ? (wave >> 1) * 32 : 0;
        const int wn = (N3 == 64) ? (wave & 1) * 32 : wave * 64;
        f32x16 acc[TMo][TNo];
#pragma unroll
        for (int i = 0; i < TMo; ++i)
#pragma unroll
            for (int j = 0; j < TNo; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // per-channel affine of this lane's 16 rows per column tile, requested before the k loop (latency off the tail)
        // (only for the narrow form: 64 more live registers would halve the occupancy of the 256-channel form)
        float scv[TNo][16], shv[TNo][16];
        auto load_affine = [&]() {
#pragma unroll
            for (int j = 0; j < TNo; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = min(wn + j * 32 + (r & 3) + 8 * (r >> 2) + hrow, a.Cout - 1);
                    scv[j][r] = a.scale[2][co];
                    shv[j][r] = a.shift[2][co];
                }
        };
        if constexpr (N3 == 64) load_affine();
        auto tile = [&](int kt, f32x4 (&fb)[2][4]) {
            const float *Ab = act + (wm + l31) * MLP_LDA + kt * MLP_BK + lh4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 fa[TMo];
#pragma unroll
                for (int i = 0; i < TMo; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * MLP_LDA + g * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TMo; ++i)
#pragma unroll
                        for (int j = 0; j < TNo; ++j)
                            if (!MLP_ABL(2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[j][g][s], fa[i][s], acc[i][j], 0, 0, 0);
            }
        };
        for (int kt = 0; kt < kt3; kt += 2) {
            load_frags(t + 1, fbB);
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(fbA);
            tile(kt, fbA);
            ++t;
            load_frags(t + 1, fbA);
            __builtin_amdgcn_sched_barrier(0);
            wait_frags(fbB);
            tile(kt + 1, fbB);
            ++t;
        }
        // The clamped prefetch past the last tile is still in flight into fbA / fbB, and the compiler does not track the asm loads:
        // it must land before the register allocator's next tenants of those registers are written.  (A four-set, three-tiles-
        // ahead pipeline for this layer read garbage for exactly this reason until the hidden layers' last prefetch was drained
        // first; once correct it was 1 % slower than this loop and was not kept.)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(fbA[0][0]), "+v"(fbA[0][1]), "+v"(fbA[0][2]), "+v"(fbA[0][3]), "+v"(fbA[1][0]), "+v"(fbA[1][1]),
                       "+v"(fbA[1][2]), "+v"(fbA[1][3]), "+v"(fbB[0][0]), "+v"(fbB[0][1]), "+v"(fbB[0][2]), "+v"(fbB[0][3]),
                       "+v"(fbB[1][0]), "+v"(fbB[1][1]), "+v"(fbB[1][2]), "+v"(fbB[1][3]));
        TRACE();
        // planar store out[n][co][pix]: the channel term rides in the SGPR offset of a buffer store, a lane whose pixel or
        // channel does not exist gets the out-of-range offset
        if constexpr (N3 != 64) load_affine();
        const int nimg = a.M / a.HW;
        const __amdgpu_buffer_rsrc_t rout =
            make_rsrc(a.out, (unsigned)(((long long)(nimg - 1) * a.out_img_stride + (long long)a.Cout * a.HW) * 4));
        const unsigned hw4 = (unsigned)a.HW * 4u;
#pragma unroll
        for (int i = 0; i < TMo; ++i) {
            const int m = m0 + wm + i * 32 + l31;
            const bool mok = m < a.M;
            const int mm = mok ? m : 0;
            const int n = mm / a.HW, pix = mm - n * a.HW;
            const unsigned pbase = (unsigned)((long long)n * a.out_img_stride + pix + (long long)hrow * a.HW) * 4u;
#pragma unroll
            for (int j = 0; j < TNo; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cu = wn + j * 32 + (r & 3) + 8 * (r >> 2);      // wave-uniform part of the channel
                    const unsigned vo = (mok && cu + hrow < a.Cout) ? pbase : M3D_BUF_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i][j][r] * scv[j][r] + shv[j][r]), rout,
                                                          vo, (unsigned)cu * hw4, 0);
                }
            }
        }
    }
    TRACE();
}

template <bool HAS_L1, int N3>
static int launch_mlp(const MlpBatch &b, int n, hipStream_t stream)
{
    constexpr size_t smem = (size_t)(MLP_BM * MLP_LDA) * sizeof(float);
    auto kern = head_mlp_kernel<HAS_L1, N3>;
    static bool attr_set = false;
    if (!attr_set) {
        M3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(cdiv(b.head[0].M, MLP_BM), n), dim3(256), smem, stream, b);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

static int fill_mlp_args(const m3d_mlp_desc *d, MlpArgs &a)
{
    M3D_REQUIRE(d->in && d->w2 && d->w3 && d->out, "head_mlp: null pointer");
    M3D_REQUIRE(d->s2 && d->t2 && d->s3 && d->t3, "head_mlp: scale/shift of layers 2 and 3 are required");
    M3D_REQUIRE((d->Cin == 128 && d->w1 && d->s1 && d->t1) || (d->Cin == 256 && !d->w1),
                "head_mlp: Cin must be 128 (3 layers, w1 given) or 256 (2 layers, w1 NULL)");
    M3D_REQUIRE(d->in_cs % 4 == 0 && d->in_cs >= d->Cin && ((uintptr_t)d->in & 15) == 0, "head_mlp: input alignment");
    M3D_REQUIRE(d->Cout >= 1 && d->Cout <= d->Cout_pad && (d->Cout_pad == 64 || d->Cout_pad == 256),
                "head_mlp: Cout_pad must be 64 or 256 (got %d)", d->Cout_pad);
    M3D_REQUIRE(d->M > 0 && d->M < (1ll << 30) && d->HW > 0 && d->M % d->HW == 0, "head_mlp: bad M / HW");
    M3D_REQUIRE(d->M * d->in_cs * 4 < (1ll << 31) &&
                ((d->M / d->HW - 1) * d->out_img_stride + (long long)d->Cout * d->HW) * 4 < (1ll << 31),
                "head_mlp: input / output views must be < 2 GiB");
    a.in = d->in; a.in_cs = d->in_cs; a.M = (int)d->M; a.HW = d->HW; a.Cin = d->Cin;
    a.w[0] = d->w1; a.w[1] = d->w2; a.w[2] = d->w3;
    a.scale[0] = d->s1; a.scale[1] = d->s2; a.scale[2] = d->s3;
    a.shift[0] = d->t1; a.shift[1] = d->t2; a.shift[2] = d->t3;
    a.out = d->out; a.out_img_stride = d->out_img_stride; a.Cout = d->Cout; a.Cout_pad = d->Cout_pad;
    static int abl = -1;
    if (abl < 0) { const char *e = getenv("M3D_ABLATE_MLP"); abl = e ? atoi(e) : 0; }
    a.ablate = abl;
    return M3D_OK;
}

extern "C" int m3d_head_mlp_forward_batched(const m3d_mlp_desc *d, int n, m3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    M3D_REQUIRE(d && n >= 1 && n <= MLP_MAX_HEADS, "head_mlp: 1..%d heads per launch (got %d)", MLP_MAX_HEADS, n);
    MlpBatch b;
    for (int i = 0; i < n; ++i) {
        const int rc = fill_mlp_args(d + i, b.head[i]);
        if (rc != M3D_OK) return rc;
        M3D_REQUIRE(d[i].M == d[0].M && (d[i].w1 != nullptr) == (d[0].w1 != nullptr) && d[i].Cout_pad == d[0].Cout_pad,
                    "head_mlp: heads of one launch must share M, depth and Cout_pad (head %d differs)", i);
    }
    for (int i = n; i < MLP_MAX_HEADS; ++i) b.head[i] = b.head[0];
#ifdef HEAD_TRACE
    b.trace = g_head_trace;
#endif
    if (d->w1) {
        if (d->Cout_pad == 64) return launch_mlp<true, 64>(b, n, stream);
        return launch_mlp<true, 256>(b, n, stream);
    }
    if (d->Cout_pad == 64) return launch_mlp<false, 64>(b, n, stream);
    return launch_mlp<false, 256>(b, n, stream);
}

extern "C" int m3d_head_mlp_forward(const m3d_mlp_desc *d, m3d_stream_t stream)
{
    return m3d_head_mlp_forward_batched(d, 1, stream);
}
