// Winograd F(2x2, 3x3) convolution for gfx950 on the fp32 MFMA pipe: 3x3, stride 1, pad 1, NHWC, fp32 throughout.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile / 4x4 input tile, summed over input channels
//
// 16 multiplies per 4 outputs instead of 36: 2.25x fewer MFMA FLOPs than the direct implicit GEMM (igemm_conv.hip) --
// the same algebraic reformulation cuDNN applies to these layers in the reference (nn.Conv2d 3x3 in
// model/pose_dla_dcn.py:96-103 and the cls head).  The 16 transform positions xi = (a, b) are 16 independent GEMMs
// M[xi] (tiles x Cout) = V[xi] (tiles x Cin) . U[xi] (Cin x Cout).
//
// One workgroup = 32 output tiles (128 output pixels) x 32 output channels, 4 waves, each wave owns 4 of the 16 xi:
//   * every thread loads 3 rows x 4 columns of a 4x4 input patch for one (tile, channel quad) (12 x 16-byte loads;
//     offsets constant over the whole K loop, SGPR channel base advances), applies its half of B^T d B in registers
//     and writes 8 of the 16 V[xi][tile][quad] to LDS;
//   * U = G g G^T is transformed offline (fp64 -> fp32) and packed in MFMA-fragment order, so every wave reads the B
//     fragments of ITS xi straight global->register (coalesced 1 KB loads, no LDS, no reuse lost);
//   * v_mfma_f32_32x32x2_f32 with the k-permuted fragment order (lane half h, step t -> k = 8g + 4h + t);
//   * the 16 accumulated M[xi] meet in LDS, A^T M A + affine/residual/LeakyReLU/sigmoid are applied per (tile, cout)
//     and written NHWC with lanes along channels.
#include "common.h"

struct WinoArgs {
    const float *in;
    const float *U;          // [16][Cout_pad/32][Cin/8][64 lanes][4]
    float *out;
    const float *scale;
    const float *shift;
    const float *res;
    int in_cs, out_cs, res_cs;
    int N, H, W, Cin, Cout, Cout_pad;
    int TH, TW, NT;          // tiles per image (rows, cols), total tiles
    int tiles_n;             // Cout_pad / 32
    int act, sigmoid_from, res_mode;
};

#define WINO_T 32            // tiles per workgroup
#define WINO_BK 16           // input channels per k-step
#define WINO_VBUF (16 * WINO_T * WINO_BK)   // floats per V buffer: V[xi][tile][16], unpadded, XOR-swizzled quads
#define WINO_LDM 33

__global__ __launch_bounds__(256, 2) void wino_kernel(const WinoArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // V[2][16][32][16]  then  M[16][32][33]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh4 = (lane >> 5) * 4, hrow = 4 * (lane >> 5);

    int tile_blk;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        tile_blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = tile_blk / a.tiles_n, bn = tile_blk - bm * a.tiles_n;
    const int t0 = bm * WINO_T, n0 = bn * 32;
    const int KS = a.Cin / WINO_BK;

    // ---- loader state: thread = (tile, channel quad, half).  half 0 produces V rows 0,1 from input rows 0..2,
    //      half 1 produces V rows 2,3 from input rows 1..3 (B^T d needs rows {0,2},{1,2} / {2,1},{1,3}) ---------
    const int half = tid & 1, item = tid >> 1;
    const int ltile = item >> 2, lquad = (item & 3) * 4;
    unsigned poff[12];
    unsigned pmask = 0;          // bit e*4+c set = position inside the image (e = 0..2 -> input row half+e)
    {
        const int t = t0 + ltile;
        const bool tv = t < a.NT;
        const int tt = tv ? t : 0;
        const int n = tt / (a.TH * a.TW), rem = tt - n * a.TH * a.TW;
        const int ty = rem / a.TW, tx = rem - ty * a.TW;
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int hi = 2 * ty - 1 + half + e, wi = 2 * tx - 1 + c;
                const bool ok = tv && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
                const int hc = min(max(hi, 0), a.H - 1), wc = min(max(wi, 0), a.W - 1);
                poff[e * 4 + c] = ((unsigned)((n * a.H + hc) * a.W + wc) * (unsigned)a.in_cs + (unsigned)lquad) * 4u;
                if (ok) pmask |= 1u << (e * 4 + c);
            }
    }
    f32x4 d[12];
    auto load_patch = [&](int ks) {
        const char *base = reinterpret_cast<const char *>(a.in + ks * WINO_BK);
#pragma unroll
        for (int i = 0; i < 12; ++i) d[i] = *reinterpret_cast<const f32x4 *>(base + poff[i]);
    };
    // V element (xi, tile, quad q) lives at ((xi*32 + tile)*16 + 4*(q ^ ((tile >> 2) & 3))): 64-byte rows without
    // padding; the XOR spreads the 16 tiles of a ds_read_b128 lane group over all 64 banks (conflict-free)
    const int wq = (((item & 3) ^ ((ltile >> 2) & 3))) * 4;
    auto transform_store = [&](int buf) {
        if (pmask != 0xFFFu) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                if (!((pmask >> i) & 1u)) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // B^T d for this thread's two V rows (per column c), then (.) B along the columns
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            f32x4 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (half == 0) t[c] = (rr == 0) ? d[0 * 4 + c] - d[2 * 4 + c] : d[1 * 4 + c] + d[2 * 4 + c];
                else           t[c] = (rr == 0) ? d[1 * 4 + c] - d[0 * 4 + c] : d[0 * 4 + c] - d[2 * 4 + c];
            }
            const f32x4 v0 = t[0] - t[2], v1 = t[1] + t[2], v2 = t[2] - t[1], v3 = t[1] - t[3];
            const int r = half * 2 + rr;
            float *vb = smem + buf * WINO_VBUF + ((r * 4) * WINO_T + ltile) * WINO_BK + wq;
            *reinterpret_cast<f32x4 *>(vb) = v0;
            *reinterpret_cast<f32x4 *>(vb + 1 * WINO_T * WINO_BK) = v1;
            *reinterpret_cast<f32x4 *>(vb + 2 * WINO_T * WINO_BK) = v2;
            *reinterpret_cast<f32x4 *>(vb + 3 * WINO_T * WINO_BK) = v3;
        }
    };

    // ---- U fragments of this wave's 4 xi: [xi][g] -----------------------------------------------------------
    const int kgroups = a.Cin / 8;
    f32x4 fb[4][2], fbn[4][2];
    auto load_u = [&](int ks, f32x4 (&dst)[4][2]) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                dst[x][g] = *reinterpret_cast<const f32x4 *>(
                    a.U + ((size_t)(((wave * 4 + x) * a.tiles_n + bn) * kgroups + ks * 2 + g) * 64 + lane) * 4);
    };

    f32x16 acc[4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    // software pipeline, ONE barrier per k-step: while the MFMAs of step ks run, the patch of step ks+1 (already in
    // registers) is transformed into the other V buffer, the patch of ks+2 and the U fragments of ks+1 are in flight
    load_patch(0);
    load_u(0, fb);
    transform_store(0);
    if (KS > 1) load_patch(1);
    __syncthreads();
    const int rq0 = (((lane >> 5)) ^ ((l31 >> 2) & 3)) * 4;          // swizzled quad of k-group 0 (q = h)
    const int rq1 = ((2 + (lane >> 5)) ^ ((l31 >> 2) & 3)) * 4;      // k-group 1 (q = 2 + h)
    for (int ks = 0; ks < KS; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < KS) load_u(ks + 1, fbn);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const float *vb = smem + buf * WINO_VBUF + ((wave * 4 + x) * WINO_T + l31) * WINO_BK;
            const f32x4 fa0 = *reinterpret_cast<const f32x4 *>(vb + rq0);
            const f32x4 fa1 = *reinterpret_cast<const f32x4 *>(vb + rq1);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s], fb[x][0][s], acc[x], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s], fb[x][1][s], acc[x], 0, 0, 0);
        }
        if (ks + 1 < KS) {
            transform_store(buf ^ 1);          // independent of the MFMAs above: the scheduler interleaves them
            if (ks + 2 < KS) load_patch(ks + 2);
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int g = 0; g < 2; ++g) fb[x][g] = fbn[x][g];
        }
        __syncthreads();
    }

    // ---- gather the 16 M[xi] in LDS: M[xi][tile][cout] (the loop ended with a barrier) ------------------------
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        float *mb = smem + (size_t)(wave * 4 + x) * WINO_T * WINO_LDM + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) mb[((r & 3) + 8 * (r >> 2) + hrow) * WINO_LDM] = acc[x][r];
    }
    __syncthreads();

    // ---- A^T M A + epilogue: thread = (tile, cout) ----------------------------------------------------------
    const int co = n0 + (tid & 31);
    const bool cok = co < a.Cout;
    const float sc = (cok && a.scale) ? a.scale[co] : 1.f;
    const float sh = (cok && a.shift) ? a.shift[co] : 0.f;
    const bool sg = a.sigmoid_from >= 0 && co >= a.sigmoid_from;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int tl = (tid >> 5) + 8 * q;
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
}

extern "C" int m3d_wino_conv3x3_forward(const m3d_conv_desc *d, m3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    M3D_REQUIRE(d && d->in && d->wgt && d->out, "wino: null pointer");
    M3D_REQUIRE(d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1, "wino: 3x3 stride 1 pad 1 only");
    M3D_REQUIRE(d->Cin % 16 == 0 && d->Cout_pad % 32 == 0 && d->Cout <= d->Cout_pad, "wino: Cin %% 16, Cout_pad %% 32");
    M3D_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0 && d->Ho == d->H && d->Wo == d->W, "wino: even H and W");
    M3D_REQUIRE(!d->out_nchw && !d->dcn_offmask && !d->wgt_img_stride, "wino: NHWC output, plain conv, shared weights");
    M3D_REQUIRE(d->in_cs % 4 == 0 && ((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->wgt & 15) == 0, "wino: alignment");
    M3D_REQUIRE((long long)d->N * d->H * d->W * d->in_cs * 4 < (1ll << 32), "wino: input view must be < 4 GiB");
    WinoArgs a;
    a.in = d->in; a.U = d->wgt; a.out = d->out; a.scale = d->scale; a.shift = d->shift; a.res = d->res;
    a.in_cs = d->in_cs; a.out_cs = d->out_cs; a.res_cs = d->res_cs;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.Cout_pad = d->Cout_pad;
    a.TH = d->H / 2; a.TW = d->W / 2; a.NT = d->N * a.TH * a.TW; a.tiles_n = d->Cout_pad / 32;
    a.act = d->act; a.sigmoid_from = d->sigmoid_from; a.res_mode = d->res_mode;
    constexpr size_t smem = (size_t)16 * WINO_T * WINO_LDM * sizeof(float);   // 67,584 B (>= 2 V buffers: 65,536 B)
    static bool attr_set = false;
    if (!attr_set) {
        M3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
        attr_set = true;
    }
    const int grid = cdiv(a.NT, WINO_T) * a.tiles_n;
    hipLaunchKernelGGL(wino_kernel, dim3(grid), dim3(256), smem, stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
