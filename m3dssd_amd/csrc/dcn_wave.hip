// Register-resident deformable convolution (DCNv2 main conv) for gfx950: ONE WAVE = 32*MT output pixels x 128 output
// channels, no LDS, no barriers -- the design of wino_wave_kernel (wino_conv.hip) applied to the modulated-deformable
// implicit GEMM of igemm_conv.hip.
//
//   * lane (pixel i = lane & 31 of each of the MT row tiles, half h = lane >> 5) gathers the four bilinear corners of
//     ITS pixel for channels 8*cg + 4h .. +3 (4 x 16-byte buffer loads, out-of-image corners read 0.0f), combines them
//     with 8 packed-fp32 ops (the modulation mask is folded into the corner weights once per tap,
//     model/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:18-47,174) and the result IS the A operand of the next MFMAs
//     (k-permuted order: lane half h, step t -> k = 8g + 4h + t);
//   * B operands: the [Cout_pad, kh*kw*Cin] tap-major weight matrix in MFMA-fragment order
//     [Cout_pad/32][K/8][h=2][r=32][t=4] (m3dssd_amd.engine.pack_frag), one coalesced 1 KB load per column tile and k-group;
//   * a step covers 32 channels of one tap (the full 128-byte line of each corner); the next step's corners are requested
//     right after the combine, the next k-group's B fragments before each group of MFMAs;
//   * epilogue (affine = folded BN + bias, residual, LeakyReLU) from the accumulators, NHWC, lanes along channels.
// Each wave is an independent unit (64-thread workgroups): the hardware balances them over the 1024 SIMDs, so the
// launcher only takes this path when there are enough waves to fill them (see m3d_dcn_wave_forward).
#include <stdlib.h>

#include "common.h"

struct DcnWaveArgs {
    const float *in, *wfrag, *scale, *shift, *res, *om;
    float *out;
    int in_cs, out_cs, res_cs, om_cs;
    int H, W, Cin, Ho, Wo, HoWo, Cout;
    int kh, kw, stride, pad, dil;
    int M, KG, CG, tiles_n;
    int act, res_mode;
    unsigned in_bytes, out_bytes, res_bytes, w_bytes;
};

template <int MT>
__global__ __launch_bounds__(64) void dcn_wave_kernel(const DcnWaveArgs a)
{
    constexpr int NT = 4;
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    int blk;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int bm = blk / a.tiles_n, bn = blk - bm * a.tiles_n;
    const int m0 = bm * 32 * MT;
    const int KK = a.kh * a.kw;

    int pix_base[MT], hi0[MT], wi0[MT];
    bool rvalid[MT];
    const float *omp[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + mt * 32 + l31;
        rvalid[mt] = m < a.M;
        const int mm = rvalid[mt] ? m : 0;
        const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        pix_base[mt] = n * a.H * a.W;
        hi0[mt] = ho * a.stride - a.pad;
        wi0[mt] = wo * a.stride - a.pad;
        omp[mt] = a.om + (size_t)mm * a.om_cs;
    }
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wfrag + (size_t)bn * NT * a.KG * 256, a.w_bytes);
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned wstride = (unsigned)a.KG * 1024u;              // bytes between column tiles

    // sampling state of the current tap (corner byte offsets incl. the channel sub-offset 4h, weights * mask) and the raw
    // (dh, dw, mask) of the next tap, fetched one tap ahead
    unsigned doff[MT][4];
    float bw[MT][4];
    float raw[MT][3];
    auto load_raw = [&](int tap) {
        const int tp = min(tap, KK - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            raw[mt][0] = omp[mt][2 * tp];
            raw[mt][1] = omp[mt][2 * tp + 1];
            raw[mt][2] = omp[mt][2 * KK + tp];
        }
    };
    auto setup_tap = [&](int tap) {
        const int ti = tap / a.kw, tj = tap - ti * a.kw;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
            int o1 = -1, o2 = -1, o3 = -1, o4 = -1;
            const float mk = raw[mt][2];
            const float h_im = (float)(hi0[mt] + ti * a.dil) + raw[mt][0];
            const float w_im = (float)(wi0[mt] + tj * a.dil) + raw[mt][1];
            if (rvalid[mt] && h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
                const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
                const int hh = hl + 1, wh = wl + 1;
                const float lh = h_im - (float)hl, lw = w_im - (float)wl;
                const float uh = 1.f - lh, uw = 1.f - lw;
                if (hl >= 0 && wl >= 0) { w1 = uh * uw; o1 = hl * a.W + wl; }
                if (hl >= 0 && wh <= a.W - 1) { w2 = uh * lw; o2 = hl * a.W + wh; }
                if (hh <= a.H - 1 && wl >= 0) { w3 = lh * uw; o3 = hh * a.W + wl; }
                if (hh <= a.H - 1 && wh <= a.W - 1) { w4 = lh * lw; o4 = hh * a.W + wh; }
            }
            bw[mt][0] = w1 * mk; bw[mt][1] = w2 * mk; bw[mt][2] = w3 * mk; bw[mt][3] = w4 * mk;
            const int o[4] = {o1, o2, o3, o4};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                doff[mt][q] = o[q] >= 0 ? ((unsigned)(pix_base[mt] + o[q]) * (unsigned)a.in_cs + (unsigned)(h * 4)) * 4u
                                        : M3D_BUF_OOB;
        }
    };

    // One super-step = 32 input channels of one tap = the whole 128-byte line of every corner: the four 16-byte chunks a
    // lane needs from a line are requested back to back, so each line crosses L2 -> L1 once (with 8-channel steps the 32-byte
    // slices of a line were fetched four times: 4 waves x 32 KB of live lines do not survive in the 32 KB L1).
    f32x4 cr[MT][4][4];                        // [row tile][corner][chunk j]: channels 32*c32 + 8j + 4h .. +3
    f32x4 bfA[NT], bfB[NT];
    auto issue_corners = [&](int c32) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) cr[mt][q][j] = buf_load_f32x4(rin, doff[mt][q], (unsigned)(c32 * 128 + j * 32));
    };
    auto issue_b = [&](int kg, f32x4 (&bf)[NT]) {
        const unsigned bsoff = (unsigned)min(kg, a.KG - 1) * 1024u;      // unconditional, clamped (see wino_kernel)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = buf_load_f32x4(rw, wlane, bsoff + nt * wstride);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    load_raw(0);
    setup_tap(0);
    load_raw(1);
    issue_corners(0);
    issue_b(0, bfA);

    const int C32 = a.Cin / 32;
    int tap = 0, c32 = 0;                      // position of the super-step being computed
    for (int kg0 = 0; kg0 < a.KG; kg0 += 4) {
        // ---- bilinear combine of the 4 chunks (mask already folded into bw): 8 packed ops per chunk ------------------
        f32x4 A[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                A[mt][j] = pk_fma_s(bw[mt][0], cr[mt][0][j], pk_fma_s(bw[mt][1], cr[mt][1][j],
                           pk_fma_s(bw[mt][2], cr[mt][2][j], pk_mul_s(bw[mt][3], cr[mt][3][j]))));
        // ---- next super-step: step to the next tap first if this was the tap's last channel block -------------------
        if (++c32 == C32) {                     // wave-uniform
            c32 = 0;
            ++tap;
            setup_tap(min(tap, KK - 1));
            load_raw(tap + 1);
        }
        issue_corners(c32);                     // after the last super-step: a redundant in-range reload, never used
        __builtin_amdgcn_sched_barrier(0);
        // ---- 4 k-groups x (4 x MT x NT) MFMAs; the B fragments of the next group are in flight meanwhile ---------------
        auto group = [&](int j, f32x4 (&bf)[NT], f32x4 (&bfn)[NT]) {
            issue_b(kg0 + j + 1, bfn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[mt][j][t], bf[nt][t], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        group(0, bfA, bfB);
        group(1, bfB, bfA);
        group(2, bfA, bfB);
        group(3, bfB, bfA);
    }

    // ---- epilogue: lane = channel n0 + nt*32 + l31, rows = pixels m0 + mt*32 + (r&3) + 8*(r>>2) + 4h ---------------------
    const __amdgpu_buffer_rsrc_t rout = make_rsrc(a.out, a.out_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(a.res ? a.res : a.out, a.res ? a.res_bytes : 0u);
    const int n0 = bn * 32 * NT;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = n0 + nt * 32 + l31;
        const bool cok = co < a.Cout;
        const float sc = (cok && a.scale) ? a.scale[co] : 1.f;
        const float sh = (cok && a.shift) ? a.shift[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int mb = m0 + mt * 32 + 4 * h;
            float rv[16];
            if (a.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    const unsigned ro = (cok && m < a.M) ? ((unsigned)m * (unsigned)a.res_cs + (unsigned)co) * 4u : M3D_BUF_OOB;
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, ro, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[mt][nt][r];
                if (a.res) v = a.res_mode ? (v + rv[r]) * sc + sh : v * sc + sh + rv[r];
                else v = v * sc + sh;
                if (a.act == 1) v = fmaxf(v, v * M3D_LEAKY_SLOPE);
                const unsigned oo = (cok && m < a.M) ? ((unsigned)m * (unsigned)a.out_cs + (unsigned)co) * 4u : M3D_BUF_OOB;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, oo, 0, 0);
            }
        }
    }
}

// Wave counts the register-resident kernel would launch for this layer; 0 = not applicable.
static int dcn_wave_plan(const m3d_conv_desc *d, int *mt, bool enforce_min)
{
    if (!d->dcn_offmask || d->out_nchw || d->wgt_img_stride || d->sigmoid_from >= 0) return 0;
    if (d->Cin % 32 != 0 || d->Cout_pad % 128 != 0) return 0;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    const long long cols = d->Cout_pad / 128;
    static int wave_min = -1;                    // tuning knob (experiments only): M3D_DCN_WAVE_MIN
    if (wave_min < 0) { const char *e = getenv("M3D_DCN_WAVE_MIN"); wave_min = e ? atoi(e) : 800; }
    if (((M + 63) / 64) * cols >= wave_min) { *mt = 2; return (int)(((M + 63) / 64) * cols); }
    if (((M + 31) / 32) * cols >= wave_min || !enforce_min) { *mt = 1; return (int)(((M + 31) / 32) * cols); }
    return 0;
}

extern "C" int m3d_dcn_wave_applicable(const m3d_conv_desc *d)
{
    int mt;
    return d ? dcn_wave_plan(d, &mt, true) : 0;
}

extern "C" int m3d_dcn_wave_forward(const m3d_conv_desc *d, m3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    M3D_REQUIRE(d && d->in && d->wgt && d->out && d->dcn_offmask, "dcn_wave: null pointer");
    int mt = 0;
    const int waves = dcn_wave_plan(d, &mt, false);       // the fill heuristic is advisory here
    M3D_REQUIRE(waves > 0, "dcn_wave: needs Cin %% 32 == 0, Cout_pad %% 128 == 0, NHWC output, shared weights, no sigmoid");
    const int ho = (d->H + 2 * d->pad - (d->dil * (d->kh - 1) + 1)) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - (d->dil * (d->kw - 1) + 1)) / d->stride + 1;
    M3D_REQUIRE(ho == d->Ho && wo == d->Wo, "dcn_wave: Ho/Wo mismatch");
    M3D_REQUIRE(d->in_cs % 4 == 0 && d->in_cs >= d->Cin && ((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->wgt & 15) == 0,
                "dcn_wave: alignment");
    const long long M = (long long)d->N * d->Ho * d->Wo;
    M3D_REQUIRE((long long)d->N * d->H * d->W * d->in_cs * 4 < (1ll << 31) && M * d->out_cs * 4 < (1ll << 31) &&
                M * d->res_cs * 4 < (1ll << 31) && (long long)d->Cout_pad * d->kh * d->kw * d->Cin * 4 < (1ll << 31),
                "dcn_wave: views must be < 2 GiB");
    DcnWaveArgs a;
    a.in = d->in; a.wfrag = d->wgt; a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.om = d->dcn_offmask;
    a.out = d->out;
    a.in_cs = d->in_cs; a.out_cs = d->out_cs; a.res_cs = d->res_cs; a.om_cs = d->dcn_om_cs;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.HoWo = d->Ho * d->Wo; a.Cout = d->Cout;
    a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.M = (int)M; a.CG = d->Cin / 8; a.KG = d->kh * d->kw * a.CG; a.tiles_n = d->Cout_pad / 128;
    a.act = d->act; a.res_mode = d->res_mode;
    a.in_bytes = (unsigned)((long long)d->N * d->H * d->W * d->in_cs * 4);
    a.out_bytes = (unsigned)(M * d->out_cs * 4);
    a.res_bytes = (unsigned)(M * d->res_cs * 4);
    a.w_bytes = (unsigned)((long long)128 * d->kh * d->kw * d->Cin * 4);
    if (mt == 2) hipLaunchKernelGGL(dcn_wave_kernel<2>, dim3(waves), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(dcn_wave_kernel<1>, dim3(waves), dim3(64), 0, stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
