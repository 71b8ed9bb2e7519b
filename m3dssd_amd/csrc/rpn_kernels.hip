// RPN-side kernels that are bandwidth/latency bound: class softmax + top-1 foreground anchor,
// offset/mask synthesis for shape_align / center_align, ANAB pyramid pooling, row softmax,
// output bundling (flatten + cat + prob) and decode of the selected rows.
#include <type_traits>

#include "common.h"

// ---------------------------------------------------------------------------------------
// M3d_inference_align.py:229-234 (softmax over classes, fg = 1 - p(bg)) and
// feturealign_mgpu.py:58-62 / 160-164 (topk k=1 -> index, max prob).  Tie rule: lowest index.
// cls_planar [B][num_classes*A][HW], channel = cls*A + a.
__global__ void anchor_select_kernel(const float *__restrict__ cls, int A, int NC, int HW, int *__restrict__ sel_idx,
                                     float *__restrict__ sel_prob, float *__restrict__ fg_all)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float *base = cls + (size_t)b * NC * A * HW + p;
    float best = -1.f;
    int bi = 0;
    for (int a = 0; a < A; ++a) {
        float l[8];
        float mx = -INFINITY;
        for (int c = 0; c < NC; ++c) {
            l[c] = base[(size_t)(c * A + a) * HW];
            mx = fmaxf(mx, l[c]);
        }
        float s = 0.f, e0 = 0.f;
        for (int c = 0; c < NC; ++c) {
            const float e = expf(l[c] - mx);
            if (c == 0) e0 = e;
            s += e;
        }
        const float fg = 1.f - e0 / s;
        if (fg_all) fg_all[((size_t)b * A + a) * HW + p] = fg;
        if (fg > best) { best = fg; bi = a; }
    }
    sel_idx[(size_t)b * HW + p] = bi;
    sel_prob[(size_t)b * HW + p] = best;
}

// 4-class fast path: 64 pixels x 4 anchor groups per workgroup (one wave per group, pixels along lanes -> 256-byte coalesced
// loads), the anchors of a group unrolled by 3 so that 12 loads are in flight; the groups cover ascending anchor ranges and are
// merged in that order with a strict '>' -> the same "lowest index wins" tie rule as the sequential kernel above, and each
// fg value is computed by the same expression (bit-identical).  The one-thread-per-pixel loop above serialised 36 x 4
// dependent load latencies on 960 waves (60 us for 35 MB).
// `key` (optional): the detection stage's sort keys [B][A][HW] -- f32_sortable(max foreground class probability), the bits
// score_keys_planar_kernel / bundle_outputs produce -- written on the way: the logits are in registers here, and the separate key
// pass re-read all of them (283 MB at bs 64).
__global__ __launch_bounds__(256) void anchor_select4_kernel(const float *__restrict__ cls, int A, int HW, int *__restrict__ sel_idx,
                                                             float *__restrict__ sel_prob, float *__restrict__ fg_all,
                                                             unsigned int *__restrict__ key)
{
    __shared__ float sbest[4][64];
    __shared__ int sidx[4][64];
    const int b = blockIdx.y, g = threadIdx.x >> 6, lp = threadIdx.x & 63;
    const int p = blockIdx.x * 64 + lp;
    const bool pv = p < HW;
    const int per = (A + 3) >> 2, a0 = g * per, a1 = min(A, a0 + per);
    const float *base = cls + (size_t)b * 4 * A * HW + (pv ? p : 0);
    float best = -1.f;
    int bi = 0;
    unsigned int *kb = key ? key + (size_t)b * A * HW + (pv ? p : 0) : nullptr;
    auto fg_of = [&](const float (&l)[4], int an) {
        // class_softmax4: the same maximum, the same exponentials and the same left-to-right sum as this kernel always used
        const f32x4 pr = class_softmax4(f32x4{l[0], l[1], l[2], l[3]});
        if (kb && pv) kb[(size_t)an * HW] = f32_sortable(fg_score(pr));
        return 1.f - pr[0];
    };
    int a = a0;
    for (; a + 3 <= a1; a += 3) {
        float l[3][4];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) l[u][c] = base[(size_t)(c * A + a + u) * HW];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float fg = fg_of(l[u], a + u);
            if (fg_all && pv) fg_all[((size_t)b * A + a + u) * HW + p] = fg;
            if (fg > best) { best = fg; bi = a + u; }
        }
    }
    for (; a < a1; ++a) {
        float l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) l[c] = base[(size_t)(c * A + a) * HW];
        const float fg = fg_of(l, a);
        if (fg_all && pv) fg_all[((size_t)b * A + a) * HW + p] = fg;
        if (fg > best) { best = fg; bi = a; }
    }
    sbest[g][lp] = best;
    sidx[g][lp] = bi;
    __syncthreads();
    if (g == 0 && pv) {
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (sbest[k][lp] > best) { best = sbest[k][lp]; bi = sidx[k][lp]; }
        sel_idx[(size_t)b * HW + p] = bi;
        sel_prob[(size_t)b * HW + p] = best;
    }
}

extern "C" int m3d_anchor_select(const float *cls_planar, int B, int A, int num_classes, int HW, int *sel_idx,
                                 float *sel_prob, float *fg_all, m3d_stream_t stream)
{
    M3D_REQUIRE(cls_planar && sel_idx && sel_prob && num_classes >= 2 && num_classes <= 8, "anchor_select: bad arguments");
    if (num_classes == 4 && A >= 4)
        hipLaunchKernelGGL(anchor_select4_kernel, dim3(cdiv(HW, 64), B), dim3(256), 0, (hipStream_t)stream, cls_planar, A, HW,
                           sel_idx, sel_prob, fg_all, (unsigned int *)nullptr);
    else
        hipLaunchKernelGGL(anchor_select_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, cls_planar, A,
                           num_classes, HW, sel_idx, sel_prob, fg_all);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_anchor_select_keys(const float *cls_planar, int B, int A, int HW, int *sel_idx, float *sel_prob,
                                      unsigned int *score_bits, m3d_stream_t stream)
{
    M3D_REQUIRE(cls_planar && sel_idx && sel_prob && score_bits && A >= 4, "anchor_select_keys: bad arguments (4 classes, A >= 4)");
    hipLaunchKernelGGL(anchor_select4_kernel, dim3(cdiv(HW, 64), B), dim3(256), 0, (hipStream_t)stream, cls_planar, A, HW, sel_idx,
                       sel_prob, (float *)nullptr, score_bits);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// Same selection on an already computed fg-probability map [B][A][HW] (stand-alone align modules).
__global__ void fg_top1_kernel(const float *__restrict__ prob, int A, int HW, int *__restrict__ idx,
                               float *__restrict__ val)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float *base = prob + (size_t)b * A * HW + p;
    float best = base[0];
    int bi = 0;
    for (int a = 1; a < A; ++a) {
        const float v = base[(size_t)a * HW];
        if (v > best) { best = v; bi = a; }
    }
    idx[(size_t)b * HW + p] = bi;
    val[(size_t)b * HW + p] = best;
}

extern "C" int m3d_fg_top1(const float *prob, int B, int A, int HW, int *idx, float *val, m3d_stream_t stream)
{
    M3D_REQUIRE(prob && idx && val && A >= 1, "fg_top1: bad arguments");
    hipLaunchKernelGGL(fg_top1_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, prob, A, HW, idx, val);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------
// mode 0, shape_align (feturealign_mgpu.py:166-183): offmask[p][0..2kk) = table[idx][.] * hard,
//                                                    offmask[p][2kk..3kk) = max fg prob
// mode 1, center_align (feturealign_mgpu.py:67-89):  offmask[p] = (off_y, off_x, prob) with
//         off_x = ((bbox_x[idx] * std_x + mean_x) * anchor_w/stride) * hard        (kk = 1)
__global__ void align_offsets_kernel(int mode, const int *__restrict__ sel_idx, const float *__restrict__ sel_prob,
                                     float thresh, const float *__restrict__ table, const float *__restrict__ bbox_x,
                                     const float *__restrict__ bbox_y, const float *__restrict__ anchor_wh, float mean_x,
                                     float std_x, float mean_y, float std_y, float *__restrict__ om, int om_cs, int A,
                                     int HW, int kk, long long box_img_stride)
{
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const size_t bp = (size_t)b * HW + p;
    const int idx = sel_idx[bp];
    const float pr = sel_prob[bp];
    const float hard = pr > thresh ? 1.f : 0.f;
    float *o = om + bp * om_cs;
    if (mode == 0) {
        if (kk == 9 && om_cs == 28 && ((uintptr_t)om & 15) == 0) {
            // the 3x3 form (shape_align): the pixel's 27 values + its pad float leave as 7 x 16-byte stores (27 4-byte stores
            // at a 112-byte lane stride took 0.073 ms for 55 MB at bs 64)
            float v[28];
#pragma unroll
            for (int k = 0; k < 18; ++k) v[k] = table[idx * 18 + k] * hard;
#pragma unroll
            for (int k = 18; k < 27; ++k) v[k] = pr;
            v[27] = 0.f;
#pragma unroll
            for (int q = 0; q < 7; ++q) *reinterpret_cast<f32x4 *>(o + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            return;
        }
        for (int k = 0; k < 2 * kk; ++k) o[k] = table[idx * 2 * kk + k] * hard;
        for (int k = 0; k < kk; ++k) o[2 * kk + k] = pr;
    } else {
        const float bx = bbox_x[(size_t)b * box_img_stride + (size_t)idx * HW + p];
        const float by = bbox_y[(size_t)b * box_img_stride + (size_t)idx * HW + p];
        const float off_x = ((bx * std_x + mean_x) * anchor_wh[idx * 2 + 0]) * hard;
        const float off_y = ((by * std_y + mean_y) * anchor_wh[idx * 2 + 1]) * hard;
        o[0] = off_y;
        o[1] = off_x;
        o[2] = pr;
    }
}

extern "C" int m3d_align_offsets(int mode, const int *sel_idx, const float *sel_prob, float thresh, const float *table,
                                 const float *bbox_x, const float *bbox_y, const float *anchor_wh, float mean_x,
                                 float std_x, float mean_y, float std_y, float *offmask, int om_cs, int B, int A, int HW,
                                 int kk, long long box_img_stride, m3d_stream_t stream)
{
    M3D_REQUIRE(sel_idx && sel_prob && offmask && om_cs >= 3 * kk, "align_offsets: bad arguments");
    M3D_REQUIRE(mode == 0 ? (table != nullptr) : (bbox_x && bbox_y && anchor_wh && kk == 1), "align_offsets: mode inputs");
    hipLaunchKernelGGL(align_offsets_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, mode, sel_idx,
                       sel_prob, thresh, table, bbox_x, bbox_y, anchor_wh, mean_x, std_x, mean_y, std_y, offmask, om_cs,
                       A, HW, kk, box_img_stride);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------
// ANAB weighted pyramid pooling (attention.py:136-147): for scale s, bin (i,j):
//   mean over the adaptive window of  feat[c] * gate_s      (AdaptiveAvgPool2d windows:
//   rows [floor(i*H/s), ceil((i+1)*H/s)) etc.).  Work is split into row-chunk items so that the
//   48x160 "size 1" bin does not serialise on one block; slots are reduced in fixed order
//   (deterministic, no float atomics).
// items[i] = (bin, h0, h1, w0, w1, slot); partial [B][n_bins][max_slots][C].
__global__ void anab_pool_partial_kernel(const float *__restrict__ kv, int kv_cs, const float *__restrict__ s, int s_cs,
                                         const int *__restrict__ items, const int *__restrict__ bin_scale,
                                         float *__restrict__ partial, int n_bins, int max_slots, int H, int W, int C)
{
    const int it = blockIdx.x, b = blockIdx.y;
    const int bin = items[it * 6 + 0], h0 = items[it * 6 + 1], h1 = items[it * 6 + 2];
    const int w0 = items[it * 6 + 3], w1 = items[it * 6 + 4], slot = items[it * 6 + 5];
    const int sc = bin_scale[bin];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                const size_t p = (size_t)(b * H + h) * W + w;
                acc += kv[p * kv_cs + c] * s[p * s_cs + sc];
            }
        partial[(((size_t)b * n_bins + bin) * max_slots + slot) * C + c] = acc;
    }
}

extern "C" int m3d_anab_pool_partial(const float *kv, int kv_cs, const float *s, int s_cs, const int *items,
                                     int n_items, const int *bin_scale, int n_bins, float *partial, int max_slots,
                                     int B, int H, int W, int C, m3d_stream_t stream)
{
    M3D_REQUIRE(kv && s && items && bin_scale && partial && n_items > 0 && n_bins > 0, "anab_pool_partial: bad arguments");
    // one thread per channel where possible (C = 296 on the M3DSSD path -> 320 threads, a single pass)
    const int threads = C <= 1024 ? ((C + 63) / 64) * 64 : 256;
    hipLaunchKernelGGL(anab_pool_partial_kernel, dim3(n_items, B), dim3(threads), 0, (hipStream_t)stream, kv, kv_cs, s,
                       s_cs, items, bin_scale, partial, n_bins, max_slots, H, W, C);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// Position of weight element (row, k) of an [R][K] matrix in MFMA-fragment order [R/32][K/8][h=2][r=32][t=4] (what
// m3d_conv_wave_forward / m3d_head_mlp_forward take as `wgt`; m3dssd_amd.engine.pack_frag on the host).
__device__ __forceinline__ size_t frag_index(int row, int k, int K)
{
    return ((size_t)((row >> 5) * (K >> 3) + (k >> 3)) * 64 + ((k >> 2) & 1) * 32 + (row & 31)) * 4 + (k & 3);
}

// Sum the slots of each bin in order, scale by 1/area, scatter into the two GEMM operand layouts (row-major
// khat[keys_pad][ck_pad], vhatT[Cv][keys_pad], or -- frag != 0 -- the same two matrices in MFMA-fragment order; frag bit 0 = khat, bit 1 = vhatT).
// channels [0, Ck) are keys, [Ck, Ck+Cv) values.
__global__ void anab_pool_finish_kernel(const float *__restrict__ partial, const int *__restrict__ bin_slots,
                                        const float *__restrict__ bin_inv_area, int n_bins, int max_slots, int Ck,
                                        int Cv, float *__restrict__ khat, int keys_pad, int ck_pad,
                                        float *__restrict__ vhatT, int frag)
{
    const int bin = blockIdx.x, b = blockIdx.y;
    const int C = Ck + Cv;
    const int ns = bin_slots[bin];
    const float inv = bin_inv_area[bin];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float *pp = partial + (((size_t)b * n_bins + bin) * max_slots) * C + c;
        float acc = 0.f;
        for (int sl = 0; sl < ns; ++sl) acc += pp[(size_t)sl * C];
        acc *= inv;
        if (c < Ck) {
            if (frag & 1) khat[(size_t)b * keys_pad * ck_pad + frag_index(bin, c, ck_pad)] = acc;
            else khat[((size_t)b * keys_pad + bin) * ck_pad + c] = acc;
        } else {
            if (frag & 2) vhatT[(size_t)b * Cv * keys_pad + frag_index(c - Ck, bin, keys_pad)] = acc;
            else vhatT[((size_t)b * Cv + (c - Ck)) * keys_pad + bin] = acc;
        }
    }
}

extern "C" int m3d_anab_pool_finish(const float *partial, const int *bin_slots, const float *bin_inv_area, int n_bins,
                                    int max_slots, int Ck, int Cv, float *khat, int keys_pad, int ck_pad, float *vhatT,
                                    int B, int frag, m3d_stream_t stream)
{
    M3D_REQUIRE(partial && bin_slots && bin_inv_area && khat && vhatT && keys_pad >= n_bins && ck_pad >= Ck,
                "anab_pool_finish: bad arguments");
    M3D_REQUIRE(!frag || (keys_pad % 32 == 0 && ck_pad % 8 == 0 && Cv % 32 == 0), "anab_pool_finish: fragment layout needs "
                "keys_pad %% 32 == 0, ck_pad %% 8 == 0, Cv %% 32 == 0");
    hipLaunchKernelGGL(anab_pool_finish_kernel, dim3(n_bins, B), dim3(256), 0, (hipStream_t)stream, partial, bin_slots,
                       bin_inv_area, n_bins, max_slots, Ck, Cv, khat, keys_pad, ck_pad, vhatT, frag);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------
// Nested fast path of the pyramid pooling for psp sizes (1, 4, 8, 16) on maps with H % 16 == 0 and W % 16 == 0 (the 48x160
// map of the 1280x384 input): the adaptive windows of the four scales then nest exactly, so the features are read ONCE
// (the generic item kernel above reads them once per scale: 291 MB instead of 73 MB at bs=8).  One workgroup per finest
// bin (H/16 x W/16 pixels), one thread per channel, four gated sums per thread (one per scale) -> fine[B][256][4][C];
// the finish kernel adds the fine partials of each coarse bin in row-major order (deterministic) and scatters into the
// two GEMM operand layouts like anab_pool_finish_kernel.
// T = element type of the K|V map: float, or __bf16 for the bf16 engine (same fp32 sums over bf16-rounded features: the pooled
// keys / values are rounded to bf16 again before they become GEMM operands).
template <typename T>
__global__ void anab_pool_nested_kernel(const T *__restrict__ kv, int kv_cs, const float *__restrict__ s, int s_cs,
                                        float *__restrict__ fine, int H, int W, int C)
{
    const int fb = blockIdx.x, b = blockIdx.y;
    const int bh = H >> 4, bw = W >> 4;
    const int h0 = (fb >> 4) * bh, w0 = (fb & 15) * bw;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int h = h0; h < h0 + bh; ++h) {
            const size_t prow = (size_t)(b * H + h) * W + w0;
#pragma unroll 5
            for (int w = 0; w < bw; ++w) {
                const float x = (float)kv[(prow + w) * kv_cs + c];
                const float *g = s + (prow + w) * s_cs;
                a0 = fmaf(x, g[0], a0);
                a1 = fmaf(x, g[1], a1);
                a2 = fmaf(x, g[2], a2);
                a3 = fmaf(x, g[3], a3);
            }
        }
        float *o = fine + (((size_t)b * 256 + fb) * 4) * C + c;
        o[0] = a0; o[C] = a1; o[2 * (size_t)C] = a2; o[3 * (size_t)C] = a3;
    }
}

__global__ void anab_pool_nested_finish_kernel(const float *__restrict__ fine, int H, int W, int Ck, int Cv,
                                               float *__restrict__ khat, int keys_pad, int ck_pad, float *__restrict__ vhatT,
                                               int frag, __bf16 *__restrict__ khat16, __bf16 *__restrict__ vhat16)
{
    // khat16 / vhat16 (optional): bf16 twins of the two outputs (same element order) for the bf16 attention kernel -- what two
    // m3d_f32_to_bf16 launches produced before
    const int bin = blockIdx.x, b = blockIdx.y;       // bins in scale-major order: 1 + 16 + 64 + 256
    const int C = Ck + Cv;
    int si, sz, local;
    if (bin < 1) { si = 0; sz = 1; local = bin; }
    else if (bin < 17) { si = 1; sz = 4; local = bin - 1; }
    else if (bin < 81) { si = 2; sz = 8; local = bin - 17; }
    else { si = 3; sz = 16; local = bin - 81; }
    const int bi = local / sz, bj = local - bi * sz, f = 16 / sz;      // f x f finest bins per bin of this scale
    const float inv = 1.f / (float)((H / sz) * (W / sz));
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        // f*f terms (256 for the whole-map bin, 64 for the scale-4 bins): 32 independent partial sums = 32 loads in flight per thread
        // (round 6; before: 8 -- the whole-map workgroup of every image walked 32 dependent round trips, ~50 us of a 74 us launch
        // that moves 78 MB), then a fixed-order tree combine (deterministic)
        float acc;
        const float *fp = fine + (((size_t)b * 256) * 4 + si) * C + c;
        if (f == 16) {                          // the whole-map bin (f is 16, 4, 2 or 1): 8 rounds of 32 independent loads = 2 rows of the fine grid
            float part[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) part[u] = 0.f;
            for (int di = 0; di < 16; di += 2) {
#pragma unroll
                for (int u = 0; u < 32; ++u) part[u] += fp[(size_t)((di + (u >> 4)) * 16 + (u & 15)) * 4 * C];
            }
#pragma unroll
            for (int w = 16; w >= 1; w >>= 1)
#pragma unroll
                for (int u = 0; u < w; ++u) part[u] += part[u + w];
            acc = part[0];
        } else {                                // f = 1, 2 or 4: at most 16 terms, all in flight at once
            float part[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) part[u] = 0.f;
#pragma unroll
            for (int di = 0; di < 4; ++di)
#pragma unroll
                for (int dj = 0; dj < 4; ++dj)
                    if (di < f && dj < f) part[di * 4 + dj] = fp[(size_t)((bi * f + di) * 16 + bj * f + dj) * 4 * C];
#pragma unroll
            for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                for (int u = 0; u < w; ++u) part[u] += part[u + w];
            acc = part[0];
        }
        acc *= inv;
        if (c < Ck) {
            const size_t o = (frag & 1) ? (size_t)b * keys_pad * ck_pad + frag_index(bin, c, ck_pad) : ((size_t)b * keys_pad + bin) * ck_pad + c;
            khat[o] = acc;
            if (khat16) khat16[o] = (__bf16)acc;
        } else {
            const size_t o = (frag & 2) ? (size_t)b * Cv * keys_pad + frag_index(c - Ck, bin, keys_pad) : ((size_t)b * Cv + (c - Ck)) * keys_pad + bin;
            vhatT[o] = acc;
            if (vhat16) vhat16[o] = (__bf16)acc;
        }
    }
}

extern "C" long long m3d_anab_pool_nested_scratch_bytes(int B, int C) { return (long long)B * 256 * 4 * C * 4; }

template <typename T>
static int anab_pool_nested_launch(const T *kv, int kv_cs, const float *s, int s_cs, int B, int H, int W, int Ck, int Cv,
                                   float *scratch, float *khat, int keys_pad, int ck_pad, float *vhatT, int frag, m3d_stream_t stream,
                                   void *khat16 = nullptr, void *vhat16 = nullptr)
{
    M3D_REQUIRE(!frag || (keys_pad % 32 == 0 && ck_pad % 8 == 0 && Cv % 32 == 0), "anab_pool_nested: fragment layout needs "
                "keys_pad %% 32 == 0, ck_pad %% 8 == 0, Cv %% 32 == 0");
    M3D_REQUIRE(kv && s && scratch && khat && vhatT, "anab_pool_nested: null pointer");
    M3D_REQUIRE(H > 0 && W > 0 && H % 16 == 0 && W % 16 == 0, "anab_pool_nested: H and W must be multiples of 16 (got %dx%d)", H, W);
    M3D_REQUIRE(keys_pad >= 337 && ck_pad >= Ck && s_cs >= 4, "anab_pool_nested: bad operand layout");
    const int C = Ck + Cv;
    const int threads = C <= 1024 ? ((C + 63) / 64) * 64 : 256;
    // (tried, round 5: 8 channels per thread + pixel lanes combined through LDS -- 16-byte loads instead of 2-byte ones -- measured
    // 0.1 ms SLOWER per bs-64 step: 37 channel groups x 6 pixel lanes leave a thread 5 dependent loads and the 33 KB of LDS four
    // workgroups per CU; this form keeps 256 x 5 independent loads in flight per workgroup)
    // (tried, round 6: a compile-time 3 x 10 bin with all 30 loads of a thread issued up front -- 0.198 ms against 0.164 for pool +
    // finish at bs 64: the pool kernel already moves 4.2 TB/s (88 us by rocprofv3); the finish kernel's dependent round trips were
    // the slow half)
    hipLaunchKernelGGL(anab_pool_nested_kernel<T>, dim3(256, B), dim3(threads), 0, (hipStream_t)stream, kv, kv_cs, s, s_cs, scratch,
                       H, W, C);
    M3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(anab_pool_nested_finish_kernel, dim3(337, B), dim3(256), 0, (hipStream_t)stream, scratch, H, W, Ck, Cv,
                       khat, keys_pad, ck_pad, vhatT, frag, (__bf16 *)khat16, (__bf16 *)vhat16);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_anab_pool_nested(const float *kv, int kv_cs, const float *s, int s_cs, int B, int H, int W, int Ck, int Cv,
                                    float *scratch, float *khat, int keys_pad, int ck_pad, float *vhatT, int frag,
                                    m3d_stream_t stream)
{
    return anab_pool_nested_launch<float>(kv, kv_cs, s, s_cs, B, H, W, Ck, Cv, scratch, khat, keys_pad, ck_pad, vhatT, frag, stream);
}

extern "C" int m3d_anab_pool_nested_bf16(const void *kv, int kv_cs, const float *s, int s_cs, int B, int H, int W, int Ck, int Cv,
                                         float *scratch, float *khat, int keys_pad, int ck_pad, float *vhatT, int frag,
                                         m3d_stream_t stream)
{
    return anab_pool_nested_launch<__bf16>(static_cast<const __bf16 *>(kv), kv_cs, s, s_cs, B, H, W, Ck, Cv, scratch, khat, keys_pad,
                                           ck_pad, vhatT, frag, stream);
}

extern "C" int m3d_anab_pool_nested_bf16_ex(const void *kv, int kv_cs, const float *s, int s_cs, int B, int H, int W, int Ck, int Cv,
                                            float *scratch, float *khat, int keys_pad, int ck_pad, float *vhatT, int frag,
                                            void *khat16, void *vhat16, m3d_stream_t stream)
{
    return anab_pool_nested_launch<__bf16>(static_cast<const __bf16 *>(kv), kv_cs, s, s_cs, B, H, W, Ck, Cv, scratch, khat, keys_pad,
                                           ck_pad, vhatT, frag, stream, khat16, vhat16);
}

// ---------------------------------------------------------------------------------------
// Row softmax (attention.py:208), one wave per row, in place; pad columns zeroed.
__global__ void softmax_rows_kernel(float *__restrict__ x, int rows, int valid, int cs)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float *r = x + (size_t)row * cs;
    float mx = -INFINITY;
    for (int j = lane; j < valid; j += 64) mx = fmaxf(mx, r[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < valid; j += 64) sum += expf(r[j] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    for (int j = lane; j < cs; j += 64) r[j] = j < valid ? expf(r[j] - mx) * inv : 0.f;
}

// Single-pass variant for rows of up to 64*NV*4 floats with cs % 4 == 0: the row is read once into registers (16-byte loads),
// same arithmetic as the kernel above (max, then sum of expf(x - max), then the quotient; the per-lane grouping of the sum
// differs, i.e. fp32 reassociation only); the three-pass kernel re-reads the row from L2 twice.
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(float *__restrict__ x, int rows, int valid, int cs)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float *r = x + (size_t)row * cs;
    f32x4 v[NV];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = (k * 64 + lane) * 4;
        v[k] = j < cs ? *reinterpret_cast<const f32x4 *>(r + j) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j + e < valid) mx = fmaxf(mx, v[k][e]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = (k * 64 + lane) * 4 + e;
            v[k][e] = j < valid ? expf(v[k][e] - mx) : 0.f;
            sum += v[k][e];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = (k * 64 + lane) * 4;
        if (j < cs) *reinterpret_cast<f32x4 *>(r + j) = v[k] * inv;
    }
}

extern "C" int m3d_softmax_rows(float *x, int rows, int valid, int cs, m3d_stream_t stream)
{
    M3D_REQUIRE(x && rows > 0 && valid > 0 && cs >= valid, "softmax_rows: bad arguments");
    if (cs % 4 == 0 && cs <= 512 && ((uintptr_t)x & 15) == 0) {
        if (cs <= 256) hipLaunchKernelGGL(softmax_rows_reg_kernel<1>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, valid, cs);
        else hipLaunchKernelGGL(softmax_rows_reg_kernel<2>, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, valid, cs);
        M3D_LAUNCH_CHECK();
        return M3D_OK;
    }
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, valid, cs);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------
// Output bundling: flatten_tensor x13 + cat + class softmax (M3d_inference_align.py:229-232,280-301).
// One thread per anchor row = (a*HW + p); planar reads are coalesced along p, row writes are 16 B.
__global__ void bundle_outputs_kernel(const float *__restrict__ cls_pl, const float *__restrict__ box_pl,
                                      float *__restrict__ cls, float *__restrict__ prob, float *__restrict__ b2,
                                      float *__restrict__ b3, unsigned int *__restrict__ key, int A, int HW)
{
    const int b = blockIdx.y;
    const int R = A * HW;
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    const int a = row / HW, p = row - a * HW;
    const float *cb = cls_pl + (size_t)b * 4 * R;
    f32x4 l;
#pragma unroll
    for (int c = 0; c < 4; ++c) l[c] = cb[(size_t)(c * A + a) * HW + p];
    const f32x4 pr = class_softmax4(l);
    const size_t o = (size_t)b * R + row;
    *reinterpret_cast<f32x4 *>(cls + o * 4) = l;
    *reinterpret_cast<f32x4 *>(prob + o * 4) = pr;
    const float *bb = box_pl + (size_t)b * 11 * R + row;
    f32x4 v2;
#pragma unroll
    for (int k = 0; k < 4; ++k) v2[k] = bb[(size_t)k * R];
    *reinterpret_cast<f32x4 *>(b2 + o * 4) = v2;
#pragma unroll
    for (int k = 0; k < 7; ++k) b3[o * 7 + k] = bb[(size_t)(4 + k) * R];
    if (key) {
        key[o] = f32_sortable(fg_score(pr));
    }
}

extern "C" int m3d_bundle_outputs(const float *cls_planar, const float *box_planar, float *cls, float *prob,
                                  float *bbox_2d, float *bbox_3d, unsigned int *score_bits, int B, int A, int HW,
                                  m3d_stream_t stream)
{
    M3D_REQUIRE(cls_planar && box_planar && cls && prob && bbox_2d && bbox_3d, "bundle_outputs: null pointer");
    hipLaunchKernelGGL(bundle_outputs_kernel, dim3(cdiv((long long)A * HW, 256), B), dim3(256), 0, (hipStream_t)stream,
                       cls_planar, box_planar, cls, prob, bbox_2d, bbox_3d, score_bits, A, HW);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// The sort keys alone, straight from the planar class logits: what the detection stage needs of the 38 MB per image that
// bundle_outputs moves when nobody reads cls / prob / bbox_2d / bbox_3d in full (m3dssd_amd.pipeline.PipelinedDetector:
// m3d_topk_decode_planar decodes its 3000 rows from the planar staging).  Same softmax arithmetic as bundle_outputs_kernel
// (class_softmax4, common.h), so the keys are the same bits.  4 consecutive rows per thread: 16-byte loads / stores.
__global__ void score_keys_planar_kernel(const float *__restrict__ cls_pl, unsigned int *__restrict__ key, int A, int HW)
{
    const int b = blockIdx.y;
    const int R = A * HW;
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (row >= R) return;
    const int a = row / HW, p = row - a * HW;                  // HW % 4 == 0: the four rows share the anchor
    const float *cb = cls_pl + (size_t)b * 4 * R;
    f32x4 lc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) lc[c] = *reinterpret_cast<const f32x4 *>(cb + (size_t)(c * A + a) * HW + p);
    u32x4 k;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f32x4 l;
#pragma unroll
        for (int c = 0; c < 4; ++c) l[c] = lc[c][e];
        k[e] = f32_sortable(fg_score(class_softmax4(l)));
    }
    *reinterpret_cast<u32x4 *>(key + (size_t)b * R + row) = k;
}

extern "C" int m3d_score_keys_planar(const float *cls_planar, unsigned int *score_bits, int B, int A, int HW, m3d_stream_t stream)
{
    M3D_REQUIRE(cls_planar && score_bits && B >= 1 && A >= 1 && HW >= 4, "score_keys_planar: bad arguments");
    M3D_REQUIRE(HW % 4 == 0 && ((uintptr_t)cls_planar & 15) == 0 && ((uintptr_t)score_bits & 15) == 0,
                "score_keys_planar: HW (%d) must be a multiple of 4 and the buffers 16-byte aligned", HW);
    hipLaunchKernelGGL(score_keys_planar_kernel, dim3(cdiv((long long)A * HW / 4, 256), B), dim3(256), 0, (hipStream_t)stream,
                       cls_planar, score_bits, A, HW);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
