// Post-forward detection stage on the device (lib/rpn_util.py:1442-1555 in the reference):
//   decode of the network outputs for the top-N-pre rows, the descending score sort that picks them, and the selection of
//   the kept rows after NMS into fixed-size blocks.
//
//  * topk_decode_kernel: ONE workgroup per image does the reference's `argsort()[::-1][:nms_topN_pre]` (rpn_util.py:1510-1544)
//    as an MSD radix SELECT over the 64-bit total order  key = (sortable score bits << 32) | (0xFFFFFFFF - row)
//    -- descending score, ascending row among equal scores -- followed by a bitonic sort of the k selected keys in LDS and
//    the decode of exactly those k rows.  Level 0 histograms the top 11 score bits of all R rows (LDS atomics on integers:
//    deterministic), rows above the threshold bin are selected, rows inside it become the candidates of the next digit
//    (ping-pong buffers in the workspace; typically a few hundred rows); the walk ends as soon as the candidates left are
//    exactly the rows still needed.  No float atomics, no data-dependent launch count: graph-capturable.
//  * select_post_kernel: keep lists of the NMS -> [B][post + 1][14] blocks (zero padded; row `post` carries the count), the
//    wire format of the multi-GPU all-gather.
#include <atomic>

#include "common.h"

#define TOPK_NT 1024
#define TOPK_MAXK 16384                  // keys of the final sort in LDS: 8 bytes each, next power of two of k (dynamic allocation)

// Decode (lib/rpn_util.py:1442-1521 + bbox_transform_inv :1137-1186), scale_factor = 1.
// Row layout out: x1,y1,x2,y2,score,cls,x3d,y3d,z3d,w3d,h3d,l3d,ry3d,anchor  (rpn_util.py:1550)
__device__ __forceinline__ void decode_values(int row, const float *__restrict__ v2 /*[4]*/, const float *__restrict__ v3 /*[7]*/,
                                              float p1, float p2, float p3, const float *__restrict__ rois,
                                              const float *__restrict__ anchors, const float *__restrict__ means,
                                              const float *__restrict__ stds, float *__restrict__ q)
{
    const float *ro = rois + (size_t)row * 5;
    const float x1 = ro[0], y1 = ro[1], x2 = ro[2], y2 = ro[3];
    const int tr = (int)ro[4];
    const float *an = anchors + tr * 9;
    const float widths = x2 - x1 + 1.0f, heights = y2 - y1 + 1.0f;
    const float ctr_x = x1 + 0.5f * widths, ctr_y = y1 + 0.5f * heights;
    float d3[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) d3[k] = v3[k] * stds[4 + k] + means[4 + k];
    const float dx = v2[0] * stds[0] + means[0];
    const float dy = v2[1] * stds[1] + means[1];
    const float dw = v2[2] * stds[2] + means[2];
    const float dh = v2[3] * stds[3] + means[3];
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    q[0] = pcx - 0.5f * pw;
    q[1] = pcy - 0.5f * ph;
    q[2] = pcx + 0.5f * pw;
    q[3] = pcy + 0.5f * ph;
    float sc = p1;
    int cl = 1;
    if (p2 > sc) { sc = p2; cl = 2; }
    if (p3 > sc) { sc = p3; cl = 3; }
    q[4] = sc;
    q[5] = (float)cl;
    q[6] = d3[0] * widths + ctr_x;
    q[7] = d3[1] * heights + ctr_y;
    q[8] = an[4] + d3[2];
    q[9] = expf(d3[3]) * an[5];
    q[10] = expf(d3[4]) * an[6];
    q[11] = expf(d3[5]) * an[7];
    q[12] = an[8] + d3[6];
    q[13] = (float)tr;
}

// rows of the bundled tensors prob [B][R][4], bbox_2d [B][R][4], bbox_3d [B][R][7]; o = img * R + row
__device__ __forceinline__ void decode_row(int row, size_t o, const float *__restrict__ prob, const float *__restrict__ b2,
                                           const float *__restrict__ b3, const float *__restrict__ rois,
                                           const float *__restrict__ anchors, const float *__restrict__ means,
                                           const float *__restrict__ stds, float *__restrict__ q)
{
    float v2[4], v3[7];
#pragma unroll
    for (int k = 0; k < 4; ++k) v2[k] = b2[o * 4 + k];
#pragma unroll
    for (int k = 0; k < 7; ++k) v3[k] = b3[o * 7 + k];
    decode_values(row, v2, v3, prob[o * 4 + 1], prob[o * 4 + 2], prob[o * 4 + 3], rois, anchors, means, stds, q);
}

// the same row read from the planar staging the heads write (cls [B][4A][HW], box [B][11][A*HW]): the class probabilities are
// recomputed from the four logits with the arithmetic of bundle_outputs (class_softmax4, common.h) -- the same bits
__device__ __forceinline__ void decode_row_planar(int row, int img, int A, int HW, const float *__restrict__ cls_pl,
                                                  const float *__restrict__ box_pl, const float *__restrict__ rois,
                                                  const float *__restrict__ anchors, const float *__restrict__ means,
                                                  const float *__restrict__ stds, float *__restrict__ q)
{
    const int R = A * HW;
    const int an = row / HW, p = row - an * HW;
    const float *cb = cls_pl + (size_t)img * 4 * R;
    f32x4 l;
#pragma unroll
    for (int c = 0; c < 4; ++c) l[c] = cb[(size_t)(c * A + an) * HW + p];
    const f32x4 pr = class_softmax4(l);
    const float *bb = box_pl + (size_t)img * 11 * R + row;
    float v2[4], v3[7];
#pragma unroll
    for (int k = 0; k < 4; ++k) v2[k] = bb[(size_t)k * R];
#pragma unroll
    for (int k = 0; k < 7; ++k) v3[k] = bb[(size_t)(4 + k) * R];
    decode_values(row, v2, v3, pr[1], pr[2], pr[3], rois, anchors, means, stds, q);
}

__global__ void decode_rows_kernel(const long long *__restrict__ rows, const float *__restrict__ prob,
                                   const float *__restrict__ b2, const float *__restrict__ b3,
                                   const float *__restrict__ rois, const float *__restrict__ anchors,
                                   const float *__restrict__ means, const float *__restrict__ stds,
                                   float *__restrict__ out, int R, int n_rows)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const int row = (int)rows[(size_t)b * n_rows + i];
    decode_row(row, (size_t)b * R + row, prob, b2, b3, rois, anchors, means, stds, out + ((size_t)b * n_rows + i) * 14);
}

extern "C" int m3d_decode_rows(const long long *rows, const float *prob, const float *bbox_2d, const float *bbox_3d,
                               const float *rois, const float *anchors, const float *means, const float *stds,
                               float *aboxes, int B, int R, int n_rows, m3d_stream_t stream)
{
    M3D_REQUIRE(rows && prob && bbox_2d && bbox_3d && rois && anchors && means && stds && aboxes && n_rows > 0,
                "decode_rows: bad arguments");
    hipLaunchKernelGGL(decode_rows_kernel, dim3(cdiv(n_rows, 256), B), dim3(256), 0, (hipStream_t)stream, rows, prob,
                       bbox_2d, bbox_3d, rois, anchors, means, stds, aboxes, R, n_rows);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
struct TopkArgs {
    const unsigned int *score_bits;   // [B][R] monotone (sortable) bits of the row score
    const float *prob, *b2, *b3, *rois, *anchors, *means, *stds;
    float *aboxes;                    // [B][k][14]
    int *rows_out;                    // [B][k] selected rows, score-descending (optional)
    unsigned long long *cand;         // workspace [B][2][R]
    const float *scale;               // [B] or null: test-time scale factor of each image (lib/rpn_util.py:1504-1506)
    int R, k;
    int A, HW;                        // planar form (A > 0): prob = cls planar [B][4A][HW], b2 = box planar [B][11][A*HW], b3 unused
};

// Exclusive prefix sum of one value per thread over the 1024-thread workgroup (16 waves): wave shuffles + one LDS hop.
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *wave_tot /*[16] LDS*/, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    unsigned inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(TOPK_NT) void topk_decode_kernel(TopkArgs a)
{
    __shared__ unsigned hist[2048];
    extern __shared__ __attribute__((aligned(16))) unsigned long long sel[];      // P = next power of two >= k keys
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned s_bin, s_above, s_nsel, s_ncand;
    const int img = blockIdx.x, tid = threadIdx.x;
    const int R = a.R, k = a.k;
    const unsigned int *sc = a.score_bits + (size_t)img * R;
    unsigned long long *candA = a.cand + (size_t)img * 2 * R, *candB = candA + R;

    // threshold bin of a 2048-bin histogram: the highest bin b with  count(digit > b) < need <= count(digit >= b)
    auto find_bin = [&](unsigned need) {
        const int j0 = 2047 - 2 * tid, j1 = j0 - 1;            // two bins per thread, walking down from the top bin
        const unsigned h0 = hist[j0], h1 = hist[j1];
        const unsigned ex = block_excl_scan(h0 + h1, wave_tot, tid);
        if (ex < need && need <= ex + h0) { s_bin = (unsigned)j0; s_above = ex; }
        else if (ex + h0 < need && need <= ex + h0 + h1) { s_bin = (unsigned)j1; s_above = ex + h0; }
        __syncthreads();
    };

    unsigned need = (unsigned)k;
    if (tid == 0) { s_nsel = 0; s_ncand = 0; }
    for (int i = tid; i < 2048; i += TOPK_NT) hist[i] = 0;
    __syncthreads();
    // ---- level 0: top 11 bits of the score over all R rows -----------------------------------------------------------
    const int R4 = R >> 2;
    const u32x4 *sc4 = reinterpret_cast<const u32x4 *>(sc);    // R*4 bytes per image: 16-byte aligned when R % 4 == 0
    const bool vec = (R & 3) == 0;
    if (vec) {
        for (int i = tid; i < R4; i += TOPK_NT) {
            const u32x4 v = sc4[i];
            atomicAdd(&hist[v[0] >> 21], 1u);
            atomicAdd(&hist[v[1] >> 21], 1u);
            atomicAdd(&hist[v[2] >> 21], 1u);
            atomicAdd(&hist[v[3] >> 21], 1u);
        }
    } else {
        for (int i = tid; i < R; i += TOPK_NT) atomicAdd(&hist[sc[i] >> 21], 1u);
    }
    __syncthreads();
    find_bin(need);
    {
        const unsigned bin = s_bin;
        auto put = [&](unsigned s, int row) {
            const unsigned d = s >> 21;
            if (d >= bin) {
                const unsigned long long key = ((unsigned long long)s << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)row);
                if (d > bin) sel[atomicAdd(&s_nsel, 1u)] = key;
                else candA[atomicAdd(&s_ncand, 1u)] = key;
            }
        };
        if (vec) {
            for (int i = tid; i < R4; i += TOPK_NT) {
                const u32x4 v = sc4[i];
                put(v[0], 4 * i); put(v[1], 4 * i + 1); put(v[2], 4 * i + 2); put(v[3], 4 * i + 3);
            }
        } else {
            for (int i = tid; i < R; i += TOPK_NT) put(sc[i], i);
        }
    }
    __threadfence_block();
    __syncthreads();
    need -= s_above;
    unsigned ncand = s_ncand;
    __syncthreads();                                           // everyone has read the counters before they are reset
    // ---- lower digits over the candidate list (global ping-pong; the workgroup is its only reader / writer) ---------
    // key bits 52..42, 41..32, then the row part: bits 31..22 are all ones for R < 2^22, so 21..11 and 10..0
    const int shifts[4] = {42, 32, 11, 0};
    const unsigned masks[4] = {0x7FFu, 0x3FFu, 0x7FFu, 0x7FFu};
    for (int lv = 0; lv < 4 && ncand != need; ++lv) {
        const int sh = shifts[lv];
        const unsigned mk = masks[lv];
        for (int i = tid; i < 2048; i += TOPK_NT) hist[i] = 0;
        if (tid == 0) s_ncand = 0;
        __syncthreads();
        for (unsigned i = tid; i < ncand; i += TOPK_NT) atomicAdd(&hist[(unsigned)(candA[i] >> sh) & mk], 1u);
        __syncthreads();
        find_bin(need);
        const unsigned bin = s_bin;
        for (unsigned i = tid; i < ncand; i += TOPK_NT) {
            const unsigned long long key = candA[i];
            const unsigned d = (unsigned)(key >> sh) & mk;
            if (d > bin) sel[atomicAdd(&s_nsel, 1u)] = key;
            else if (d == bin) candB[atomicAdd(&s_ncand, 1u)] = key;
        }
        __threadfence_block();
        __syncthreads();
        need -= s_above;
        ncand = s_ncand;
        unsigned long long *t = candA; candA = candB; candB = t;
        __syncthreads();
    }
    // the candidates left are exactly the rows still needed (keys are unique, so the walk always ends here)
    {
        const unsigned base = s_nsel;
        for (unsigned i = tid; i < need; i += TOPK_NT) sel[base + i] = candA[i];
    }
    // ---- bitonic sort of the k keys, descending (padding keys 0 sort last) -------------------------------------------
    int P = 1;
    while (P < k) P <<= 1;
    for (int i = k + tid; i < P; i += TOPK_NT) sel[i] = 0ULL;
    __syncthreads();
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += TOPK_NT) {
                const int lo = 2 * t - (t & (stride - 1));     // index with bit `stride` clear
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = sel[lo], y = sel[hi];
                if ((x < y) == desc) { sel[lo] = y; sel[hi] = x; }
            }
            __syncthreads();
        }
    }
    // ---- decode exactly the k selected rows --------------------------------------------------------------------------
    for (int i = tid; i < k; i += TOPK_NT) {
        const int row = (int)(0xFFFFFFFFu - (unsigned)sel[i]);
        if (a.rows_out) a.rows_out[(size_t)img * k + i] = row;
        float *q = a.aboxes + ((size_t)img * k + i) * 14;
        if (a.A > 0) decode_row_planar(row, img, a.A, a.HW, a.prob, a.b2, a.rois, a.anchors, a.means, a.stds, q);
        else decode_row(row, (size_t)img * R + row, a.prob, a.b2, a.b3, a.rois, a.anchors, a.means, a.stds, q);
        if (a.scale) {
            // `coords_2d[:, 0:4] /= scale_factor; coords_3d[:, 0:2] /= scale_factor` BEFORE the sort and the NMS, as the reference
            // does it (lib/rpn_util.py:1504-1506): the +1 convention of the NMS areas is not scale invariant, so an IoU next to
            // nms_thres can fall on the other side when the division comes after the NMS.  float32 divisions like torch's.
            const float sf = a.scale[img];
#pragma unroll
            for (int c = 0; c < 4; ++c) q[c] = __fdiv_rn(q[c], sf);
            q[6] = __fdiv_rn(q[6], sf);
            q[7] = __fdiv_rn(q[7], sf);
        }
    }
}

static int topk_launch(const TopkArgs &a, int B, hipStream_t stream)
{
    int P = 1;
    while (P < a.k) P <<= 1;
    const int lds = P * (int)sizeof(unsigned long long);             // <= 128 KB (+ 8 KB of static LDS)
    if (lds > 32768) {
        // The raised dynamic-LDS limit is a PER-DEVICE attribute of the kernel: a process that drives several GPUs (nn.DataParallel
        // replica engines, one engine per device) needs it on each of them -- set it on the current device, once per device ordinal.
        static std::atomic<int> state[64];                             // 0 unknown, 1 raised, 2 refused
        int dev = 0;
        M3D_HIP(hipGetDevice(&dev));
        int st = (dev >= 0 && dev < 64) ? state[dev].load(std::memory_order_acquire) : 0;
        if (st == 0) {
            const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&topk_decode_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                TOPK_MAXK * (int)sizeof(unsigned long long)) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            st = ok ? 1 : 2;
            if (dev >= 0 && dev < 64) state[dev].store(st, std::memory_order_release);
        }
        M3D_REQUIRE(st == 1, "topk_decode: device %d cannot reserve %d bytes of LDS for k = %d", dev, lds, a.k);
    }
    hipLaunchKernelGGL(topk_decode_kernel, dim3(B), dim3(TOPK_NT), lds, stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" long long m3d_topk_decode_workspace_bytes(int B, int R)
{
    return (long long)B * 2 * R * (long long)sizeof(unsigned long long);
}

extern "C" int m3d_topk_decode(const unsigned int *score_bits, const float *prob, const float *bbox_2d, const float *bbox_3d,
                               const float *rois, const float *anchors, const float *means, const float *stds, float *aboxes,
                               int *rows_out, void *workspace, long long workspace_bytes, int B, int R, int k,
                               m3d_stream_t stream)
{
    return m3d_topk_decode_scaled(score_bits, prob, bbox_2d, bbox_3d, rois, anchors, means, stds, nullptr, aboxes, rows_out, workspace,
                                  workspace_bytes, B, R, k, stream);
}

extern "C" int m3d_topk_decode_scaled(const unsigned int *score_bits, const float *prob, const float *bbox_2d, const float *bbox_3d,
                                      const float *rois, const float *anchors, const float *means, const float *stds,
                                      const float *scale, float *aboxes, int *rows_out, void *workspace, long long workspace_bytes,
                                      int B, int R, int k, m3d_stream_t stream)
{
    M3D_REQUIRE(score_bits && prob && bbox_2d && bbox_3d && rois && anchors && means && stds && aboxes && workspace,
                "topk_decode: null pointer");
    M3D_REQUIRE(B >= 1 && R >= 1 && R < (1 << 22), "topk_decode: R (%d) must be in [1, 2^22)", R);
    M3D_REQUIRE(k >= 1 && k <= R && k <= TOPK_MAXK, "topk_decode: k (%d) must be in [1, min(R, %d)]", k, TOPK_MAXK);
    if (workspace_bytes < m3d_topk_decode_workspace_bytes(B, R)) {
        m3d_set_error("topk_decode: workspace of %lld bytes, %lld needed", workspace_bytes, m3d_topk_decode_workspace_bytes(B, R));
        return M3D_E_WORKSPACE;
    }
    TopkArgs a;
    a.score_bits = score_bits; a.prob = prob; a.b2 = bbox_2d; a.b3 = bbox_3d; a.rois = rois; a.anchors = anchors;
    a.means = means; a.stds = stds; a.aboxes = aboxes; a.rows_out = rows_out; a.cand = (unsigned long long *)workspace;
    a.R = R; a.k = k; a.scale = scale; a.A = 0; a.HW = 0;
    return topk_launch(a, B, (hipStream_t)stream);
}

extern "C" int m3d_topk_decode_planar(const unsigned int *score_bits, const float *cls_planar, const float *box_planar,
                                      const float *rois, const float *anchors, const float *means, const float *stds,
                                      const float *scale, float *aboxes, int *rows_out, void *workspace, long long workspace_bytes,
                                      int B, int A, int HW, int k, m3d_stream_t stream)
{
    M3D_REQUIRE(score_bits && cls_planar && box_planar && rois && anchors && means && stds && aboxes && workspace,
                "topk_decode_planar: null pointer");
    M3D_REQUIRE(B >= 1 && A >= 1 && HW >= 1 && (long long)A * HW < (1 << 22), "topk_decode_planar: A * HW must be in [1, 2^22)");
    const int R = A * HW;
    M3D_REQUIRE(k >= 1 && k <= R && k <= TOPK_MAXK, "topk_decode_planar: k (%d) must be in [1, min(R, %d)]", k, TOPK_MAXK);
    if (workspace_bytes < m3d_topk_decode_workspace_bytes(B, R)) {
        m3d_set_error("topk_decode_planar: workspace of %lld bytes, %lld needed", workspace_bytes, m3d_topk_decode_workspace_bytes(B, R));
        return M3D_E_WORKSPACE;
    }
    TopkArgs a;
    a.score_bits = score_bits; a.prob = cls_planar; a.b2 = box_planar; a.b3 = nullptr; a.rois = rois; a.anchors = anchors;
    a.means = means; a.stds = stds; a.aboxes = aboxes; a.rows_out = rows_out; a.cand = (unsigned long long *)workspace;
    a.R = R; a.k = k; a.scale = scale; a.A = A; a.HW = HW;
    return topk_launch(a, B, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// Kept rows -> fixed-size blocks.  block [B][post + 1][14]: rows [0, min(num, post)) = aboxes[keep[j]], the rest zero;
// row `post` = (count, 0, ...).  counts [B] gets the same count as int32.
__global__ void select_post_kernel(const float *__restrict__ aboxes, const int *__restrict__ keep, const int *__restrict__ num,
                                   int n, int post, float *__restrict__ block, int *__restrict__ counts)
{
    const int b = blockIdx.x;
    const int cnt = min(num[b], post);
    for (int e = threadIdx.x; e < (post + 1) * 14; e += blockDim.x) {
        const int j = e / 14, c = e - j * 14;
        float v = 0.f;
        if (j < cnt) v = aboxes[((size_t)b * n + keep[(size_t)b * n + j]) * 14 + c];
        else if (j == post && c == 0) v = (float)cnt;
        block[(size_t)b * (post + 1) * 14 + e] = v;
    }
    if (threadIdx.x == 0 && counts) counts[b] = cnt;
}

extern "C" int m3d_select_post(const float *aboxes, const int *keep, const int *num_keep, int B, int n, int post, float *block,
                               int *counts, m3d_stream_t stream)
{
    M3D_REQUIRE(aboxes && keep && num_keep && block && B >= 1 && n >= 1 && post >= 1, "select_post: bad arguments");
    hipLaunchKernelGGL(select_post_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, aboxes, keep, num_keep, n, post, block,
                       counts);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
