// Greedy NMS on the device (lib/nms/nms_kernel.cu restated for wave64, zero host round trips).
//
//  * nms_mask_kernel: one wave per (row tile, col tile) of 64x64 boxes, upper triangle only (the
//    greedy pass never reads tiles left of the diagonal, nms_kernel.cu:133-137).  IoU uses the
//    reference's exact fp32 expression with the "+1" pixel convention and a strict '>' test
//    (nms_kernel.cu:24-32,71); this file is compiled with -ffp-contract=off and HIP's default
//    correctly-rounded fp32 division, so kept indices are bit-identical to the CPU oracle.
//  * nms_reduce_kernel: one workgroup per image replaces the host loop of nms_kernel.cu:124-141.
//    Lane j of wave 0 owns the 64-bit "removed" word of column tile j (n <= 4096; nms_reduce_big_kernel up to 16384; the host-pointer
//    twin `_nms` takes larger inputs over the reference's own route: device masks + greedy pass on the host).  Per row tile: the
//    64 sequential decisions run on scalar-broadcast words (v_readlane) while the four waves already
//    hold the tile's 64 mask rows (prefetched one tile ahead) and OR in those whose box was kept.
#include <stdlib.h>

#include "common.h"

#define NMS_TPB 64

__device__ __forceinline__ float dev_iou(const float *a, const float *b)
{
#pragma clang fp contract(off)   // Sa + Sb must not become fma(w, h, Sb): keep IEEE op-by-op like the oracle
    const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
    const float interS = width * height;
    const float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    const float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

// Mask tile without the division in all but a sliver of cases, same bits as dev_iou(a, b) > thresh (correctly rounded fp32 division,
// strict '>'): q = RN(interS / U) > t holds when interS >= next(t) * U and fails when interS <= t * U in the reals; p = RN(t * U)
// is within 2^-24 of t * U (p normal), so outside the band p * (1 -+ 2^-20) the comparison of interS with p decides.  A lane (row
// box) that meets a pair inside the band -- or a non-positive U or threshold (degenerate boxes) -- redoes its row with the exact
// expression afterwards.  The 64 column boxes (+ their areas) sit in LDS; the loop is fully unrolled and branch-free, the "redo"
// flags accumulate in an SGPR pair.  v_max / v_min are issued directly: on loaded values fmaxf costs a canonicalising extra
// instruction each (same result for the non-NaN boxes the exact path does not take over: a NaN makes the band test fail).
__device__ __forceinline__ float nms_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float nms_vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

__global__ __launch_bounds__(NMS_TPB) void nms_mask_kernel(int n, int box_stride, float thresh,
                                                          const float *__restrict__ boxes_all,
                                                          unsigned long long *__restrict__ mask_all)
{
#pragma clang fp contract(off)
    const int row_start = blockIdx.y, col_start = blockIdx.x, img = blockIdx.z;
    if (col_start < row_start) return;
    const int col_blocks = (n + NMS_TPB - 1) / NMS_TPB;
    const float *boxes = boxes_all + (size_t)img * n * box_stride;
    unsigned long long *mask = mask_all + (size_t)img * n * col_blocks;
    const int row_size = min(n - row_start * NMS_TPB, NMS_TPB);
    const int col_size = min(n - col_start * NMS_TPB, NMS_TPB);
    __shared__ __attribute__((aligned(16))) float bb[NMS_TPB * 4];
    __shared__ float bs[NMS_TPB];
    const int t = threadIdx.x;
    {
        float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f;            // past the end: a unit box (its bits are masked off below)
        if (t < col_size) {
            const float *s = boxes + (size_t)(NMS_TPB * col_start + t) * box_stride;
            x0 = s[0]; y0 = s[1]; x1 = s[2]; y1 = s[3];
        }
        *reinterpret_cast<f32x4 *>(bb + t * 4) = f32x4{x0, y0, x1, y1};
        bs[t] = (x1 - x0 + 1) * (y1 - y0 + 1);
    }
    __syncthreads();
    const int cur = NMS_TPB * row_start + min(t, row_size - 1);
    const float *cp = boxes + (size_t)cur * box_stride;
    const float cb[4] = {cp[0], cp[1], cp[2], cp[3]};
    const int start = (row_start == col_start) ? t + 1 : 0;
    unsigned long long bits = 0;
    unsigned long long redo = thresh > 0.f ? 0ULL : ~0ULL;       // wave-uniform lane mask
    if (redo == 0) {
        const float Sa = (cb[2] - cb[0] + 1) * (cb[3] - cb[1] + 1);
        unsigned int lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < NMS_TPB; ++i) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(bb + i * 4);
            const float Sb = bs[i];
            const float w = fmaxf(nms_vmin(cb[2], b[2]) - nms_vmax(cb[0], b[0]) + 1, 0.f);
            const float h = fmaxf(nms_vmin(cb[3], b[3]) - nms_vmax(cb[1], b[1]) + 1, 0.f);
            const float interS = w * h;
            const float U = Sa + Sb - interS;
            const float p = thresh * U;
            const bool decided = (p > 1e-30f) & (fabsf(interS - p) > p * 9.5367431640625e-07f);   // 2^-20
            redo |= __builtin_amdgcn_ballot_w64(!decided);
            if (i < 32) lo |= interS > p ? 1u << (i & 31) : 0u;
            else hi |= interS > p ? 1u << (i & 31) : 0u;
        }
        bits = ((unsigned long long)hi << 32) | lo;
    }
    if (redo >> t & 1ULL) {                                         // rare: this row with the reference's own expression
        bits = 0;
        for (int i = 0; i < col_size; ++i)
            if (dev_iou(cb, bb + i * 4) > thresh) bits |= 1ULL << i;
    }
    if (t < row_size) {
        unsigned long long valid = col_size < NMS_TPB ? (1ULL << col_size) - 1ULL : ~0ULL;
        valid &= start < NMS_TPB ? ~0ULL << start : 0ULL;
        mask[(size_t)cur * col_blocks + col_start] = bits & valid;
    }
}

// Round 1-4 form (one correctly rounded division per pair), kept as the A/B baseline: M3D_NMS_DIV=1.
__global__ __launch_bounds__(NMS_TPB) void nms_mask_div_kernel(int n, int box_stride, float thresh,
                                                              const float *__restrict__ boxes_all,
                                                              unsigned long long *__restrict__ mask_all)
{
    const int row_start = blockIdx.y, col_start = blockIdx.x, img = blockIdx.z;
    if (col_start < row_start) return;
    const int col_blocks = (n + NMS_TPB - 1) / NMS_TPB;
    const float *boxes = boxes_all + (size_t)img * n * box_stride;
    unsigned long long *mask = mask_all + (size_t)img * n * col_blocks;
    const int row_size = min(n - row_start * NMS_TPB, NMS_TPB);
    const int col_size = min(n - col_start * NMS_TPB, NMS_TPB);
    __shared__ float bb[NMS_TPB * 4];
    const int t = threadIdx.x;
    if (t < col_size) {
        const float *s = boxes + (size_t)(NMS_TPB * col_start + t) * box_stride;
        bb[t * 4 + 0] = s[0]; bb[t * 4 + 1] = s[1]; bb[t * 4 + 2] = s[2]; bb[t * 4 + 3] = s[3];
    }
    __syncthreads();
    if (t < row_size) {
        const int cur = NMS_TPB * row_start + t;
        const float *cp = boxes + (size_t)cur * box_stride;
        const float cb[4] = {cp[0], cp[1], cp[2], cp[3]};
        unsigned long long bits = 0;
        const int start = (row_start == col_start) ? t + 1 : 0;
        for (int i = start; i < col_size; ++i)
            if (dev_iou(cb, bb + i * 4) > thresh) bits |= 1ULL << i;
        mask[(size_t)cur * col_blocks + col_start] = bits;
    }
}

// One workgroup (4 waves) per image.  Wave 0 runs the inherently sequential greedy decisions of a 64-box tile on
// scalar-broadcast words; all four waves stream the tile's 64 suppression rows (16 rows each, lane = column tile) and
// the rows of tile blk+1 are already in flight while tile blk is being decided, so memory latency is off the
// critical path (the one-wave version paid two dependent latencies per tile: 264 us for 3000 boxes).
__global__ __launch_bounds__(256) void nms_reduce_kernel(int n, const unsigned long long *__restrict__ mask_all,
                                                         int *__restrict__ keep_all, int *__restrict__ num_keep)
{
    __shared__ unsigned long long diag_s[64 * NMS_TPB];     // mask[i][i / 64] for every box
    __shared__ unsigned long long part_s[4][64];
    __shared__ unsigned long long keep_s;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col_blocks = (n + NMS_TPB - 1) / NMS_TPB;
    const unsigned long long *mask = mask_all + (size_t)img * n * col_blocks;
    int *keep = keep_all + (size_t)img * n;

    for (int i = tid; i < n; i += 256) diag_s[i] = mask[(size_t)i * col_blocks + i / NMS_TPB];

    unsigned long long rows[16], rown[16];
    auto load_rows = [&](int blk, unsigned long long (&dst)[16]) {
        const int nb = min(NMS_TPB, n - blk * NMS_TPB);
        const bool col_ok = lane > blk && lane < col_blocks;
        const unsigned long long *rp = mask + (size_t)(blk * NMS_TPB + wave * 16) * col_blocks + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dst[r] = (col_ok && wave * 16 + r < nb) ? rp[(size_t)r * col_blocks] : 0ULL;
    };
    load_rows(0, rows);
    unsigned long long remv = 0;   // wave 0, lane j: removed bits of column tile j
    int base = 0;
    __syncthreads();
    for (int blk = 0; blk < col_blocks; ++blk) {
        const int nb = min(NMS_TPB, n - blk * NMS_TPB);
        if (blk + 1 < col_blocks) load_rows(blk + 1, rown);
        if (wave == 0) {
            const unsigned long long diag = lane < nb ? diag_s[blk * NMS_TPB + lane] : 0ULL;
            const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
            const unsigned int rlo = (unsigned int)remv, rhi = (unsigned int)(remv >> 32);
            unsigned long long r = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)rhi, blk) << 32) |
                                   (unsigned int)__builtin_amdgcn_readlane((int)rlo, blk);
            // boxes past the end of a ragged last tile count as already removed; the 64 decisions are fully unrolled
            // (constant lane index for v_readlane, no loop-carried branch): a short scalar chain per box
            if (nb < NMS_TPB) r |= ~0ULL << nb;
            unsigned long long keepbits = 0;
#pragma unroll
            for (int i = 0; i < NMS_TPB; ++i) {
                const unsigned long long d = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
                                             (unsigned int)__builtin_amdgcn_readlane((int)dlo, i);
                const bool alive = !((r >> i) & 1ULL);
                keepbits |= alive ? (1ULL << i) : 0ULL;
                r |= alive ? d : 0ULL;
            }
            if ((keepbits >> lane) & 1ULL)
                keep[base + __popcll(keepbits & ((1ULL << lane) - 1ULL))] = blk * NMS_TPB + lane;
            base += __popcll(keepbits);
            if (lane == 0) keep_s = keepbits;
        }
        __syncthreads();
        const unsigned long long kb = keep_s >> (wave * 16);
        unsigned long long acc = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc |= ((kb >> r) & 1ULL) ? rows[r] : 0ULL;
        part_s[wave][lane] = acc;
        __syncthreads();
        if (wave == 0) remv |= part_s[0][lane] | part_s[1][lane] | part_s[2][lane] | part_s[3][lane];
#pragma unroll
        for (int r = 0; r < 16; ++r) rows[r] = rown[r];
    }
    if (tid == 0) num_keep[img] = base;
}

// More than 4096 boxes per image (up to 16384: not the path's configuration -- nms_topN_pre is 3000 -- but the reference has no
// limit): the same greedy pass with the "removed" words in LDS instead of one per lane, the diagonal word of a tile read when
// the tile is decided, and the 256 threads OR-ing the kept rows into one column tile each.  No prefetch: a few ms at 16384 boxes.
__global__ __launch_bounds__(256) void nms_reduce_big_kernel(int n, const unsigned long long *__restrict__ mask_all,
                                                             int *__restrict__ keep_all, int *__restrict__ num_keep)
{
    __shared__ unsigned long long remv_s[256];
    __shared__ unsigned long long keep_s;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col_blocks = (n + NMS_TPB - 1) / NMS_TPB;
    const unsigned long long *mask = mask_all + (size_t)img * n * col_blocks;
    int *keep = keep_all + (size_t)img * n;
    remv_s[tid] = 0ULL;
    int base = 0;
    __syncthreads();
    for (int blk = 0; blk < col_blocks; ++blk) {
        const int nb = min(NMS_TPB, n - blk * NMS_TPB);
        if (wave == 0) {
            const unsigned long long diag = lane < nb ? mask[(size_t)(blk * NMS_TPB + lane) * col_blocks + blk] : 0ULL;
            const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
            unsigned long long r = remv_s[blk];
            if (nb < NMS_TPB) r |= ~0ULL << nb;
            unsigned long long keepbits = 0;
#pragma unroll
            for (int i = 0; i < NMS_TPB; ++i) {
                const unsigned long long d = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
                                             (unsigned int)__builtin_amdgcn_readlane((int)dlo, i);
                const bool alive = !((r >> i) & 1ULL);
                keepbits |= alive ? (1ULL << i) : 0ULL;
                r |= alive ? d : 0ULL;
            }
            if ((keepbits >> lane) & 1ULL)
                keep[base + __popcll(keepbits & ((1ULL << lane) - 1ULL))] = blk * NMS_TPB + lane;
            base += __popcll(keepbits);
            if (lane == 0) keep_s = keepbits;
        }
        __syncthreads();
        const unsigned long long kb = keep_s;
        for (int c = blk + 1 + tid; c < col_blocks; c += 256) {
            unsigned long long acc = 0ULL;
            for (int r = 0; r < nb; ++r)
                if ((kb >> r) & 1ULL) acc |= mask[(size_t)(blk * NMS_TPB + r) * col_blocks + c];
            remv_s[c] |= acc;
        }
        __syncthreads();
    }
    if (tid == 0) num_keep[img] = base;
}

extern "C" long long m3d_nms_workspace_bytes(int B, int n)
{
    const long long cb = (n + NMS_TPB - 1) / NMS_TPB;
    return (long long)B * n * cb * (long long)sizeof(unsigned long long);
}

extern "C" int m3d_nms_sorted_dev(const float *boxes_dev, int B, int n, int box_stride, float thresh, void *mask_ws,
                                  int *keep_dev, int *num_keep_dev, m3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    M3D_REQUIRE(num_keep_dev && B >= 1, "nms: null pointer / bad batch");
    M3D_REQUIRE(n >= 0 && n <= 256 * NMS_TPB, "nms: n (%d) must be <= 16384", n);
    if (n == 0) {   // empty input: nothing kept (boxes/keep may legitimately be null)
        M3D_HIP(hipMemsetAsync(num_keep_dev, 0, sizeof(int) * B, stream));
        return M3D_OK;
    }
    M3D_REQUIRE(boxes_dev && mask_ws && keep_dev, "nms: null pointer");
    M3D_REQUIRE(box_stride >= 4, "nms: box_stride must be >= 4");
    const int cb = (n + NMS_TPB - 1) / NMS_TPB;
    static const int div_form = []() { const char *e = getenv("M3D_NMS_DIV"); return e ? atoi(e) : 0; }();
    if (div_form) hipLaunchKernelGGL(nms_mask_div_kernel, dim3(cb, cb, B), dim3(NMS_TPB), 0, stream, n, box_stride, thresh, boxes_dev,
                                     (unsigned long long *)mask_ws);
    else hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, B), dim3(NMS_TPB), 0, stream, n, box_stride, thresh, boxes_dev,
                            (unsigned long long *)mask_ws);
    M3D_LAUNCH_CHECK();
    if (n <= 64 * NMS_TPB)
        hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(256), 0, stream, n, (const unsigned long long *)mask_ws, keep_dev, num_keep_dev);
    else
        hipLaunchKernelGGL(nms_reduce_big_kernel, dim3(B), dim3(256), 0, stream, n, (const unsigned long long *)mask_ws, keep_dev, num_keep_dev);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// Exact twin of the reference's host-pointer entry (lib/nms/gpu_nms.hpp:1-2, nms_kernel.cu:91-144):
// boxes_host sorted by descending score, keep_out sized boxes_num, synchronous, errors to stdout only.
extern "C" void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
                     float nms_overlap_thresh, int device_id)
{
    *num_out = 0;
    if (boxes_num <= 0) return;
    if (boxes_dim < 4) { printf("_nms: boxes_dim = %d, need x1, y1, x2, y2 (>= 4 columns)\n", boxes_dim); return; }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device_id) {
        if (hipSetDevice(device_id) != hipSuccess) { printf("_nms: hipSetDevice(%d) failed\n", device_id); return; }
    }
    float *boxes_dev = nullptr;
    void *mask_dev = nullptr;
    int *keep_dev = nullptr, *num_dev = nullptr;
    const size_t bbytes = (size_t)boxes_num * boxes_dim * sizeof(float);
    // every HIP failure records its own message: the report at the end must not print a stale one
    auto hip_ok = [](hipError_t e, const char *what) {
        if (e != hipSuccess) m3d_set_error("_nms: %s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    bool ok = hip_ok(hipMalloc(&boxes_dev, bbytes), "hipMalloc(boxes)") &&
              hip_ok(hipMalloc(&mask_dev, (size_t)m3d_nms_workspace_bytes(1, boxes_num)), "hipMalloc(mask)") &&
              hip_ok(hipMalloc(&keep_dev, sizeof(int) * boxes_num), "hipMalloc(keep)") &&
              hip_ok(hipMalloc(&num_dev, sizeof(int)), "hipMalloc(num)");
    ok = ok && hip_ok(hipMemcpy(boxes_dev, boxes_host, bbytes, hipMemcpyHostToDevice), "hipMemcpy(boxes -> device)");
    if (ok && boxes_num > 256 * NMS_TPB) {
        // The reference has no row limit (nms_kernel.cu:91-144); the on-device greedy reduce keeps one 64-bit "removed" word per
        // lane (4096 boxes) or in LDS (16384).  Larger inputs take the reference's own route: bitmask tiles on the device, greedy pass on the host
        // over the upper triangle (nms_kernel.cu:124-141) -- the only words the pass reads are the ones the mask kernel writes.
        const int cb = (boxes_num + NMS_TPB - 1) / NMS_TPB;
        hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, 1), dim3(NMS_TPB), 0, nullptr, boxes_num, boxes_dim, nms_overlap_thresh,
                           boxes_dev, (unsigned long long *)mask_dev);
        unsigned long long *mask_host = (unsigned long long *)malloc((size_t)boxes_num * cb * sizeof(unsigned long long));
        unsigned long long *remv = (unsigned long long *)calloc(cb, sizeof(unsigned long long));
        ok = mask_host && remv && hip_ok(hipGetLastError(), "mask kernel launch") &&
             hip_ok(hipMemcpy(mask_host, mask_dev, (size_t)boxes_num * cb * sizeof(unsigned long long), hipMemcpyDeviceToHost),
                    "hipMemcpy(mask -> host)");
        int kept = 0;
        for (int i = 0; ok && i < boxes_num; ++i) {
            const int nblock = i / NMS_TPB, inblock = i % NMS_TPB;
            if (!(remv[nblock] & (1ULL << inblock))) {
                keep_out[kept++] = i;
                const unsigned long long *p = mask_host + (size_t)i * cb;
                for (int j = nblock; j < cb; ++j) remv[j] |= p[j];
            }
        }
        if (ok) *num_out = kept;
        else if (!mask_host || !remv) m3d_set_error("_nms: out of host memory");
        free(mask_host); free(remv);
    } else {
        ok = ok && m3d_nms_sorted_dev(boxes_dev, 1, boxes_num, boxes_dim, nms_overlap_thresh, mask_dev, keep_dev, num_dev,
                                      nullptr) == M3D_OK;
        ok = ok && hip_ok(hipMemcpy(num_out, num_dev, sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy(num -> host)");
        ok = ok && hip_ok(hipMemcpy(keep_out, keep_dev, sizeof(int) * (*num_out), hipMemcpyDeviceToHost), "hipMemcpy(keep -> host)");
    }
    if (!ok) { printf("_nms: %s\n", m3d_last_error()); *num_out = 0; }
    (void)hipFree(boxes_dev); (void)hipFree(mask_dev); (void)hipFree(keep_dev); (void)hipFree(num_dev);
}
