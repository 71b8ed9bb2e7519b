// Greedy NMS on the device (lib/nms/nms_kernel.cu restated for wave64, zero host round trips).
//
//  * nms_mask_kernel: one wave per (row tile, col tile) of 64x64 boxes, upper triangle only (the
//    greedy pass never reads tiles left of the diagonal, nms_kernel.cu:133-137).  IoU uses the
//    reference's exact fp32 expression with the "+1" pixel convention and a strict '>' test
//    (nms_kernel.cu:24-32,71); this file is compiled with -ffp-contract=off and HIP's default
//    correctly-rounded fp32 division, so kept indices are bit-identical to the CPU oracle.
//  * nms_reduce_kernel: one wave per image replaces the host loop of nms_kernel.cu:124-141.
//    Lane j owns the 64-bit "removed" word of column tile j (n <= 4096).  Per row tile: the 64
//    sequential decisions run on scalar-broadcast words (v_readlane), then all 64 mask rows of
//    the tile are streamed (independent loads) and OR-ed in if their box was kept.
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

__global__ __launch_bounds__(NMS_TPB) void nms_mask_kernel(int n, int box_stride, float thresh,
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

__global__ __launch_bounds__(64) void nms_reduce_kernel(int n, const unsigned long long *__restrict__ mask_all,
                                                        int *__restrict__ keep_all, int *__restrict__ num_keep)
{
    const int img = blockIdx.x, lane = threadIdx.x;
    const int col_blocks = (n + NMS_TPB - 1) / NMS_TPB;
    const unsigned long long *mask = mask_all + (size_t)img * n * col_blocks;
    int *keep = keep_all + (size_t)img * n;
    unsigned long long remv = 0;   // lane j: removed bits of column tile j
    int base = 0;
    for (int blk = 0; blk < col_blocks; ++blk) {
        const int nb = min(NMS_TPB, n - blk * NMS_TPB);
        const unsigned long long diag = lane < nb ? mask[(size_t)(blk * NMS_TPB + lane) * col_blocks + blk] : 0ULL;
        const unsigned int dlo = (unsigned int)diag, dhi = (unsigned int)(diag >> 32);
        const unsigned int rlo = (unsigned int)remv, rhi = (unsigned int)(remv >> 32);
        unsigned long long r = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)rhi, blk) << 32) |
                               (unsigned int)__builtin_amdgcn_readlane((int)rlo, blk);
        unsigned long long keepbits = 0;
        for (int i = 0; i < nb; ++i) {
            const unsigned long long d = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
                                         (unsigned int)__builtin_amdgcn_readlane((int)dlo, i);
            if (!((r >> i) & 1ULL)) {
                keepbits |= 1ULL << i;
                r |= d;
            }
        }
        if ((keepbits >> lane) & 1ULL)
            keep[base + __popcll(keepbits & ((1ULL << lane) - 1ULL))] = blk * NMS_TPB + lane;
        base += __popcll(keepbits);
        // propagate the kept boxes' suppression rows to the later column tiles: all 64 row words of this lane's
        // column are fetched with independent loads (one memory latency per tile, not 64), then OR-ed if kept
        const int j = lane;
        if (j > blk && j < col_blocks) {
            const unsigned long long *rowp = mask + (size_t)(blk * NMS_TPB) * col_blocks + j;
            unsigned long long rows[NMS_TPB];
#pragma unroll
            for (int i = 0; i < NMS_TPB; ++i) rows[i] = (i < nb) ? rowp[(size_t)i * col_blocks] : 0ULL;
            unsigned long long acc = 0;
#pragma unroll
            for (int i = 0; i < NMS_TPB; ++i) acc |= ((keepbits >> i) & 1ULL) ? rows[i] : 0ULL;
            remv |= acc;
        }
    }
    if (lane == 0) num_keep[img] = base;
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
    M3D_REQUIRE(n >= 0 && n <= 64 * NMS_TPB, "nms: n (%d) must be <= 4096", n);
    if (n == 0) {   // empty input: nothing kept (boxes/keep may legitimately be null)
        M3D_HIP(hipMemsetAsync(num_keep_dev, 0, sizeof(int) * B, stream));
        return M3D_OK;
    }
    M3D_REQUIRE(boxes_dev && mask_ws && keep_dev, "nms: null pointer");
    M3D_REQUIRE(box_stride >= 4, "nms: box_stride must be >= 4");
    const int cb = (n + NMS_TPB - 1) / NMS_TPB;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb, B), dim3(NMS_TPB), 0, stream, n, box_stride, thresh, boxes_dev,
                       (unsigned long long *)mask_ws);
    M3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(B), dim3(64), 0, stream, n, (const unsigned long long *)mask_ws, keep_dev,
                       num_keep_dev);
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
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device_id) {
        if (hipSetDevice(device_id) != hipSuccess) { printf("_nms: hipSetDevice(%d) failed\n", device_id); return; }
    }
    float *boxes_dev = nullptr;
    void *mask_dev = nullptr;
    int *keep_dev = nullptr, *num_dev = nullptr;
    const size_t bbytes = (size_t)boxes_num * boxes_dim * sizeof(float);
    bool ok = hipMalloc(&boxes_dev, bbytes) == hipSuccess &&
              hipMalloc(&mask_dev, (size_t)m3d_nms_workspace_bytes(1, boxes_num)) == hipSuccess &&
              hipMalloc(&keep_dev, sizeof(int) * boxes_num) == hipSuccess &&
              hipMalloc(&num_dev, sizeof(int)) == hipSuccess;
    ok = ok && hipMemcpy(boxes_dev, boxes_host, bbytes, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && m3d_nms_sorted_dev(boxes_dev, 1, boxes_num, boxes_dim, nms_overlap_thresh, mask_dev, keep_dev, num_dev,
                                  nullptr) == M3D_OK;
    ok = ok && hipMemcpy(num_out, num_dev, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && hipMemcpy(keep_out, keep_dev, sizeof(int) * (*num_out), hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) { printf("_nms: %s\n", m3d_last_error()); *num_out = 0; }
    (void)hipFree(boxes_dev); (void)hipFree(mask_dev); (void)hipFree(keep_dev); (void)hipFree(num_dev);
}
