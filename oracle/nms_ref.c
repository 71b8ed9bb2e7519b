/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/dcn_im2col.c header for the rule).
 *
 * Plain-C CPU restatement of the reference's GPU NMS (CUDA, not buildable here):
 *   devIoU ("+1" pixel convention) ......... lib/nms/nms_kernel.cu:24-32
 *   64x64 tile suppression bitmask ......... lib/nms/nms_kernel.cu:34-78
 *     - diagonal tile starts at threadIdx+1   :66-69
 *     - suppress iff IoU  >  thresh (strict)  :71
 *   sequential greedy reduce on the host ... lib/nms/nms_kernel.cu:124-141
 *   caller-side descending score sort ...... lib/nms/gpu_nms.pyx:24-31
 *
 * All IoU arithmetic is IEEE fp32 with NO FMA contraction.  The expression DOES hold
 * products that feed an add/subtract -- interS = width*height goes into Sa + Sb - interS,
 * and Sa, Sb are themselves products (a2-a0+1)*(a3-a1+1) (nms_kernel.cu:27-31) -- so a
 * compiler free to contract would fuse them (fma(-width, height, Sa+Sb), ...) and change
 * the last bit.  This file, the device kernel (csrc/nms.hip) and the pinned target
 * (lib/nms/py_cpu_nms.py: numpy, one rounding per operation) all evaluate it unfused:
 * this file is built with -ffp-contract=off, csrc/nms.hip carries `#pragma clang fp contract(off)` around
 * the expression -- both are load-bearing.  (An nvcc build of
 * the reference may contract it -- nvcc's default is --fmad=true; there is no such build
 * to compare with here, so parity is pinned on the unfused numpy form.)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TPB 64
#define DIVUP(m, n) ((m) / (n) + ((m) % (n) > 0))

static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline float fmin2(float a, float b) { return a < b ? a : b; }

/* nms_kernel.cu:24-32 */
float oracle_iou(const float *a, const float *b)
{
    float left = fmax2(a[0], b[0]), right = fmin2(a[2], b[2]);
    float top = fmax2(a[1], b[1]), bottom = fmin2(a[3], b[3]);
    float width = fmax2(right - left + 1, 0.f), height = fmax2(bottom - top + 1, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

/*
 * boxes: [n, dim>=4] ALREADY sorted by descending score (what _nms receives,
 * gpu_nms.pyx:28-29).  keep_out sized n; returns indices into the sorted array.
 * Mirrors _nms's signature minus the device id (nms_kernel.cu:91-92).
 */
void oracle_nms_sorted(int *keep_out, int *num_out, const float *boxes, int n, int dim,
                       float thresh)
{
    const int col_blocks = DIVUP(n, TPB);
    uint64_t *mask = (uint64_t *)calloc((size_t)n * col_blocks + 1, sizeof(uint64_t));
    /* nms_kernel.cu:34-78: one (row tile, col tile) pair per block */
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < n; ++i) {
        const int row_tile = i / TPB, tid = i % TPB;
        for (int cb = 0; cb < col_blocks; ++cb) {
            const int col_size = (n - cb * TPB) < TPB ? (n - cb * TPB) : TPB;
            int start = (row_tile == cb) ? tid + 1 : 0;
            uint64_t t = 0;
            for (int j = start; j < col_size; ++j)
                if (oracle_iou(boxes + (size_t)i * dim, boxes + (size_t)(cb * TPB + j) * dim) > thresh)
                    t |= 1ULL << j;
            mask[(size_t)i * col_blocks + cb] = t;
        }
    }
    /* nms_kernel.cu:124-141 */
    uint64_t *remv = (uint64_t *)calloc(col_blocks + 1, sizeof(uint64_t));
    int num_to_keep = 0;
    for (int i = 0; i < n; ++i) {
        int nblock = i / TPB, inblock = i % TPB;
        if (!(remv[nblock] & (1ULL << inblock))) {
            keep_out[num_to_keep++] = i;
            const uint64_t *p = mask + (size_t)i * col_blocks;
            for (int j = nblock; j < col_blocks; ++j)
                remv[j] |= p[j];
        }
    }
    *num_out = num_to_keep;
    free(mask);
    free(remv);
}
