/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the
 * product path (m3dssd_amd/, model/, lib/); only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * Plain-C CPU restatement of the reference's modulated deformable im2col, the
 * first half of dcn_v2_cuda_forward.  The reference has no CPU implementation
 * (model/DCNv2/src/dcn_v2.c:14 prints "only implemented in GPU") and its CUDA
 * sources cannot be built here (TH/THC + nvcc), so this file restates the
 * algorithm from the reference's kernel text:
 *
 *   bilinear sample ........ model/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:18-47
 *   im2col index math ...... model/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:129-178
 *   sample gate (> -1, < H)  model/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:165
 *   val * mask, col layout . model/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:174-175
 *   output size ............ model/DCNv2/src/dcn_v2_cuda.c:40-41
 *
 * The second half (bias via GEMM-with-ones, dcn_v2_cuda.c:72-78; weight GEMM
 * W[Co, C*kh*kw] x columns, dcn_v2_cuda.c:90-96) is done by the caller
 * (oracle/dcn.py) with a BLAS matmul, like the reference's cuBLAS call.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stddef.h>

/* dcn_v2_im2col_cuda.cu:18-47 */
static float bilinear_at(const float *plane, int data_width, int height, int width,
                         float h, float w)
{
    int h_low = (int)floorf(h);
    int w_low = (int)floorf(w);
    int h_high = h_low + 1;
    int w_high = w_low + 1;

    float lh = h - (float)h_low;
    float lw = w - (float)w_low;
    float hh = 1.0f - lh, hw = 1.0f - lw;

    float v1 = 0.0f, v2 = 0.0f, v3 = 0.0f, v4 = 0.0f;
    if (h_low >= 0 && w_low >= 0)
        v1 = plane[h_low * data_width + w_low];
    if (h_low >= 0 && w_high <= width - 1)
        v2 = plane[h_low * data_width + w_high];
    if (h_high <= height - 1 && w_low >= 0)
        v3 = plane[h_high * data_width + w_low];
    if (h_high <= height - 1 && w_high <= width - 1)
        v4 = plane[h_high * data_width + w_high];

    float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/*
 * One image (the reference calls its kernel with batch_size = 1 from a host
 * loop, dcn_v2_cuda.c:61,80).
 *   im      [C, H, W]
 *   offset  [dg * 2*kh*kw, Ho, Wo]   channel 2k = dh of tap k, 2k+1 = dw
 *   mask    [dg * kh*kw,   Ho, Wo]
 *   col     [C*kh*kw, Ho*Wo]         row index = c*kh*kw + i*kw + j
 */
void oracle_dcn_im2col(const float *im, const float *offset, const float *mask,
                       int C, int H, int W, int Ho, int Wo,
                       int kh, int kw, int pad_h, int pad_w,
                       int stride_h, int stride_w, int dil_h, int dil_w,
                       int deformable_group, float *col)
{
    const int cpg = C / deformable_group;
    const size_t plane_o = (size_t)Ho * Wo;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const int g = c / cpg;
        const float *im_c = im + (size_t)c * H * W;
        const float *off_g = offset + (size_t)g * 2 * kh * kw * plane_o;
        const float *msk_g = mask + (size_t)g * kh * kw * plane_o;
        for (int ho = 0; ho < Ho; ++ho) {
            for (int wo = 0; wo < Wo; ++wo) {
                const int h_in = ho * stride_h - pad_h;
                const int w_in = wo * stride_w - pad_w;
                for (int i = 0; i < kh; ++i) {
                    for (int j = 0; j < kw; ++j) {
                        const int k = i * kw + j;
                        const float dh = off_g[(size_t)(2 * k) * plane_o + (size_t)ho * Wo + wo];
                        const float dw = off_g[(size_t)(2 * k + 1) * plane_o + (size_t)ho * Wo + wo];
                        const float m = msk_g[(size_t)k * plane_o + (size_t)ho * Wo + wo];
                        float val = 0.0f;
                        const float h_im = (float)(h_in + i * dil_h) + dh;
                        const float w_im = (float)(w_in + j * dil_w) + dw;
                        if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                            val = bilinear_at(im_c, W, H, W, h_im, w_im);
                        col[((size_t)c * kh * kw + k) * plane_o + (size_t)ho * Wo + wo] = val * m;
                    }
                }
            }
        }
    }
}
