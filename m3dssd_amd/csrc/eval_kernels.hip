// KITTI AP evaluator natives.
//
//  * m3d_rotate_iou_eval: the reference's numba.cuda kernel `rotate_iou_kernel_eval` (lib/eval/rotate_iou.py:12-262,
//    launched by rotate_iou_gpu_eval :264-326) as a HIP kernel: one thread per (box, query box) pair, float32 corner /
//    intersection / vertex-sort arithmetic, the triangle fan summed in float64 (numba promotes at `/ 2.0`), ratio stored as
//    float32.  Contraction is off so every product is rounded like the reference's scalar float32 code.
//  * m3d_eval_statistics / m3d_eval_fused_statistics: `compute_statistics_jit` and `fused_compute_statistics`
//    (lib/eval/eval.py:152-333), the greedy GT <-> detection matching that the reference compiles with numba; host C++
//    (sequential per image, a few hundred operations per pair).
//  * m3d_eval_image_box_overlap, m3d_eval_d3_overlap: `image_box_overlap` and `d3_box_overlap_kernel` (eval.py:84-141), host.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"

#pragma clang fp contract(off)

__device__ __forceinline__ double tri_area(const float *a, const float *b, const float *c)
{
    return (double)((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0;
}

__device__ void rbbox_corners(float *corners, const float *rb)
{
    const float angle = rb[4];
    const float a_cos = cosf(angle), a_sin = sinf(angle);
    const float cx = rb[0], cy = rb[1], xd = rb[2], yd = rb[3];
    const float xs[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2};
    const float ys[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        corners[2 * i] = a_cos * xs[i] + a_sin * ys[i] + cx;
        corners[2 * i + 1] = -a_sin * xs[i] + a_cos * ys[i] + cy;
    }
}

__device__ bool point_in_quad(float px, float py, const float *c)
{
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
    const float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

__device__ bool segment_intersection(const float *p1, const float *p2, int i, int j, float *out)
{
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1];
    const float B0 = p1[2 * ((i + 1) & 3)], B1 = p1[2 * ((i + 1) & 3) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1];
    const float D0 = p2[2 * ((j + 1) & 3)], D1 = p2[2 * ((j + 1) & 3) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const bool acd = DA1 * CA0 > CA1 * DA0;
    const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        const bool abc = CA1 * BA0 > BA1 * CA0;
        const bool abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            const float DC0 = D0 - C0, DC1 = D1 - C1;
            const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
            const float DH = BA1 * DC0 - BA0 * DC1;
            const float Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
            out[0] = Dx / DH;
            out[1] = Dy / DH;
            return true;
        }
    }
    return false;
}

__device__ double rotated_intersection(const float *rb1, const float *rb2)
{
    float c1[8], c2[8], pts[16];
    rbbox_corners(c1, rb1);
    rbbox_corners(c2, rb2);
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        if (point_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { pts[2 * n] = c1[2 * i]; pts[2 * n + 1] = c1[2 * i + 1]; ++n; }
        if (point_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { pts[2 * n] = c2[2 * i]; pts[2 * n + 1] = c2[2 * i + 1]; ++n; }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float t[2];
            // two convex quadrilaterals meet in at most 8 vertices; the reference's 16-float scratch has no room for more
            // (coincident corners of identical boxes would be appended a second time there: duplicates add no area)
            if (segment_intersection(c1, c2, i, j, t) && n < 8) { pts[2 * n] = t[0]; pts[2 * n + 1] = t[1]; ++n; }
        }
    if (n > 0) {                                   // sort_vertex_in_convex_polygon
        float cx = 0.f, cy = 0.f;
        for (int i = 0; i < n; ++i) { cx += pts[2 * i]; cy += pts[2 * i + 1]; }
        cx = (float)((double)cx / n);
        cy = (float)((double)cy / n);
        float vs[8];
        for (int i = 0; i < n; ++i) {
            float v0 = pts[2 * i] - cx, v1 = pts[2 * i + 1] - cy;
            const float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d;
            v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i) {
            if (vs[i - 1] > vs[i]) {
                const float temp = vs[i], tx = pts[2 * i], ty = pts[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > temp) {
                    vs[j] = vs[j - 1];
                    pts[2 * j] = pts[2 * j - 2];
                    pts[2 * j + 1] = pts[2 * j - 1];
                    --j;
                }
                vs[j] = temp;
                pts[2 * j] = tx;
                pts[2 * j + 1] = ty;
            }
        }
    }
    double area = 0.0;
    for (int i = 0; i < n - 2; ++i) area += fabs(tri_area(pts, pts + 2 * i + 2, pts + 2 * i + 4));
    return area;
}

__global__ void rotate_iou_eval_kernel(const float *__restrict__ boxes, int N, const float *__restrict__ qboxes, int K,
                                       int criterion, float *__restrict__ iou)
{
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= (long long)N * K) return;
    const int n = (int)(idx / K), k = (int)(idx - (long long)n * K);
    float r1[5], r2[5];                            // devRotateIoUEval(query box, box): rbox1 = query box
#pragma unroll
    for (int e = 0; e < 5; ++e) { r1[e] = qboxes[(size_t)k * 5 + e]; r2[e] = boxes[(size_t)n * 5 + e]; }
    const float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
    const double ai = rotated_intersection(r1, r2);
    double v;
    if (criterion == -1) v = ai / ((double)(a1 + a2) - ai);
    else if (criterion == 0) v = ai / (double)a1;
    else if (criterion == 1) v = ai / (double)a2;
    else v = ai;
    iou[idx] = (float)v;
}

extern "C" int m3d_rotate_iou_eval(const float *boxes_dev, int N, const float *qboxes_dev, int K, int criterion, float *iou_dev,
                                   m3d_stream_t stream)
{
    M3D_REQUIRE(N >= 0 && K >= 0, "rotate_iou_eval: negative size");
    if (N == 0 || K == 0) return M3D_OK;
    M3D_REQUIRE(boxes_dev && qboxes_dev && iou_dev, "rotate_iou_eval: null pointer");
    const long long total = (long long)N * K;
    hipLaunchKernelGGL(rotate_iou_eval_kernel, dim3(cdiv(total, 128)), dim3(128), 0, (hipStream_t)stream, boxes_dev, N, qboxes_dev,
                       K, criterion, iou_dev);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side (float64, like the numpy arrays the reference's jitted functions receive).
extern "C" int m3d_eval_image_box_overlap(const double *boxes, int N, const double *q, int K, int criterion, double *out)
{
    M3D_REQUIRE(N >= 0 && K >= 0 && (N == 0 || K == 0 || (boxes && q && out)), "image_box_overlap: bad arguments");
    for (long long i = 0; i < (long long)N * K; ++i) out[i] = 0.0;
    for (int k = 0; k < K; ++k) {
        const double *qb = q + (size_t)k * 4;
        const double qarea = (qb[2] - qb[0]) * (qb[3] - qb[1]);
        for (int n = 0; n < N; ++n) {
            const double *b = boxes + (size_t)n * 4;
            const double iw = fmin(b[2], qb[2]) - fmax(b[0], qb[0]);
            if (iw > 0) {
                const double ih = fmin(b[3], qb[3]) - fmax(b[1], qb[1]);
                if (ih > 0) {
                    double ua;
                    if (criterion == -1) ua = (b[2] - b[0]) * (b[3] - b[1]) + qarea - iw * ih;
                    else if (criterion == 0) ua = (b[2] - b[0]) * (b[3] - b[1]);
                    else if (criterion == 1) ua = qarea;
                    else ua = 1.0;
                    out[(size_t)n * K + k] = iw * ih / ua;
                }
            }
        }
    }
    return M3D_OK;
}

// boxes [N][7], qboxes [K][7] = (x, y, z, l, h, w, ry) camera coordinates; rinc [N][K] holds the BEV intersection areas on
// entry and the 3-D IoU on return (eval.py:112-141).
extern "C" int m3d_eval_d3_overlap(const double *boxes, int N, const double *qboxes, int K, double *rinc, int criterion)
{
    M3D_REQUIRE(N >= 0 && K >= 0 && (N == 0 || K == 0 || (boxes && qboxes && rinc)), "d3_overlap: bad arguments");
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < K; ++j) {
            double &r = rinc[(size_t)i * K + j];
            if (r > 0) {
                const double *b = boxes + (size_t)i * 7, *q = qboxes + (size_t)j * 7;
                const double iw = fmin(b[1], q[1]) - fmax(b[1] - b[4], q[1] - q[4]);
                if (iw > 0) {
                    const double a1 = b[3] * b[4] * b[5], a2 = q[3] * q[4] * q[5];
                    const double inc = iw * r;
                    double ua;
                    if (criterion == -1) ua = a1 + a2 - inc;
                    else if (criterion == 0) ua = a1;
                    else if (criterion == 1) ua = a2;
                    else ua = inc;
                    r = inc / ua;
                } else {
                    r = 0.0;
                }
            }
        }
    return M3D_OK;
}

// compute_statistics_jit (eval.py:152-272).  overlaps [det_size][ov_stride] (entry [j][i] = detection j vs ground truth i),
// gt_datas [gt_size][5] (bbox, alpha), dt_datas [det_size][6] (bbox, alpha, score).  stats[4] = tp, fp, fn, similarity;
// thresholds_out (may be null) receives the scores of the true positives, *n_thresholds their count.
static void compute_statistics(const double *overlaps, long long ov_stride, const double *gt_datas, int gt_size,
                               const double *dt_datas, int det_size, const long long *ignored_gt, const long long *ignored_det,
                               const double *dc_bboxes, int n_dc, int metric, double min_overlap, double thresh, bool compute_fp,
                               bool compute_aos, double *stats, double *thresholds_out, int *n_thresholds)
{
    std::vector<char> assigned(det_size, 0), ign_thr(det_size, 0);
    std::vector<double> delta;
    if (compute_fp)
        for (int i = 0; i < det_size; ++i)
            if (dt_datas[(size_t)i * 6 + 5] < thresh) ign_thr[i] = 1;
    const double NO_DET = -10000000;
    long long tp = 0, fp = 0, fn = 0;
    double similarity = 0;
    int nthr = 0;
    for (int i = 0; i < gt_size; ++i) {
        if (ignored_gt[i] == -1) continue;
        int det_idx = -1;
        double valid_detection = NO_DET, max_overlap = 0;
        bool assigned_ignored = false;
        for (int j = 0; j < det_size; ++j) {
            if (ignored_det[j] == -1 || assigned[j] || ign_thr[j]) continue;
            const double overlap = overlaps[(size_t)j * ov_stride + i], sc = dt_datas[(size_t)j * 6 + 5];
            if (!compute_fp && overlap > min_overlap && sc > valid_detection) {
                det_idx = j;
                valid_detection = sc;
            } else if (compute_fp && overlap > min_overlap && (overlap > max_overlap || assigned_ignored) && ignored_det[j] == 0) {
                max_overlap = overlap;
                det_idx = j;
                valid_detection = 1;
                assigned_ignored = false;
            } else if (compute_fp && overlap > min_overlap && valid_detection == NO_DET && ignored_det[j] == 1) {
                det_idx = j;
                valid_detection = 1;
                assigned_ignored = true;
            }
        }
        if (valid_detection == NO_DET && ignored_gt[i] == 0) {
            ++fn;
        } else if (valid_detection != NO_DET && (ignored_gt[i] == 1 || ignored_det[det_idx] == 1)) {
            assigned[det_idx] = 1;
        } else if (valid_detection != NO_DET) {
            ++tp;
            if (thresholds_out) thresholds_out[nthr] = dt_datas[(size_t)det_idx * 6 + 5];
            ++nthr;
            if (compute_aos) delta.push_back(gt_datas[(size_t)i * 5 + 4] - dt_datas[(size_t)det_idx * 6 + 4]);
            assigned[det_idx] = 1;
        }
    }
    if (compute_fp) {
        for (int i = 0; i < det_size; ++i)
            if (!(assigned[i] || ignored_det[i] == -1 || ignored_det[i] == 1 || ign_thr[i])) ++fp;
        long long nstuff = 0;
        if (metric == 0) {
            for (int i = 0; i < n_dc; ++i) {
                const double *qb = dc_bboxes + (size_t)i * 4;
                for (int j = 0; j < det_size; ++j) {
                    if (assigned[j] || ignored_det[j] == -1 || ignored_det[j] == 1 || ign_thr[j]) continue;
                    // image_box_overlap(dt_bboxes, dc_bboxes, criterion 0)[j][i]
                    const double *b = dt_datas + (size_t)j * 6;
                    double ov = 0.0;
                    const double iw = fmin(b[2], qb[2]) - fmax(b[0], qb[0]);
                    if (iw > 0) {
                        const double ih = fmin(b[3], qb[3]) - fmax(b[1], qb[1]);
                        if (ih > 0) ov = iw * ih / ((b[2] - b[0]) * (b[3] - b[1]));
                    }
                    if (ov > min_overlap) {
                        assigned[j] = 1;
                        ++nstuff;
                    }
                }
            }
        }
        fp -= nstuff;
        if (compute_aos) {
            double s = 0.0;                       // np.sum over [0]*fp + [(1 + cos(delta)) / 2]
            for (double d : delta) s += (1.0 + cos(d)) / 2.0;
            similarity = (tp > 0 || fp > 0) ? s : -1;
        }
    }
    stats[0] = (double)tp; stats[1] = (double)fp; stats[2] = (double)fn; stats[3] = similarity;
    if (n_thresholds) *n_thresholds = nthr;
}

extern "C" int m3d_eval_statistics(const double *overlaps, long long ov_stride, const double *gt_datas, int gt_size,
                                   const double *dt_datas, int det_size, const long long *ignored_gt, const long long *ignored_det,
                                   const double *dc_bboxes, int n_dc, int metric, double min_overlap, double thresh, int compute_fp,
                                   int compute_aos, double *stats, double *thresholds_out, int *n_thresholds)
{
    M3D_REQUIRE(stats && gt_size >= 0 && det_size >= 0, "eval_statistics: bad arguments");
    compute_statistics(overlaps, ov_stride, gt_datas, gt_size, dt_datas, det_size, ignored_gt, ignored_det, dc_bboxes, n_dc, metric,
                       min_overlap, thresh, compute_fp != 0, compute_aos != 0, stats, thresholds_out, n_thresholds);
    return M3D_OK;
}

// fused_compute_statistics (eval.py:287-333): one part = n_images images whose overlaps form ONE [sum dt][sum gt] matrix
// (row-major, ov_stride = sum gt); pr [n_thresholds][4] is accumulated into.
extern "C" int m3d_eval_fused_statistics(const double *overlaps, long long ov_stride, double *pr, const long long *gt_nums,
                                         const long long *dt_nums, const long long *dc_nums, int n_images, const double *gt_datas,
                                         const double *dt_datas, const double *dontcares, const long long *ignored_gts,
                                         const long long *ignored_dets, int metric, double min_overlap, const double *thresholds,
                                         int n_thresholds, int compute_aos)
{
    M3D_REQUIRE(pr && gt_nums && dt_nums && dc_nums && n_images >= 0, "eval_fused_statistics: bad arguments");
    long long gt_num = 0, dt_num = 0, dc_num = 0;
    for (int i = 0; i < n_images; ++i) {
        for (int t = 0; t < n_thresholds; ++t) {
            double st[4];
            compute_statistics(overlaps + (size_t)dt_num * ov_stride + gt_num, ov_stride, gt_datas + (size_t)gt_num * 5, (int)gt_nums[i],
                               dt_datas + (size_t)dt_num * 6, (int)dt_nums[i], ignored_gts + gt_num, ignored_dets + dt_num,
                               dontcares + (size_t)dc_num * 4, (int)dc_nums[i], metric, min_overlap, thresholds[t], true,
                               compute_aos != 0, st, nullptr, nullptr);
            pr[(size_t)t * 4 + 0] += st[0];
            pr[(size_t)t * 4 + 1] += st[1];
            pr[(size_t)t * 4 + 2] += st[2];
            if (st[3] != -1) pr[(size_t)t * 4 + 3] += st[3];
        }
        gt_num += gt_nums[i];
        dt_num += dt_nums[i];
        dc_num += dc_nums[i];
    }
    return M3D_OK;
}
