// Post-NMS 3-D refinement of the detections: one thread per detection row runs the reference's
// per-box loop of test_kitti_3d in float64 --
//   convertAlpha2Rot (lib/util.py:516-524), hill_climb on the yaw / depth with step halving (lib/rpn_util.py:652-708) scored by
//   test_projection (-L1 distance between the 2-D box and the bounding box of the projected 3-D box, :2015-2050, project_3d
//   :921-970), convertRot2Alpha (lib/util.py:527-535) and the final camera-space centre (:1836-1845)
// -- which the reference does in Python / numpy on the host (<= 40 boxes x ~14 projections per image).  Contraction is off and
// every dot product is accumulated left to right; against numpy the results agree to ~1e-12 (BLAS may fuse / reorder), far
// inside the 6 decimals of the KITTI result format.
// numpy scalar semantics of the reference's environment (requirements.txt:50 pins numpy==1.18.1: value-based promotion) are
// mirrored: `box = aboxes[i]` holds np.float32 scalars, so x3d*z3d, y3d*z3d (rpn_util.py:1825,1836 and test_projection :2025,
// where cx, cy, z arrive as np.float32), the box width / height (:1813-1814) and x + w - 1 (:2022-2023) are ROUNDED TO FLOAT32
// before they widen into the float64 arithmetic; np.float32 + Python float (the angle conversions, ry3d +- step) is float64.
#include "common.h"

#pragma clang fp contract(off)

struct RefineArgs {
    const float *aboxes;     // [B][K][14]: x1 y1 x2 y2 score cls x3d y3d z3d w3d h3d l3d alpha anchor (lib/rpn_util.py:1550)
    const int *counts;       // [B] valid rows per image
    const double *p2;        // [B][16] row-major 4x4 projection matrices
    const double *p2_inv;    // [B][16] their inverses (np.linalg.inv on the host, as the reference computes them)
    double *out;             // [B][K][16]: valid, cls, alpha, x1, y1, x2, y2, h3d, w3d, l3d, x3d, y3d, z3d, ry3d, score, 0
    const float *scale;      // [B] or null: rows are divided back to the original image scale first (lib/rpn_util.py:1506-1507; callers that want the reference order -- before the NMS -- pass the factors to m3d_topk_decode_scaled and NULL here)
    const float *clip_wh;    // [B][2] = (imW, imH) or null: then the 2-D box is clipped to the image (:1533-1538); <= 0: not clipped
    int B, K, hill_climbing;
    double score_thresh, step_r_init, r_lim, step_z_init, z_lim, min_ol_dif;
};

#define REF_PI 3.141592653589793

__device__ __forceinline__ double wrap_pi(double a)
{
    while (a > REF_PI) a -= REF_PI * 2;
    while (a < (-REF_PI)) a += REF_PI * 2;
    return a;
}

// test_projection: returns ol, sets *invalid
// bx, by, bw, bh, cx, cy are float32 values (the row's np.float32 scalars); z_f32: z is still the row's float32 depth (always,
// for the depth step of 0 the reference passes) -> the products round to float32 like np.float32 * np.float32
__device__ double test_projection_dev(const double *p2, const double *pi, float bxf, float byf, float bwf, float bhf, float cxf,
                                      float cyf, double z, bool z_f32, double w3d, double h3d, double l3d, double rot,
                                      bool *invalid)
{
    const double bx = bxf, by = byf;
    const double x2 = (double)((bxf + bwf) - 1.0f), y2 = (double)((byf + bhf) - 1.0f);
    const double v0 = z_f32 ? (double)(cxf * (float)z) : (double)cxf * z;
    const double v1 = z_f32 ? (double)(cyf * (float)z) : (double)cyf * z;
    const double X = pi[0] * v0 + pi[1] * v1 + pi[2] * z + pi[3] * 1.0;
    const double Y = pi[4] * v0 + pi[5] * v1 + pi[6] * z + pi[7] * 1.0;
    const double Z = pi[8] * v0 + pi[9] * v1 + pi[10] * z + pi[11] * 1.0;
    const double c = cos(rot), s = sin(rot);
    const double xs[8] = {0, l3d, l3d, l3d, l3d, 0, 0, 0};
    const double ys[8] = {0, 0, h3d, h3d, 0, 0, h3d, h3d};
    const double zs[8] = {0, 0, 0, w3d, w3d, w3d, w3d, 0};
    double xn = 1e300, yn = 1e300, xm = -1e300, ym = -1e300;
    bool inv = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double xc = xs[k] + (-l3d / 2), yc = ys[k] + (-h3d / 2), zc = zs[k] + (-w3d / 2);
        const double rx = c * xc + 0.0 * yc + s * zc + X;
        const double ry = 0.0 * xc + 1.0 * yc + 0.0 * zc + Y;
        const double rz = (-s) * xc + 0.0 * yc + c * zc + Z;
        inv = inv || (rz <= 0);
        const double u = p2[0] * rx + p2[1] * ry + p2[2] * rz + p2[3] * 1.0;
        const double v = p2[4] * rx + p2[5] * ry + p2[6] * rz + p2[7] * 1.0;
        const double q = p2[8] * rx + p2[9] * ry + p2[10] * rz + p2[11] * 1.0;
        const double px = u / q, py = v / q;
        xn = fmin(xn, px); xm = fmax(xm, px);
        yn = fmin(yn, py); ym = fmax(ym, py);
    }
    *invalid = inv;
    return -(fabs(bx - xn) + fabs(by - yn) + fabs(x2 - xm) + fabs(y2 - ym));
}

__global__ void refine3d_kernel(const RefineArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B * a.K) return;
    const int b = i / a.K, k = i - b * a.K;
    double *o = a.out + (size_t)i * 16;
    const float *r = a.aboxes + (size_t)i * 14;
    const double score = (double)r[4];
    if (k >= a.counts[b] || !(score >= a.score_thresh)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = 0.0;
        return;
    }
    const double *p2 = a.p2 + (size_t)b * 16, *pi = a.p2_inv + (size_t)b * 16;
    float x1f = r[0], y1f = r[1], x2f = r[2], y2f = r[3], x3f = r[6], y3f = r[7];
    if (a.scale) {                                     // np.float32 divisions, as `aboxes[:, 0:4] /= scale_factor` does them
        const float sf = a.scale[b];
        x1f /= sf; y1f /= sf; x2f /= sf; y2f /= sf; x3f /= sf; y3f /= sf;
    }
    if (a.clip_wh) {
        const float cw = a.clip_wh[2 * b], ch = a.clip_wh[2 * b + 1];
        if (cw > 0.f) { x1f = fminf(fmaxf(x1f, 0.f), cw - 1.f); x2f = fminf(fmaxf(x2f, 0.f), cw - 1.f); }
        if (ch > 0.f) { y1f = fminf(fmaxf(y1f, 0.f), ch - 1.f); y2f = fminf(fmaxf(y2f, 0.f), ch - 1.f); }
    }
    const double x1 = x1f, y1 = y1f, x2 = x2f, y2 = y2f;
    double z3d = r[8];
    bool z_f32 = true;                                                         // z3d is still the row's np.float32
    const double w3d = r[9], h3d = r[10], l3d = r[11];
    double ry = r[12];
    {
        const double v0 = (double)(x3f * r[8]), v1 = (double)(y3f * r[8]);     // np.float32 products (rpn_util.py:1825)
        const double X = pi[0] * v0 + pi[1] * v1 + pi[2] * z3d + pi[3] * 1.0;
        const double Z = pi[8] * v0 + pi[9] * v1 + pi[10] * z3d + pi[11] * 1.0;
        ry = wrap_pi(ry + atan2(-Z, X) + 0.5 * REF_PI);                       // convertAlpha2Rot
    }
    if (a.hill_climbing) {
        const float bw = (x2f - x1f) + 1.0f, bh = (y2f - y1f) + 1.0f;        // np.float32 (rpn_util.py:1813-1814)
        const float x1 = x1f, y1 = y1f, x3d = x3f, y3d = y3f;
        double step_z = a.step_z_init, step_r = a.step_r_init;
        bool invalid;
        double ol_best = test_projection_dev(p2, pi, x1, y1, bw, bh, x3d, y3d, z3d, z_f32, w3d, h3d, l3d, ry, &invalid);
        if (!invalid) {
            int guard = 0;
            while ((step_z > a.z_lim || step_r > a.r_lim) && ++guard < 100000) {
                if (step_z > a.z_lim) {
                    bool in_neg, in_pos;
                    const double ol_neg = test_projection_dev(p2, pi, x1, y1, bw, bh, x3d, y3d, z3d - step_z, false, w3d, h3d, l3d, ry, &in_neg);
                    const double ol_pos = test_projection_dev(p2, pi, x1, y1, bw, bh, x3d, y3d, z3d + step_z, false, w3d, h3d, l3d, ry, &in_pos);
                    if (((ol_pos - ol_best) <= a.min_ol_dif) && ((ol_neg - ol_best) <= a.min_ol_dif)) step_z = step_z * 0.5;
                    else if ((ol_pos - ol_best) > a.min_ol_dif && ol_pos > ol_neg && !in_pos) { z3d += step_z; z_f32 = false; ol_best = ol_pos; }
                    else if ((ol_neg - ol_best) > a.min_ol_dif && !in_neg) { z3d -= step_z; z_f32 = false; ol_best = ol_neg; }
                    else step_z = step_z * 0.5;
                }
                if (step_r > a.r_lim) {
                    bool in_neg, in_pos;
                    const double ol_neg = test_projection_dev(p2, pi, x1, y1, bw, bh, x3d, y3d, z3d, z_f32, w3d, h3d, l3d, ry - step_r, &in_neg);
                    const double ol_pos = test_projection_dev(p2, pi, x1, y1, bw, bh, x3d, y3d, z3d, z_f32, w3d, h3d, l3d, ry + step_r, &in_pos);
                    if (((ol_pos - ol_best) <= a.min_ol_dif) && ((ol_neg - ol_best) <= a.min_ol_dif)) step_r = step_r * 0.5;
                    else if ((ol_pos - ol_best) > a.min_ol_dif && ol_pos > ol_neg && !in_pos) { ry += step_r; ol_best = ol_pos; }
                    else if ((ol_neg - ol_best) > a.min_ol_dif && !in_neg) { ry -= step_r; ol_best = ol_neg; }
                    else step_r = step_r * 0.5;
                }
            }
            ry = wrap_pi(ry);
        }
    }
    const double v0 = z_f32 ? (double)(x3f * (float)z3d) : (double)x3f * z3d;   // rpn_util.py:1836
    const double v1 = z_f32 ? (double)(y3f * (float)z3d) : (double)y3f * z3d;
    const double X = pi[0] * v0 + pi[1] * v1 + pi[2] * z3d + pi[3] * 1.0;
    const double Y = pi[4] * v0 + pi[5] * v1 + pi[6] * z3d + pi[7] * 1.0;
    const double Z = pi[8] * v0 + pi[9] * v1 + pi[10] * z3d + pi[11] * 1.0;
    const double alpha = wrap_pi(ry - atan2(-Z, X) - 0.5 * REF_PI);          // convertRot2Alpha
    o[0] = 1.0; o[1] = (double)r[5]; o[2] = alpha; o[3] = x1; o[4] = y1; o[5] = x2; o[6] = y2;
    o[7] = h3d; o[8] = w3d; o[9] = l3d; o[10] = X; o[11] = Y + h3d / 2; o[12] = Z; o[13] = ry; o[14] = score; o[15] = 0.0;
}

extern "C" int m3d_refine_3d_ex(const float *aboxes, const int *counts, int B, int K, const double *p2, const double *p2_inv,
                                const float *scale, const float *clip_wh, double score_thresh, int hill_climbing,
                                double step_r_init, double r_lim, double *out, m3d_stream_t stream)
{
    M3D_REQUIRE(aboxes && counts && p2 && p2_inv && out && B >= 1 && K >= 1, "refine_3d: bad arguments");
    M3D_REQUIRE(step_r_init >= 0 && r_lim >= 0, "refine_3d: negative step / limit");
    RefineArgs a;
    a.aboxes = aboxes; a.counts = counts; a.p2 = p2; a.p2_inv = p2_inv; a.out = out; a.B = B; a.K = K;
    a.scale = scale; a.clip_wh = clip_wh;
    a.hill_climbing = hill_climbing; a.score_thresh = score_thresh; a.step_r_init = step_r_init; a.r_lim = r_lim;
    a.step_z_init = 0.0; a.z_lim = 0.0; a.min_ol_dif = 0.0;                   // the values test_kitti_3d passes (:1833)
    hipLaunchKernelGGL(refine3d_kernel, dim3(cdiv(B * K, 64)), dim3(64), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

extern "C" int m3d_refine_3d(const float *aboxes, const int *counts, int B, int K, const double *p2, const double *p2_inv,
                             double score_thresh, int hill_climbing, double step_r_init, double r_lim, double *out,
                             m3d_stream_t stream)
{
    return m3d_refine_3d_ex(aboxes, counts, B, K, p2, p2_inv, nullptr, nullptr, score_thresh, hill_climbing, step_r_init, r_lim,
                            out, stream);
}
