// ANAB attention of the fp32 path in ONE launch (model/module/attention.py:207-211 + the BatchNorm / LeakyReLU that follows the
// block in M3d_inference_align.py): for every pixel
//     out = act( softmax_k(q . khat_k) @ vhat  (+ x, scale, shift) )
// with the per-image pooled keys khat [keys][Ck] and values vhat^T [Cv][keys] (337 keys of the 1/4/8/16 pyramid).  The three-launch
// form (logits GEMM -> row softmax -> P.V GEMM, rounds 1-4) writes the logits (7680 x 384 fp32 per image), reads them twice and
// writes them again as probabilities: 0.21 ms per bs-8 step around 12 GFLOP.  Here the logits never leave the registers:
//
//   Workgroup = 256 threads = 4 waves, 128 pixels of one image; a wave owns 32 pixels.  D = A.B with rows = keys (QK) or value
//   channels (PV) and columns = pixels (v_mfma_f32_32x32x2f32), so a lane holds ONE pixel: the softmax reductions over the keys are
//   in-lane plus one exchange between the two half-waves.  ONE pass over the keys in tiles of 32 with a running maximum (sum and
//   accumulators rescaled when a tile raises it: the online form of csrc/bf16_anab.hip); khat / vhat^T tiles are staged in LDS by
//   the workgroup, the next tile's pieces in flight in registers while the current one is multiplied.
//     * QK: the two k of an MFMA step are summation indices only, so step 4t + j takes k = 8t + j (lanes 0-31) and 8t + 4 + j (lanes
//       32-63): a lane's operands of four steps are 4 CONSECUTIVE floats -- one ds_read_b128 of the khat row, one 16-byte load of q.
//     * PV: register r = 4i + j of the logit tile is key 8i + 4h + j of lane half h -- exactly the two k of an MFMA step whose A
//       operand reads vhat^T[cv][8i + 4h + j]: the exponentials ARE the B operands, no exchange; the A operands are four runs of
//       four consecutive keys per lane.
//   Per tile and wave 84 + 64 MFMAs (9 472 cycles): the kernel is MFMA-bound by construction; two workgroups per CU.
#include <stdlib.h>

#include "common.h"

#define AF_CV 128
#define AF_KT 32                         // keys per tile
#define AF_VROW (AF_KT * 4 + 16)         // bytes per vhat^T tile row in LDS (32 keys + 16 pad)

struct AnabF32Args {
    const float *q, *khat, *vhat, *res, *scale, *shift;
    float *out;
    int q_cs, k_cs, HW, Ck, keys, keys_pad, res_cs, out_cs, res_mode, act;
};

template <int CK>                        // key / query channels (multiple of 8)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void anab_attend_f32_kernel(const AnabF32Args a)
{
    constexpr int KROW = CK * 4 + 16;    // bytes per khat tile row in LDS (+16: rows of 8 neighbouring keys start in different banks)
    constexpr int NS4 = CK / 8;          // groups of four MFMA steps
    __shared__ __attribute__((aligned(16))) unsigned char lds[AF_KT * KROW + AF_CV * AF_VROW];
    unsigned char *Ks = lds, *Vs = lds + AF_KT * KROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_per_img = a.HW / 128;
    const int img = blockIdx.x / tiles_per_img;
    const int mq = blockIdx.x * 128 + wave * 32 + l31;            // this lane's pixel (linear over the batch)

    // ---- q: B operands of the QK steps, lane (pixel, h) holds k = 8t + 4h + {0..3} ---------------------------------------------------
    f32x4 qf[NS4];
    {
        const float *qp = a.q + (size_t)mq * a.q_cs + 4 * lh;
#pragma unroll
        for (int t = 0; t < NS4; ++t) qf[t] = *reinterpret_cast<const f32x4 *>(qp + 8 * t);
    }
    const float *kimg = a.khat + (size_t)img * a.keys_pad * a.k_cs;
    const float *vimg = a.vhat + (size_t)img * AF_CV * a.keys_pad;

    // staging: khat tile = 32 rows x CK / 4 pieces of 16 B; vhat^T tile = 128 rows x 8 pieces
    constexpr int KP = AF_KT * (CK / 4), KPT = (KP + 255) / 256;
    f32x4 kr[KPT], vr[4];
    auto stage_load = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < KPT; ++p) {
            const int i = tid + 256 * p, row = i / (CK / 4), c = i - row * (CK / 4);
            if (i < KP) kr[p] = *reinterpret_cast<const f32x4 *>(kimg + (size_t)(AF_KT * t + row) * a.k_cs + c * 4);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = tid + 256 * p, row = i >> 3, c = i & 7;
            vr[p] = *reinterpret_cast<const f32x4 *>(vimg + (size_t)row * a.keys_pad + AF_KT * t + c * 4);
        }
    };
    auto stage_store = [&]() __attribute__((always_inline)) {
        __syncthreads();                                        // every wave is done with the previous tile
#pragma unroll
        for (int p = 0; p < KPT; ++p) {
            const int i = tid + 256 * p, row = i / (CK / 4), c = i - row * (CK / 4);
            if (i < KP) *reinterpret_cast<f32x4 *>(Ks + row * KROW + c * 16) = kr[p];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = tid + 256 * p, row = i >> 3, c = i & 7;
            *reinterpret_cast<f32x4 *>(Vs + row * AF_VROW + c * 16) = vr[p];
        }
        __syncthreads();
    };

    f32x16 o[AF_CV / 32];
#pragma unroll
    for (int j = 0; j < AF_CV / 32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    float l = 0.f, m = -INFINITY;
    const int T = (a.keys + AF_KT - 1) / AF_KT;
    stage_load(0);
    stage_store();
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) stage_load(t + 1);                       // (wave-uniform) in flight under this tile's MFMAs
        // ---- S tile: rows = the 32 keys of the tile, columns = the wave's pixels ---------------------------------------------------
        f32x16 s;                                               // (one chain: a 16-pass MFMA fills the pipe, a dependent one follows at no cost)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const unsigned char *kb = Ks + l31 * KROW + lh * 16;
#pragma unroll
        for (int g = 0; g < NS4; ++g) {
            const f32x4 kf = *reinterpret_cast<const f32x4 *>(kb + g * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[g][j], s, 0, 0, 0);
        }
        if (AF_KT * t + AF_KT > a.keys) {                        // (wave-uniform) the ragged last tile: keys past the end count as -inf
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = AF_KT * t + 8 * (r >> 2) + 4 * lh + (r & 3);
                s[r] = key < a.keys ? s[r] : -INFINITY;
            }
        }
        float mt = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mn = fmaxf(m, mt);                           // finite from the first tile on (keys >= 1)
        const float alpha = expf(m - mn);                        // 0 for the first tile (m = -inf), 1 when the maximum stands
        float e[16], ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            e[r] = expf(s[r] - mn);                              // (-inf -> 0)
            ls += e[r];
        }
        l = l * alpha + ls;
        m = mn;
#pragma unroll
        for (int j = 0; j < AF_CV / 32; ++j) o[j] *= alpha;
        // ---- O += vhat_t^T . e: step r uses keys 8 (r / 4) + 4 h + r % 4 -- the lane's own e[r] is the B operand ----------------
#pragma unroll
        for (int j = 0; j < AF_CV / 32; ++j) {
            const unsigned char *vb = Vs + (32 * j + l31) * AF_VROW + lh * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 vf = *reinterpret_cast<const f32x4 *>(vb + i * 32);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) o[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[jj], e[4 * i + jj], o[j], 0, 0, 0);
            }
        }
        if (t + 1 < T) stage_store();
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;

    // ---- epilogue: lane = pixel, register r of block j = channel 32 j + 8 (r / 4) + 4 h + r % 4: 16-byte pieces -------------------
    const float slope = a.act ? M3D_LEAKY_SLOPE : 1.f;
    float *op = a.out + (size_t)mq * a.out_cs;
    const float *rp = a.res ? a.res + (size_t)mq * a.res_cs : nullptr;
#pragma unroll
    for (int j = 0; j < AF_CV / 32; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 32 * j + 8 * i + 4 * lh;
            f32x4 v = {o[j][4 * i] * inv, o[j][4 * i + 1] * inv, o[j][4 * i + 2] * inv, o[j][4 * i + 3] * inv};
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (a.scale) sc = *reinterpret_cast<const f32x4 *>(a.scale + c);
            if (a.shift) sh = *reinterpret_cast<const f32x4 *>(a.shift + c);
            if (rp) {
                const f32x4 rv = *reinterpret_cast<const f32x4 *>(rp + c);
                v = a.res_mode ? (v + rv) * sc + sh : v * sc + sh + rv;
            } else {
                v = v * sc + sh;
            }
            v = __builtin_elementwise_max(v, v * slope);
            *reinterpret_cast<f32x4 *>(op + c) = v;
        }
}

extern "C" int m3d_anab_attend_f32(const float *q, int q_cs, const float *khat, int k_cs, const float *vhatT, int B, int HW, int Ck,
                                   int keys, int keys_pad, int Cv, const float *res, int res_cs, int res_mode, const float *scale,
                                   const float *shift, int act, float *out, int out_cs, m3d_stream_t stream)
{
    M3D_REQUIRE(q && khat && vhatT && out, "anab_attend_f32: null pointer");
    M3D_REQUIRE((Ck == 168 || Ck == 64 || Ck == 128) && Cv == AF_CV, "anab_attend_f32: built for Ck in {64, 128, 168}, Cv = %d (got %d, %d)", AF_CV, Ck, Cv);
    M3D_REQUIRE(B >= 1 && HW >= 128 && HW % 128 == 0, "anab_attend_f32: H*W must be a multiple of 128 (got %d)", HW);
    M3D_REQUIRE(keys >= 1 && keys <= keys_pad && keys_pad % 32 == 0, "anab_attend_f32: keys <= keys_pad, keys_pad %% 32 == 0");
    M3D_REQUIRE(q_cs % 4 == 0 && q_cs >= Ck && k_cs % 4 == 0 && k_cs >= Ck && out_cs % 4 == 0 && (!res || res_cs % 4 == 0),
                "anab_attend_f32: row strides must be multiples of 4 floats");
    M3D_REQUIRE((((uintptr_t)q | (uintptr_t)khat | (uintptr_t)vhatT | (uintptr_t)out | (uintptr_t)res | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0,
                "anab_attend_f32: 16-byte aligned views");
    M3D_REQUIRE((long long)B * HW < 0x7FFFFFFFLL, "anab_attend_f32: too many pixels");
    AnabF32Args a;
    a.q = q; a.khat = khat; a.vhat = vhatT; a.res = res; a.scale = scale; a.shift = shift; a.out = out;
    a.q_cs = q_cs; a.k_cs = k_cs; a.HW = HW; a.Ck = Ck; a.keys = keys; a.keys_pad = keys_pad; a.res_cs = res_cs; a.out_cs = out_cs;
    a.res_mode = res_mode; a.act = act ? 1 : 0;
    const dim3 grid(B * (HW / 128));
    if (Ck == 168) hipLaunchKernelGGL(anab_attend_f32_kernel<168>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (Ck == 128) hipLaunchKernelGGL(anab_attend_f32_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(anab_attend_f32_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
