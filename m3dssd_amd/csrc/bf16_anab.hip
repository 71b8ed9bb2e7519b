// ANAB attention of the bf16 path in ONE launch (model/module/attention.py:207-211 + the BatchNorm / LeakyReLU that follows
// the block in M3d_inference_align.py): for every pixel
//     out = act( (softmax_k(q . khat_k) @ vhat + x) * scale + shift )
// with the per-image pooled keys khat [keys][Ck] and values vhat^T [Cv][keys] (337 keys of the 1/4/8/16 pyramid).  The three
// launch form (logits GEMM -> row softmax -> P.V GEMM) writes the fp32 logits (7680 x 384 per image: 755 MB at bs = 64), reads
// them back, writes the bf16 probabilities and reads those back: 0.74 ms of HBM round trips per step around 0.12 TFLOP.
//
//   Workgroup = 256 threads = 4 waves, 128 pixels of one image; a wave owns 32 pixels.  D = A.B with rows = keys (QK) or value
//   channels (PV) and columns = pixels, so a lane holds ONE pixel: the softmax reductions over the keys are in-lane plus one
//   exchange between the two half-waves.  The keys are walked in tiles of 32 (khat / vhat^T tiles staged in LDS by the whole
//   workgroup, 20 KB).  Round-5 form (ONLINE, below): ONE pass with a running maximum.  Rounds 2-4 (M3D_ANAB_ONLINE=0): two passes,
//     pass 1   S = khat_t . q (12 MFMAs, K = 192)             -> running row maximum m
//     pass 2   S again, e = exp(S - m), l += sum e, O += vhat_t^T . bf16(e)   (the 16 exponentials of a lane ARE the B fragments of
//              the two K = 16 steps: register r = 4i + j of the accumulator is key 8i + 4*lh + j, so step u takes registers
//              8u .. 8u + 7 and the A fragment reads the matching two 4-key runs of the vhat^T row)
//   then O / l goes through the shared conv epilogue (residual before the affine, LeakyReLU, 16-byte bf16 stores).
//   The q fragments of the wave's 32 pixels (12 x 16 bytes per lane) stay in registers for both passes.
//   Differences to the three-launch form: the logits are recomputed instead of stored (same MFMA sequence, same values), and the
//   bf16 rounding is applied to exp(S - m) instead of exp(S - m) / l (the division happens in fp32 on the accumulator).
#include <stdlib.h>

#include "bf16_tile.h"

#ifndef AN_ABL
#define AN_ABL 0                          // diagnostic builds: phase ablations of the key loop (timing only)
#endif

#define AN_CKP 192            // padded key / query channels (Ck = 168)
#define AN_CV 128
#define AN_KROW (AN_CKP * 2)  // bytes per khat row
#define AN_VROW 72            // bytes per vhat^T tile row in LDS (32 keys = 64 B + 8: the 32 rows of a half-wave read hit 64 banks once)

struct AnabArgs {
    const void *q;            // bf16 [B*HW][q_cs], channels [0, 192) (168.. are zero)
    const void *khat;         // bf16 [B][keys_pad][192]
    const void *vhat;         // bf16 [B][128][keys_pad]
    int q_cs, HW, keys, keys_pad;
    Bf16Args ep;              // out / res / scale / shift / act of the epilogue
};

// ONLINE (round 5, default): ONE pass over the keys with a running maximum -- when a tile raises the maximum, the sum and the
// accumulators are rescaled by exp(m_old - m_new) (64 multiplies per tile) -- instead of a pass for the maximum and a second one
// that recomputes the logits; the next tile's khat / vhat pieces are in flight (registers) while the current tile is multiplied,
// and the 12 logit MFMAs run as two independent chains (two workgroups per CU instead of three: 20 + 16 more registers).  22 -> 11 staged tiles per workgroup, 352 -> 220 MFMAs per wave.
template <bool ONLINE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ONLINE ? 2 : 3, ONLINE ? 2 : 3))) void bf16_anab_attend_kernel(const AnabArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[32 * AN_KROW + AN_CV * AN_VROW];
    unsigned char *Ks = lds, *Vs = lds + 32 * AN_KROW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_per_img = a.HW / 128;
    const int img = blockIdx.x / tiles_per_img;
    const int m0 = blockIdx.x * 128;                       // first pixel (linear index) of the workgroup
    const int mq = m0 + wave * 32 + l31;                   // this lane's pixel

    // ---- q fragments: B operand, lane (pixel, lh) holds channels 16s + 8*lh .. + 7 ----------------------------------------------
    bf16x8 qf[AN_CKP / 16];
    {
        const __bf16 *qp = (const __bf16 *)a.q + (size_t)mq * a.q_cs + 8 * lh;
#pragma unroll
        for (int s = 0; s < AN_CKP / 16; ++s) qf[s] = *reinterpret_cast<const bf16x8 *>(qp + 16 * s);
    }
    const unsigned char *kimg = (const unsigned char *)a.khat + (size_t)img * a.keys_pad * AN_KROW;
    const unsigned char *vimg = (const unsigned char *)a.vhat + (size_t)img * AN_CV * a.keys_pad * 2;

    // staging maps.  khat tile: 32 rows x 24 pieces of 16 B = 768 pieces, 3 per thread; piece index XOR-swizzled by (row >> 1) & 7
    // inside its group of 8 (fragment reads of 16 rows then cover all 64 banks).  vhat^T tile: 128 rows x 64 B = 512 pieces, 2 per thread.
    auto stage_load = [&](int t, bool with_v, u32x4 (&kr)[3], u32x4 (&vr)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int i = tid + 256 * p, row = i / 24, c = i - row * 24;
            kr[p] = *reinterpret_cast<const u32x4 *>(kimg + (size_t)(32 * t + row) * AN_KROW + c * 16);
        }
        if (with_v) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int i = tid + 256 * p, row = i >> 2, c = i & 3;
                vr[p] = *reinterpret_cast<const u32x4 *>(vimg + ((size_t)row * a.keys_pad + 32 * t) * 2 + c * 16);
            }
        }
    };
    auto stage_store = [&](bool with_v, const u32x4 (&kr)[3], const u32x4 (&vr)[2]) __attribute__((always_inline)) {
        __syncthreads();                                    // every wave is done with the previous tile
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int i = tid + 256 * p, row = i / 24, c = i - row * 24;
            *reinterpret_cast<u32x4 *>(Ks + row * AN_KROW + (((c & ~7) | ((c ^ (row >> 1)) & 7)) << 4)) = kr[p];
        }
        if (with_v) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int i = tid + 256 * p, row = i >> 2, c = i & 3;
                u32x2 *dst = reinterpret_cast<u32x2 *>(Vs + row * AN_VROW + c * 16);
                dst[0] = u32x2{vr[p][0], vr[p][1]};
                dst[1] = u32x2{vr[p][2], vr[p][3]};
            }
        }
        __syncthreads();
    };
    auto stage = [&](int t, bool with_v) __attribute__((always_inline)) {
        u32x4 kr[3], vr[2];
        stage_load(t, with_v, kr, vr);
        stage_store(with_v, kr, vr);
    };
    // S tile: rows = the 32 keys of tile t, columns = the wave's pixels
    const int ksw = (l31 >> 1) & 7;
    auto logits = [&]() __attribute__((always_inline)) {
        f32x16 s, s2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; s2[r] = 0.f; }
        const unsigned char *kb = Ks + l31 * AN_KROW;
#pragma unroll
        for (int st = 0; st < AN_CKP / 16; ++st) {
            const int c = 2 * st + lh;
            const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kb + (((c & ~7) | ((c ^ ksw) & 7)) << 4));
            if (ONLINE && (st & 1)) s2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s2, 0, 0, 0);
            else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);
        }
        if (ONLINE) s += s2;
        return s;
    };
    const int T = (a.keys + 31) / 32;

    f32x16 o[AN_CV / 32];
#pragma unroll
    for (int j = 0; j < AN_CV / 32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    float l = 0.f;
    if constexpr (ONLINE) {
        // ---- one pass: running maximum m (the same in both half-waves of a pixel), rescaled sum and accumulators ------------------
        float m = -INFINITY;
        stage(0, true);
        u32x4 kr[3], vr[2];
        for (int t = 0; t < T; ++t) {
#if !(AN_ABL & 1)
            if (t + 1 < T) stage_load(t + 1, true, kr, vr);             // (wave-uniform) in flight under this tile's MFMAs
#endif
            f32x16 s = logits();
            if (32 * t + 32 > a.keys) {                          // (wave-uniform) the ragged last tile: keys past the end count as -inf
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * t + 8 * (r >> 2) + 4 * lh + (r & 3);
                    s[r] = key < a.keys ? s[r] : -INFINITY;
                }
            }
            float mt = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float mn = fmaxf(m, mt);                       // finite from the first tile on (keys >= 1)
            // exponentials of non-positive arguments through v_exp_f32 (2 instructions; expf's range handling is 10+ and the loop is
            // bound by instruction issue): 2 ulp of fp32 before the value is rounded to bf16 / summed
            const float alpha = __expf(m - mn);                  // 0 for the first tile (m = -inf), 1 when the maximum stands
            float e[16], ls = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if AN_ABL & 2
                e[r] = s[r] - mn;
#else
                e[r] = __expf(s[r] - mn);                        // (-inf - finite = -inf -> 0)
#endif
                ls += e[r];
            }
            l = l * alpha + ls;
            m = mn;
#pragma unroll
            for (int j = 0; j < AN_CV / 32; ++j) o[j] *= alpha;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 pb;
#pragma unroll
                for (int w = 0; w < 4; ++w) pb[w] = pack_bf16(e[8 * u + 2 * w], e[8 * u + 2 * w + 1]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pb);
                // A fragment of value-channel row 32j + l31: keys 16u + 4*lh + {0..3} and 16u + 8 + 4*lh + {0..3} of the tile
                const unsigned char *vb = Vs + l31 * AN_VROW + (16 * u + 4 * lh) * 2;
#pragma unroll
                for (int j = 0; j < AN_CV / 32; ++j) {
                    const u32x2 lo = *reinterpret_cast<const u32x2 *>(vb + j * 32 * AN_VROW);
                    const u32x2 hi = *reinterpret_cast<const u32x2 *>(vb + j * 32 * AN_VROW + 16);
                    const u32x4 vw = {lo[0], lo[1], hi[0], hi[1]};
                    o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), pf, o[j], 0, 0, 0);
                }
            }
#if !(AN_ABL & 1)
            if (t + 1 < T) stage_store(true, kr, vr);
#endif
        }
    } else {
        // ---- pass 1: row maximum -------------------------------------------------------------------------------------------
        float mx = -INFINITY;
        for (int t = 0; t < T; ++t) {
            stage(t, false);
            const f32x16 s = logits();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * t + 8 * (r >> 2) + 4 * lh + (r & 3);
                if (key < a.keys) mx = fmaxf(mx, s[r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // ---- pass 2: exponentials, their sum, O += vhat_t^T . e ----------------------------------------------------------------
        for (int t = 0; t < T; ++t) {
            stage(t, true);
            const f32x16 s = logits();
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * t + 8 * (r >> 2) + 4 * lh + (r & 3);
                e[r] = key < a.keys ? expf(s[r] - mx) : 0.f;
                l += e[r];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x4 pb;
#pragma unroll
                for (int w = 0; w < 4; ++w) pb[w] = pack_bf16(e[8 * u + 2 * w], e[8 * u + 2 * w + 1]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, pb);
                // A fragment of value-channel row 32j + l31: keys 16u + 4*lh + {0..3} and 16u + 8 + 4*lh + {0..3} of the tile
                const unsigned char *vb = Vs + l31 * AN_VROW + (16 * u + 4 * lh) * 2;
#pragma unroll
                for (int j = 0; j < AN_CV / 32; ++j) {
                    const u32x2 lo = *reinterpret_cast<const u32x2 *>(vb + j * 32 * AN_VROW);
                    const u32x2 hi = *reinterpret_cast<const u32x2 *>(vb + j * 32 * AN_VROW + 16);
                    const u32x4 vw = {lo[0], lo[1], hi[0], hi[1]};
                    o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vw), pf, o[j], 0, 0, 0);
                }
            }
        }
    }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x16 acc[AN_CV / 32][1];
#pragma unroll
    for (int j = 0; j < AN_CV / 32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][0][r] = o[j][r] * inv;
    const int mpix[1] = {mq};
    conv_epilogue<AN_CV / 32, 1>(a.ep, acc, mpix, 0, 0, lh, 0);
}

extern "C" int m3d_anab_attend_bf16(const void *q, int q_cs, const void *khat, const void *vhatT, int B, int HW, int Ck_pad, int keys,
                                    int keys_pad, int Cv, const void *res, int res_cs, const float *scale, const float *shift, int act,
                                    void *out, int out_cs, m3d_stream_t stream)
{
    M3D_REQUIRE(q && khat && vhatT && out, "anab_attend_bf16: null pointer");
    M3D_REQUIRE(Ck_pad == AN_CKP && Cv == AN_CV, "anab_attend_bf16: built for Ck_pad = %d, Cv = %d (got %d, %d)", AN_CKP, AN_CV, Ck_pad, Cv);
    M3D_REQUIRE(B >= 1 && HW >= 128 && HW % 128 == 0, "anab_attend_bf16: H*W must be a multiple of 128 (got %d)", HW);
    M3D_REQUIRE(keys >= 1 && keys <= keys_pad && keys_pad % 32 == 0, "anab_attend_bf16: keys <= keys_pad, keys_pad %% 32 == 0");
    M3D_REQUIRE(q_cs % 8 == 0 && q_cs >= AN_CKP && out_cs % 8 == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                ((uintptr_t)khat & 15) == 0 && ((uintptr_t)vhatT & 15) == 0, "anab_attend_bf16: 16-byte aligned bf16 rows");
    if (res) M3D_REQUIRE(res_cs % 4 == 0 && ((uintptr_t)res & 7) == 0, "anab_attend_bf16: residual view alignment");
    M3D_REQUIRE((long long)B * HW < 0x7FFFFFFFLL, "anab_attend_bf16: too many pixels");
    AnabArgs a = {};
    a.q = q; a.khat = khat; a.vhat = vhatT; a.q_cs = q_cs; a.HW = HW; a.keys = keys; a.keys_pad = keys_pad;
    a.ep.out = out; a.ep.out_cs = out_cs; a.ep.out_mode = 0; a.ep.Cout = Cv; a.ep.scale = scale; a.ep.shift = shift;
    a.ep.res = res; a.ep.res_cs = res_cs; a.ep.res_mode = 1; a.ep.act = act ? 1 : 0; a.ep.sigmoid_from = -1;
    static const int online = []() { const char *e = getenv("M3D_ANAB_ONLINE"); return e ? atoi(e) : 1; }();   // 0: the two-pass form (A/B)
    if (online) hipLaunchKernelGGL(bf16_anab_attend_kernel<true>, dim3(B * (HW / 128)), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(bf16_anab_attend_kernel<false>, dim3(B * (HW / 128)), dim3(256), 0, (hipStream_t)stream, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
