// Internal helpers shared by the HIP translation units of libm3dssd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "m3dssd_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- DCNv2 sampling state without lane masks in SGPRs --------------------------------------------------------------
// dcn_v2_im2col_cuda.cu:18-47,150-178: a sample at (h_im, w_im) is taken iff h_im > -1, w_im > -1, h_im < H, w_im < W, and each
// of its four corners contributes iff it lies inside the image.  Written the obvious way (compares, &&, ?:) hipcc builds the
// decisions as v_cmp -> s_and_b64 -> v_cndmask chains on 64-bit lane masks held in SGPRs; in the deformable kernels -- 255
// VGPRs, two waves per SIMD, MFMA / LDS / gather traffic around the sampling code -- about one (pixel, tap) state in 10^5..10^6
// then came out with ONE corner dropped (weight 0, offset 0) in lanes 48-63 of a wave: run-to-run different outputs
// (tools/dcn_determinism.py resolves the difference into per-corner contributions: coefficient -1.00 on one corner, residual
// at bf16 rounding level).  Which corner / tap and how often changed with every recompile (0 of 800 launches for one build, 100 %
// of the launches for another), one workgroup per CU never showed it, a 24-bit multiply or a branch-free form moved it but did
// not remove it, and two isolated reproducers (tools/ubench/vcmp_sand_hazard.hip, vmul_divergent.hip) are clean: the
// observation is consistent with bits 48-63 of a VALU-written SGPR mask being read stale, the root cause is not established.
// What removes it in every build tried (4 variants x 4 shapes x 300 launches): no lane mask at all.  Every decision is a sign
// bit smeared over the register by an arithmetic shift the optimiser cannot see through (`x & ~(y >> 31)` written in C++ is
// turned back into v_cmp + v_cndmask), applied with AND / OR.
__device__ __forceinline__ int sign_smear(int x)         // x < 0 ? ~0 : 0, as one opaque v_ashrrev_i32
{
    int m;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(x));
    return m;
}
// Corner weights (uh*uw, uh*lw, lh*uw, lh*lw -- zero for a dropped corner), corner pixel offsets inside the image plane
// (hl*W + wl, ... -- valid only where kept) and the drop masks (all ones: corner not sampled).  `drop_all`: all ones forces the
// whole sample off (a lane without a pixel).
__device__ __forceinline__ void dcn_corners(float h_im, float w_im, int H, int W, int drop_all, float (&w)[4], int (&o)[4],
                                            int (&drop)[4])
{
    const float tin = fminf(fminf(h_im, w_im) + 1.f, -fmaxf(h_im - (float)H, w_im - (float)W));     // > 0: inside
    const unsigned xin = __float_as_uint(tin);
    // fminf / fmaxf return the other operand for a NaN: a NaN coordinate must fail the test like the reference's compares do
    // (`h_im > -1 && ...` is false for NaN, dcn_v2_im2col_cuda.cu:165).  x - x is +0 for finite x and NaN for NaN / inf.
    const unsigned nonfinite = __float_as_uint((h_im - h_im) + (w_im - w_im)) & 0x7fffffffu;
    const int out = sign_smear((int)((xin - 1u) | xin)) | sign_smear(-(int)nonfinite) | drop_all;      // tin <= 0 (or -0.0)
    // far outside the image the float -> int conversions saturate: the index arithmetic below is unsigned (wrap-around, no
    // signed overflow for the optimiser to reason about); its results are masked by `out` there
    const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
    const float lh = h_im - (float)hl, lw = w_im - (float)wl;
    const float uh = 1.f - lh, uw = 1.f - lw;
    const int hr = (int)((unsigned)(H - 2) - (unsigned)hl), wr = (int)((unsigned)(W - 2) - (unsigned)wl);   // >= 0: high row / column inside
    drop[0] = sign_smear(hl | wl) | out;
    drop[1] = sign_smear(hl | wr) | out;
    drop[2] = sign_smear(hr | wl) | out;
    drop[3] = sign_smear(hr | wr) | out;
    const unsigned row0 = (unsigned)__mul24(hl, W) + (unsigned)wl, row1 = row0 + (unsigned)W;
    o[0] = (int)row0; o[1] = (int)(row0 + 1u); o[2] = (int)row1; o[3] = (int)(row1 + 1u);
    const float wf[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = __uint_as_float(__float_as_uint(wf[q]) & ~(unsigned)drop[q]);
}

// ---- raw buffer access (gfx950) -------------------------------------------------------------------------------
// A buffer resource makes the address of a load  base(SGPR x4) + voffset(VGPR, 32 bit) + soffset(SGPR): no per-load
// 64-bit VALU address arithmetic, and a lane whose voffset is >= num_records reads 0.0f -- out-of-image taps cost no
// v_cndmask.  Every VALU instruction saved matters: on this part a VALU instruction of ANY wave of a SIMD delays that
// SIMD's MFMA stream by ~8 cycles (tools/ubench/mfma_valu_overlap.hip), i.e. VALU work does not hide under MFMAs.
#define M3D_BUF_OOB 0x80000000u          // voffset of a masked lane (views are < 2 GiB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
// ---- the ">64-bit store data" hazard of gfx950 --------------------------------------------------------------------------------
// A 12- / 16-byte buffer / global store reads its data registers AFTER it has issued: a VALU write of those registers within the
// next 1 (buffer store, SGPR soffset) or 2 (global store, literal soffset) instructions lands in the stored data (element 1 of
// lanes 12-15 / 28-31 / 44-47 / 60-63 first; tools/ubench/vmem_war_hazards.hip reproduces it: 5 % wrong dwords at 0 wait states).
// hipcc (ROCm 7.2) does not insert the wait states for gfx950 -- round 4 found `buffer_store_dwordx4 v[4:7]` directly followed by
// `v_pk_fma_f32 v[4:5]` in the K-pair F(4x4) epilogue.  tools/check_isa_hazards.py scans every kernel's ISA for the pattern (part
// of the build check); where it fires, the store goes through these helpers: the wait states sit inside the same asm statement.
__device__ __forceinline__ void buf_store_f32x4_nop(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset)
{
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(voffset), "s"(r), "s"(soffset) : "memory");
}
__device__ __forceinline__ void global_store_u32x4_nop(void *p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ f32x4 buf_load_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voffset, unsigned soffset)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, 0));
}
// f32x4 add / subtract as two packed v_pk_add_f32.  hipcc scalarises a <4 x float> fsub into four v_sub_f32 (and folds
// shuffles / fneg back into that fsub), so the subtraction is spelled in asm; it is not volatile: free to schedule.
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b)
{
    const f32x2 lo = pk_sub2(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub2(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// scalar * f32x4 (+ f32x4) as two packed ops
__device__ __forceinline__ f32x4 pk_mul_s(float w, f32x4 x)
{
    const f32x2 w2 = {w, w};
    const f32x2 lo = w2 * __builtin_shufflevector(x, x, 0, 1), hi = w2 * __builtin_shufflevector(x, x, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_fma_s(float w, f32x4 x, f32x4 c)
{
    const f32x2 w2 = {w, w};
    const f32x2 lo = __builtin_elementwise_fma(w2, __builtin_shufflevector(x, x, 0, 1), __builtin_shufflevector(c, c, 0, 1));
    const f32x2 hi = __builtin_elementwise_fma(w2, __builtin_shufflevector(x, x, 2, 3), __builtin_shufflevector(c, c, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_add(f32x4 a, f32x4 b)
{
    const f32x2 lo = __builtin_shufflevector(a, a, 0, 1) + __builtin_shufflevector(b, b, 0, 1);
    const f32x2 hi = __builtin_shufflevector(a, a, 2, 3) + __builtin_shufflevector(b, b, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

#define M3D_LEAKY_SLOPE 0.01f

void m3d_set_error(const char *fmt, ...);
// channel-slice forms of m3d_nchw_to_nhwc / m3d_pack_conv_weight (internal: the deformable groups of m3d_dcn_v2_forward)
int m3d_nchw_to_nhwc_slice(const float *in, int Ctot, int c0, float *out, int N, int C, int H, int W, int out_cs, m3d_stream_t stream);
int m3d_pack_conv_weight_slice(const float *w, int Ctot, int c0, float *packed, int Cout, int Cout_pad, int Cin, int Cin_pad, int kh,
                               int kw, m3d_stream_t stream);

#define M3D_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            m3d_set_error(__VA_ARGS__); \
            return M3D_E_ARG;         \
        }                             \
    } while (0)

#define M3D_HIP(call)                                                             \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) {                                                  \
            m3d_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                          __FILE__, __LINE__);                                    \
            return M3D_E_HIP;                                                     \
        }                                                                         \
    } while (0)

#define M3D_LAUNCH_CHECK()                                                          \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            m3d_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                          __FILE__, __LINE__);                                      \
            return M3D_E_HIP;                                                       \
        }                                                                           \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int imin(int a, int b) { return a < b ? a : b; }

// LeakyReLU as max(v, slope*v) (identical for 0 < slope < 1, signed zeros included): 2 VALU ops instead of 3
__device__ __forceinline__ float leaky(float v) { return fmaxf(v, v * M3D_LEAKY_SLOPE); }
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// ---- class softmax + row score shared by bundle_outputs / score_keys_planar (rpn_kernels.hip) and the planar decode
// (detect_kernels.hip): one definition, so the probabilities a row is SORTED by and the ones it is DECODED with are the same bits
// whichever kernel computes them (M3d_inference_align.py:229-232: softmax over the 4 class logits of an anchor row).
__device__ __forceinline__ f32x4 class_softmax4(f32x4 l)
{
    const float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
    f32x4 e;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { e[c] = expf(l[c] - mx); s += e[c]; }
    f32x4 pr;
#pragma unroll
    for (int c = 0; c < 4; ++c) pr[c] = e[c] / s;
    return pr;
}
__device__ __forceinline__ float fg_score(f32x4 pr) { return fmaxf(fmaxf(pr[1], pr[2]), pr[3]); }
// monotone unsigned image of a float: a > b as floats <=> f32_sortable(a) > f32_sortable(b)
__device__ __forceinline__ unsigned int f32_sortable(float f)
{
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- deterministic split-K reduction shared by the LDS-tiled and the wave-granular implicit GEMMs (igemm_conv.hip) -----------
// ws holds `splits` raw partial sums [split][M][Cout_pad]; they are added in split order and the conv epilogue is applied.
struct SplitkReduceArgs {
    const float *ws, *scale, *shift, *res;
    float *out;
    int M, Cout, Cout_pad, splits, out_cs, res_cs, res_mode, act, sigmoid_from;
};
int m3d_launch_splitk_reduce(const SplitkReduceArgs &a, hipStream_t stream);
