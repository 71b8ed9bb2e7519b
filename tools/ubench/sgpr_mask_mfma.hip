// Third isolated reproducer attempt for the dropped-corner event of the DCNv2 sampling code.  What the
// bisection of the real kernel says (tools/dcn_determinism.py on stripped builds): the event needs v_mfma instructions in the
// kernel (two builds without them: 0 differing launches of 400; with them 100 %) and two waves on a SIMD (one workgroup per CU:
// never), not the gathers.  vcmp_sand_hazard.hip could not see that: its MFMA companions are the odd waves of each workgroup
// and wave i of a workgroup lives on SIMD i % 4 -- tester and companion never shared a SIMD.  Here EVERY wave alternates an MFMA
// burst with the select chains (v_cmp -> s_and_b64 -> v_cndmask on SGPR lane masks, compiler generated) and checks them against
// the sign-smear form the kernels use now; register use is padded so that exactly two waves fit a SIMD.
//   hipcc --offload-arch=gfx950 -O3 sgpr_mask_mfma.hip -o sgpr_mask_mfma.bin && ./sgpr_mask_mfma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int sign_smear(int x)
{
    int m;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(x));
    return m;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void probe(const float *__restrict__ in, unsigned long long *mism, int iters, int H, int W, int n)
{
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned long long bad = 0;
    f32x16 acc[10];                                    // 160 accumulator registers: two waves per SIMD, not more
    for (int j = 0; j < 10; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (__bf16)in[(tid + e) & 255]; fb[e] = (__bf16)in[(tid * 3 + e) & 255]; }
    size_t idx = ((size_t)blockIdx.x * 256 + tid) * 3;
    for (int i = 0; i < iters; ++i) {
        const float h0 = in[idx % n], w0 = in[(idx + 1) % n], mk = in[(idx + 2) % n];
        idx += (size_t)gridDim.x * 768;
        const float h_im = h0 * (float)(H + 10) - 5.f, w_im = w0 * (float)(W + 20) - 10.f;     // straddles the image
        // form A: selects on compare results (lane masks in SGPRs, combined with s_and_b64)
        const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
        const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
        const float lh = h_im - (float)hl, lw = w_im - (float)wl, uh = 1.f - lh, uw = 1.f - lw;
        const int hh = hl + 1, wh = wl + 1;
        const bool c1 = inside && hl >= 0 && wl >= 0, c2 = inside && hl >= 0 && wh <= W - 1;
        const bool c3 = inside && hh <= H - 1 && wl >= 0, c4 = inside && hh <= H - 1 && wh <= W - 1;
        const float wa[4] = {c1 ? uh * uw * mk : 0.f, c2 ? uh * lw * mk : 0.f, c3 ? lh * uw * mk : 0.f, c4 ? lh * lw * mk : 0.f};
        const int oa[4] = {c1 ? hl * W + wl : 0, c2 ? hl * W + wh : 0, c3 ? hh * W + wl : 0, c4 ? hh * W + wh : 0};
        // the kernel's K-step between building and using the state
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 10; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j], 0, 0, 0);
        lds[(tid + i * 64) & 8191] = acc[i % 10][i & 15];
        // form B: no lane masks
        const float tin = fminf(fminf(h_im, w_im) + 1.f, -fmaxf(h_im - (float)H, w_im - (float)W));
        const int xin = (int)__float_as_uint(tin);
        const int out = sign_smear((xin - 1) | xin);
        const int hr = H - 2 - hl, wr = W - 2 - wl;
        const int k[4] = {sign_smear(hl | wl) | out, sign_smear(hl | wr) | out, sign_smear(hr | wl) | out, sign_smear(hr | wr) | out};
        const float wf[4] = {uh * uw * mk, uh * lw * mk, lh * uw * mk, lh * lw * mk};
        const int of[4] = {hl * W + wl, hl * W + wh, hh * W + wl, hh * W + wh};
        bool diff = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned wb = __float_as_uint(wf[q]) & ~(unsigned)k[q];
            const int ob = of[q] & ~k[q];
            diff = diff || (__float_as_uint(wa[q]) != wb && !(wa[q] == 0.f && __uint_as_float(wb) == 0.f)) || oa[q] != ob;
        }
        const unsigned long long m = __ballot(diff);
        bad |= m;
        if (m && lane == 0) atomicAdd(&mism[0], (unsigned long long)__popcll(m));
    }
    float t = 0.f;
    for (int j = 0; j < 10; ++j) t += acc[j][j];
    if (t == 123.456f) mism[1] = 1;
    if (bad && lane == 0) atomicOr(&mism[2], bad);
}

int main()
{
    const int n = 1 << 20, iters = 1000, H = 48, W = 160;
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.f; }
    float *d;
    unsigned long long *m, hm[3] = {0, 0, 0};
    if (hipMalloc(&d, n * 4) != hipSuccess || hipMalloc(&m, 24) != hipSuccess) return 1;
    (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(m, hm, 24, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 20; ++rep) probe<<<4096, 256>>>(d, m, iters, H, W, n);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(hm, m, 24, hipMemcpyDeviceToHost);
    printf("states evaluated: %.3g, lanes where the two forms differ: %llu, lane mask of the differences: %016llx\n",
           20.0 * 4096 * 256 * iters, hm[0], hm[2]);
    return 0;
}
