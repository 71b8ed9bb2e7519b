// Attempt to reproduce, in isolation, the wrong-predicate event seen in the bf16 DCNv2 sampling code (the
// kernels now avoid SGPR lane masks there altogether, csrc/common.h dcn_corners): a four-compare range test compiled to back-to-back v_cmp -> s_and_b64 chains gave lanes 48-63 of a wave the wrong
// predicate about once per thousand workgroups.  Every wave evaluates the test in that form (A) and in the single-compare form
// (B) on the same operands, `iters` times with fresh operands, while the other waves of the workgroup stream MFMAs / LDS
// traffic; any lane where A != B is counted.
// Blind spot (found later): the companions are the odd waves of each workgroup, and wave i of a workgroup runs on SIMD i % 4,
// so a tester never shares a SIMD with an MFMA companion -- sgpr_mask_mfma.hip closes it (and is clean as well).
//   hipcc --offload-arch=gfx950 -O3 vcmp_sand_hazard.hip -o vcmp_sand_hazard.bin && ./vcmp_sand_hazard.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void probe(const float *__restrict__ in, unsigned long long *mism, int iters, float H, float W, int n)
{
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long bad = 0;
    if (wave & 1) {                                      // companion waves: MFMA + LDS traffic on the same SIMDs / CU
        f32x16 acc = {0};
        float a = in[tid & 255], b = in[(tid * 7) & 255];
        for (int i = 0; i < iters * 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            lds[(tid + i * 64) & 4095] = acc[i & 15];
            a += lds[(tid * 3 + i) & 4095] * 1e-9f;
        }
        if (acc[0] == 123.456f) mism[1] = 1;
        return;
    }
    size_t idx = ((size_t)blockIdx.x * 256 + tid) * 2;
    for (int i = 0; i < iters; ++i) {
        const float h0 = in[idx % n], w0 = in[(idx + 1) % n];
        idx += (size_t)gridDim.x * 512;
        const float hi = h0 * 60.f - 5.f, wi = w0 * 180.f - 10.f;     // straddles -1 .. H, -1 .. W
        // form A: four compares, combined by the compiler into v_cmp / s_and chains
        float ra = 0.f;
        if (hi > -1.f && wi > -1.f && hi < H && wi < W) {
            const int hl = (int)floorf(hi), wl = (int)floorf(wi);
            const float lh = hi - (float)hl, lw = wi - (float)wl;
            if (hl >= 0 && wl >= 0) ra += (1.f - lh) * (1.f - lw);
            if (hl >= 0 && wl + 1 <= (int)W - 1) ra += (1.f - lh) * lw;
            if (hl + 1 <= (int)H - 1 && wl >= 0) ra += lh * (1.f - lw);
            if (hl + 1 <= (int)H - 1 && wl + 1 <= (int)W - 1) ra += lh * lw;
        }
        // form B: one compare per decision
        float rb = 0.f;
        if (fminf(fminf(hi, wi) + 1.f, -fmaxf(hi - H, wi - W)) > 0.f) {
            const int hl = (int)floorf(hi), wl = (int)floorf(wi);
            const float lh = hi - (float)hl, lw = wi - (float)wl;
            const int hr = (int)H - 2 - hl, wr = (int)W - 2 - wl;
            if ((hl | wl) >= 0) rb += (1.f - lh) * (1.f - lw);
            if ((hl | wr) >= 0) rb += (1.f - lh) * lw;
            if ((hr | wl) >= 0) rb += lh * (1.f - lw);
            if ((hr | wr) >= 0) rb += lh * lw;
        }
        const unsigned long long m = __ballot(__float_as_uint(ra) != __float_as_uint(rb));
        bad |= m;
        if (m && lane == 0) atomicAdd(&mism[0], (unsigned long long)__popcll(m));
    }
    if (bad && lane == 0) atomicOr(&mism[2], bad);
}

int main()
{
    const int n = 1 << 20;
    std::vector<float> h(n);
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) * (1.0f / 16777216.0f); }
    float *d; unsigned long long *m;
    hipMalloc(&d, n * 4); hipMalloc(&m, 64);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(m, 0, 64);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d, m, 200, 48.f, 160.f, n);
        hipDeviceSynchronize();
        unsigned long long r[3];
        hipMemcpy(r, m, 24, hipMemcpyDeviceToHost);
        printf("run %d: %llu lane-evaluations differ between the two forms (lane mask %016llx) out of %.1f M\n", rep, r[0], r[2],
               4096.0 * 128 * 200 / 1e6);
    }
    return 0;
}
