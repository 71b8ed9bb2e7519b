// Reproducers for write-after-read hazards between a memory instruction and a VALU write of ITS OWN operand registers issued
// 0..2 instructions later (gfx950, ROCm 7.2).  Companion of store_data_war.hip (the case found in the K-pair F(4x4) epilogue:
// hipcc emitted `buffer_store_dwordx4 v[4:7], ...` directly followed by `v_pk_fma_f32 v[4:5], ...` and the stored tile was wrong
// in element 1 of lanes 12-15 / 28-31 / 44-47 / 60-63).  Each case issues
//     <memory instruction using v[100:103] as data and/or v104 / v[104:105] as address>
//     <0, 1 or 2 s_nop states>
//     <VALU write of those registers>
// from every wave of a 1024 x 256 launch and counts wrong dwords:
//   st128 / st64 / st32   buffer_store_dwordx4 / x2 / dword, data overwritten by v_pk_mul_f32 (v_mov_b32 for the dword)
//   gst128                global_store_dwordx4, data overwritten
//   lds128                ds_write_b128, data overwritten
//   staddr                buffer_store_dwordx4, its voffset register overwritten by v_mov_b32 (data stays)
//   ldaddr                buffer_load_dwordx4, its voffset register overwritten by v_mov_b32
// build: hipcc --offload-arch=gfx950 -O3 vmem_war_hazards.hip -o vmem_war_hazards.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

enum { ST128, ST64, ST32, GST128, LDS128, STADDR, LDADDR };
static const char *case_name[] = {"buffer_store_dwordx4 data", "buffer_store_dwordx2 data", "buffer_store_dword data",
                                  "global_store_dwordx4 data", "ds_write_b128 data", "buffer_store_dwordx4 voffset",
                                  "buffer_load_dwordx4 voffset"};

#define NOPSTR(N) ((N) == 0 ? "" : (N) == 1 ? "s_nop 0\n" : "s_nop 1\n")

template <int CASE, int NOPS>
__global__ __launch_bounds__(256) void k(float *out, const float *src, int iters, float a)
{
    __shared__ __attribute__((aligned(16))) float lds[256 * 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t wave_base = ((size_t)blockIdx.x * 4 + wave) * (size_t)iters * 256;      // floats
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + wave_base, 0, (unsigned)iters * 1024u, 0x00020000);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1u << 20, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    const double two = __builtin_bit_cast(double, (unsigned long long)0x4000000040000000ull);
    const unsigned laddr = (unsigned)(uintptr_t)&lds[threadIdx.x * 4];
    for (int it = 0; it < iters; ++it) {
        const float val = a + (float)it;
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(it * 1024);
        float *gp = out + wave_base + (size_t)it * 256 + lane * 4;
#define SETUP "v_mov_b32 v100, %0\nv_mov_b32 v101, %0\nv_mov_b32 v102, %0\nv_mov_b32 v103, %0\nv_mov_b32 v104, %1\ns_nop 4\n"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "memory"
#define EMIT(NS)                                                                                                               \
        if constexpr (CASE == ST128)                                                                                            \
            asm volatile(SETUP "buffer_store_dwordx4 v[100:103], v104, %2, %3 offen\n" NS                                       \
                         "v_pk_mul_f32 v[100:101], v[100:101], %4\nv_pk_mul_f32 v[102:103], v[102:103], %4\n"                  \
                         :: "v"(val), "v"(voff), "s"(r), "s"(so), "v"(two) : CLOB);                                             \
        else if constexpr (CASE == ST64)                                                                                        \
            asm volatile(SETUP "buffer_store_dwordx2 v[100:101], v104, %2, %3 offen\n" NS                                       \
                         "v_pk_mul_f32 v[100:101], v[100:101], %4\n"                                                           \
                         :: "v"(val), "v"(voff), "s"(r), "s"(so), "v"(two) : CLOB);                                             \
        else if constexpr (CASE == ST32)                                                                                        \
            asm volatile(SETUP "buffer_store_dword v100, v104, %2, %3 offen\n" NS "v_mov_b32 v100, %4\n"                       \
                         :: "v"(val), "v"(voff), "s"(r), "s"(so), "v"(-1.0f) : CLOB);                                           \
        else if constexpr (CASE == GST128)                                                                                      \
            asm volatile(SETUP "global_store_dwordx4 %2, v[100:103], off\n" NS                                                  \
                         "v_pk_mul_f32 v[100:101], v[100:101], %3\nv_pk_mul_f32 v[102:103], v[102:103], %3\n"                  \
                         :: "v"(val), "v"(voff), "v"(gp), "v"(two) : CLOB);                                                     \
        else if constexpr (CASE == LDS128)                                                                                      \
            asm volatile(SETUP "ds_write_b128 %2, v[100:103]\n" NS                                                              \
                         "v_pk_mul_f32 v[100:101], v[100:101], %3\nv_pk_mul_f32 v[102:103], v[102:103], %3\n"                  \
                         :: "v"(val), "v"(voff), "v"(laddr), "v"(two) : CLOB);                                                  \
        else if constexpr (CASE == STADDR)                                                                                      \
            asm volatile(SETUP "buffer_store_dwordx4 v[100:103], v104, %2, %3 offen\n" NS "v_mov_b32 v104, %4\n"               \
                         :: "v"(val), "v"(voff), "s"(r), "s"(so), "v"(0x7fffff00u) : CLOB);       /* out of range: store dropped */ \
        else                                                                                                                    \
            asm volatile(SETUP "buffer_load_dwordx4 v[100:103], v104, %2, %3 offen\n" NS "v_mov_b32 v104, %4\n"                \
                         "s_waitcnt vmcnt(0)\nbuffer_store_dwordx4 v[100:103], %1, %5, %3 offen\ns_nop 4\n"                    \
                         :: "v"(val), "v"(voff), "s"(rs), "s"(so), "v"(0x7fffff00u), "s"(r) : CLOB);   /* OOB load returns 0 */
        if constexpr (NOPS == 0) { EMIT("") } else if constexpr (NOPS == 1) { EMIT("s_nop 0\n") } else { EMIT("s_nop 1\n") }
        if constexpr (CASE == LDS128) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const float4 v = *reinterpret_cast<float4 *>(&lds[threadIdx.x * 4]);
            *reinterpret_cast<float4 *>(gp) = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CASE, int NOPS>
void run(int iters)
{
    const int grid = 1024;
    const size_t n = (size_t)grid * 4 * iters * 256;
    float *d, *src;
    hipMalloc(&d, n * 4);
    hipMemset(d, 0, n * 4);
    hipMalloc(&src, 1 << 20);
    std::vector<float> hs((1 << 20) / 4);
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = 1.0f + (float)(i / 256);       // LDADDR: word i of iteration it reads 1 + it
    hipMemcpy(src, hs.data(), 1 << 20, hipMemcpyHostToDevice);
    k<CASE, NOPS><<<grid, 256>>>(d, src, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<float> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    const int words = CASE == ST64 ? 2 : (CASE == ST32 ? 1 : 4);
    size_t bad = 0, tot = 0;
    long long lane_hist[64] = {0}, elem_hist[4] = {0};
    for (size_t w = 0; w < (size_t)grid * 4; ++w)
        for (int it = 0; it < iters; ++it)
            for (int i = 0; i < 256; ++i) {
                if ((i & 3) >= words) continue;
                ++tot;
                if (h[(w * iters + it) * 256 + i] != 1.0f + (float)it) { ++bad; ++lane_hist[i >> 2]; ++elem_hist[i & 3]; }
            }
    printf("%-30s %d wait state(s) before the VALU write: %9zu wrong dwords of %zu", case_name[CASE], NOPS, bad, tot);
    if (bad) {
        printf("  [elements x/y/z/w: %lld %lld %lld %lld; lanes:", elem_hist[0], elem_hist[1], elem_hist[2], elem_hist[3]);
        for (int l = 0; l < 64; ++l) if (lane_hist[l]) printf(" %d", l);
        printf("]");
    }
    printf("\n");
    hipFree(d); hipFree(src);
}

int main()
{
    const int it = 64;
#define ALLN(C) run<C, 0>(it); run<C, 1>(it); run<C, 2>(it);
    ALLN(ST128) ALLN(ST64) ALLN(ST32) ALLN(GST128) ALLN(LDS128) ALLN(STADDR) ALLN(LDADDR)
    return 0;
}
