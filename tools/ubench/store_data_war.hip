// Reproducer: a VALU write of the data registers of a buffer_store_dwordx4 that was issued 0..3 instructions earlier.
// Found in round 4 in the K-pair F(4x4) epilogue (wino44_conv.hip): hipcc (ROCm 7.2, gfx950) emitted
//     buffer_store_dwordx4 v[4:7], v48, s[28:31], s0 offen
//     v_pk_fma_f32 v[4:5], v[38:39], v[32:33], v[42:43]
// back to back, and the stored tile had wrong values in element 1 of lanes 12-15 / 28-31 / 44-47 / 60-63 of exactly that store:
// the store reads its data registers AFTER it has issued, and the packed FMA had already overwritten part of them.  Older
// targets document this as the ">64-bit store data" hazard (1-2 wait states, inserted by the compiler); the compiler inserts
// nothing for gfx950.  This probe issues the pair with 0..3 s_nop states between them, under a stream of other stores, for
// several overwriting instructions, and counts wrong dwords in memory.
// build: hipcc --offload-arch=gfx950 -O3 store_data_war.hip -o store_data_war.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

enum { OW_PK_FMA, OW_PK_MOV, OW_MOV, OW_FMA };

#define NOP0 ""
#define NOP1 "s_nop 0\n"
#define NOP2 "s_nop 1\n"
#define NOP3 "s_nop 2\n"

template <int OW, int NOPS, bool SGPR_SOFF>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float two)
{
    // every (block, wave, iteration) stores 64 lanes x 16 bytes of the value a + iteration; the registers are then overwritten
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t wave_base = ((size_t)blockIdx.x * 4 + wave) * (size_t)iters * 256;      // floats
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + wave_base, 0, (unsigned)iters * 1024u, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    for (int it = 0; it < iters; ++it) {
        const float val = a + (float)it;
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(it * 1024);
#define BODY(NOPSTR)                                                                                                       \
        if constexpr (SGPR_SOFF) {                                                                                          \
            if constexpr (OW == OW_PK_FMA)                                                                                  \
                asm volatile("v_mov_b32 v100, %0\nv_mov_b32 v101, %0\nv_mov_b32 v102, %0\nv_mov_b32 v103, %0\ns_nop 4\n"   \
                             "buffer_store_dwordx4 v[100:103], %1, %2, %3 offen\n" NOPSTR                                    \
                             "v_pk_fma_f32 v[100:101], v[100:101], %4, %4\nv_pk_fma_f32 v[102:103], v[102:103], %4, %4\n"   \
                             :: "v"(val), "v"(voff), "s"(r), "s"(so), "v"(__builtin_bit_cast(double, (unsigned long long)0x4000000040000000ull)) \
                             : "v100", "v101", "v102", "v103", "memory");                                                   \
            else if constexpr (OW == OW_PK_MOV)                                                                             \
                asm volatile("v_mov_b32 v100, %0\nv_mov_b32 v101, %0\nv_mov_b32 v102, %0\nv_mov_b32 v103, %0\ns_nop 4\n"   \
                             "buffer_store_dwordx4 v[100:103], %1, %2, %3 offen\n" NOPSTR                                    \
                             "v_pk_mul_f32 v[100:101], v[100:101], %4\nv_pk_mul_f32 v[102:103], v[102:103], %4\n"                           \
                             :: "v"(val), "v"(voff), "s"(r), "s"(so), "v"(__builtin_bit_cast(double, (unsigned long long)0x4000000040000000ull)) \
                             : "v100", "v101", "v102", "v103", "memory");                                                   \
            else if constexpr (OW == OW_MOV)                                                                                \
                asm volatile("v_mov_b32 v100, %0\nv_mov_b32 v101, %0\nv_mov_b32 v102, %0\nv_mov_b32 v103, %0\ns_nop 4\n"   \
                             "buffer_store_dwordx4 v[100:103], %1, %2, %3 offen\n" NOPSTR                                    \
                             "v_mov_b32 v101, %4\nv_mov_b32 v100, %4\nv_mov_b32 v103, %4\nv_mov_b32 v102, %4\n"             \
                             :: "v"(val), "v"(voff), "s"(r), "s"(so), "v"(two) : "v100", "v101", "v102", "v103", "memory"); \
            else                                                                                                            \
                asm volatile("v_mov_b32 v100, %0\nv_mov_b32 v101, %0\nv_mov_b32 v102, %0\nv_mov_b32 v103, %0\ns_nop 4\n"   \
                             "buffer_store_dwordx4 v[100:103], %1, %2, %3 offen\n" NOPSTR                                    \
                             "v_fma_f32 v101, v101, %4, %4\nv_fma_f32 v100, v100, %4, %4\nv_fma_f32 v103, v103, %4, %4\n"   \
                             "v_fma_f32 v102, v102, %4, %4\n"                                                              \
                             :: "v"(val), "v"(voff), "s"(r), "s"(so), "v"(two) : "v100", "v101", "v102", "v103", "memory"); \
        } else {                                                                                                            \
            const unsigned vo2 = voff + so;                                                                                 \
            asm volatile("v_mov_b32 v100, %0\nv_mov_b32 v101, %0\nv_mov_b32 v102, %0\nv_mov_b32 v103, %0\ns_nop 4\n"       \
                         "buffer_store_dwordx4 v[100:103], %1, %2, 0 offen\n" NOPSTR                                        \
                         "v_pk_fma_f32 v[100:101], v[100:101], %3, %3\nv_pk_fma_f32 v[102:103], v[102:103], %3, %3\n"       \
                         :: "v"(val), "v"(vo2), "s"(r), "v"(__builtin_bit_cast(double, (unsigned long long)0x4000000040000000ull)) \
                         : "v100", "v101", "v102", "v103", "memory");                                                       \
        }
        if constexpr (NOPS == 0) { BODY(NOP0) } else if constexpr (NOPS == 1) { BODY(NOP1) } else if constexpr (NOPS == 2) { BODY(NOP2) } else { BODY(NOP3) }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static const char *ow_name[] = {"v_pk_fma_f32 x2", "v_pk_mul_f32 x2", "v_mov_b32 x4", "v_fma_f32 x4"};

template <int OW, int NOPS, bool SGPR_SOFF>
void run(int iters)
{
    const int grid = 1024;
    const size_t n = (size_t)grid * 4 * iters * 256;
    float *d;
    hipMalloc(&d, n * 4);
    hipMemset(d, 0, n * 4);
    k<OW, NOPS, SGPR_SOFF><<<grid, 256>>>(d, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    std::vector<float> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    long long lane_hist[64] = {0}, elem_hist[4] = {0};
    for (size_t w = 0; w < (size_t)grid * 4; ++w)
        for (int it = 0; it < iters; ++it)
            for (int i = 0; i < 256; ++i) {
                const float v = h[(w * iters + it) * 256 + i];
                if (v != 1.0f + (float)it) { ++bad; ++lane_hist[i >> 2]; ++elem_hist[i & 3]; }
            }
    printf("%-16s soffset %-7s %d wait state(s) between store and overwrite: %zu wrong dwords of %zu", ow_name[OW],
           SGPR_SOFF ? "SGPR" : "literal", NOPS, bad, n);
    if (bad) {
        printf("  [elements x/y/z/w: %lld %lld %lld %lld; lanes:", elem_hist[0], elem_hist[1], elem_hist[2], elem_hist[3]);
        for (int l = 0; l < 64; ++l) if (lane_hist[l]) printf(" %d", l);
        printf("]");
    }
    printf("\n");
    hipFree(d);
}

int main()
{
    const int it = 64;
#define ALLN(OW, S) run<OW, 0, S>(it); run<OW, 1, S>(it); run<OW, 2, S>(it); run<OW, 3, S>(it);
    ALLN(OW_PK_FMA, true) ALLN(OW_PK_MOV, true) ALLN(OW_MOV, true) ALLN(OW_FMA, true)
    run<OW_PK_FMA, 0, false>(it); run<OW_PK_FMA, 1, false>(it);
    return 0;
}
