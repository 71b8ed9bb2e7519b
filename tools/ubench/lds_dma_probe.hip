// LDS-DMA probe (gfx950): buffer_load_dwordx4 ... offen lds  via __builtin_amdgcn_raw_ptr_buffer_load_lds.
// Questions: (1) where does lane l's 16 bytes land (M0 base + l*16?), (2) what does an out-of-range lane write (zeros?),
// (3) are per-lane source offsets arbitrary.   hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o lds_dma_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(const unsigned *in, unsigned nbytes, unsigned *out)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(in), 0, nbytes, 0x00020000);
    const unsigned lane = threadIdx.x;
    // lane l reads source chunk (l * 7 + 3) % 64 (arbitrary permutation); lanes 5 and 40 are sent out of range
    unsigned voff = ((lane * 7 + 3) % 64) * 16u;
    if (lane == 5 || lane == 40) voff = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + 256), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}

int main()
{
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
    unsigned *din, *dout;
    hipMalloc(&din, 4096);
    hipMalloc(&dout, 8192);
    hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, din, 1024u, dout);   // only the first 1024 bytes (64 chunks) are in range
    std::vector<unsigned> o(2048);
    hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const unsigned src = ((l * 7 + 3) % 64) * 4;
        for (int e = 0; e < 4; ++e) {
            const unsigned got = o[256 + l * 4 + e];
            const unsigned want = (l == 5 || l == 40) ? 0u : 1000 + src + e;
            if (got != want) { if (bad < 8) printf("lane %d elem %d: got %u (0x%x) want %u\n", l, e, got, got, want); ++bad; }
        }
    }
    int touched = 0;
    for (int i = 0; i < 2048; ++i) if ((i < 256 || i >= 512) && o[i] != 0xDEADBEEFu) ++touched;
    printf("lane-linear 16 B per lane at base + lane*16, out-of-range lanes write zeros: %s (mismatches %d, stray writes %d)\n",
           bad == 0 && touched == 0 ? "YES" : "NO", bad, touched);
    return 0;
}
