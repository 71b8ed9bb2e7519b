// 1x1 DCNv2 of the bf16 path (`center_align`: model/module/feturealign_mgpu.py:48-99 -- ONE bilinear sample per pixel with the
// pixel's own (dy, dx) and modulation mask, a 128 -> 128 channel mix, + the input as residual; dcn_v2_im2col_cuda.cu:18-47,118-180
// with kernel_h = kernel_w = 1).
//
// Why its own kernel: on the generic deformable implicit-GEMM tile (bf16_conv.hip, K = 128 = two 64-channel K-steps) the two
// center_align launches of a bs-64 step took 0.112 ms each -- 0.12 of the matrix peak AND 0.19 of the HBM roof, the worst family
// of the step by either roof -- because a 2-step K loop is all prologue: offsets -> sampling state -> corner gather ->
// LDS -> MFMA -> epilogue run as one dependent chain per workgroup with a barrier between every link and nothing to overlap it
// with.  The op itself is HBM-bound: 126 MB in + 126 MB out per launch = 0.04 ms at 6.3 TB/s, 16 GFLOP = 0.006 ms of MFMA.
//   * a workgroup = 128 consecutive pixels (256 threads, two workgroups per CU: 69 KB of LDS each);
//   * thread p < 128 builds the sampling state of pixel p ONCE (`dcn_corners`, common.h: the reference's corner rules) and hands
//     it to the 16 gather threads of that pixel through LDS -- while all 256 threads already have the 128 x 128 weight tile in flight;
//   * the gather is line-shaped: 16 consecutive threads fetch the 256-byte channel row of one corner of one pixel (16 bytes
//     each), all 32 corner pieces of a thread (8 pixels x 4 corners) are in flight at once;
//   * combine in fp32 with the mask folded into the corner weights, ONE rounding to bf16 (same expression as the generic kernel's
//     store_tile), into an XOR-swizzled [128 pixels][128 channels] bf16 tile = the B operand of v_mfma_f32_32x32x16_bf16;
//   * 8 K-steps x 4 channel blocks per wave (32 pixels x 128 channels), weights from the LDS tile; epilogue = bf16_tile.h
//     (affine, residual, activation, whole 256-byte rows out).
#include "bf16_tile.h"

#define D1_ROWB 256                       // bytes per LDS row (128 bf16 channels)
#define D1_A 0                            // [128 pixels][256 B]   sampled tile, later the output tile
#define D1_W (128 * D1_ROWB)              // [128 channels][256 B] weights
#define D1_ST (2 * 128 * D1_ROWB)         // [128 pixels][8 words] sampling state: 4 byte offsets, 4 weights
#define D1_SS (D1_ST + 128 * 32)          // [2][128] floats: scale | shift
#define D1_LDS (D1_SS + 2 * 128 * 4)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bf16_dcn1x1_kernel(const Bf16Args a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[D1_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    int tile = blockIdx.x;
    {                                                   // XCD-contiguous tile order: neighbouring pixel rows share an L2
        const int nt = gridDim.x, q = nt >> 3, r = nt & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = tile * 128;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(a.in, a.in_bytes);
    const __amdgpu_buffer_rsrc_t rwgt = make_rsrc(a.wgt, a.wgt_bytes);

    // ---- weights: 128 rows x 256 bytes = 2048 pieces of 16 bytes, 8 per thread, in flight from the first instruction -----------
    u32x4 rw[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i, row = q >> 4, ch = q & 15;
        rw[i] = buf_load_u32x4(rwgt, ((unsigned)row * (unsigned)(a.KT * 64) + (unsigned)ch * 8u) * 2u, 0);
    }
    // ---- sampling state of pixel m0 + tid (threads 0..127) -------------------------------------------------------------------
    if (tid < 128) {
        const int m = m0 + tid;
        const bool ok = m < a.M;
        const int inv = sign_smear(a.M - 1 - m);
        const int mm = ok ? m : 0;
        const float *omp = a.om + (size_t)mm * a.om_cs;
        const float dh = omp[0], dw = omp[1], mk = omp[2];
        const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
        const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
        float wq[4];
        int oq[4], drop[4];
        dcn_corners((float)ho + dh, (float)wo + dw, a.H, a.W, inv, wq, oq, drop);
        u32x4 ob;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned off = ((unsigned)(n * a.H * a.W) + (unsigned)oq[q]) * (unsigned)a.in_cs * 2u;      // garbage where dropped: masked next
            ob[q] = (off & ~(unsigned)drop[q]) | (M3D_BUF_OOB & (unsigned)drop[q]);                          // a dropped corner reads 0
        }
        *reinterpret_cast<u32x4 *>(lds + D1_ST + tid * 32) = ob;
        *reinterpret_cast<f32x4 *>(lds + D1_ST + tid * 32 + 16) = f32x4{wq[0] * mk, wq[1] * mk, wq[2] * mk, wq[3] * mk};
        const bool cok = tid < a.Cout;
        reinterpret_cast<float *>(lds + D1_SS)[tid] = (cok && a.scale) ? a.scale[tid] : (cok ? 1.f : 0.f);
        reinterpret_cast<float *>(lds + D1_SS)[128 + tid] = (cok && a.shift) ? a.shift[tid] : 0.f;
    }
    __syncthreads();

    // ---- gather: pass q covers pixels 16 q .. 16 q + 15, thread = (pixel tid >> 4, 16-byte piece tid & 15) -------------------------
    const int gp = tid >> 4, gc = tid & 15;
    u32x4 cr[8][4];
    f32x4 cw[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = 16 * q + gp;
        const u32x4 ov = *reinterpret_cast<const u32x4 *>(lds + D1_ST + r * 32);
        cw[q] = *reinterpret_cast<const f32x4 *>(lds + D1_ST + r * 32 + 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) cr[q][c] = buf_load_u32x4(rin, ov[c] + (unsigned)gc * 16u, 0);       // OOB marker + < 256 stays out of range
    }
    // weights -> LDS (their loads are the oldest in flight), chunk XOR-swizzled by the row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i, row = q >> 4, ch = q & 15;
        *reinterpret_cast<u32x4 *>(lds + D1_W + row * D1_ROWB + ((ch ^ (row & 15)) << 4)) = rw[i];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = 16 * q + gp;
        u32x4 v;
        // (w1*v1 + w2*v2 + w3*v3 + w4*v4) with the modulation mask folded into the corner weights, fp32, then one rounding to bf16
        // (dcn_v2_im2col_cuda.cu:44-46,174) -- the expression of bf16_conv.hip's store_tile
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 v1 = unpack_bf16(cr[q][0][e]), v2 = unpack_bf16(cr[q][1][e]);
            const f32x2 v3 = unpack_bf16(cr[q][2][e]), v4 = unpack_bf16(cr[q][3][e]);
            const f32x2 s = v1 * cw[q][0] + v2 * cw[q][1] + v3 * cw[q][2] + v4 * cw[q][3];
            v[e] = pack_bf16(s[0], s[1]);
        }
        *reinterpret_cast<u32x4 *>(lds + D1_A + r * D1_ROWB + ((gc ^ (r & 15)) << 4)) = v;
    }
    __syncthreads();

    // ---- 32 pixels x 128 channels per wave: D rows = channels (A operand = weights), columns = pixels ---------------------------
    f32x16 acc[4][1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][0][r] = 0.f;
    const unsigned char *Pb = lds + D1_A + (wave * 32 + l31) * D1_ROWB;
    const unsigned char *Wb = lds + D1_W + l31 * D1_ROWB;
    const int sw = l31 & 15;                               // rows are (multiple of 32) + l31: the swizzle term is per lane
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int co = ((2 * s + lh) ^ sw) << 4;
        const bf16x8 fp = *reinterpret_cast<const bf16x8 *>(Pb + co);
        bf16x8 fw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const bf16x8 *>(Wb + j * 32 * D1_ROWB + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fp, acc[j][0], 0, 0, 0);
    }
    __syncthreads();                                       // every wave is done with the sampled tile: it becomes the output tile

    const int p = wave * 32 + l31;
    int mpix[1] = {m0 + p < a.M ? m0 + p : -1}, lrow[1] = {p};
    conv_epilogue_fast<4, 1>(a, acc, mpix, lrow, 0, 0, lh, reinterpret_cast<const float *>(lds + D1_SS), 128, lds + D1_A);
    __syncthreads();
    const int M = a.M;
    store_otile<128, 128, 256>(a, lds + D1_A, 0, 0, tid, [&](int row) { return m0 + row < M ? m0 + row : -1; });
}

// 1 if the kernel serves the descriptor (M3D_BF16_DCN1X1=0: the generic deformable tile, A/B)
int dcn1x1_applicable(const m3d_conv_bf16_desc *d)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("M3D_BF16_DCN1X1"); on = e ? atoi(e) : 1; }
    if (!on || !d->dcn_offmask || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->groups != 1 || d->wgt_img_stride) return 0;
    if (d->Cin != 128 || d->Cout_pad != 128 || d->Kpad != 128 || d->out_mode != 0 || d->sigmoid_from >= 0 || d->dcn_om_cs < 3) return 0;
    return 1;
}

int launch_dcn1x1(const Bf16Args &a, const m3d_conv_bf16_desc *d, hipStream_t st)
{
    (void)d;
    hipLaunchKernelGGL(bf16_dcn1x1_kernel, dim3(cdiv(a.M, 128)), dim3(256), 0, st, a);
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}
