// libm3dssd_hip.so: error reporting, events, and the drop-in for dcn_v2_cuda_forward.
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void m3d_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *m3d_last_error(void) { return g_err; }
extern "C" int m3d_abi_version(void) { return M3D_ABI_VERSION; }
// "name:sha256[:16];..." of every source this library was built from (build/src_hash.h, written by the Makefile): lets a
// measurement taken with one build (profiles/*_hbm_traffic.json) be told apart from the build that is loaded now.
#include "build/src_hash.h"
extern "C" const char *m3d_source_hashes(void) { return M3D_SRC_HASHES; }

// ------------------------------------------------------------------------------------------
extern "C" int m3d_event_create(void **ev)
{
    hipEvent_t e;
    M3D_HIP(hipEventCreate(&e));
    *ev = (void *)e;
    return M3D_OK;
}
extern "C" int m3d_event_record(void *ev, m3d_stream_t stream)
{
    M3D_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return M3D_OK;
}
extern "C" int m3d_event_elapsed_ms(void *start, void *stop, float *ms)
{
    M3D_HIP(hipEventSynchronize((hipEvent_t)stop));
    M3D_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return M3D_OK;
}
extern "C" int m3d_event_destroy(void *ev)
{
    M3D_HIP(hipEventDestroy((hipEvent_t)ev));
    return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// Shader clock the chip actually holds over a window (bench.py: `sclk_under_step_ghz`): ONE wave samples s_memtime (one tick per
// shader cycle) and s_memrealtime (constant 100 MHz) when it starts, sleeps until `ticks` of the 100 MHz clock have passed and
// samples both again; run on a side stream while the step replays on the main one, d(memtime) / d(realtime) x 100 MHz is the
// clock the MFMA peak of that window has to be priced at (the 157.3 TFLOP/s behind `roofline.frac` assume 2.4 GHz; under the fp32
// kernels the part holds 2.0-2.1 at ~1.2 kW: DESIGN.md).  out = {memtime0, realtime0, memtime1, realtime1}.
__global__ void clock_probe_kernel(long long *out, long long ticks)
{
    if (threadIdx.x != 0) return;
    const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(127);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    out[0] = c0; out[1] = r0; out[2] = __builtin_readcyclecounter(); out[3] = r1;
}

extern "C" int m3d_clock_probe(long long *out4_dev, double seconds, m3d_stream_t stream)
{
    M3D_REQUIRE(out4_dev && seconds > 0 && seconds <= 1.0, "clock_probe: bad arguments (window <= 1 s)");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out4_dev, (long long)(seconds * 1e8));
    M3D_LAUNCH_CHECK();
    return M3D_OK;
}

// ------------------------------------------------------------------------------------------
// Drop-in for dcn_v2_cuda_forward (model/DCNv2/src/dcn_v2_cuda.c:10-102) on NCHW tensors.
// The reference loops over images on the host and round-trips a `columns` scratch through HBM;
// here the whole batch is one fused gather+GEMM launch.  NCHW<->NHWC conversion and weight packing
// happen in the caller-provided workspace (the engine path keeps everything NHWC and skips them).
static inline long long rup(long long a, long long b) { return (a + b - 1) / b * b; }

struct DcnWs {
    long long in_off, om_off, w_off, out_off, out2_off, total;
    int cp, co_pad, om_cs, out_cs, ho, wo;
};

// `c` = channels of ONE deformable group (the whole input when deformable_group == 1); groups > 1 add a second output
// buffer: group g accumulates onto group g-1's result through the residual input of the conv epilogue (ping-pong).
static DcnWs dcn_ws(int n, int c, int h, int w, int co, int kh, int kw, int stride, int pad, int dil, int groups)
{
    DcnWs s;
    s.cp = (int)rup(c, 32);   // deformable tiles use BK = 32
    s.co_pad = (int)rup(co, 64);
    s.ho = (h + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
    s.wo = (w + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
    s.om_cs = (int)rup(3 * kh * kw, 4);
    s.out_cs = (int)rup(co, 4);
    long long o = 0;
    s.in_off = o;  o += rup((long long)n * h * w * s.cp * 4, 256);
    s.om_off = o;  o += rup((long long)n * s.ho * s.wo * s.om_cs * 4, 256);
    s.w_off = o;   o += rup((long long)s.co_pad * kh * kw * s.cp * 4, 256);
    s.out_off = o; o += rup((long long)n * s.ho * s.wo * s.out_cs * 4, 256);
    s.out2_off = o;
    if (groups > 1) o += rup((long long)n * s.ho * s.wo * s.out_cs * 4, 256);
    s.total = o;
    return s;
}

extern "C" long long m3d_dcn_v2_workspace_bytes(int batch, int channels, int height, int width, int channels_out,
                                                int kernel_h, int kernel_w, int stride, int pad, int dilation)
{
    return dcn_ws(batch, channels, height, width, channels_out, kernel_h, kernel_w, stride, pad, dilation, 1).total;
}

extern "C" long long m3d_dcn_v2_workspace_bytes_grouped(int batch, int channels, int height, int width, int channels_out,
                                                        int kernel_h, int kernel_w, int stride, int pad, int dilation,
                                                        int deformable_group)
{
    if (deformable_group < 1 || channels % deformable_group) return -1;
    return dcn_ws(batch, channels / deformable_group, height, width, channels_out, kernel_h, kernel_w, stride, pad, dilation,
                  deformable_group).total;
}

extern "C" int m3d_dcn_v2_forward(const float *input, const float *weight, const float *bias, const float *offset,
                                  const float *mask, float *output, int batch, int channels, int height, int width,
                                  int channels_out, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h,
                                  int pad_w, int dilation_h, int dilation_w, int deformable_group, void *workspace,
                                  long long workspace_bytes, m3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    M3D_REQUIRE(input && weight && bias && offset && mask && output && workspace, "dcn_v2_forward: null pointer");
    M3D_REQUIRE(deformable_group >= 1 && channels % deformable_group == 0,
                "dcn_v2_forward: deformable_group (%d) must divide the input channels (%d)", deformable_group, channels);
    M3D_REQUIRE(stride_h == stride_w && pad_h == pad_w && dilation_h == dilation_w,
                "dcn_v2_forward: anisotropic stride/pad/dilation not supported");
    M3D_REQUIRE(((uintptr_t)workspace & 255) == 0, "dcn_v2_forward: workspace must be 256-byte aligned");
    const int G = deformable_group, cg = channels / G;
    const DcnWs s = dcn_ws(batch, cg, height, width, channels_out, kernel_h, kernel_w, stride_h, pad_h, dilation_h, G);
    if (workspace_bytes < s.total) {
        m3d_set_error("dcn_v2_forward: workspace %lld < %lld bytes (m3d_dcn_v2_workspace_bytes%s)", workspace_bytes, s.total,
                      G > 1 ? "_grouped" : "");
        return M3D_E_WORKSPACE;
    }
    M3D_REQUIRE(s.ho > 0 && s.wo > 0, "dcn_v2_forward: empty output");
    char *ws = (char *)workspace;
    float *in_nhwc = (float *)(ws + s.in_off), *om = (float *)(ws + s.om_off);
    float *wp = (float *)(ws + s.w_off);
    float *outs[2] = {(float *)(ws + s.out_off), (float *)(ws + s.out2_off)};
    const int kk = kernel_h * kernel_w;
    int rc;
    // Deformable group g (dcn_v2_im2col_cuda.cu:139-156): input channels [g*cg, (g+1)*cg) are sampled at the positions of
    // offset channels [g*2kk, (g+1)*2kk) with mask channels [g*kk, (g+1)*kk); the weights are not grouped, so the output
    // is the sum over g of a deformable conv of that channel slice with weight[:, g*cg:(g+1)*cg] -- one fused gather + GEMM
    // launch per group, accumulated through the epilogue's residual input, bias added once.
    for (int g = 0; g < G; ++g) {
        if (s.cp != cg) M3D_HIP(hipMemsetAsync(in_nhwc, 0, (size_t)batch * height * width * s.cp * 4, stream));
        if ((rc = m3d_nchw_to_nhwc_slice(input, channels, g * cg, in_nhwc, batch, cg, height, width, s.cp, stream))) return rc;
        if ((rc = m3d_nchw_to_nhwc_slice(offset, 2 * kk * G, g * 2 * kk, om, batch, 2 * kk, s.ho, s.wo, s.om_cs, stream))) return rc;
        if ((rc = m3d_nchw_to_nhwc_slice(mask, kk * G, g * kk, om + 2 * kk, batch, kk, s.ho, s.wo, s.om_cs, stream))) return rc;
        if ((rc = m3d_pack_conv_weight_slice(weight, channels, g * cg, wp, channels_out, s.co_pad, cg, s.cp, kernel_h, kernel_w,
                                             stream)))
            return rc;
        m3d_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.in = in_nhwc; d.in_cs = s.cp; d.N = batch; d.H = height; d.W = width; d.Cin = s.cp;
        d.wgt = wp; d.Cout = channels_out; d.Cout_pad = s.co_pad;
        d.kh = kernel_h; d.kw = kernel_w; d.stride = stride_h; d.pad = pad_h; d.dil = dilation_h;
        d.Ho = s.ho; d.Wo = s.wo; d.out = outs[g & 1]; d.out_cs = s.out_cs;
        d.shift = g == 0 ? bias : nullptr;   // bias GEMM-with-ones of dcn_v2_cuda.c:72-78 folded into the epilogue
        if (g > 0) { d.res = outs[(g - 1) & 1]; d.res_cs = s.out_cs; d.res_mode = 0; }
        d.sigmoid_from = -1;
        d.dcn_offmask = om; d.dcn_om_cs = s.om_cs;
        if ((rc = m3d_conv2d_forward(&d, stream))) return rc;
    }
    return m3d_nhwc_to_nchw(outs[(G - 1) & 1], s.out_cs, output, batch, channels_out, s.ho, s.wo, stream);
}
