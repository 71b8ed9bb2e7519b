/*
 * m3dssd_hip.h -- C ABI of libm3dssd_hip.so: the MI355X (gfx950) native hot path of M3DSSD.
 *
 * Plain pointers and sizes only (no torch types).  All device pointers are BORROWED (the
 * caller -- torch, or any other allocator -- owns them); nothing here allocates device
 * memory except the legacy host-pointer `_nms` twin, which mirrors the reference's own
 * behaviour.  Every entry point is stream-ordered on the `hipStream_t` it is given,
 * re-entrant per stream, and returns an int status: 0 = ok, <0 = M3D_E_* (the host layer
 * maps these to Python RuntimeError, like THError/THArgCheck did in the reference).
 *
 * Reference interfaces replaced (paths relative to mumianyuxin/M3DSSD):
 *   m3d_dcn_v2_forward ......... void dcn_v2_cuda_forward(THCudaTensor *input, *weight, *bias, *ones,
 *                                 *offset, *mask, *output, *columns, kernel_h, kernel_w, stride_h,
 *                                 stride_w, pad_h, pad_w, dilation_h, dilation_w, deformable_group)
 *                                 model/DCNv2/src/dcn_v2_cuda.h:9-17, dcn_v2_cuda.c:10-102,
 *                                 kernel model/DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:118-180
 *   _nms / m3d_nms_sorted_dev .. void _nms(int* keep_out, int* num_out, const float* boxes_host,
 *                                 int boxes_num, int boxes_dim, float nms_overlap_thresh, int device_id)
 *                                 lib/nms/gpu_nms.hpp:1-2, lib/nms/nms_kernel.cu:91-144 (kernel :34-78)
 *   m3d_conv2d_forward ......... the cuDNN/ATen nn.Conv2d + BatchNorm2d(eval) + LeakyReLU + residual
 *                                 chains of model/pose_dla_dcn.py:107-121,261-269,379-389 and the RPN
 *                                 heads model/M3d_inference_align.py:66-210, plus (deformable mode)
 *                                 the fused im2col+GEMM of dcn_v2_cuda.c:80-96 without `columns`
 *   m3d_wino_conv3x3_forward ... the same Conv2d 3x3 stride-1 layers through Winograd F(2x2,3x3)
 *   m3d_head_mlp_forward ....... the 3-layer 1x1 heads, model/M3d_inference_align.py:77-210 (one launch per head)
 *   m3d_stem_conv7x7 ........... DLA.base_layer, model/pose_dla_dcn.py:336-340
 *   m3d_conv3x3_c16 ............ DLA.level0 (3x3 16->16 at full resolution), model/pose_dla_dcn.py:341-342
 *   m3d_maxpool2x2 ............. Tree.downsample nn.MaxPool2d(2,2), pose_dla_dcn.py:306,316
 *   m3d_upsample2x_add ......... IDAUp: depthwise ConvTranspose2d(4, s2, p1) + skip add,
 *                                 pose_dla_dcn.py:536-538,550-552
 *   m3d_anchor_select .......... softmax over classes + fg_prob + topk(k=1)/max/hard mask,
 *                                 M3d_inference_align.py:229-234, feturealign_mgpu.py:58-62,160-164
 *   m3d_align_offsets .......... offset/mask synthesis of shape_align (feturealign_mgpu.py:119-136,
 *                                 166-183) and center_align (:67-89)
 *   m3d_anab_pool* ............. PAPAModule weighted adaptive average pooling, attention.py:136-147
 *   m3d_softmax_rows ........... nn.Softmax(dim=-1) on the 337-key logits, attention.py:208
 *   m3d_bundle_outputs ......... flatten_tensor x13 + torch.cat + softmax, M3d_inference_align.py:280-301,
 *                                 lib/rpn_util.py:892-901 (+ the score key used by top-k)
 *   m3d_decode_rows ............ lib/rpn_util.py:1442-1521 (im_detect_3d decode) and
 *                                 bbox_transform_inv :1137-1186, for the top-N-pre rows
 */
#ifndef M3DSSD_HIP_H
#define M3DSSD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *m3d_stream_t; /* hipStream_t */

enum {
    M3D_OK = 0,
    M3D_E_ARG = -1,      /* bad argument / unsupported shape (message via m3d_last_error) */
    M3D_E_HIP = -2,      /* a HIP runtime call or kernel launch failed */
    M3D_E_WORKSPACE = -3 /* workspace too small */
};

/* Thread-local text of the last error returned on this thread. */
const char *m3d_last_error(void);
/* Library/ABI version.  Policy: the number moves whenever a caller built against the previous header could misbehave --
 * a signature or struct layout changes, the CONTRACT of an argument changes (a workspace size rule, a row limit, a
 * precondition), or entry points are added that the Python binding resolves at load time.  A binding checks it once
 * after dlopen and refuses a library of another version (m3dssd_amd/_hip.py).
 *   4 -> 5 (round 6): m3d_conv_bf16_desc.dcn_ws sized by m3d_conv_bf16_dcn_ws_bytes(N, Ho, Wo) (was ">= 1024 bytes"),
 *                     m3d_nms_sorted_dev / m3d_topk_decode accept up to 16 384 rows per image (was 4 096), 15 entry
 *                     points added in round 5 (anab_attend_*, head_mlp2 / tail2 / qkvs bf16, tree_entry, frontend2, ...),
 *                     the round-4 experimental forms (bf16_wino2, bf16_frontend, bf16_head_mlp) left the product library. */
#define M3D_ABI_VERSION 5
int m3d_abi_version(void);
/* "file:sha256[:16];file:sha256[:16];..." of the sources (csrc .hip / .h files and this header) the loaded library was built from. */
const char *m3d_source_hashes(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), NHWC activations.
 *   GEMM view: M = N*Ho*Wo pixels, N = Cout, K = kh*kw*Cin (tap-major, channel-minor).
 *   out = act( conv(in, wgt) * scale[c] + shift[c] + res )   (res_mode 0; scale/shift/res optional)
 *   out = act( (conv(in, wgt) + res) * scale[c] + shift[c] )  (res_mode 1)
 * Deformable mode (dcn_offmask != NULL): the A operand is the modulated bilinear gather of
 * DCNv2 (dcn_v2_im2col_cuda.cu:18-47,129-178) computed on the fly -- no `columns` buffer.
 * ------------------------------------------------------------------------------------------ */
typedef struct m3d_conv_desc {
    const float *in;          /* NHWC view: in[(n*H + h)*W + w][c], pixel stride in_cs floats   */
    int in_cs;
    int N, H, W, Cin;         /* Cin % 16 == 0                                                  */
    const float *wgt;         /* packed [Cout_pad][K], K = (i*kw + j)*Cin + c; rows >= Cout = 0 */
    long long wgt_img_stride; /* 0, or floats between per-image weight sets (needs Ho*Wo % BM == 0) */
    int Cout, Cout_pad;       /* Cout_pad % 32 == 0                                             */
    int kh, kw, stride, pad, dil;
    int Ho, Wo;
    float *out;
    int out_cs;               /* NHWC pixel stride (out_nchw == 0)                              */
    int out_nchw;             /* 1: planar out[n*out_img_stride + c*Ho*Wo + p]                  */
    long long out_img_stride;
    const float *scale;       /* [Cout] or NULL (=1)                                            */
    const float *shift;       /* [Cout] or NULL (=0)                                            */
    const float *res;         /* NHWC residual view or NULL                                     */
    int res_cs;
    int res_mode;             /* 0: acc*scale+shift+res   1: (acc+res)*scale+shift (BN after the add) */
    int act;                  /* 0 none, 1 LeakyReLU(0.01)                                      */
    int sigmoid_from;         /* channels >= this get sigmoid instead of act; <0: none          */
    const float *dcn_offmask; /* NHWC [.., 3*kh*kw]: 2k=dh, 2k+1=dw, 2*kh*kw+k=mask; NULL=plain */
    int dcn_om_cs;
    float *splitk_ws;         /* optional scratch for split-K (small-M layers); NULL = never split   */
    long long splitk_ws_bytes;
} m3d_conv_desc;

int m3d_conv2d_forward(const m3d_conv_desc *d, m3d_stream_t stream);
/* Split-K plan of m3d_conv2d_forward for this descriptor: *splits (1 = no split) and the scratch bytes
 * (splits * N*Ho*Wo * Cout_pad * 4) the caller should provide through splitk_ws to enable it.  A layer whose
 * output tiles cannot give each of the 256 CUs a workgroup is split along K; partial sums are added in split
 * order by a second launch (deterministic), which also applies the epilogue. */
int m3d_conv2d_splitk_plan(const m3d_conv_desc *d, int *splits, long long *ws_bytes);

/* Wave-granular convolution / deformable convolution (csrc/dcn_wave.hip): same descriptor as m3d_conv2d_forward
 * (dcn_offmask optional), except that `wgt` is the packed [Cout_pad, kh*kw*Cin] matrix in MFMA-fragment order
 * [Cout_pad/32][kh*kw*Cin/8][h=2][r=32][t=4].  Each wave owns 32 pixels x 128 channels (64 when Cout_pad is an odd multiple of 64) and works alone (no workgroup
 * barrier).  m3d_conv_wave_applicable returns the number of waves the layer yields, or 0 when it does not apply
 * (needs Cin % 32 == 0, in_cs % 32 == 0, 128-byte aligned input, Cout_pad % 64 == 0, NHWC output; per-image weights
 * -- wgt_img_stride in floats between fragment-packed sets -- need Ho*Wo % 32 == 0) or when there are too few waves to fill the chip; the caller then stays on m3d_conv2d_forward. */
int m3d_conv_wave_applicable(const m3d_conv_desc *d);
/* Thin layers (too few 32 x 128 tiles) can still take this path split along K across waves: *splits and the scratch bytes to
 * pass through splitk_ws / splitk_ws_bytes (partials are reduced in split order by a second launch, which applies the
 * epilogue).  Without a workspace the layer is simply not split. */
int m3d_conv_wave_splitk_plan(const m3d_conv_desc *d, int *splits, long long *ws_bytes);
int m3d_conv_wave_forward(const m3d_conv_desc *d, m3d_stream_t stream);

/* Winograd F(2x2,3x3) variant for 3x3 / stride 1 / pad 1 / even H,W plain convolutions (same descriptor; `wgt`
 * must point to the Winograd-transformed weights U = G g G^T packed in fragment order
 * [16 xi][Cout_pad/32][Cin/8][64][4], see m3dssd_amd/engine.py:pack_wino).  2.25x fewer MFMA FLOPs, fp32. */
int m3d_wino_conv3x3_forward(const m3d_conv_desc *d, m3d_stream_t stream);
/* Which kernel m3d_wino_conv3x3_forward runs for this descriptor: 1 = register-resident one-wave kernel (32 tiles x 32
 * channels x 16 transform positions in 512 registers, no LDS), 0 = LDS kernel (64 tiles x 32 channels per 512-thread
 * workgroup) -- chosen when the layer yields too few waves for the 1024 SIMDs or needs the sigmoid epilogue. */
int m3d_wino_conv3x3_variant(const m3d_conv_desc *d);
/* Thin layers can still use the wave kernel split along K across waves (and only the split form has the sigmoid epilogue):
 * *splits and the scratch bytes to pass through splitk_ws / splitk_ws_bytes; partials are reduced in split order by a second
 * launch.  m3d_wino_conv3x3_variant answers for the descriptor as given (with or without a workspace). */
int m3d_wino_conv3x3_splitk_plan(const m3d_conv_desc *d, int *splits, long long *ws_bytes);
/* Same with the kernel chosen by the caller: variant -1 = automatic, 0 = LDS kernel, 1 = wave kernel (tests, tuning). */
int m3d_wino_conv3x3_forward_ex(const m3d_conv_desc *d, int variant, m3d_stream_t stream);
/* Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32 (csrc/wino44_conv.hip): 4x fewer MFMA FLOPs than the direct 3x3 convolution
 * (F(2x2,3x3): 2.25x).  Same descriptor and epilogue (scale / shift, residual, LeakyReLU) as m3d_wino_conv3x3_forward; `wgt`
 * points to U = G g G^T (6x6 per filter, fp64 -> fp32) packed [Cout_pad/32][Cin/16][36 xi][2][64][4], see
 * m3dssd_amd/engine.py:pack_wino44.  Needs H % 4 == W % 4 == 0, Cin % 16 == 0, Cout_pad % 64 == 0, no sigmoid channels, NHWC
 * output (m3d_wino44_applicable = 1).  fp32 rounding error ~7x that of direct summation (rms 1.5e-6 of the output scale at
 * Cin = 128). */
int m3d_wino44_applicable(const m3d_conv_desc *d);
int m3d_wino44_conv3x3_forward(const m3d_conv_desc *d, m3d_stream_t stream);
/* nb = 16-channel blocks per wave: 1 or 0 = 64-channel workgroups, two per CU (round 4: the faster form on every layer); 2 =
 * 128-channel workgroups, one per CU with the whole register file (needs Cout_pad % 128 == 0; round 3's form, kept for A/B runs).
 * Layers whose 64-channel workgroups would fill less than ~60 % of the 512 CU slots run "K-pair" workgroups (512 threads: the two
 * halves of the input channels side by side, accumulators traded through LDS at the end): m3d_wino44_kpair() says whether. */
int m3d_wino44_conv3x3_forward_ex(const m3d_conv_desc *d, int nb, m3d_stream_t stream);
int m3d_wino44_kpair(const m3d_conv_desc *d);
/* Split-K for maps whose 16-tile strips x 64-channel blocks do not fill the chip (512 -> 512 @ 12x40 at bs 8):
 * *splits slices of >= 64 input channels run as gridDim.z, raw partial outputs go to splitk_ws ([splits][N*H*W][Cout_pad]
 * fp32, *ws_bytes), a second launch adds them in slice order and applies the epilogue.  *splits = 1 / *ws_bytes = 0: no split.
 * m3d_wino44_conv3x3_forward and _ex with nb == 0 or nb == the split form (1 = 64-channel workgroups by default) split when the
 * descriptor carries a workspace of at least *ws_bytes; _ex with the other nb runs unsplit.  The split form and the fill
 * threshold are process-wide and read ONCE from the environment (M3D_W44_SPLIT_NB = 1 | 2, M3D_W44_SPLIT_FILL = workgroups a
 * layer must reach to run unsplit): size the workspace with m3d_wino44_splitk_plan in the process that launches, not from a
 * table computed under another setting. */
int m3d_wino44_splitk_plan(const m3d_conv_desc *d, int *splits, long long *ws_bytes);
/* Same as _ex, and every thread of the launch first touches a few 128-byte lines of [touch, touch + touch_bytes): the engine
 * passes the U tensor of the NEXT F(4x4) layer, which is then in the memory-side cache instead of HBM when that layer's B
 * fragments (four transform positions of lookahead) ask for it -- 256 -> 256 @ 24x80: 0.087 -> 0.063 ms in the network.
 * touch = NULL: none. */
int m3d_wino44_conv3x3_forward_touch(const m3d_conv_desc *d, int nb, const void *touch, long long touch_bytes, m3d_stream_t stream);

/* Touches one dword of every 128-byte line of [p, p + bytes): warms the memory-side cache with a weight tensor ahead of a
 * kernel whose operand lookahead does not cover an HBM round trip (the engine issues it before the F(4x4,3x3) layers). */
int m3d_cache_touch(const void *p, long long bytes, m3d_stream_t stream);
/* Fed-input upload as a kernel (replaces the per-frame im.cuda() of lib/rpn_util.py:1427-1429 for the pipelined detector):
 * copies `bytes` from the 16-byte-aligned pinned-host buffer whose address the 8-byte word *src_slot holds -- src_slot itself
 * lives in pinned host memory and is read when the kernel RUNS, so one captured graph serves a different frame set every replay
 * -- to dst (device, 16-byte aligned).  *src_slot == NULL: no copy. */
int m3d_upload_indirect(const void *const *src_slot, void *dst, long long bytes, m3d_stream_t stream);

/* Average launch geometry chosen for a descriptor (for roofline bookkeeping / tests). */
int m3d_conv2d_tile(const m3d_conv_desc *d, int *bm, int *bn, int *bk, int *grid);

/* ------------------------------------------------------------------------------------------
 * bf16 path (BASELINE.json configs[2]: bs = 64, bf16 storage, MFMA v_mfma_f32_32x32x16_bf16, fp32 accumulation).
 * Same operator as m3d_conv2d_forward -- conv (+ optional DCNv2 gather) + folded BatchNorm / bias + residual +
 * LeakyReLU / sigmoid in ONE launch -- on bf16 NHWC activations and bf16 weights packed [Cout_pad][Kpad],
 * K index = (i*kw + j)*Cin + c, zero padded to Kpad % 64 == 0 and Cout_pad % 32 == 0.  scale / shift / offsets / masks
 * stay fp32.  A kernel larger than 1x1 needs a power-of-two Cin; Cin % 8 == 0, in_cs % 8 == 0.
 * `groups` > 1 runs that many independent problems of identical geometry in one launch (the RPN heads that read the same
 * map): group g reads in + g*in_group_off, wgt + g*wgt_group_off, scale/shift + g*ss_group_off and writes
 * out + g*out_group_off (element units of the respective type).
 * ------------------------------------------------------------------------------------------ */
typedef struct m3d_conv_bf16_desc {
    const void *in;           /* bf16 NHWC view, pixel stride in_cs elements                                 */
    int in_cs;
    int N, H, W, Cin;
    const void *wgt;          /* bf16 [Cout_pad][Kpad]                                                       */
    long long wgt_img_stride; /* 0, or bf16 elements between per-image weight sets (needs Ho*Wo % 128 == 0)  */
    int Cout, Cout_pad, Kpad;
    int kh, kw, stride, pad;
    int Ho, Wo;
    void *out;
    int out_cs;               /* NHWC pixel stride in output elements (modes 0, 1)                           */
    int out_mode;             /* 0: bf16 NHWC   1: fp32 NHWC   2: fp32 planar out[n*out_img_stride + c*Ho*Wo + p] */
    long long out_img_stride;
    const float *scale;       /* [groups][Cout] or NULL (=1)                                                 */
    const float *shift;       /* [groups][Cout] or NULL (=0)                                                 */
    const void *res;          /* bf16 NHWC residual view or NULL                                             */
    int res_cs;
    int res_mode;             /* 0: acc*scale+shift+res   1: (acc+res)*scale+shift                           */
    int act;                  /* 0 none, 1 LeakyReLU(0.01)                                                   */
    int sigmoid_from;         /* channels >= this get sigmoid instead of act; <0: none                       */
    const float *dcn_offmask; /* fp32 NHWC [.., 3*kh*kw] (2k = dh, 2k+1 = dw, 2*kh*kw + k = mask); NULL = plain conv */
    int dcn_om_cs;
    int groups;
    long long in_group_off, wgt_group_off, out_group_off;
    int ss_group_off;
    /* Deformable 3x3 / stride 1 / pad 1 only, all three optional (NULL / 0 = implicit-GEMM kernel as before): an fp16 copy of
     * `wgt` (same [Cout_pad][Kpad] layout; bf16 -> fp16 is exact for |w| in [6.1e-5, 65504]) and m3d_conv_bf16_dcn_ws_bytes(N, Ho,
     * Wo) bytes of device scratch enable the LDS-patch kernel (csrc/bf16_dcn_patch.hip): every 16 x 16 (H % 16 == 0) or 8 x 16
     * (H % 8 == 0) pixel tile samples from an LDS-resident fp16 window sized to its own largest |offset| when that is <= 9 (<= 6)
     * and raises its word in dcn_ws otherwise; the implicit-GEMM kernel launched behind it recomputes the tiles that did
     * (W % 16 == 0, Cin % 32 == 0, Cout_pad % 128 == 0) -- decided on the device per tile, no host synchronisation.
     * Mixed rounding: a recomputing workgroup rewrites its whole 128-pixel tile, so which of the two kernels' roundings (fp16
     * window vs bf16 gather; both inside the bf16 tolerance) a pixel carries depends on the OFFSET DATA of its neighbourhood;
     * for given inputs the output is deterministic.  `res` must not alias `out` on this path (checked: M3D_E_ARG). */
    const void *wgt_f16;
    void *dcn_ws;
    long long dcn_ws_bytes;
    /* Plain 3x3 / stride 1 / pad 1, optional (NULL = halo-tile / implicit-GEMM kernels as before): the same weights in the
     * fragment order of csrc/bf16_conv_wide.hip -- [Cout_pad/128][Cin/32][9 taps][2 K-steps of 16][4 blocks of 32 channels]
     * [64 lanes][8] with lane = 32 * (k / 8) + (channel % 32) (m3dssd_amd/engine_bf16.py:PackedBf16.wave3x3) -- enable the
     * 128 x 128 wave-tile kernel on maps with H % 8 == 0, W % 16 == 0, Cin % 32 == 0, Cout_pad % 128 == 0, bf16 NHWC output. */
    const void *wgt_wave;
} m3d_conv_bf16_desc;
int m3d_conv_bf16_forward(const m3d_conv_bf16_desc *d, m3d_stream_t stream);
/* Which kernel m3d_conv_bf16_forward launches for `d` (profiling labels; no launch): 0 = implicit-GEMM tile
 * (bf16_conv_kernel), 1 = 3x3 halo tile of 8 x 16 pixels, 2 = 3x3 halo tile of 8 x 32 pixels (bf16_conv3x3_halo_kernel),
 * 3 / 4 = deformable 3x3 with the sampling window in LDS, 8 x 16 / 16 x 16 pixel patches (bf16_dcn_patch_kernel; the
 * implicit-GEMM kernel is launched behind it and recomputes, on the device, the tiles whose offsets do not fit),
 * 5 = 3x3 with 128 x 128 wave tiles (bf16_conv3x3_wide_kernel; needs wgt_wave),
 * 6 = deformable 1x1, 128 -> 128 channels, bf16 NHWC output (bf16_dcn1x1_kernel: center_align, feturealign_mgpu.py:48-99),
 * 8 = 3x3 64 -> 64 channels on maps that tile 8 x 32 pixels, persistent workgroups with the weights resident in LDS
 *     (bf16_conv3x3_c64_kernel: DLA level2). */
/* Bytes of dcn_ws the LDS-patch DCNv2 kernel needs for N x Ho x Wo output pixels (one flag word per pixel tile; 0 = the map does not tile). */
long long m3d_conv_bf16_dcn_ws_bytes(int N, int Ho, int Wo);
int m3d_conv_bf16_variant(const m3d_conv_bf16_desc *d);

/* Entry of a DLA tree of the bf16 path in one launch (model/pose_dla_dcn.py:314-327, :107-121): bottom = MaxPool2d(2, 2)(x) (written
 * when `bottom` is not NULL), res = affine(Conv1x1(bottom)) (Tree.project), t = LeakyReLU(affine(Conv3x3 stride 2 pad 1 (x)))
 * (tree1.conv1) -- x is read once.  Cin % 32 == 0, Cout % 64 == 0, even H / W.  wfrag fp16 [Cout/32][Cin/32][10][2][64 lanes][8]:
 * slice ws, chunk c, tap t (0..8 = 3x3 taps row-major, 9 = the 1x1), block b, lane l (row r = l % 16, k-group l / 16), element e =
 * scale[ch] * W[ch][32 c + 8 (l / 16) + e][tap] with ch = 32 ws + 8 (r / 4) + 4 b + r % 4 (m3dssd_amd/engine_bf16.py:
 * pack_tree_entry); shift1 / shiftp fp32 [Cout]; all views bf16 NHWC with pixel strides in elements (% 8 == 0). */
typedef struct m3d_tree_entry_bf16_desc {
    const void *in;
    int in_cs, N, H, W, Cin, Cout;
    const void *wfrag;
    const float *shift1, *shiftp;
    void *t; int t_cs;
    void *res; int res_cs;
    void *bottom; int bottom_cs;
} m3d_tree_entry_bf16_desc;
int m3d_tree_entry_bf16_applicable(const m3d_tree_entry_bf16_desc *d);
int m3d_tree_entry_bf16_forward(const m3d_tree_entry_bf16_desc *d, m3d_stream_t stream);

/* Fused 3-layer RPN head of the bf16 path (model/M3d_inference_align.py:77-210): [1x1 128 -> 256, affine, LeakyReLU] ->
 * [1x1 256 -> 256, affine, LeakyReLU] -> [1x1 256 -> Cout, affine] per 128-pixel tile in ONE launch, hidden activations in LDS.
 * `groups` heads that read the same map share the launch: weights bf16 row-major [groups][256][128], [groups][256][256],
 * [groups][Cout_pad = 64][256]; scale / shift fp32 [groups][256], [groups][256], [groups][Cout]; output planar fp32
 * out[g*out_group_off + img*out_img_stride + c*HW + p]. */
typedef struct m3d_head_bf16_desc {
    const void *in;           /* bf16 [M][in_cs], first 128 channels used */
    int in_cs;
    long long M;
    int Cin;                  /* 128 */
    const void *w1, *w2, *w3;
    const float *s1, *t1, *s2, *t2, *s3, *t3;
    int Cout, Cout_pad;       /* Cout <= 64, Cout_pad == 64 */
    float *out;
    long long out_group_off, out_img_stride;
    int HW;
    int groups;
} m3d_head_bf16_desc;
int m3d_head_mlp_bf16_forward(const m3d_head_bf16_desc *d, m3d_stream_t stream);
/* Round-5 form of the fused head (csrc/bf16_head_mlp2.hip): the weights of layers 1 / 2 live in registers for the whole launch (wave w
 * of the 8-wave workgroup owns output channels [32w, 32w + 32)), hidden activations fp16 in LDS, layers 2 / 3 on fp16 MFMA.  The
 * caller folds the BatchNorm scales into the weights and packs them (m3dssd_amd/engine_bf16.py: pack_head2):
 *   w1f bf16 [groups][8 waves][8 K-steps][64 lanes][8], w2f fp16 [groups][8][16][64][8]: lane l of wave w, K-step s, element e =
 *       scale[ch] * W[ch][16 s + 8 (l / 32) + e] with ch = 32 w + 16 ((r % 8) / 4) + 4 (r / 8) + r % 4, r = l % 32;
 *   w3 fp16 [groups][64][256] row-major, scale folded, rows >= Cout zero; t1 / t2 [groups][256], t3 [groups][64] fp32 shifts.
 * Same input / output conventions as m3d_head_mlp_bf16_forward (128 input channels, Cout <= 64, planar fp32 output). */
typedef struct m3d_head2_bf16_desc {
    const void *in;           /* bf16 [M][in_cs], first 128 channels used */
    int in_cs;
    long long M;
    const void *w1f, *w2f, *w3;
    const float *t1, *t2, *t3;
    int Cout;
    float *out;
    long long out_group_off, out_img_stride;
    int HW;
    int groups;
} m3d_head2_bf16_desc;
int m3d_head_mlp2_bf16_forward(const m3d_head2_bf16_desc *d, m3d_stream_t stream);
/* The two 1x1 layers behind the 3x3 convolution of the class head (M3d_inference_align.py:66-76: 256 -> 256 + affine + LeakyReLU,
 * 256 -> Cout <= 256 + affine) in one launch, same scheme: waf bf16 / wbf fp16 fragments [8 waves][16 K-steps][64 lanes][8] in the
 * layout of w2f above (scales folded, rows >= Cout of wbf zero), t1 / t2 fp32 [256] shifts (t2 past Cout ignored); input bf16
 * [M][in_cs >= 256]; output planar fp32 out[img * out_img_stride + c * HW + p]. */
typedef struct m3d_tail2_bf16_desc {
    const void *in;
    int in_cs;
    long long M;
    const void *waf, *wbf;
    const float *t1, *t2;
    int Cout;
    float *out;
    long long out_img_stride;
    int HW;
} m3d_tail2_bf16_desc;
int m3d_head_tail2_bf16_forward(const m3d_tail2_bf16_desc *d, m3d_stream_t stream);
/* The bias-free 1x1 projections of ANAB (model/module/attention.py:169-173, 183-200) over one 128-channel input in one launch:
 * wf bf16 fragments [16 row blocks][8 K-steps][64 lanes][8] (the layout of w1f above with the row blocks in the place of the waves)
 * of the stacked matrix [query rows padded with zeros to q_rows | key | value (kv_rows in all) | gates (s_rows) | zeros up to 512];
 * q bf16 [M][q_cs] gets all q_rows (the padding rows as zeros), kv bf16 [M][kv_cs], s fp32 [M][s_cs] = sigmoid(gates). */
typedef struct m3d_qkvs_bf16_desc {
    const void *in;
    int in_cs;
    long long M;
    const void *wf;
    void *q; int q_cs, q_rows;
    void *kv; int kv_cs, kv_rows;
    float *s; int s_cs, s_rows;
} m3d_qkvs_bf16_desc;
int m3d_anab_qkvs_bf16_forward(const m3d_qkvs_bf16_desc *d, m3d_stream_t stream);

/* HBM-bound helpers of the bf16 path: NHWC bf16 views (pixel strides in bf16 elements, multiples of 8), fp32 arithmetic.
 * m3d_stem_conv7x7_bf16: DLA.base_layer from the fp32 [N][3][H][W] image (is_u8 = 0; img_h/img_w/mean3/stds3 ignored) or
 * from uint8 BGR frames [N][img_h][img_w][3] with the reference's test-time Preprocess fused into the loads (is_u8 = 1,
 * mean3 / stds3 host pointers as in m3d_stem_conv7x7_u8); wgt [7*7*3][16] fp32. */
int m3d_stem_conv7x7_bf16(const void *img, int is_u8, int img_h, int img_w, const float *mean3, const float *stds3,
                          const float *wgt, const float *scale, const float *shift, void *out, int out_cs, int N, int H, int W,
                          m3d_stream_t stream);
/* Fused front end of the bf16 path (csrc/bf16_frontend2.hip): base_layer 7x7 3->16 -> level0 3x3 16->16 -> level1 3x3/2 16->32,
 * each + folded BN + LeakyReLU (pose_dla_dcn.py:336-345,391-397) in ONE launch: the image is read once, only the level1 map
 * [N][H/2][W/2][out_cs >= 32] bf16 is written.  img / is_u8 / img_h / img_w / mean3 / stds3 as in m3d_stem_conv7x7_bf16.  fp16 inside the kernel (the tiles never leave LDS), the
 * BatchNorm scales folded into fp16 weights by the caller, the shifts added as the C operand of the MFMA chains; stem on
 * v_mfma_f32_32x32x16_f16 with two adjacent output pixels per column.  Operands (m3dssd_amd/engine_bf16.py: pack_frontend_f16):
 *   w_stem_frag fp16 [7 tap rows][2 K-steps][64 lanes][8]: lane l, element e = weight * scale of channel 4 * ((l % 32) / 8) +
 *               (l % 4) [(l % 32) % 8 < 4: pixel shift 0, else 1], colour e % 4 (3 = zero), tap column 4 * kstep + 2 * (l / 32)
 *               + e / 4 - shift (outside [0, 7): zero); colour slot 3 of tap (0, 0) = the channel's BatchNorm shift (the kernel
 *               writes 1.0 into that slot of every image pixel; t_stem itself is not read);
 *   w_l0 / w_l1 fp16 [16 | 32][160], k = tap * 16 + c (k >= 144 zero), scale folded; t_* fp32 shifts [16], [16], [32].
 */
int m3d_frontend2_bf16_forward(const void *img, int is_u8, int img_h, int img_w, const float *mean3, const float *stds3,
                               const void *w_stem_frag, const float *t_stem, const void *w_l0, const float *t_l0,
                               const void *w_l1, const float *t_l1, void *out, int out_cs, int N, int H, int W,
                               m3d_stream_t stream);
/* ANAB attention of the bf16 path in one launch (model/module/attention.py:207-211 + the BatchNorm / LeakyReLU after the block):
 * out[p] = act((softmax_k(q[p] . khat[k]) @ vhat + res[p]) * scale + shift) per image, replacing the logits GEMM / row softmax /
 * P.V GEMM sequence (m3d_conv_bf16_forward with per-image weights, m3d_softmax_rows_bf16).  q bf16 [B*HW][q_cs] (channels
 * [Ck, Ck_pad) zero), khat bf16 [B][keys_pad][Ck_pad], vhatT bf16 [B][Cv][keys_pad] (rows >= keys ignored), res bf16 [B*HW][res_cs]
 * or NULL, scale / shift fp32 [Cv] or NULL, out bf16 [B*HW][out_cs].  Built for Ck_pad = 192, Cv = 128; HW % 128 == 0.
 * PRECONDITION (not checked): the padding -- khat rows [keys, keys_pad) and vhatT columns [keys, keys_pad) -- must hold FINITE
 * values (zeros; m3d_anab_pool_nested* leave what the caller allocated, the engine allocates with zeros): the padded keys get
 * probability 0, and 0 * (Inf | NaN) in the P.V product would be NaN. */
int m3d_anab_attend_bf16(const void *q, int q_cs, const void *khat, const void *vhatT, int B, int HW, int Ck_pad, int keys,
                         int keys_pad, int Cv, const void *res, int res_cs, const float *scale, const float *shift, int act,
                         void *out, int out_cs, m3d_stream_t stream);
int m3d_maxpool2x2_bf16(const void *in, int in_cs, void *out, int out_cs, int N, int H, int W, int C, m3d_stream_t stream);
int m3d_upsample2x_add_bf16(const void *in, int in_cs, const float *wgt /*[4][4][C] fp32*/, const void *skip, int skip_cs,
                            void *out, int out_cs, int N, int H, int W, int C, m3d_stream_t stream);
int m3d_f32_to_bf16(const float *src, void *dst, long long n /* % 8 == 0 */, m3d_stream_t stream);
/* softmax over the first `valid` fp32 columns of each row (pixel stride cs) -> bf16 probabilities (row stride out_cs <= 512,
 * columns [valid, out_cs) zeroed): the P operand of the bf16 P.V GEMM of ANAB (attention.py:208-209). */
int m3d_softmax_rows_bf16(const float *x, int rows, int valid, int cs, void *out, int out_cs, m3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused RPN head (model/M3d_inference_align.py:77-210): per-pixel MLP
 *   [1x1 Cin->256 + affine + LeakyReLU] -> 1x1 256->256 + affine + LeakyReLU -> 1x1 256->Cout + affine
 * in ONE launch; hidden activations stay in LDS.  Cin = 128 with w1 given (3 layers) or Cin = 256 with
 * w1 = NULL (2 layers: `in` is already the first hidden activation, e.g. after the 3x3 cls conv).
 * Weights are packed in MFMA-fragment order: element W[J*32 + r][G*8 + h*4 + t] (row-tile J, k-group G,
 * r < 32, h < 2, t < 4) at float index ((J*(K/8) + G)*64 + h*32 + r)*4 + t; rows zero-padded to 256 for
 * w1/w2 and to Cout_pad in {64, 256} for w3.  Output is planar: out[n*out_img_stride + c*HW + p].
 * ------------------------------------------------------------------------------------------ */
typedef struct m3d_mlp_desc {
    const float *in;          /* NHWC view [M][in_cs], first Cin channels used                     */
    int in_cs;
    long long M;              /* pixels = N*H*W                                                    */
    int Cin;
    const float *w1, *s1, *t1;
    const float *w2, *s2, *t2;
    const float *w3, *s3, *t3;
    int Cout, Cout_pad;
    float *out;
    long long out_img_stride;
    int HW;
} m3d_mlp_desc;
int m3d_head_mlp_forward(const m3d_mlp_desc *d, m3d_stream_t stream);
/* n (<= 16) independent heads of the same depth, M and Cout_pad in ONE launch (grid.y = head): the regression heads
 * that read the same aligned feature map (M3d_inference_align.py:139-176) have no mutual dependency. */
int m3d_head_mlp_forward_batched(const m3d_mlp_desc *d, int n, m3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Drop-in for dcn_v2_cuda_forward: NCHW contiguous fp32 device tensors, exactly the reference
 * argument meaning.  `workspace` replaces the reference's `ones`/`columns` scratch tensors
 * (dcn_v2_func.py:28): the caller allocates m3d_dcn_v2_workspace_bytes() bytes.
 * deformable_group = G > 1 (model/DCNv2/test.py:169-179; the M3DSSD path itself uses 1): G must divide `channels`;
 * offset is [N, G*2*kh*kw, Ho, Wo], mask [N, G*kh*kw, Ho, Wo] (dcn_v2_im2col_cuda.cu:139-156); the workspace then comes
 * from m3d_dcn_v2_workspace_bytes_grouped (returns -1 for an invalid G).
 * ------------------------------------------------------------------------------------------ */
long long m3d_dcn_v2_workspace_bytes(int batch, int channels, int height, int width, int channels_out,
                                     int kernel_h, int kernel_w, int stride, int pad, int dilation);
long long m3d_dcn_v2_workspace_bytes_grouped(int batch, int channels, int height, int width, int channels_out,
                                             int kernel_h, int kernel_w, int stride, int pad, int dilation,
                                             int deformable_group);
int m3d_dcn_v2_forward(const float *input, const float *weight, const float *bias, const float *offset,
                       const float *mask, float *output, int batch, int channels, int height, int width,
                       int channels_out, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h,
                       int pad_w, int dilation_h, int dilation_w, int deformable_group, void *workspace,
                       long long workspace_bytes, m3d_stream_t stream);

/* Weight packing: [Cout, Cin, kh, kw] (torch layout) -> [Cout_pad, kh*kw*Cin_pad] (tap-major, zero pad). */
int m3d_pack_conv_weight(const float *w, float *packed, int Cout, int Cout_pad, int Cin, int Cin_pad, int kh,
                         int kw, m3d_stream_t stream);
/* Layout changes at the op boundary. */
int m3d_nchw_to_nhwc(const float *in, float *out, int N, int C, int H, int W, int out_cs, m3d_stream_t stream);
int m3d_nhwc_to_nchw(const float *in, int in_cs, float *out, int N, int C, int H, int W, m3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Backbone helpers (NHWC, channel-sliced views allowed through *_cs).
 * ------------------------------------------------------------------------------------------ */
/* 7x7, 3->16, stride 1, pad 3 on an NCHW image; fused affine (folded BN) + LeakyReLU; NHWC out. */
int m3d_stem_conv7x7(const float *img_nchw, const float *wgt /*[7*7*3][16]*/, const float *scale,
                     const float *shift, float *out, int out_cs, int N, int H, int W, m3d_stream_t stream);
/* Test-time input path of the reference: uint8 BGR frames [N][img_h][img_w][3] (what cv2.imread returns),
 * zero border at the bottom / right up to H x W, /255, -mean, /stds in float32 (mean / stds: HOST pointers to 3 floats, indexed
 * by BGR channel position exactly as lib/augmentations.py:44-57 applies them), BGR -> RGB, HWC -> CHW
 * (augmentations.py:472-501, dataloader.py:943-950).  Bit-identical to the numpy arithmetic.  m3d_preprocess_u8 writes the
 * [N][3][H][W] float tensor; m3d_stem_conv7x7_u8 feeds the stem directly (no float image in HBM: 1.5 MB instead of 5.9 MB
 * per 384x1280 frame cross the boundary). */
int m3d_preprocess_u8(const unsigned char *frames_bgr, int N, int img_h, int img_w, const float *mean3, const float *stds3,
                      float *out_nchw, int H, int W, m3d_stream_t stream);
int m3d_stem_conv7x7_u8(const unsigned char *frames_bgr, int img_h, int img_w, const float *mean3, const float *stds3,
                        const float *wgt, const float *scale, const float *shift, float *out, int out_cs, int N, int H, int W,
                        m3d_stream_t stream);
/* 3x3, 16 -> 16, stride 1, pad 1 (DLA level0, pose_dla_dcn.py:341-342) as a direct VALU convolution: NHWC in/out,
 * wgt [(i*3+j)*16 + cin][16 cout], fused affine (folded BN) + LeakyReLU. */
int m3d_conv3x3_c16(const float *in, int in_cs, const float *wgt, const float *scale, const float *shift, float *out,
                    int out_cs, int N, int H, int W, m3d_stream_t stream);
/* The same layer as Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32 (csrc/wino44_conv.hip: 4x fewer MFMA FLOPs; fp32 rounding ~7x
 * that of direct summation, as for m3d_wino44_conv3x3_forward): `U` = G g G^T of the 16 x 16 filters packed [36 xi][64 lanes =
 * 16 * (cin / 4) + cout][cin % 4] (m3dssd_amd/engine.py:pack_wino44_c16); H % 4 == W % 4 == 0. */
int m3d_conv3x3_c16_wino(const float *in, int in_cs, const float *U, const float *scale, const float *shift, float *out,
                         int out_cs, int N, int H, int W, m3d_stream_t stream);
int m3d_maxpool2x2(const float *in, int in_cs, float *out, int out_cs, int N, int H, int W, int C,
                   m3d_stream_t stream);
/* out = ConvTranspose2d_depthwise(in, wgt[4][4][C], stride 2, pad 1) + skip ;  in is [N,H,W,C], out/skip [N,2H,2W,C] */
int m3d_upsample2x_add(const float *in, int in_cs, const float *wgt, const float *skip, int skip_cs,
                       float *out, int out_cs, int N, int H, int W, int C, m3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Alignment stages.
 * ------------------------------------------------------------------------------------------ */
/* cls_planar [B][num_classes*A][HW] (class-major).  Writes per pixel: top-1 fg anchor (lowest
 * index among equals), its fg probability, and fg_all [B][A][HW] (may be NULL). */
int m3d_anchor_select(const float *cls_planar, int B, int A, int num_classes, int HW, int *sel_idx,
                      float *sel_prob, float *fg_all, m3d_stream_t stream);
/* The same selection for 4 classes that also writes the detection stage's sort keys score_bits [B][A*HW] (the bits
 * m3d_score_keys_planar / m3d_bundle_outputs produce) while the logits are in registers. */
int m3d_anchor_select_keys(const float *cls_planar, int B, int A, int HW, int *sel_idx, float *sel_prob,
                           unsigned int *score_bits, m3d_stream_t stream);
/* Top-1 over anchors of a given fg-probability map [B][A][HW] (lowest index among equals). */
int m3d_fg_top1(const float *prob, int B, int A, int HW, int *idx, float *val, m3d_stream_t stream);
/* mode 0 = shape_align (table [A][2*kk] -> offmask [B*HW][3*kk], kk=9), mode 1 = center_align
 * (bbox_x/bbox_y planar: value of image b, anchor a, pixel p at b*box_img_stride + a*HW + p;
 *  anchor_wh [A][2] = (w/stride, h/stride), mean/std xy). */
int m3d_align_offsets(int mode, const int *sel_idx, const float *sel_prob, float thresh, const float *table,
                      const float *bbox_x, const float *bbox_y, const float *anchor_wh, float mean_x,
                      float std_x, float mean_y, float std_y, float *offmask, int om_cs, int B, int A, int HW,
                      int kk, long long box_img_stride, m3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ANAB (asymmetric non-local block).
 * ------------------------------------------------------------------------------------------ */
/* items: int32 [n_items][6] = (bin, h0, h1, w0, w1, slot); grid = n_items x B.
 * kv: NHWC view of the K|V channels (C = Ck + Cv), s: NHWC view of the sigmoid gates, bin_scale[bin]
 * = which gate weights that bin; partial [B][n_bins][max_slots][C]. */
int m3d_anab_pool_partial(const float *kv, int kv_cs, const float *s, int s_cs, const int *items, int n_items,
                          const int *bin_scale, int n_bins, float *partial, int max_slots, int B, int H, int W,
                          int C, m3d_stream_t stream);
/* Reduce slots, divide by bin area, scatter into GEMM operand layouts:
 *   khat [B][keys_pad][ck_pad]  (row = key j, col = channel)   -- "weights" of the logits GEMM
 *   vhatT[B][Cv][keys_pad]      (row = channel, col = key j)   -- "weights" of the P.V GEMM   */
int m3d_anab_pool_finish(const float *partial, const int *bin_slots, const float *bin_inv_area, int n_bins,
                         int max_slots, int Ck, int Cv, float *khat, int keys_pad, int ck_pad, float *vhatT,
                         int B, int frag, m3d_stream_t stream);
/* (frag bit 0 / bit 1 in either finish: khat / vhatT is written in MFMA-fragment order [R/32][K/8][2][32][4], the `wgt` layout of
 * m3d_conv_wave_forward, instead of row-major.)
 * Both steps in one call for psp sizes (1, 4, 8, 16) on maps with H % 16 == 0 and W % 16 == 0, where the adaptive windows of
 * the four scales nest: the features are read once instead of once per scale.  `scratch` holds
 * m3d_anab_pool_nested_scratch_bytes(B, Ck + Cv) bytes; `s` = the 4 gate channels (scale order), bins in scale-major order
 * (1 + 16 + 64 + 256 = 337 keys).  Same outputs as m3d_anab_pool_partial + m3d_anab_pool_finish up to fp32 summation order. */
long long m3d_anab_pool_nested_scratch_bytes(int B, int C);
int m3d_anab_pool_nested(const float *kv, int kv_cs, const float *s, int s_cs, int B, int H, int W, int Ck, int Cv,
                         float *scratch, float *khat, int keys_pad, int ck_pad, float *vhatT, int frag, m3d_stream_t stream);
/* The same with the K|V map stored as bf16 (bf16 engine: the K|V conv writes bf16 NHWC, the 4 gate channels come from a separate
 * fp32 conv); fp32 sums, same outputs up to the rounding of the features. */
int m3d_anab_pool_nested_bf16(const void *kv, int kv_cs, const float *s, int s_cs, int B, int H, int W, int Ck, int Cv,
                              float *scratch, float *khat, int keys_pad, int ck_pad, float *vhatT, int frag, m3d_stream_t stream);
/* ... that also writes bf16 twins of khat / vhatT (same element order; either may be NULL): the operands of m3d_anab_attend_bf16,
 * without the two m3d_f32_to_bf16 launches. */
int m3d_anab_pool_nested_bf16_ex(const void *kv, int kv_cs, const float *s, int s_cs, int B, int H, int W, int Ck, int Cv,
                                 float *scratch, float *khat, int keys_pad, int ck_pad, float *vhatT, int frag,
                                 void *khat16, void *vhat16, m3d_stream_t stream);
/* The attention of ANAB in ONE launch on fp32 MFMA (csrc/anab_attend.hip; attention.py:207-211 + the BatchNorm / activation behind the
 * block): out[p] = act(softmax_k(q[p] . khat[k]) @ vhat (+ res[p], scale, shift)) per image, replacing the logits GEMM, the row softmax
 * and the P.V GEMM (the logits never reach memory; one pass over the keys with a running maximum).  q [B*HW][q_cs] (first Ck
 * channels), khat [B][keys_pad][k_cs] and vhatT [B][Cv][keys_pad] row-major as m3d_anab_pool_nested / _finish write them with
 * frag = 0; Ck in {64, 128, 168}, Cv = 128, HW % 128 == 0, keys_pad % 32 == 0; res_mode as in m3d_conv_desc (0: + res behind the
 * affine, 1: before it); scale / shift / res may be NULL.
 * PRECONDITION (not checked): khat rows [keys, keys_pad) and vhatT columns [keys, keys_pad) hold finite values (zeros), as for
 * m3d_anab_attend_bf16. */
int m3d_anab_attend_f32(const float *q, int q_cs, const float *khat, int k_cs, const float *vhatT, int B, int HW, int Ck,
                        int keys, int keys_pad, int Cv, const float *res, int res_cs, int res_mode, const float *scale,
                        const float *shift, int act, float *out, int out_cs, m3d_stream_t stream);
/* In-place softmax over the first `valid` columns of each row; columns [valid, cs) are zeroed. */
int m3d_softmax_rows(float *x, int rows, int valid, int cs, m3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Outputs, decode, NMS.
 * ------------------------------------------------------------------------------------------ */
/* planar staging (cls [B][4A][HW], box [B][11][A][HW] in the order x,y,w,h,x3d,y3d,z3d,w3d,h3d,l3d,rY3d)
 * -> cls/prob [B][A*HW][4], bbox_2d [B][A*HW][4], bbox_3d [B][A*HW][7], and (optional) score_bits [B][A*HW]: the
 * max fg class probability of the row as monotone unsigned bits (a > b as floats <=> bits(a) > bits(b)), the sort
 * key of m3d_topk_decode. */
int m3d_bundle_outputs(const float *cls_planar, const float *box_planar, float *cls, float *prob, float *bbox_2d,
                       float *bbox_3d, unsigned int *score_bits, int B, int A, int HW, m3d_stream_t stream);
/* The sort keys alone (score_bits [B][A*HW]) from the planar class logits, with the softmax arithmetic of m3d_bundle_outputs (the
 * same bits): for callers that only run the detection stage (m3d_topk_decode_planar) and never read cls / prob / bbox_2d /
 * bbox_3d in full -- 5.5 MB per image instead of the 38 MB the bundling moves.  HW % 4 == 0, 16-byte aligned buffers. */
int m3d_score_keys_planar(const float *cls_planar, unsigned int *score_bits, int B, int A, int HW, m3d_stream_t stream);
/* The reference's `argsort()[::-1][:nms_topN_pre]` + decode (lib/rpn_util.py:1442-1544) in one launch, one workgroup
 * per image: radix select of the k rows with the largest (score, -row) -- descending score, ascending row among equal
 * scores: a total order, unlike the reference's unstable argsort -- bitonic sort of those k, decode of exactly those rows
 * -> aboxes [B][k][14] (x1,y1,x2,y2,score,cls,x3d,y3d,z3d,w3d,h3d,l3d,ry3d,anchor), score-descending; rows_out [B][k]
 * (optional) gets the selected row ids.  1 <= k <= min(R, 16384), R < 2^22; workspace: m3d_topk_decode_workspace_bytes. */
long long m3d_topk_decode_workspace_bytes(int B, int R);
int m3d_topk_decode(const unsigned int *score_bits, const float *prob, const float *bbox_2d, const float *bbox_3d,
                    const float *rois /*[R][5]*/, const float *anchors /*[A][9]*/, const float *means /*[11]*/,
                    const float *stds /*[11]*/, float *aboxes, int *rows_out, void *workspace, long long workspace_bytes,
                    int B, int R, int k, m3d_stream_t stream);
/* Same with the test-time scale factor of every image (scale [B] device floats, or NULL = 1): the 2-D corners and the projected
 * 3-D centre are divided by it right after the decode -- BEFORE the NMS that follows, as im_detect_3d does
 * (lib/rpn_util.py:1504-1506; the +1 area convention of the NMS is not scale invariant). */
int m3d_topk_decode_scaled(const unsigned int *score_bits, const float *prob, const float *bbox_2d, const float *bbox_3d,
                           const float *rois, const float *anchors, const float *means, const float *stds, const float *scale,
                           float *aboxes, int *rows_out, void *workspace, long long workspace_bytes, int B, int R, int k,
                           m3d_stream_t stream);
/* m3d_topk_decode_scaled reading the planar staging of the heads instead of the bundled tensors (cls_planar [B][4A][HW],
 * box_planar [B][11][A*HW]; score_bits from m3d_score_keys_planar or m3d_bundle_outputs): row = a*HW + p like
 * lib/rpn_util.py:892-901 flattens it; class probabilities of the k decoded rows are recomputed from their logits.  Output
 * identical (bit for bit) to m3d_bundle_outputs + m3d_topk_decode_scaled. */
int m3d_topk_decode_planar(const unsigned int *score_bits, const float *cls_planar, const float *box_planar, const float *rois,
                           const float *anchors, const float *means, const float *stds, const float *scale, float *aboxes,
                           int *rows_out, void *workspace, long long workspace_bytes, int B, int A, int HW, int k,
                           m3d_stream_t stream);
/* Decode `n_rows` selected rows per image -> aboxes [B][n_rows][14]
 * (x1,y1,x2,y2,score,cls,x3d,y3d,z3d,w3d,h3d,l3d,ry3d,anchor). */
int m3d_decode_rows(const long long *rows /*[B][n_rows] row ids*/, const float *prob, const float *bbox_2d,
                    const float *bbox_3d, const float *rois /*[R][5]*/, const float *anchors /*[A][9]*/,
                    const float *means /*[11]*/, const float *stds /*[11]*/, float *aboxes, int B, int R,
                    int n_rows, m3d_stream_t stream);

/* Greedy NMS on device.  boxes_dev [B][n][box_stride>=4] sorted by descending score; mask_ws needs
 * B*n*ceil(n/64) uint64.  keep_dev [B][n] int32 (kept positions, ascending), num_keep_dev [B].
 * n <= 16384 (the reference's _nms is unbounded -- the host-pointer twin `_nms` below is too; the path calls this with
 * nms_topN_pre = 3000 rows): up to 4096 rows one wave holds the whole "removed" bit set of an image in registers (64 lanes x 64
 * bits), up to 16384 it lives in LDS; larger n returns M3D_E_ARG. */
long long m3d_nms_workspace_bytes(int B, int n);
int m3d_nms_sorted_dev(const float *boxes_dev, int B, int n, int box_stride, float thresh, void *mask_ws,
                       int *keep_dev, int *num_keep_dev, m3d_stream_t stream);
/* Kept rows of the NMS -> fixed-size blocks (lib/rpn_util.py:1547-1555 `aboxes[keep][:nms_topN_post]`): block
 * [B][post + 1][14], rows [0, min(num_keep, post)) = aboxes[keep[j]], the rest zero, row `post` = (count, 0, ...) -- the
 * message of the multi-GPU all-gather; counts [B] (optional) gets min(num_keep, post). */
int m3d_select_post(const float *aboxes /*[B][n][14]*/, const int *keep /*[B][n]*/, const int *num_keep /*[B]*/, int B, int n,
                    int post, float *block, int *counts, m3d_stream_t stream);
/* Exact twin of the reference's _nms (lib/nms/gpu_nms.hpp): host pointers, synchronous. */
void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
          float nms_overlap_thresh, int device_id);

/* ------------------------------------------------------------------------------------------
 * Post-NMS 3-D refinement (lib/rpn_util.py:1801-1847 per-box loop of test_kitti_3d): convertAlpha2Rot,
 * hill_climb on the yaw (step_r_init, halving down to r_lim; the depth step is 0 as in the reference call) scored by
 * test_projection, convertRot2Alpha, camera-space centre.  One thread per row, float64.
 *   aboxes [B][K][14] fp32 rows (x1 y1 x2 y2 score cls x3d y3d z3d w3d h3d l3d alpha anchor), counts [B],
 *   p2 / p2_inv [B][16] row-major 4x4 doubles (device), out [B][K][16] doubles:
 *   valid, cls, alpha, x1, y1, x2, y2, h3d, w3d, l3d, x3d, y3d (bottom centre), z3d, ry3d, score, 0
 *   (valid = 0 and zeros for rows past counts[b] or with score < score_thresh) -- the fields of the KITTI result line
 *   '{cls} -1 -1 {alpha} {x1} {y1} {x2} {y2} {h} {w} {l} {x} {y} {z} {ry} {score}' (lib/rpn_util.py:1848-1849).
 * ------------------------------------------------------------------------------------------ */
int m3d_refine_3d(const float *aboxes, const int *counts, int B, int K, const double *p2, const double *p2_inv,
                  double score_thresh, int hill_climbing, double step_r_init, double r_lim, double *out, m3d_stream_t stream);
/* The same with the two per-image steps im_detect_3d / test_kitti_3d apply between NMS and the loop folded in, so that the call can
 * sit in a captured graph right behind m3d_select_post: scale [B] fp32 (device, or NULL) -- x1 y1 x2 y2 x3d y3d are divided by
 * scale[b] in float32 first (`coords_2d[:, 0:4] /= scale_factor`, lib/rpn_util.py:1506-1507 -- the reference does it BEFORE the NMS: pass the factors to m3d_topk_decode_scaled and NULL here to reproduce that); clip_wh [B][2] fp32 = (imW, imH)
 * (device, or NULL; an entry <= 0 disables it) -- the 2-D box is then clipped to [0, imW - 1] x [0, imH - 1] (:1533-1538).  The
 * rows themselves are not modified.  K may include the count row of a m3d_select_post block (it lies past counts[b]). */
int m3d_refine_3d_ex(const float *aboxes, const int *counts, int B, int K, const double *p2, const double *p2_inv,
                     const float *scale, const float *clip_wh, double score_thresh, int hill_climbing, double step_r_init,
                     double r_lim, double *out, m3d_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * KITTI AP evaluator natives (lib/eval/eval.py, lib/eval/rotate_iou.py of the reference, which compiles them
 * with numba / numba.cuda).
 * m3d_rotate_iou_eval: rotate_iou_gpu_eval (rotate_iou.py:264-326).  boxes [N][5], qboxes [K][5] = (cx, cy, dx, dy, angle)
 *   float32 DEVICE pointers; iou [N][K] float32 device: entry [n][k] = devRotateIoUEval(qboxes[k], boxes[n], criterion) with
 *   criterion -1 IoU, 0 / qbox area, 1 / box area, 2 the intersection area itself.
 * The remaining entry points are HOST functions on float64 / int64 numpy-style arrays (synchronous, no device work):
 * m3d_eval_image_box_overlap: image_box_overlap (eval.py:84-113), boxes [N][4], q [K][4] -> out [N][K].
 * m3d_eval_d3_overlap: d3_box_overlap_kernel (eval.py:116-141) on boxes [N][7] / qboxes [K][7] = (x, y, z, l, h, w, ry);
 *   rinc [N][K] holds the BEV intersection areas on entry, the 3-D overlaps on return.
 * m3d_eval_statistics: compute_statistics_jit (eval.py:152-272) for one image.  overlaps [det][ov_stride] (detection j vs
 *   ground truth i at [j*ov_stride + i]), gt_datas [gt][5] = bbox, alpha; dt_datas [det][6] = bbox, alpha, score;
 *   ignored_* int64 (0 evaluate, 1 ignore, -1 other class); stats[4] = tp, fp, fn, similarity; thresholds_out (optional,
 *   room for gt_size values) gets the scores of the true positives, *n_thresholds their number.
 * m3d_eval_fused_statistics: fused_compute_statistics (eval.py:287-333) over a part of n_images images whose overlaps form
 *   one [sum det][sum gt] matrix; pr [n_thresholds][4] is accumulated into (tp, fp, fn, similarity).
 * ------------------------------------------------------------------------------------------ */
int m3d_rotate_iou_eval(const float *boxes_dev, int N, const float *qboxes_dev, int K, int criterion, float *iou_dev,
                        m3d_stream_t stream);
int m3d_eval_image_box_overlap(const double *boxes, int N, const double *q, int K, int criterion, double *out);
int m3d_eval_d3_overlap(const double *boxes, int N, const double *qboxes, int K, double *rinc, int criterion);
int m3d_eval_statistics(const double *overlaps, long long ov_stride, const double *gt_datas, int gt_size, const double *dt_datas,
                        int det_size, const long long *ignored_gt, const long long *ignored_det, const double *dc_bboxes, int n_dc,
                        int metric, double min_overlap, double thresh, int compute_fp, int compute_aos, double *stats,
                        double *thresholds_out, int *n_thresholds);
int m3d_eval_fused_statistics(const double *overlaps, long long ov_stride, double *pr, const long long *gt_nums,
                              const long long *dt_nums, const long long *dc_nums, int n_images, const double *gt_datas,
                              const double *dt_datas, const double *dontcares, const long long *ignored_gts,
                              const long long *ignored_dets, int metric, double min_overlap, const double *thresholds,
                              int n_thresholds, int compute_aos);

/* ------------------------------------------------------------------------------------------
 * Instrumentation: HIP-event timing of a launch sequence on a stream (used by bench.py).
 * ------------------------------------------------------------------------------------------ */
/* One wave samples the shader-cycle counter and the 100 MHz wall clock at both ends of a `seconds`-long window (<= 1 s):
 * out4_dev = {cycles0, realtime0, cycles1, realtime1}; (cycles1 - cycles0) / (realtime1 - realtime0) x 100 MHz = the shader clock the
 * chip held while whatever ran next to it on other streams (bench.py: sclk_under_step_ghz). */
int m3d_clock_probe(long long *out4_dev, double seconds, m3d_stream_t stream);
int m3d_event_create(void **ev);
int m3d_event_record(void *ev, m3d_stream_t stream);
int m3d_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */
int m3d_event_destroy(void *ev);

#ifdef __cplusplus
}
#endif
#endif /* M3DSSD_HIP_H */
