/* yv_hip.h - C ABI of the MI355X (gfx950) hot-path library `libyvhip.so`.
 *
 * Drop-in boundary of the detect -> NMS/inflate/crop -> ViT-classify path of
 * Voyager0587/yolov8-vit.  The reference has NO native/FFI interface (SURVEY.md
 * section 8(b)): its boundary is a Python call surface.  Each entry point below
 * therefore cites the reference *Python site* whose arithmetic it replaces; the
 * Python host side (yolov8-vit_amd/) binds these symbols with ctypes and mirrors
 * the reference's names.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on
 *    it (no allocation, no synchronisation, graph-capturable);
 *  - the caller owns all buffers, including `ws` workspaces whose minimum size
 *    is returned by the matching `*_ws_bytes` function;
 *  - return value: 0 = YV_OK, negative = error (yv_error_string), never throws;
 *  - bf16 tensors are raw uint16_t storage; "NHWC view" = base pointer + pixel
 *    stride `ld` in elements (lets producers write straight into concat
 *    buffers and consumers read channel slices: no concat/split kernels).
 */
#ifndef YV_HIP_H
#define YV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YV_OK 0
#define YV_ERR_ARG (-1)        /* bad size / null pointer / unsupported shape */
#define YV_ERR_LIMIT (-2)      /* exceeds a documented kernel limit           */
#define YV_ERR_WORKSPACE (-3)  /* workspace too small                          */
#define YV_ERR_LAUNCH (-4)     /* hipLaunch failed (hipGetLastError != 0)     */

int yv_version(void);
const char* yv_error_string(int code);
/* 1 if the current HIP device is gfx950, 0 otherwise, <0 on HIP error. */
int yv_device_is_gfx950(void);
/* Tuning knobs (process-wide, not part of the reference surface): "linear_variant" (0 register-staged
 * 128x128, 1 LDS-DMA 128x128, 2 256x128, 3 256x256, 4 128x256), "linear_group_m" (M tiles per L2 group). */
int yv_set_option(const char* key, int value);
/* Current value of a knob.  "linear_p8_cus" (workgroups of the persistent classifier GEMMs; 0 = every CU) is PER THREAD: it
 * applies to the launches the calling thread makes (a pipelined runner lowers it around its own classifier submissions so that
 * kernels of its other streams find free CUs, and restores it; other threads and later callers are not affected). */
int yv_get_option(const char* key, int* value);

/* MXFP8 linears (BASELINE.json configs[4], FP8 classifier GEMMs; OCP e4m3 bytes + one E8M0 scale per 32 consecutive K
 * elements of a row, consumed by the block-scaled gfx950 MFMA).
 * Scale arrays are K-step-major: (K/128, rows_pad, 4) bytes - the four block scales of one 128-deep K step of a row form
 * one dword, a tile's 128 dwords are contiguous (one LDS-DMA half); rows_pad = row count rounded up to 128.
 * yv_quant_mxfp8: x (rows, K) bf16 -> q (rows, K) bytes (row stride ldq) + scales; scale exponent
 *   e = ceil(log2(amax/448)) per block, q = RNE_e4m3(x * 2^-e); K a multiple of 128.
 * yv_linear_mxfp8: out[M,N] = (Aq*2^sa)[M,K] . (Wq*2^sw)[N,K]^T with the epilogues of yv_linear (bias, GELU, f32 residual
 *   read-modify-write, f32 output). */
/* yv_linear_mxfp8 whose OUTPUT is again an MXFP8 operand (bias / GELU applied, rounded to bf16, then quantised in the
 * epilogue): the fc1 -> fc2 hand-off of the MLP without a bf16 round trip through HBM.  N a multiple of 128. */
int yv_linear_mxfp8_q(const void* Aq, long long lda, const void* Ascale, long long a_rows_pad, const void* Wq, const void* Wscale,
                      long long w_rows_pad, const float* bias, int M, int N, int K, int flags, const int32_t* m_dev, int m_mul,
                      void* out_q, long long ldq, void* out_scales, long long out_rows_pad, void* stream);

/* yv_attention with the output written directly in the MXFP8 operand format of the proj GEMM (same numbers as
 * yv_attention followed by yv_quant_mxfp8); H even. */
int yv_attention_mxfp8(const void* qkv, int R, int N, int H, float scale, void* out_q, long long ldq, void* out_scales,
                       long long rows_pad, const int32_t* r_dev, void* stream);

/* yv_layernorm with the output written directly in the MXFP8 operand format (same numbers as yv_layernorm followed by
 * yv_quant_mxfp8: the bf16 rounding is kept); D a multiple of 128. */
int yv_layernorm_mxfp8(const float* x, size_t ldx, const float* gamma, const float* beta, int rows, int D, float eps, void* q,
                       size_t ldq, void* scales, long long rows_pad, const int32_t* count_dev, int rows_per_count,
                       void* stream);

/* Diagnostic: ONE v_mfma_scale_f32_16x16x128_f8f6f4 on caller-provided register images: a, b (64 lanes x 32 bytes),
 * sa, sb (64 x int32 scale registers), opsel 0..3 for both; d (64 lanes x 4 f32). */
int yv_mx_probe(const void* a, const void* b, const void* sa, const void* sb, int opsel, void* d, void* stream);

int yv_quant_mxfp8(const void* x, long long ldx, long long rows, int K, void* q, long long ldq, void* scales,
                   long long rows_pad, void* stream);
int yv_linear_mxfp8(const void* Aq, long long lda, const void* Ascale, long long a_rows_pad, const void* Wq, const void* Wscale,
                    long long w_rows_pad, const float* bias, int M, int N, int K, void* out, int ldo, int flags,
                    const int32_t* m_dev, int m_mul, void* stream);

/* Measurement hook (bench.py): the NEXT LDS-DMA GEMM launch issued by the calling thread (yv_linear / yv_linear_ex /
 * yv_linear_nn) records its start / stop timestamps into these hipEvent_t handles through hipExtLaunchKernel, i.e. from
 * the kernel's own dispatch packet.  Either may be null; the setting is consumed by that launch. */
int yv_set_launch_timing(void* start_event, void* stop_event);
/* Registers (ws != NULL) or removes a caller-owned f32 scratch buffer for split-K partial sums used by launches
 * on `stream`.  One buffer per stream: launches of a stream are ordered, different streams must not share one. */
int yv_set_workspace(void* stream, void* ws, size_t bytes);

/* ------------------------------------------------------------------ boxes */

/* custom_nms(boxes, scores, iou_threshold=0.45)  -- README.md:62-84 (== tech.md:72-94).
 * Class-agnostic greedy NMS, strict `<`, returns ORIGINAL indices in
 * (score desc, index asc) order.  Batched: set s uses rows [s*n_max, s*n_max+count[s]).
 * boxes (S,n_max,4) f32 xyxy; scores (S,n_max) f32; counts (S) i32 or NULL (= n_max);
 * keep (S,n_max) i32 (tail filled with -1); num_keep (S) i32.  n_max <= 16384. */
size_t yv_custom_nms_ws_bytes(int n_sets, int n_max);
int yv_custom_nms(const float* boxes, const float* scores, const int32_t* counts, int n_sets, int n_max,
                  float iou_threshold, int32_t* keep, int32_t* num_keep, void* ws, size_t ws_bytes, void* stream);

/* EfficientNMS_TRT output contract -- tech.md:41-47, test.ipynb:20-24 (KAT-2).
 * boxes (B,A,4) f32 xyxy; scores (B,A,nc) f32 (already sigmoid).  Outputs:
 * num_dets (B,1) i32, out_boxes (B,max_out,4) f32, out_scores (B,max_out) f32,
 * out_labels (B,max_out) i32, zero padded, sorted by score.  pre_topk <= 4096. */
int yv_efficient_nms(const float* boxes, const float* scores, int B, int A, int nc, float score_threshold,
                     float iou_threshold, int max_out, int pre_topk, int32_t* num_dets, float* out_boxes,
                     float* out_scores, int32_t* out_labels, void* stream);

/* Same contract (tech.md:41-47, test.ipynb:20-24), multi-workgroup form: the candidate filter streams the scores at
 * HBM rate over (A*nc / 4096, B) workgroups; one workgroup per image then resolves a prefix of the ranking (all the
 * sequential scan looks at before the max_out-th kept box: radix cut, rank sort, 64-wide tiles), and only images it
 * cannot finish go through the exact top-pre_topk select, per-class greedy NMS and a merge - bit-identical to
 * yv_efficient_nms.  `ws`: caller-owned scratch of yv_efficient_nms_ws_bytes() bytes, 256-byte aligned, private to the
 * call while it is in flight.  This is the form the pipeline uses; the single-kernel form above needs no scratch. */
size_t yv_efficient_nms_ws_bytes(int B, int A, int nc, int max_out, int pre_topk);
int yv_efficient_nms_ws(const float* boxes, const float* scores, int B, int A, int nc, float score_threshold,
                        float iou_threshold, int max_out, int pre_topk, int32_t* num_dets, float* out_boxes,
                        float* out_scores, int32_t* out_labels, void* ws, size_t ws_bytes, void* stream);

/* det_postprocess + coordinate restore + score filter + int cast
 * (YOLOTensorRT_yolodet_py_解读.md:82-99), then custom_nms dedupe (README.md:41,62-84),
 * then crop_image's integer inflate/clamp (utils/trainClass.py:70-93, eval branch).
 * Per image b: ratio[b], dwdh[b] (2 f32), img_wh[b] (2 i32: original width,height).
 * coord_mode 0 = trunc toward zero (Python int()), 1 = round-half-even then int.
 * Outputs per image (slots = max_out of the NMS stage, e.g. 100):
 *   det_count (B) i32            number of reported detections (score order)
 *   det_box (B,slots,4) i32      xmin,ymin,xmax,ymax in original-image pixels
 *   det_score (B,slots) f32, det_label (B,slots) i32
 *   crop_rect (B,slots,4) i32    inflated+clamped rect (x0,y0,x1,y1), right/bottom exclusive
 *   crop_ok (B,slots) i32        0 when the rect is degenerate (PIL would raise)
 * max_crops: cap of detections kept per image (<=0: no cap).                 */
int yv_postprocess_dets(const int32_t* num_dets, const float* bboxes, const float* scores, const int32_t* labels,
                        int B, int slots, const float* ratio, const float* dwdh, const int32_t* img_wh,
                        float conf_threshold, float dedupe_iou, int coord_mode, int max_crops,
                        int32_t* det_count, int32_t* det_box, float* det_score, int32_t* det_label,
                        int32_t* crop_rect, int32_t* crop_ok, void* stream);

/* Device-side compaction of the per-image crop lists into one batch:
 * crop_list (cap,6) i32 rows {image, x0, y0, x1, y1, slot}; crop_total (1) i32.
 * Order: image ascending, then detection (score) order.  Rows >= total are zeroed. */
int yv_compact_crops(const int32_t* det_count, const int32_t* crop_rect, const int32_t* crop_ok, int B, int slots,
                     int cap, int32_t* crop_list, int32_t* crop_total, void* stream);
/* The same, for a classifier that runs the crop list as `parts` (<= 64) equal slices of ceil(cap / parts) entries on concurrent
 * streams (reference site: the per-image loop of YOLOTensorRT inferdet.main, YOLOTensorRT_yolodet_py_解读.md:57-116 - batch
 * assembly has no reference counterpart): crop_total (1 + parts) i32 = {total, crops in slice 0, crops in slice 1, ...}, so
 * that no host-side or torch arithmetic on the device-resident count is needed between detect and classify. */
int yv_compact_crops_split(const int32_t* det_count, const int32_t* crop_rect, const int32_t* crop_ok, int B, int slots,
                           int cap, int parts, int32_t* crop_list, int32_t* crop_total, void* stream);

/* crop + A.Resize(224,224,INTER_NEAREST) + A.Normalize(.5,.5) + HWC->CHW
 * (utils/trainClass.py:92,218-221,265-266; app.py:39-42).
 * images: (B,H,W,3) u8 RGB with per-image byte stride img_stride; crop_list as above;
 * crop_total: device i32 (NULL = all `cap` rows valid).
 * layout 0: out = (cap,3,S,S) f32 CHW     layout 1: out = (cap,3,S,S) bf16 CHW
 * layout 2: out = (cap*(S/P)^2, 3*P*P) bf16 patch-major rows (col = c*P*P + py*P + px),
 *           the A operand of the patch-embed GEMM.  S = out_size (224), P = patch. */
int yv_crop_resize_norm(const uint8_t* images, int B, int H, int W, size_t img_stride, const int32_t* crop_list,
                        const int32_t* crop_total, int cap, int out_size, int patch, int layout, void* out,
                        void* stream);
/* Diagnostics: forced number of 16-row groups a block of the crop kernel walks (0 = chosen from the crop count). */
int yv_crop_debug(int groups_per_block);

/* letterbox (YOLOTensorRT_yolodet_py_解读.md:67-69): src (B,Hc,Wc,3) u8 canvas holding image b in its top-left
 * w x h corner; geom (B,6) i32 rows {w, h, nw, nh, left, top} (host-computed);
 * out (B,S,S,3) u8: bilinear resample into the window, 114 elsewhere. */
int yv_letterbox(const uint8_t* src, int B, int Hc, int Wc, const int32_t* geom, int S, uint8_t* out, void* stream);

/* Training augmentation of classifier crops fused with the patch-embed operand builder: replaces the stochastic part
 * of data_transforms['train'] (utils/trainClass.py:199-216: HorizontalFlip, RandomCrop+PadIfNeeded, ShiftScaleRotate,
 * ChannelShuffle, OneOf[GridDistortion|ElasticTransform], CoarseDropout) that follows Resize + Normalize.
 * x (B,3,S,S) f32 normalised crops; one host-drawn record per sample:
 *   geo (B, 6 + 2S) f32: inverse affine {a0..a5} (u = a0*cx + a1*cy + a2, v = a3*cx + a4*cy + a5), then the per-axis
 *                        distortion tables lutx[S], luty[S] (identity: lut[i] = i);
 *   idx (B, 36 + 2S) i32: source channel of output channel 0..2, hole count (0..8), 8 holes {x1,y1,x2,y2} (exclusive
 *                        right/bottom), then integer column / row tables mapx[S], mapy[S] (flip, crop offset + reflected pad).
 * Sampling is bilinear with BORDER_REFLECT_101 in the flipped / crop-padded frame; holes are filled with 0.
 * out (B*(S/P)^2, 3*P*P) bf16 patch-major rows (same layout as yv_crop_resize_norm layout 2). */
int yv_augment_patchify(const float* x, int B, int S, int P, const float* geo, const int32_t* idx, void* out, void* stream);

/* Detector training augmentation, what `model.train()` (utils/trainYolo.py:28) applies by default: Mosaic(4) ->
 * RandomPerspective(scale, translate) -> HSV gains -> horizontal flip in one gather pass per output image.
 * tiles (n_tiles,S,S,3) u8: sources resized to long side S in the top-left corner of their slot (yv_letterbox, fill 114).
 *   rec_f (B,6) f32: inverse affine, output pixel -> mosaic-canvas coordinate;
 *   rec_i (B,34) i32: {tiles used (1..4), flip}, then per tile {id, x1a, y1a, x2a, y2a, x1b, y1b, 0}: the canvas rectangle
 *                     [x1a,x2a) x [y1a,y2a) shows tile `id` starting at its pixel (x1b, y1b);
 *   lut (B,3,256) u8: hue / saturation / value tables applied in 8-bit HSV (H in [0,180)).
 * Bilinear; canvas pixels outside every rectangle are 114.  out (B,S,S,3) u8 NHWC (the detector's input layout). */
int yv_mosaic_augment(const uint8_t* tiles, int n_tiles, int B, int S, const float* rec_f, const int32_t* rec_i,
                      const uint8_t* lut, uint8_t* out, void* stream);

/* DFL decode + anchors + sigmoid (docs/YOLO_TensorRT_Technical.md:14-30,72-77).
 * Per scale s (3 scales, strides 8/16/32): box logits (B,Hs,Ws,64) f32 and class
 * logits (B,Hs,Ws,cls_ld) f32, NHWC.  Outputs boxes (B,A,4) f32 xyxy input pixels,
 * scores (B,A,nc) f32; anchor order = scale, then y*W+x. */
int yv_detect_decode(const float* box0, const float* box1, const float* box2, const float* cls0, const float* cls1,
                     const float* cls2, int cls_ld, int B, int size, int nc, float* boxes, float* scores,
                     void* stream);

/* Fused Detect tail: the last 1 x 1 convolutions of both branches (ultralytics Detect cv2.i.2: 64 -> 4 x reg_max box logits,
 * cv3.i.2: c3 -> nc class logits; TensorRT builder docs/YOLO_TensorRT_Technical.md:160-212) + yv_detect_decode, three scales in
 * one launch; bit-identical to the unfused sequence.  feat_s (B*Hs*Ws, ld) bf16: box-branch features in channels 0..63, class
 * branch features in 64..64+c3-1.  w2[s] (64,64) bf16, b2[s] (64) f32, w3[s] (16,c3) bf16 with rows >= nc zero, b3[s] (16) f32
 * (host arrays of three device pointers).  c3 in {64,128,192}, nc <= 16; otherwise YV_ERR_LIMIT (use the unfused path). */
int yv_detect_tail(const void* feat0, const void* feat1, const void* feat2, int ld, int c3, const void* const* w2,
                   const float* const* b2, const void* const* w3, const float* const* b3, int B, int size, int nc,
                   float* boxes, float* scores, void* stream);

/* ------------------------------------------------------------ dense math */

/* Operand view for the implicit-GEMM kernel: NHWC bf16 tensor slice.
 * up = 1: the tensor is read through a nearest 2x upsample (K4). */
typedef struct yv_view {
    const void* ptr; /* bf16 */
    int ld;          /* elements between consecutive pixels */
    int c;           /* channels read from this view        */
    int up;          /* 0 | 1                               */
} yv_view;

/* epilogue flags */
#define YV_EPI_BIAS 1
#define YV_EPI_SILU 2
#define YV_EPI_GELU 4       /* exact erf GELU */
#define YV_EPI_RES_F32 8    /* out_f32 += (read-modify-write residual stream) */
#define YV_EPI_RES_BF16 16  /* + bf16 residual view (Bottleneck shortcut)     */
#define YV_EPI_OUT_F32 32   /* store f32 instead of bf16                      */
#define YV_EPI_POSEMB 64    /* patch-embed row remap + pos_embed add          */
#define YV_EPI_SAVE_PRE 128 /* also store the pre-activation (bf16) to `aux`   */
#define YV_EPI_GELU_BWD 256 /* out = acc * gelu'(aux)  (MLP backward)         */
#define YV_EPI_OUT_MXFP8 512 /* internal to yv_linear_mxfp8_q: e4m3 + E8M0 output */

/* Conv2d(k in {1,3}, stride in {1,2}, pad k/2) + folded-BN bias + SiLU as an
 * implicit GEMM on MFMA (ultralytics Conv/C2f/SPPF/Detect convs; structure per
 * docs/YOLO_TensorRT_Technical.md:160-212, fusion per test.ipynb:25,1285).
 * in0 (+ optional in1 = channel concat, 1x1 only) are (B,Hin,Win,*) views;
 * weight (Cout, k*k*Cin) bf16 with K order (ky,kx,cin); bias (Cout) f32;
 * out = (B,Hout,Wout,*) view with pixel stride out_ld (bf16, or f32 with YV_EPI_OUT_F32);
 * res = optional bf16 residual view (same pixel grid as out).
 * Limits: the fused 2x upsample (yv_view.up) is a property of 1x1 inputs (YV_ERR_ARG with ksize 3); the kernel
 * addresses each source and the weights with 32-bit byte offsets: a batch whose source passes 2 GB is taken in
 * sub-batches by this entry, one IMAGE (Hin*Win*ld*2) and the weights (Cout*k*k*Cin*2) must stay below 2 GB
 * (YV_ERR_LIMIT). */
int yv_conv2d(const yv_view* in0, const yv_view* in1, int B, int Hout, int Wout, int ksize, int stride,
              const void* weight, const float* bias, int Cout, void* out, int out_ld, const void* res, int res_ld,
              int flags, void* stream);

/* One launch for a whole C2f block of the detector backbone (ultralytics C2f with shortcut, as in layers model.2 / model.4 of the
 * YOLOv8 models the reference loads - utils/utils.py:126, test.ipynb): cv1 (1x1, 2c -> 2c), n bottlenecks of two 3x3 layers
 * c -> c with the residual add, cv2 (1x1, (2+n)c -> 2c), SiLU after every layer (BN folded).  x (B,H,W,>=2c) bf16 with pixel
 * stride ldx, out (B,H,W,>=2c) bf16 with pixel stride ldo; weights (Cout, k*k*Cin) bf16 with K order (ky,kx,cin) and f32
 * biases exactly as yv_conv2d takes them; w_m / b_m: 2n entries {m0.cv1, m0.cv2, m1.cv1, m1.cv2}.  c in {16, 32}, n in {1, 2}
 * (anything else: YV_ERR_ARG - run the block layer by layer).  Same arithmetic per output as the layer-by-layer path
 * (K order, f32 accumulation, bias -> SiLU -> + residual -> one bf16 rounding). */
int yv_c2f_fused(const void* x, long long ldx, int B, int H, int W, int c, int n, const void* w_cv1, const float* b_cv1,
                 const void* const* w_m, const float* const* b_m, const void* w_cv2, const float* b_cv2, void* out,
                 long long ldo, void* stream);
/* Diagnostics: buf = 64 x 8 uint64 on the device receives cycle stamps (phase boundaries of wave 0 of the first 64 workgroups)
 * of every following yv_c2f_fused launch; NULL switches them off. */
int yv_c2f_debug(void* buf);

/* Same, with a caller-owned f32 workspace: deep small-resolution layers then run split-K (K range sliced over
 * several workgroups per tile + a reduce/epilogue pass).  ws_bytes >= 8 * M * Cout * 4 enables every split. */
int yv_conv2d_ws(const yv_view* in0, const yv_view* in1, int B, int Hout, int Wout, int ksize, int stride,
                 const void* weight, const float* bias, int Cout, void* out, int out_ld, const void* res, int res_ld,
                 int flags, void* ws, size_t ws_bytes, void* stream);

/* Linear: out[M,N] = A[M,K] @ W[N,K]^T (+bias)(+GELU)(+residual) on MFMA
 * (timm Attention.qkv/proj, Mlp.fc1/fc2, head; README.md:21-35).
 * A (M,K) bf16 row stride lda; W (N,K) bf16; bias (N) f32.
 * m_dev: optional device i32; rows at run time = min(M, m_dev[0]*m_mul) (tiles beyond exit):
 *        the crop count of a batch is only known on the device (no host sync).
 * YV_EPI_POSEMB: row m -> out row (m/tok)*(tok+1)+1+(m%tok), adds pos[(1+m%tok)*N + n]. */
int yv_linear(const void* A, int lda, const void* W, const float* bias, int M, int N, int K, void* out, int ldo,
              const float* pos, int tok, int flags, const int32_t* m_dev, int m_mul, void* stream);

/* Training form of yv_linear: res_f32 = separate f32 residual source (out = res_f32 + acc + bias, so the
 * residual stream of every layer can be kept for backward); aux/ldaux = bf16 side buffer for
 * YV_EPI_SAVE_PRE / YV_EPI_GELU_BWD.  Requires K % 64 == 0, N > 64 and 16-byte aligned rows. */
int yv_linear_ex(const void* A, int lda, const void* W, const float* bias, int M, int N, int K, void* out, int ldo,
                 int flags, const float* res_f32, void* aux, int ldaux, void* stream);

/* LayerNorm over the last dim (timm blocks.*.norm1/2, norm; eps 1e-6; README.md:21-29).
 * x (rows, D) f32 with row stride ldx -> y (rows, D) bf16 row stride ldy.
 * count_dev: optional device i32; rows at run time = min(rows, count_dev[0]*rows_per_count). */
int yv_layernorm(const float* x, size_t ldx, const float* gamma, const float* beta, int rows, int D, float eps,
                 void* y, size_t ldy, const int32_t* count_dev, int rows_per_count, void* stream);

/* Fused attention forward, non-causal: softmax(Q K^T * scale) V (timm Attention; README.md:21-23).
 * qkv (R*N, 3*H*64) bf16 as produced by the qkv Linear ([q|k|v], head-major);
 * out (R*N, H*64) bf16.  head dim 64; N <= 256 is one K/V tile (ViT-x/16: 197), longer sequences
 * (ViT-B/8: 785) are tiled with an online softmax. */
int yv_attention(const void* qkv, int R, int N, int H, float scale, void* out, const int32_t* r_dev, void* stream);

/* Diagnostic builds of the attention kernel (0 = normal; 1 no K/V loads, 2 no compute, 3 no V^T writes). */
int yv_attention_debug(int ablate);

/* cls rows of the token stream: x[r*(tok+1), :] = cls + pos[0]  (timm cls_token + pos_embed) */
int yv_cls_rows(const float* cls, const float* pos, int R, int tok, int D, float* x, void* stream);

/* Network_Wrapper.fc on backbone logits + argmax (utils/utils.py:64-72, utils/trainClass.py:112):
 * w1t = fc.1.weight TRANSPOSED, (1000,128) f32 (coalesced rows); w2 = fc.3.weight (nc,128).
 * feats (R, ldf) f32 (first 1000 used) -> logits (R,nc) f32 [accumulated when accumulate!=0,
 * scaled by `scale`: mean-of-logits ensemble over model_list], labels (R) i32 (argmax, first max). */
int yv_wrapper_head(const float* feats, int ldf, const float* w1t, const float* b1, const float* w2, const float* b2,
                    int R, int nc, float scale, int accumulate, float* logits, int32_t* labels,
                    const int32_t* r_dev, void* stream);

/* SPPF: three chained 5x5/s1/p2 max-pools (== windows 5/9/13) in one pass (model.9).
 * buf: (B,H,W,ld) bf16; reads channels [0,c), writes [c,2c), [2c,3c), [3c,4c). */
int yv_sppf_pool(void* buf, int B, int H, int W, int ld, int c, void* stream);

/* Stem: blob (u8 RGB /255, 解读.md:70-74) + Conv 3x3 s2 + bias + SiLU (model.0).
 * images (B,H,W,3) u8; weight (27,Cout) f32, row = (ky*3+kx)*3+c; Cout in {16,32,48};
 * out (B,H/2,W/2,Cout) bf16, ld = out_ld. */
int yv_stem_conv(const uint8_t* images, int B, int H, int W, const float* weight, const float* bias, int Cout,
                 void* out, int out_ld, void* stream);

/* ------------------------------------------------------------- training */

/* Forward attention that also returns the per-(crop, head, query) log2-sum-exp for the backward pass. */
int yv_attention_train(const void* qkv, int R, int N, int H, float scale, void* out, float* lse, void* stream);

/* Attention backward (timm Attention, README.md:21-23): qkv/out/dout as in the forward, lse from
 * yv_attention_train; writes dqkv (R*N, 3*H*64) bf16; delta_ws: R*H*N floats of scratch. */
int yv_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, int R, int N, int H,
                     float scale, void* dqkv, float* delta_ws, void* stream);

/* out[M,N] (bf16) = A[M,K] . Wkn[K,N] with the weight in reduction-major layout (row stride ldw): the data
 * gradient dX = dY . W reads the (N_w, K_w) weight as it is stored, through transposing LDS reads (no W^T copy).
 * flags: YV_EPI_BIAS, YV_EPI_GELU_BWD (aux = saved pre-activation). */
int yv_linear_nn(const void* A, int lda, const void* Wkn, int ldw, const float* bias, int M, int N, int K, void* out,
                 int ldo, int flags, void* aux, int ldaux, void* stream);

/* Weight gradient dW (N,K) f32 = dY^T . X with dY (T,N) and X (T,K) bf16 token-major (T = tokens, a multiple of 64
 * whose tail rows are ZERO): the fragments are columns of the LDS tiles, read with ds_read_b64_tr_b16; no transposed
 * copies.  Split over T when a workspace is registered for the stream (deterministic). */
int yv_wgrad(const void* dY, int ldy, const void* X, int ldx, int T, int N, int K, float* dW, int ldw, void* stream);

/* Weight gradient of a 3x3 / stride 1 / pad 1 convolution with no im2col buffer: dW (N, 9*Cin) f32, column
 * ((dy+1)*3 + dx+1)*Cin + ci.  Operands over the zero-padded pixel grid (B, H+2, W+2), T = that pixel count rounded up to
 * a multiple of 64: dYp (T, N) bf16, ZERO on the ring and the tail rows; Xp (T, Cin) bf16 DENSE (row stride Cin), zero on
 * the ring, with pitch + 1 = W + 3 readable rows of finite values before its first and after its last row (tap offsets
 * reach there, always multiplied by a zero row of dYp).  yv_view_op mode 6 writes both layouts. */
int yv_wgrad_conv3(const void* dYp, int ldy, const void* Xp, int Cin, int pitch, int T, int N, float* dW, int ldw,
                   void* stream);

/* out_t[c][r] = in[r][c] (bf16), rows of out_t zero padded up to the next multiple of 64 (ld_out >= that). */
int yv_transpose_bf16(const void* in, int rows, int cols, long long ld_in, void* out_t, long long ld_out, void* stream);

/* f32 master weight (N,K) -> bf16 (N,K) working copy and bf16 transposed (K, ld_t >= round64(N)) copy. */
int yv_cast_weights(const float* w, int N, int K, void* w_bf16, void* wt_bf16, long long ld_t, void* stream);

/* x f32 (rows, cols) -> optional bf16 copy y and optional column sums (bias gradients; deterministic two-stage).
 * ws: yv_colsum_ws_floats(rows, cols) floats. */
size_t yv_colsum_ws_floats(int rows, int cols);
int yv_cast_colsum(const float* x, int rows, int cols, void* y_bf16, float* colsum, int accumulate, float* ws,
                   void* stream);
int yv_colsum_bf16(const void* x, int rows, int cols, long long ld, float* colsum, int accumulate, float* ws,
                   void* stream);

/* LayerNorm backward: dx += dLN(dy) (f32 stream), dgamma / dbeta (D) f32; statistics recomputed from x. */
size_t yv_layernorm_bwd_ws_floats(int rows, int D);
int yv_layernorm_bwd(const float* x, long long ldx, const float* gamma, const void* dy, long long lddy, int rows, int D,
                     float eps, float* dx, long long lddx, float* dgamma, float* dbeta, float* ws, void* stream);

/* out (N,D) = sum over the R crops of dx (R,N,D): d pos_embed (row 0 is also d cls_token). */
int yv_token_reduce(const float* dx, int R, int N, int D, float* out, void* stream);

/* Network_Wrapper.fc backward (utils/utils.py:64-72): feats (R,ldf) f32 backbone logits, dlogits (R,nc) ->
 * dW1 (128,1000), db1, dW2 (nc,128), db2 (f32) and d feats (R,ldd) bf16 (columns >= 1000 zero).
 * ws: 2*R*128 floats. */
int yv_head_bwd(const float* feats, int ldf, const float* w1t, const float* b1, const float* w2, const float* dlogits,
                int R, int nc, float* dw1, float* db1, float* dw2, float* db2, void* dfeats_bf16, int ldd, float* ws,
                void* stream);

/* build_loss = LSCE(0.1)/6 + Focal(1,2,'mean')*5/6 (utils/trainClass.py:46-66,162-185,362-370)
 * loss = w_lsce*LSCE + w_focal*Focal (build_loss: 1/6, 5/6; each term alone: (1,0) / (0,1)).
 * logits (B,nc) f32, labels (B) i32 -> loss (1) f32, grad (B,nc) f32 (d loss / d logits). */
int yv_loss_fwd_bwd(const float* logits, const int32_t* labels, int B, int nc, float w_lsce, float w_focal,
                    float* loss, float* grad, void* stream);

/* torch.optim.SGD(momentum, weight_decay) step (utils/trainClass.py:442-443), fp32, in place:
 * g = g*grad_scale + wd*p; m = first ? g : mu*m + g; p -= lr*m.  (grad_scale: 1/world_size after a SUM all-reduce) */
int yv_sgd_step(float* p, const float* g, float* m, size_t n, float lr, float momentum, float weight_decay,
                float grad_scale, int first, void* bf16_mirror /* optional: bf16 copy of the updated p */, void* stream);

/* Batched bf16 transpose dst[b] (cols, rows) = src[b] (rows, cols)^T; matrix b starts `*_stride` elements after matrix b-1.
 * rows, cols multiples of 8, 16-byte aligned.  The trainer (autograd of every nn.Linear of timm's Block, reference loop
 * utils/trainClass.py:403-407 `loss.backward()`) keeps a transposed bf16 mirror of the block weights so that the data
 * gradient dX = dY . W runs as an ordinary yv_linear on W^T. */
int yv_transpose_bf16_batched(const void* src, void* dst, int rows, int cols, int batch, long long src_stride,
                              long long dst_stride, void* stream);

/* Optimisers ultralytics builds for `optimizer='auto'` (the reference calls `.train(..., lr0=1e-4, lrf=1e-4)` with that
 * default, utils/trainYolo.py:33), element-wise in the operation order of torch's single-tensor implementations:
 * kind 1 = torch.optim.SGD(momentum=beta1, nesterov=True, weight_decay) (v unused), kind 2 = torch.optim.AdamW(betas,
 * eps, weight_decay).  step counts from 1 (bias corrections; step 1 initialises the state).  g is scaled by grad_scale. */
int yv_optim_step(int kind, float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float grad_scale, int step, void* bf16_mirror, void* stream);

/* dst = a*dst + b*src (gradient accumulation over `accumulate` batches: a = b = 1). */
int yv_axpby(float* dst, const float* src, size_t n, float a, float b, void* stream);

/* ultralytics ModelEMA: ema = decay*ema + (1-decay)*src. */
int yv_ema_update(float* ema, const float* src, size_t n, float decay, void* stream);

/* ---- detector training (the ultralytics trainer behind utils/trainYolo.py:13-35) --------
 * Conv = conv(no bias) -> BatchNorm2d(eps 1e-3, momentum 0.03) -> SiLU, un-folded; activations are NHWC bf16 views:
 * (rows = B*H*W, C) with a row stride `ld` (elements), C a multiple of 8, pointers 16-byte aligned. */

/* u8 RGB pixels -> bf16 value/255, channels padded 3 -> 8 with zeros (`blob`, YOLOTensorRT_yolodet_py_解读.md:70-74). */
int yv_blob_nhwc8(const void* images_u8, long long pixels, void* out_bf16, void* stream);

/* Per-channel batch statistics of z (T,C): mean, rstd = 1/sqrt(biased var + eps); optional running estimates
 * (run = (1-momentum)*run + momentum*{mean, unbiased var}, both or neither).  ws: yv_bn_ws_floats(T,C) floats. */
size_t yv_bn_ws_floats(long long T, int C);
int yv_bn_stats(const void* z, long long ldz, long long T, int C, float eps, float momentum, float* mean, float* rstd,
                float* run_mean, float* run_var, float* ws, size_t ws_floats, void* stream);

/* a = act(gamma*(z-mean)*rstd + beta) [+ res]   (act 1 = SiLU, 0 = identity); a, res bf16 views. */
int yv_bn_act_fwd(const void* z, long long ldz, long long T, int C, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, const void* res, long long ldres, void* out, long long ldo,
                  int act, void* stream);

/* Backward of the above: g = da*act'(u); dbeta = sum g; dgamma = sum g*xhat;
 * dz = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) with batch statistics, gamma*rstd*g with frozen ones. */
int yv_bn_act_bwd(const void* da, long long ldda, const void* z, long long ldz, long long T, int C, const float* mean,
                  const float* rstd, const float* gamma, const float* beta, int act, int batch_stats, float* dgamma,
                  float* dbeta, void* dz, long long lddz, float* ws, size_t ws_floats, void* stream);

/* Element-wise view operations on (B,H,W,C) bf16 views.  mode 0 copy, 1 dst += src, 2 nearest-2x upsample
 * (dst (B,2H,2W) <- src (B,H,W)), 3 its adjoint accumulated (dst (B,H,W) += 2x2 block sums of src (B,2H,2W)),
 * 4 zero insertion (dst (B,2H,2W): [2y][2x] = src[y][x], 0 elsewhere: stride-2 data gradient), 5 zero fill,
 * 6 zero-ring padding (dst (B,H+2,W+2): interior = src (B,H,W), ring = 0: the operand layout of yv_wgrad_conv3). */
int yv_view_op(int mode, const void* src, long long ld_src, void* dst, long long ld_dst, int B, int H, int W, int C,
               void* stream);

/* din += adjoint of max_pool2d(k 5, s 1, p 2) at dout, the maximum of a window being its first one in scan order.
 * ws: B*H*W*C bytes (per-window argmax offsets). */
int yv_maxpool5_bwd(const void* x, long long ldx, const void* dout, long long lddo, void* din, long long lddi, int B, int H,
                    int W, int C, void* ws, size_t ws_bytes, void* stream);

/* col (B*Hout*Wout, 9*C) = 3x3 / pad 1 patches of x (B,Hin,Win,C), K order (ky,kx,c): the X operand of yv_wgrad. */
int yv_im2col3(const void* x, long long ldx, int B, int Hin, int Win, int C, int stride, void* col, void* stream);

/* wd (Cin, taps, Cout) = 180-degree tap flip + in/out transpose of w (Cout, taps, Cin) bf16: the data gradient of a
 * stride-1 conv is yv_conv2d(dz, wd); of a stride-2 conv the same over the zero-inserted dz (yv_view_op mode 4). */
int yv_conv_weight_dgrad(const void* w, int Cout, int taps, int Cin, void* wd, void* stream);

/* v8 detection loss and its gradient (the objective of `YOLO(pt).train(...)`, utils/trainYolo.py:33; published
 * ultralytics v8DetectionLoss restated in oracle/yolo_train.py - parity unpinned): DFL decode, TaskAlignedAssigner
 * (topk 10, alpha 0.5, beta 6; top-k ties by ascending anchor index among the anchors inside the box), CIoU + DFL + BCE,
 * normalised by the batch sum of target scores, times B.
 * box[s] (B*h_s*h_s, 64) / cls[s] (B*h_s*h_s, ncp) f32 logits of the three scales (h_s = size / {8,16,32}) and the
 * matching gradient buffers; gt_boxes (B,G,4) xyxy input pixels, gt_labels (B,G), gt_counts (B) valid boxes per image;
 * loss (4) = {total*B, box, cls, dfl}.  ws: yv_detect_loss_ws_bytes(B, A, G), A = sum h_s^2. */
size_t yv_detect_loss_ws_bytes(int B, int A, int G);
int yv_detect_loss(const float* const* box, const float* const* cls, float* const* dbox, float* const* dcls, int B, int size,
                   int nc, int ncp, const float* gt_boxes, const int32_t* gt_labels, const int32_t* gt_counts, int G,
                   float gain_box, float gain_cls, float gain_dfl, float* loss, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YV_HIP_H */
