/*
 * odbox -- C-ABI of the MI355X (gfx950) box-processing library, libodbox.so.
 *
 * The reference (bioothod/object_detection) has no FFI of its own: the path sits behind
 * plain Python module functions on tf.Tensor.  Each entry point below names
 * the reference function (file:line under the reference tree) whose arithmetic it replaces;
 * object_detection_amd/{anchors,bndbox,metric,detection,cocoeval}.py bind them with ctypes
 * under the reference's own Python signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch-ROCm storage) unless noted;
 *     row-major, fp32 / int32 / int8, caller owns every buffer;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream)
 *     and the call returns without synchronising; no allocation happens inside;
 *   - return 0 on success, a negative ODBOX_E* code on failure; odbox_last_error() returns
 *     the calling thread's last message; no C++ exception crosses the boundary;
 *   - safe to call concurrently from several host threads on different streams.
 */
#ifndef ODBOX_H
#define ODBOX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 200: the signatures of odbox_match_encode (trailing out_gt_best) and odbox_gt_prepare_xywh / _poly (`rows`) changed
 * against 100, and the first 256 bytes of the decode+NMS workspace became the library's epoch / ready line.  A caller
 * built against another header must not call in: compare odbox_version() with ODBOX_VERSION first. */
#define ODBOX_VERSION 201            /* 0.2.1: 200 + odbox_set_nms_fused_tail (no signature of 200 changed) */
#define ODBOX_EINVAL (-1)            /* bad argument (shape, NULL, limit exceeded) */
#define ODBOX_EHIP   (-2)            /* a HIP runtime call failed */
#define ODBOX_ENOSPC (-3)            /* workspace too small */

#define ODBOX_MAX_CLASSES   4096     /* class id is 12 bits of the candidate key */
#define ODBOX_MAX_ANCHORS   1048576  /* anchor id is 20 bits of the candidate key */
#define ODBOX_MAX_RET       4096     /* kept list of one (image, class) lives in LDS */
#define ODBOX_COCO_MAX_THRS  16      /* IoU thresholds of one odbox_coco_match call */
#define ODBOX_COCO_MAX_AREAS 8       /* area ranges of one odbox_coco_match call */
#define ODBOX_COCO_MAX_GT    1024    /* GT rows per image held in registers by the matcher */

int odbox_version(void);
const char *odbox_last_error(void);

/* How odbox_match_encode writes its dense outputs (no counterpart in the reference: a property of this
 * GPU's memory system).  Through the caches is right when the stores land in lines the 256 MB Infinity Cache
 * still holds (one buffer rewritten call after call with nothing in between) or when the consumer reads the
 * targets next; around them (nontemporal stores) when the buffer is cold or large.  The library cannot see
 * what passed through the cache since a buffer was last written, so AUTO (default) is a function of the call's
 * shape alone and assumes a cold buffer: labels beyond 512 MB (GT lists of at most 80 rows) and label-stream
 * shapes (those the lockstep sweep serves, see odbox_match_encode_plan) go around the caches, everything else
 * through them.  The same call always launches the same kernel -- also under HIP graph capture -- and
 * odbox_match_encode_plan tells which.  CACHE / STREAM force one or the other (say CACHE when the consumer
 * reads the targets straight away or one buffer is rewritten in a tight loop) -- except that label rows
 * narrower than 64 floats (C < 63) always go through the caches: streamed, their tiles are too small to be
 * worth a block each.  Process-wide; returns the previous policy, or ODBOX_EINVAL.  Results are bit-identical
 * under every policy. */
#define ODBOX_POLICY_AUTO   0
#define ODBOX_POLICY_CACHE  1
#define ODBOX_POLICY_STREAM 2
int odbox_set_output_policy(int policy);

/* Which kernel odbox_match_encode would launch for dense labels of this shape under the current output
 * policy (no counterpart in the reference; for tests, benchmarks and logs -- the results are the same
 * bits on every plan).  TILES: one block per 256-anchor tile, stores through the caches.  TILES_STREAM:
 * 64-anchor tiles, nontemporal stores.  SWEEP: streamed outputs with wide label rows (C + 1 >= 69), at
 * most 64 GT rows, B <= 8192, B * G <= 20480 or B * G * 20480 <= the label bytes, 16-byte aligned labels, no per-GT output and at most 768 MB of labels: the label tensor
 * is written as one chip-wide lockstep sweep of 4 KB pieces by dedicated store waves while other waves
 * of the same blocks match and patch (the fastest order into cold HBM).  Returns ODBOX_EINVAL on a bad shape.
 * odbox_match_encode_last_plan: the plan the calling thread's last odbox_match_encode actually took
 * (-1 before the first call). */
#define ODBOX_PLAN_TILES        0
#define ODBOX_PLAN_TILES_STREAM 1
#define ODBOX_PLAN_SWEEP        2
int odbox_match_encode_plan(int B, int N, int G, int C, int labels_aligned16, int want_gt_best);
int odbox_match_encode_last_plan(void);

/* anchors.py:44-87 AnchorGenerator.tile_anchors_over_feature_map:
 * out[(i*fw + j)*A + a] = base[a] + ((j+.5)*stride, (i+.5)*stride, (j+.5)*stride, (i+.5)*stride). */
int odbox_anchor_tile(const float *base /*[A,4]*/, int A, int fh, int fw, float stride,
                      float *out /*[fh*fw*A,4]*/, void *stream);

/* anchors.py:116-213 anchor_targets_bbox, with compute_gt_annotations (:215-274),
 * bndbox.py:254-299 bbox_overlap and anchors.py:277-299 bbox_transform fused.
 * out_regression [B,N,5] and out_labels [B,N,C+1] are the reference's two return values;
 * out_match_idx (argmax GT per anchor), out_state (-1 ignore / 0 background / 1 foreground)
 * and out_max_iou are compact side outputs.  Any out_* may be NULL (skipped); gt_labels may
 * be NULL when out_labels is NULL.  img_h / img_w = +inf disables the out-of-image test
 * (that is compute_gt_annotations on its own).
 * out_gt_best [B,G] (8-byte aligned, or NULL): for every ground-truth row the anchor with the
 * largest IoU over all N anchors and that IoU -- argmax / max of bndbox.py:254-299's matrix over its
 * ANCHOR axis, ties to the lowest anchor, padded rows (anchor 0, -1).  An extra output: the
 * reference never forces a match from it (anchors.py:253-272 is per-anchor only; its one per-GT
 * argmax is the diagnostic at coco.py:399-402), so no other output changes by a bit. */
typedef struct odbox_gt_best { int32_t anchor; float iou; } odbox_gt_best_t;
int odbox_match_encode(const float *anchors /*[N,4]*/, const float *gt_boxes /*[B,G,4]*/,
                       const int32_t *gt_labels /*[B,G]*/, int B, int N, int G, int C,
                       float img_h, float img_w, float neg_thr, float pos_thr,
                       float *out_regression /*[B,N,5]*/, float *out_labels /*[B,N,C+1]*/,
                       int32_t *out_match_idx /*[B,N]*/, int8_t *out_state /*[B,N]*/,
                       float *out_max_iou /*[B,N]*/, odbox_gt_best_t *out_gt_best /*[B,G]*/, void *stream);

/* Fused matcher + loss, the consumer of the targets: metric.py:97-123 (ModelMetric.__call__) with
 * focal_loss (metric.py:9-38: clip to [1e-6, 1-1e-6], alpha_t, pow(1-pt, gamma), log) and
 * tf.keras.losses.Huber(delta, reduction=SUM) evaluated straight from the match, so neither
 * [B,N,C+1] nor [B,N,5] is ever materialised.
 * out_loss[4] = { dist_loss, class_loss, npos, hits }: both losses divided by npos (the number of
 * foreground anchors) when npos != 0, exactly like metric.py:108-123; hits = the number of
 * foreground anchors whose highest predicted class (first maximum) is the matched class, i.e.
 * hits / npos is the CategoricalAccuracy that metric.py:128-130 accumulates.
 * grad_boxes [B,N,4] / grad_scores [B,N,C] (both or neither): gradients of the UNNORMALISED sums
 * w.r.t. pred_boxes / pred_scores (zero for ignored anchors, background boxes and clipped
 * scores); divide by npos to get the gradients of the returned losses.
 * pred_scores are probabilities (the reference calls focal_loss with from_logits=False).
 * workspace: odbox_match_loss_workspace(B, N) bytes, 16-byte aligned. */
size_t odbox_match_loss_workspace(int B, int N);
int odbox_match_loss(const float *anchors /*[N,4]*/, const float *gt_boxes /*[B,G,4]*/,
                     const int32_t *gt_labels /*[B,G]*/, int B, int N, int G, int C,
                     float img_h, float img_w, float neg_thr, float pos_thr,
                     const float *pred_boxes /*[B,N,4]*/, const float *pred_scores /*[B,N,C]*/,
                     float gamma, float alpha, float huber_delta, float *out_loss /*[4]*/,
                     float *grad_boxes /*[B,N,4] or NULL*/, float *grad_scores /*[B,N,C] or NULL*/,
                     void *workspace, size_t workspace_bytes, void *stream);

/* odbox_match_loss for callers that cannot allocate.  No counterpart in the reference: it exists for
 * code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc, which can
 * neither size nor own a workspace.  Same arguments, kernels and results as odbox_match_loss, minus
 * the workspace: the per-tile partial sums live in a 1 MiB array in the library's own device memory
 * (one copy per GPU, held until the process ends), whose address is looked up once per device on
 * the first call there (make that first call outside graph capture).  On EVERY call the stream's device
 * must be the current device (ODBOX_EINVAL otherwise): the kernels run on the stream's GPU with the address
 * of the current GPU's copy.  Limit: B * ceil(N / 256) <= ODBOX_NOALLOC_MAX_TILES, else ODBOX_ENOSPC.  The call
 * itself allocates nothing.  Every call on one GPU shares that GPU's array: calls that overlap on
 * one GPU (different streams without an ordering between them) race; one stream is always fine. */
#define ODBOX_NOALLOC_MAX_TILES 65536
int odbox_match_loss_noalloc(const float *anchors /*[N,4]*/, const float *gt_boxes /*[B,G,4]*/,
                             const int32_t *gt_labels /*[B,G]*/, int B, int N, int G, int C,
                             float img_h, float img_w, float neg_thr, float pos_thr,
                             const float *pred_boxes /*[B,N,4]*/, const float *pred_scores /*[B,N,C]*/,
                             float gamma, float alpha, float huber_delta, float *out_loss /*[4]*/,
                             float *grad_boxes /*[B,N,4] or NULL*/, float *grad_scores /*[B,N,C] or NULL*/,
                             void *stream);

/* The backward pass of the same losses for autograd-style callers (what tf.GradientTape does to
 * metric.py:97-123 in train.py):
 *   grad_boxes  = (*upstream_dist  / n) * d dist_sum  / d pred_boxes
 *   grad_scores = (*upstream_class / n) * d class_sum / d pred_scores,   n = *normalizer, or 1 if it is 0
 * All three scalars live in DEVICE memory and are read by the kernel: pass out_loss + 2 of the
 * forward call as normalizer and the upstream gradients as autograd hands them over, so nothing
 * visits the host and the [B,N,C] gradient is written once, already final.  A NULL upstream means
 * that loss is unused (its gradient is written as zeros); a NULL normalizer means 1.  The match is
 * recomputed from the same inputs (cheaper than saving it: the kernel is bound by the score
 * stream).  No workspace; same alignment rules as odbox_match_loss. */
int odbox_match_loss_backward(const float *anchors /*[N,4]*/, const float *gt_boxes /*[B,G,4]*/,
                              const int32_t *gt_labels /*[B,G]*/, int B, int N, int G, int C,
                              float img_h, float img_w, float neg_thr, float pos_thr,
                              const float *pred_boxes /*[B,N,4]*/, const float *pred_scores /*[B,N,C]*/,
                              float gamma, float alpha, float huber_delta,
                              const float *upstream_dist /*device scalar or NULL*/,
                              const float *upstream_class /*device scalar or NULL*/,
                              const float *normalizer /*device scalar or NULL*/,
                              float *grad_boxes /*[B,N,4]*/, float *grad_scores /*[B,N,C]*/, void *stream);

/* The same two entry points for fp16 predictions: the reference's --use_fp16 run (detection.py:
 * 296-300, mixed_float16) hands fp16 model outputs to metric.py, which casts them to fp32 before
 * any arithmetic (metric.py:16-17; Keras' Huber does the same).  Here the four prediction /
 * gradient tensors are IEEE binary16 in memory: widened on load (exact), all arithmetic in fp32,
 * gradients rounded to nearest even on store; out_loss, upstream and normalizer stay fp32.  Half the bytes
 * of the dominant stream.  pred_boxes / grad_boxes 8-byte aligned, scores 2-byte aligned with
 * grad_scores on the 16-byte phase of pred_scores. */
int odbox_match_loss_f16(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                         int B, int N, int G, int C, float img_h, float img_w, float neg_thr,
                         float pos_thr, const void *pred_boxes /*f16 [B,N,4]*/,
                         const void *pred_scores /*f16 [B,N,C]*/, float gamma, float alpha,
                         float huber_delta, float *out_loss /*[4]*/, void *grad_boxes /*f16 or NULL*/,
                         void *grad_scores /*f16 or NULL*/, void *workspace, size_t workspace_bytes,
                         void *stream);
int odbox_match_loss_backward_f16(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                                  int B, int N, int G, int C, float img_h, float img_w, float neg_thr,
                                  float pos_thr, const void *pred_boxes, const void *pred_scores,
                                  float gamma, float alpha, float huber_delta,
                                  const float *upstream_dist, const float *upstream_class,
                                  const float *normalizer /*three fp32 device scalars, each may be NULL*/,
                                  void *grad_boxes, void *grad_scores, void *stream);

/* The same pair for bfloat16 predictions (no counterpart in the reference, which only has an fp16
 * mode: bfloat16 is the native mixed-precision format of this GPU).  Same rules as the fp16 pair:
 * widened on load (a 16-bit shift, exact), fp32 arithmetic, gradients rounded to nearest even. */
int odbox_match_loss_bf16(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                          int B, int N, int G, int C, float img_h, float img_w, float neg_thr,
                          float pos_thr, const void *pred_boxes /*bf16 [B,N,4]*/,
                          const void *pred_scores /*bf16 [B,N,C]*/, float gamma, float alpha,
                          float huber_delta, float *out_loss /*[4]*/, void *grad_boxes /*bf16 or NULL*/,
                          void *grad_scores /*bf16 or NULL*/, void *workspace, size_t workspace_bytes,
                          void *stream);
int odbox_match_loss_backward_bf16(const float *anchors, const float *gt_boxes, const int32_t *gt_labels,
                                   int B, int N, int G, int C, float img_h, float img_w, float neg_thr,
                                   float pos_thr, const void *pred_boxes, const void *pred_scores,
                                   float gamma, float alpha, float huber_delta,
                                   const float *upstream_dist, const float *upstream_class,
                                   const float *normalizer, void *grad_boxes, void *grad_scores, void *stream);

/* metric.py:9-38 focal_loss(y_true, y_pred, gamma, alpha, from_logits=False, reduction='sum') on
 * dense targets (y_true == 1 marks the positive class): out[1] = sum over all n elements
 * (out has room for 4 floats: out[0], out[2], out[3] are written as 0).  grad [n] or NULL:
 * d sum / d y_pred.  Used for the whole-image head (metric.py:132-134). */
size_t odbox_focal_loss_workspace(size_t n);
int odbox_focal_loss(const float *y_true /*[n]*/, const float *y_pred /*[n]*/, size_t n, float gamma, float alpha,
                     float *out /*[4]*/, float *grad /*[n] or NULL*/, void *workspace, size_t workspace_bytes,
                     void *stream);

/* metric.py:9-38 focal_loss row by row, for the two forms the summed entry point above leaves out: from_logits = 1
 * (metric.py:19-20: tf.nn.softmax over the last axis first) and the un-reduced return (metric.py:33-38 with any
 * reduction other than 'sum' / 'mean': one sum over the class axis per row).  y_true / y_pred [rows, C];
 * out_rows [rows]; grad [rows, C] or NULL: d out_rows[row] / d y_pred (through the softmax when from_logits). */
int odbox_focal_loss_rows(const float *y_true /*[rows,C]*/, const float *y_pred /*[rows,C]*/, size_t rows, int C,
                          float gamma, float alpha, int from_logits, float *out_rows /*[rows]*/,
                          float *grad /*[rows,C] or NULL*/, void *stream);

/* bndbox.py:254-299 bbox_overlap: dense [B,N,G] IoU, -1 for padded GT rows.
 * anchors_batched != 0: anchors is [B,N,4]; else [N,4] shared by every image. */
int odbox_bbox_overlap(const float *anchors, const float *gt_boxes /*[B,G,4]*/, int B, int N, int G,
                       int anchors_batched, float *out_iou /*[B,N,G]*/, void *stream);

/* anchors.py:277-299 bbox_transform: anchors [N,4] (anchors_batched == 0) or [B,N,4]. */
int odbox_bbox_transform(const float *anchors, int anchors_batched, const float *gt_boxes /*[B,N,4]*/,
                         int B, int N, float *out /*[B,N,4]*/, void *stream);

/* bndbox.py:71-99 regress_bndboxes: boxes [N,4] (boxes_batched == 0) or [B,N,4]. */
int odbox_regress_boxes(const float *boxes, int boxes_batched, const float *regressors /*[B,N,4]*/,
                        int B, int N, float *out /*[B,N,4]*/, void *stream);

/* bndbox.py:102-115 clip_boxes: x in [0, img_w-1], y in [0, img_h-1]; n boxes, may be in place. */
int odbox_clip_boxes(const float *boxes /*[n,4]*/, size_t n, float img_h, float img_w,
                     float *out /*[n,4]*/, void *stream);

/* bndbox.py:10-41 scale_boxes (divide == 0: x*fx, y*fy) and bndbox.py:44-68
 * normalize_bndboxes (divide != 0: x/fx, y/fy). */
int odbox_scale_boxes(const float *boxes /*[n,4]*/, size_t n, float fx, float fy, int divide,
                      float *out /*[n,4]*/, void *stream);

/* The score tier of odbox_decode_nms (no counterpart in the reference; the results are the same bits with it on or
 * off).  Greedy NMS visits a class in score order and a candidate is only suppressed by higher-scoring ones, so the
 * kept boxes above a score are decided by the candidates above that score alone.  Pass 1 therefore runs partition /
 * NMS / top-k on every image's best factor * max_ret candidates (ties at the edge included); an image whose tier
 * keeps at least max_ret boxes is finished, the others are done again on their whole list by pass 2 (one more
 * launch, in which blocks of finished images leave at once).  factor is the LARGEST tier pass 1 tries: the fused pass-1
 * block (odbox_set_nms_fused_tail) starts on the best factor / 4 * max_ret candidates and grows the tier only when
 * those keep too few.  On BASELINE configs[4] (4 500 candidates per image, max_ret 100) every image finishes in pass 1
 * and a call takes 197-201 us instead of 228 without a tier; a batch in which EVERY image falls through to pass 2 takes
 * 270-280 us.  factor: default 8; 0 turns the tier off (a caller whose images keep fewer than one candidate in eight).
 * Process-wide; returns the previous factor, or ODBOX_EINVAL. */
int odbox_set_nms_tier(int factor);

/* Pass 1 of odbox_decode_nms as ONE launch (no counterpart in the reference; the same bits either way).  on = 1
 * (default): tier selection, class partition, per-class NMS and the final top-k of an image run in one block, one
 * candidate per lane, whenever the tier fits it (max_ret <= 113 at the default tier factor; C <= 2048); a call is then
 * three launches: score scan, this kernel, pass 2.  on = 0: the separate partition / NMS / final kernels, as for
 * shapes the fused block does not hold.  Process-wide; returns the previous setting, or ODBOX_EINVAL. */
int odbox_set_nms_fused_tail(int on);

/* Bytes of device workspace odbox_decode_nms needs for these sizes (0 on bad arguments). */
size_t odbox_decode_nms_workspace(int B, int N, int C, int max_ret, int pre_nms_topk);

/* encoder.py:200-214 inference tail = bndbox.py:71-99 regress_bndboxes -> :102-115 clip_boxes
 * -> :117-252 nms (class argmax, score filter, per-class tf.image.non_max_suppression, concat,
 * top_k, zero pad).  anchors == NULL: regress_or_boxes already holds decoded boxes (the plain
 * bndbox.nms entry).  pre_nms_topk == 0 is the reference's unbounded behaviour; > 0 keeps the
 * best pre_nms_topk candidates per image by (score desc, anchor asc) before the class
 * partition (a documented extension, no counterpart in the reference).
 * out_keep_anchor [B,max_ret] (anchor id of every detection, -1 padding) and out_count [B]
 * may be NULL.  workspace must be 256-byte aligned. */
int odbox_decode_nms(const float *anchors /*[N,4] or NULL*/, const float *regress_or_boxes /*[B,N,4]*/,
                     const float *scores /*[B,N,C]*/, int B, int N, int C,
                     float img_h, float img_w, int do_clip,
                     float score_thr, float iou_thr, int max_ret, int pre_nms_topk,
                     float *out_boxes /*[B,max_ret,4]*/, float *out_scores /*[B,max_ret]*/,
                     int32_t *out_labels /*[B,max_ret]*/, int32_t *out_keep_anchor /*[B,max_ret]*/,
                     int32_t *out_count /*[B]*/, void *workspace, size_t workspace_bytes, void *stream);

/* Element types of model outputs handed to odbox_decode_nms_mixed. */
#define ODBOX_DTYPE_F32 0
#define ODBOX_DTYPE_F16 1   /* IEEE binary16 */
#define ODBOX_DTYPE_BF16 2  /* bfloat16 */

/* odbox_decode_nms for a model that emits 16-bit outputs (PyTorch autocast on this GPU: bfloat16; the
 * reference's own mixed-precision mode, detection.py:296-300: float16).  Same path, same outputs (fp32 boxes and
 * scores, int32 labels), same workspace: regress_or_boxes [B,N,4] of box_dtype and scores [B,N,C] of score_dtype
 * are widened exactly to fp32 where they are read, so the result is what odbox_decode_nms gives for the widened
 * tensors, bit for bit, at half the bytes of the score stream that bounds the call.  No counterpart in the
 * reference (encoder.py:200-214 runs the tail in fp32).  16-bit scores are read two classes per dword: C must
 * be even (ODBOX_EINVAL otherwise).  16-bit boxes must be 8-byte aligned. */
int odbox_decode_nms_mixed(const float *anchors /*[N,4] or NULL*/, const void *regress_or_boxes /*[B,N,4]*/,
                           int box_dtype, const void *scores /*[B,N,C]*/, int score_dtype, int B, int N, int C,
                           float img_h, float img_w, int do_clip,
                           float score_thr, float iou_thr, int max_ret, int pre_nms_topk,
                           float *out_boxes /*[B,max_ret,4]*/, float *out_scores /*[B,max_ret]*/,
                           int32_t *out_labels /*[B,max_ret]*/, int32_t *out_keep_anchor /*[B,max_ret]*/,
                           int32_t *out_count /*[B]*/, void *workspace, size_t workspace_bytes, void *stream);

/* ---- evaluation side: the step after the all-gather of detections ---- */

/* evaluate.py:14-28 _COCO_result / :31-55 _COCO_gt_annot: xyxy -> (x, y, w, h) with fp32 w = x1-x0,
 * h = y1-y0; out_area_f32 = h*w in fp32 (the GT annotation's 'area', :41); out_area_f64 =
 * (double)w*(double)h (the detection's 'area', :177).  Any output may be NULL. */
int odbox_coco_records(const float *boxes /*[n,4] xyxy*/, size_t n, float *out_xywh /*[n,4]*/,
                       float *out_area_f32 /*[n]*/, double *out_area_f64 /*[n]*/, void *stream);

/* cocoeval.py:167-194 computeIoU = pycocotools maskUtils.iou on boxes (bbIou, iscrowd = 0), in
 * fp64 like the C original: out[b,d,g] for every detection slot d and GT row g of image b.
 * self.ious[img, cat] is the sub-matrix of the category's rows / slots in score order. */
int odbox_coco_iou(const float *dt_boxes /*[B,R,4] xyxy*/, const float *gt_boxes /*[B,G,4] xyxy*/,
                   int B, int R, int G, double *out_iou /*[B,R,G]*/, void *stream);

/* cocoeval.py:239-317 evaluateImg for every (image, category, area range, IoU threshold) at once,
 * with _prepare (:88-123) and evaluate.py:93-110 folded in: GT rows with label -1 are padding and
 * dropped; detection rows are NOT filtered (zero-score padding rows are category-0 detections,
 * like the reference feeds them); rows / slots whose label is outside [0, C) are not evaluated.
 * Each (image, category) detection list is ordered by (score desc, slot asc) = the reference's
 * stable mergesort, cut at max_det (= maxDets[-1]).
 *   out_dt_match  [B,A,T,R] int32  1 + matched GT row (index into the image's G rows), 0 = none
 *   out_dt_ignore [B,A,T,R] uint8  dtIgnore
 *   out_gt_match  [B,A,T,G] int32  1 + matched detection slot, 0 = none          (may be NULL)
 *   out_gt_ignore [B,A,G]   uint8  0 counted / 1 ignored (area range) / 255 absent (may be NULL)
 *   out_dt_rank   [B,R]     int32  position in its (image, category) list, -1 = not evaluated
 * iou_thrs_h [T] and area_rng_h [A,2] are HOST arrays of doubles (cocoeval.py:510-513).
 * COCO ids follow from the indices: detection id = 1 + b*R + slot (evaluate.py:178); GT id =
 * 1 + number of kept GT rows before it (evaluate.py:53,141). */
size_t odbox_coco_match_workspace(int B, int R);
int odbox_coco_match(const float *dt_boxes /*[B,R,4]*/, const float *dt_scores /*[B,R]*/,
                     const int32_t *dt_labels /*[B,R]*/, const float *gt_boxes /*[B,G,4]*/,
                     const int32_t *gt_labels /*[B,G]*/, int B, int R, int G, int C,
                     const double *iou_thrs_h /*[T]*/, int T, const double *area_rng_h /*[A,2]*/, int A,
                     int max_det, int32_t *out_dt_match, uint8_t *out_dt_ignore, int32_t *out_gt_match,
                     uint8_t *out_gt_ignore, int32_t *out_dt_rank, void *workspace, size_t workspace_bytes,
                     void *stream);

/* ---- ground-truth preparation: the step that feeds the matcher ---- */

/* detection.py:97-141 prepare_example (box arithmetic of the evaluation path) + :257-262
 * padded_batch.  Images are ragged: image b owns rows row_offsets[b] .. row_offsets[b+1] of
 * boxes_xywh / labels (device int32 [B+1]).  Per row: x0 += (max(H,W) - W)/2, y0 += (max(H,W) - H)/2
 * (square pad), 4 corners, each coordinate / max(H,W) * image_size, optional round to fp16
 * (half_round != 0: the mixed-precision input pipeline), polygon2bbox (:77-95), rows beyond
 * max_items cut, missing rows = -1 boxes / -1 labels.  mode 0 = (xmin,ymin,xmax,ymax),
 * 1 = want_yx, 2 = want_wh.  orig_hw [B,2] = (height, width) of the original image as floats.
 * rows = length of boxes_xywh / labels (both may be NULL when it is 0); the offsets are read on the
 * device and clamped into [0, rows], so a bad table gives wrong rows, never a fault. */
int odbox_gt_prepare_xywh(const float *boxes_xywh /*[rows,4]*/, const int32_t *labels /*[rows]*/,
                          const int32_t *row_offsets /*[B+1]*/, int rows, const float *orig_hw /*[B,2]*/, int B,
                          int max_items, float image_size, int half_round, int mode,
                          float *out_boxes /*[B,max_items,4]*/, int32_t *out_labels /*[B,max_items]*/,
                          void *stream);

/* The same from corner polygons [rows,4,2] on a square canvas of side canvas_size[b] (what the
 * training augmentation hands back, detection.py:126,131-141). */
int odbox_gt_prepare_poly(const float *polys /*[rows,4,2]*/, const int32_t *labels /*[rows]*/,
                          const int32_t *row_offsets /*[B+1]*/, int rows, const float *canvas_size /*[B]*/, int B,
                          int max_items, float image_size, int half_round, int mode,
                          float *out_boxes /*[B,max_items,4]*/, int32_t *out_labels /*[B,max_items]*/,
                          void *stream);

/* ---- output side of inference: detections back to original image pixels ---- */

/* infdet.py:140-174 (run_eval's per-detection loop): the model saw the image padded to a square
 * and resized to image_size; per image  max_side = max(imh, imw), pad = (max_side - im{h,w}) / 2,
 * square_scale = max_side / image_size, and each box  bb.astype(float) * square_scale  with
 * pad_x / pad_y subtracted -- Python floats, so fp64 here, bit for bit.  The loop stops at the
 * first score <= 0 (:161-162): out_count[b] = that index, rows from there on are written as 0.
 * orig_hw [B,2] int32 = (height, width) of the original image (cv2 image.shape[:2]). */
int odbox_restore_boxes(const float *boxes /*[B,R,4]*/, const float *scores /*[B,R]*/,
                        const int32_t *orig_hw /*[B,2]*/, int B, int R, int image_size,
                        double *out_boxes /*[B,R,4] f64*/, int32_t *out_count /*[B]*/, void *stream);

/* ---- legacy / offline IoU conventions, NumPy float64 arithmetic ---- */

/* mode 0: coco.py:169-191 calc_iou(box, boxes, area) for K boxes at once -- boxes_a [K,4] and
 * boxes_b [N,4] in (cx, cy, h, w) layout, area_b [N] = np_anchor_areas (an input, not recomputed);
 * mode 1: calculate_metrics.py:22-45 calc_ious(pred_boxes, true_box) -- boxes_a [K,4] true boxes and
 * boxes_b [N,4] predictions, both xyxy, area_b ignored.  Both add one pixel to the intersection's
 * width and height and none to the areas (so an IoU can exceed 1), in float64 with the reference's
 * association.  out_iou [K,N]. */
int odbox_legacy_iou(const double *boxes_a /*[K,4]*/, const double *boxes_b /*[N,4]*/,
                     const double *area_b /*[N], mode 0*/, int K, int N, int mode,
                     double *out_iou /*[K,N]*/, void *stream);

/* coco.py:379-441, the sequential per-GT matcher of COCO_Iterable.process_image, for B images:
 * anchors [N,4] (cx, cy, h, w) float64 with their areas [N]; gt_xywh [B,G,4] rows of (x0, y0, w, h)
 * visited in order, rows with w <= 3 or h <= 3 skipped (:418; -1 padding rows fall under it);
 * gt_labels [B,G] = the converted category of each row (self.cats[cat_id]).  An anchor takes a row's
 * box (cx, cy, h, w) and label when iou > iou_thr (0.5 in the reference) and iou > its running
 * maximum, which is kept ROUNDED to float32 like the reference's max_ious array.
 * out_boxes [B,N,4] = true_bboxes (the anchor itself where nothing matched), out_labels [B,N] =
 * true_labels as float64 (0 = nothing), out_max_iou [B,N] float32 (may be NULL), out_label_sum [B]
 * = true_labels.sum() per image (may be NULL; 0 is the reference's "there are no true labels"). */
int odbox_legacy_match(const double *anchors_cxcyhw /*[N,4]*/, const double *anchor_areas /*[N]*/,
                       const double *gt_xywh /*[B,G,4]*/, const int32_t *gt_labels /*[B,G]*/,
                       int B, int N, int G, double iou_thr, double *out_boxes /*[B,N,4]*/,
                       double *out_labels /*[B,N]*/, float *out_max_iou /*[B,N]*/,
                       double *out_label_sum /*[B]*/, void *stream);

/* ---- multi-GPU plumbing (the reference has no counterpart) ---- */

/* The fixed-size detection records that ONE all-gather moves between ranks: n = images * max_ret
 * rows of 6 words (x0, y0, x1, y1, score, label bit-cast to float).  pack: the three outputs of
 * odbox_decode_nms -> records; unpack: the gathered records -> three tensors. */
int odbox_pack_records(const float *boxes /*[n,4]*/, const float *scores /*[n]*/, const int32_t *labels /*[n]*/,
                       size_t n, float *records /*[n,6]*/, void *stream);
int odbox_unpack_records(const float *records /*[n,6]*/, size_t n, float *boxes /*[n,4]*/, float *scores /*[n]*/,
                         int32_t *labels /*[n]*/, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ODBOX_H */
