/*
 * ron_hip.h -- C ABI of libron_hip.so: the MI355X (gfx950) RON-320 inference hot path.
 *
 * The reference (HiKapok/RON_Tensorflow) is pure Python/TensorFlow-1 and has NO FFI of its
 * own; every entry point below replaces the reference *Python* interface quoted next to it
 * (file:line relative to the reference repository).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add to call them.
 *
 * Conventions
 *   - Plain C: pointers, sizes, PODs.  No torch / TF types.  `stream` is a hipStream_t passed
 *     as void* (NULL = default stream).  All device pointers are HIP device pointers.
 *   - Every function returns 0 on success or a negative ron_status; the message is available
 *     through ron_last_error() (per thread when no context is involved).
 *   - Nothing here synchronises with the host unless stated; work is enqueued on `stream`.
 *   - Ownership: the caller owns every input/output buffer; a ron_ctx owns its weights,
 *     anchors and workspace.  One ron_ctx per device; calls on one ctx are not re-entrant.
 *     Everything a ctx owns - activations, split-K scratch, ron_detect's head buffers and
 *     post-processing workspace - is allocated by ron_create / ron_finalize_weights /
 *     ron_clone for max_batch: no entry point that enqueues work allocates or frees.
 *     The post-processing workspace of a ctx is cleaned by the kernels that use it, in
 *     stream order: the ron_detect calls of ONE ctx must all go to one stream, or be ordered
 *     by the caller (events); two batches in flight take two contexts (ron_clone).  A
 *     ron_detect that returns an error leaves the workspace to be re-zeroed by the next call.
 *   - Tensor layout: NHWC, row-major, fp32 at the boundary.  Boxes are (ymin, xmin, ymax, xmax)
 *     in normalised image coordinates.  Head tensors are ordered coarse -> fine
 *     (block7 5x5, block6 10x10, block5 20x20, block4 40x40), RONParams.feat_layers,
 *     nets/ron_vgg_320.py:101.
 */
#ifndef RON_HIP_H_
#define RON_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RON_MAX_LAYERS 8
#define RON_MAX_ANCHORS_PER_CELL 16
#define RON_MAX_GT 256             /* ground-truth boxes per image ron_bboxes_matching accepts */
#define RON_MAX_TOPK 512           /* rows a detection list can hold (np_methods top_k = 400) */
#define RON_MAX_CLASSES 128        /* RONParams.num_classes (background included) the entry points accept: VOC 21, COCO 81 / 91 */

typedef enum {
  RON_OK = 0,
  RON_ERR_INVALID = -1,      /* bad argument                                   */
  RON_ERR_HIP = -2,          /* a HIP runtime call failed                      */
  RON_ERR_STATE = -3,        /* call order (e.g. forward before finalize)      */
  RON_ERR_UNKNOWN_NAME = -4, /* unknown variable / end-point / network name    */
  RON_ERR_UNSUPPORTED = -5
} ron_status;

/* REDUCEDFC / FULL: RON-320 bodies (nets/ron_vgg_320.py:510-580 / :434-508); SSD512: nets/ssd_vgg_512.py:364-460 */
typedef enum { RON_VARIANT_REDUCEDFC = 0, RON_VARIANT_FULL = 1, RON_VARIANT_SSD512 = 2 } ron_variant;
/* Arithmetic of the conv stack (activations + weights in HBM, what the MFMAs consume); accumulation is fp32 in all of them.
 *   F32   : v_mfma_f32_16x16x4_f32, exact fp32 products (parity mode, 157 TFLOP/s matrix peak)
 *   BF16  : v_mfma_f32_16x16x32_bf16 (benchmark mode of BASELINE config 2)
 *   F16   : v_mfma_f32_16x16x32_f16
 *   F16X3 : split precision -- every value is stored as two f16 planes hi = rnd(v), lo = rnd(v - hi) (22 mantissa bits,
 *           4 bytes per element: 32 hi followed by 32 lo per 128-byte channel chunk) and a product is three f16 MFMAs
 *           hi*hi + lo*hi + hi*lo into the fp32 accumulator (the dropped lo*lo term is 2^-22 relative); weights carry a
 *           per-layer power-of-two scale (undone in the epilogue) so that their lo plane stays a normal f16.  fp32-grade
 *           head tensors (detections within 1e-4 of the fp32 CPU reference) at a third of the f16 matrix peak.
 *           VALID ACTIVATION RANGE (activations carry no scale): 22 bits hold for 2^-3 <= |v| < 65504.  Above, a value
 *           saturates instead of overflowing: up to 131008 it is kept as 65504 + rest (f16 spacing of 32), beyond it clips;
 *           NaN stays NaN.  Below 2^-3 the lo plane is an f16 subnormal (the matrix core does not flush it): an ABSOLUTE error
 *           floor of 2^-25 per stored value, i.e. a tensor whose whole scale is 2^-10 keeps ~15 bits.  RON / SSD activations on
 *           unscaled mean-subtracted images are O(1) .. O(10^3).  tests/test_gpu_conv.py holds both regimes against float64. */
typedef enum { RON_DTYPE_F32 = 0, RON_DTYPE_BF16 = 1, RON_DTYPE_F16 = 2, RON_DTYPE_F16X3 = 3 } ron_dtype;

const char* ron_last_error(void);
/* ABI version of the library (bumped on any signature change). */
int ron_abi_version(void);
/* Host utility: CRC32C of a buffer (chain with `crc`, 0 first) -- the checksum of TensorFlow V2 checkpoint files
 * (tf.train.Saver(write_version=2), ron_net.py:395-398), used by ron_tensorflow_amd/checkpoint.py. */
uint32_t ron_crc32c(const void* data, uint64_t nbytes, uint32_t crc);

/* ------------------------------------------------------------------------------------------
 * Anchors.  Replaces ron_anchor_one_layer / ron_anchors_all_layers / RONNet.anchors
 * (nets/ron_vgg_320.py:285-333, :336-355, :162-171).  Pure host function, bit-exact with the
 * reference's numpy float32 arithmetic.
 *   y, x : [feat_h * feat_w]   (the reference returns them as [H, W, 1])
 *   h, w : [n_sizes * n_ratios], anchor a = i_ratio * n_sizes + j_size
 * ---------------------------------------------------------------------------------------- */
int ron_anchor_one_layer(int img_h, int img_w, int feat_h, int feat_w,
                         const double* sizes, int n_sizes,
                         const double* ratios, int n_ratios,
                         double step, double offset,
                         float* y, float* x, float* h, float* w);

/* SSD flavour: ssd_anchor_one_layer (nets/ssd_vgg_512.py:286-338); A = n_sizes + n_ratios anchors per cell:
 * [sizes[0] square, sqrt(sizes[0]*sizes[1]) square, sizes[0] at each ratio]. */
int ron_ssd_anchor_one_layer(int img_h, int img_w, int feat_h, int feat_w,
                             const double* sizes, int n_sizes,
                             const double* ratios, int n_ratios,
                             double step, double offset,
                             float* y, float* x, float* h, float* w);

/* ------------------------------------------------------------------------------------------
 * Head tensors of one batch (device pointers, fp32).  This is what RONNet.net() returns as
 * Python lists (nets/ron_vgg_320.py:136-154, :580) and what bboxes_decode / detected_bboxes /
 * np_methods.ssd_bboxes_select consume.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t num_layers;
  int32_t num_classes;                      /* C, background included (21)                  */
  int32_t feat_h[RON_MAX_LAYERS];
  int32_t feat_w[RON_MAX_LAYERS];
  int32_t num_anchors[RON_MAX_LAYERS];      /* A per cell (10)                              */
  const float* cls[RON_MAX_LAYERS];         /* [N,H,W,A,C]  logits or probabilities         */
  const float* obj[RON_MAX_LAYERS];         /* [N,H,W,A,2] logits | [N,H,W,A,1] prob | NULL */
  const float* loc[RON_MAX_LAYERS];         /* [N,H,W,A,4]  raw offsets or decoded boxes    */
  /* anchors (device): y, x [H*W]; h, w [A].  Needed only when loc holds raw offsets. */
  const float* anchor_y[RON_MAX_LAYERS];
  const float* anchor_x[RON_MAX_LAYERS];
  const float* anchor_h[RON_MAX_LAYERS];
  const float* anchor_w[RON_MAX_LAYERS];
} ron_heads;

/* flags for ron_post_cfg.input_flags */
#define RON_IN_CLS_IS_PROB 1u   /* cls holds softmax probabilities (RONNet.net()[0])          */
#define RON_IN_OBJ_IS_PROB 2u   /* obj holds P(object) [N,H,W,A,1] (RONNet.net()[2])          */
#define RON_IN_LOC_DECODED 4u   /* loc already holds decoded boxes (RONNet.bboxes_decode)     */
/* bits 30 and 31 of input_flags are reserved for the library: callers pass them as 0 */

typedef struct {
  float objectness_thres;   /* eval_ron_network.py:66-67,227-229 (0.03); ignored if obj NULL */
  float select_threshold;   /* np_methods.py:59 / eval_ron_network.py:64-65; 0 = the arg-max
                             * branch (np_methods.py:82-89): one candidate per anchor       */
  float nms_threshold;      /* np_methods.py:229                                             */
  int32_t top_k;            /* np_methods.py:137 (400); <= RON_MAX_TOPK                      */
  float bbox_img[4];        /* clip reference + resize box (notebook cell 8): [0,0,1,1]      */
  float prior_scaling[4];   /* np_methods.py:25: [0.1,0.1,0.2,0.2]                           */
  uint32_t input_flags;
} ron_post_cfg;

/* Fixed-capacity detection list per image; rows >= count[i] are zero. */
typedef struct {
  int32_t capacity;         /* rows per image, >= top_k                                      */
  int32_t* classes;         /* [N, capacity]                                                 */
  float* scores;            /* [N, capacity]                                                 */
  float* bboxes;            /* [N, capacity, 4]                                              */
  int32_t* anchor_index;    /* [N, capacity]  flat index into the concatenated anchor list   */
  int32_t* count;           /* [N]                                                           */
} ron_detections;

/* Bytes of device scratch ron_post_np needs for a batch of n images. */
int64_t ron_post_np_workspace_bytes(const ron_heads* heads, int n);

/*
 * np_methods pipeline on the device (one call per batch):
 *   [softmax + objectness gate]  nets/ron_vgg_320.py:572-576, eval_ron_network.py:227-229
 *   decode                        np_methods.py:23-53   (== ssd_common.py:448-474)
 *   select (score > thr, c >= 1)  np_methods.py:56-131
 *   clip to bbox_img              np_methods.py:153-164
 *   sort, keep top_k              np_methods.py:137-150 (order: score desc, position asc)
 *   greedy class-aware IoU NMS    np_methods.py:186-205, :229-242
 *   resize by bbox_img            np_methods.py:167-183
 * `out` receives the kept boxes; `sorted_out` (optional, may be NULL) the list after the
 * top_k cut and before NMS; `n_candidates` (optional) [N] the select count per image.
 */
int ron_post_np(const ron_heads* heads, int n, const ron_post_cfg* cfg,
                void* workspace, int64_t workspace_bytes,
                ron_detections* out, ron_detections* sorted_out, int32_t* n_candidates,
                void* stream);

/*
 * The last three steps alone, on explicit candidate lists (np_methods.bboxes_sort ->
 * bboxes_nms, np_methods.py:137-150, :229-242): classes [N, n_in] int32, scores [N, n_in],
 * bboxes [N, n_in, 4]; n_valid [N] (or NULL = n_in everywhere).  No clip / resize.
 * Scores may be any float: negative ones sort below positive ones, -0 ties with +0, NaN sorts
 * last (the order of np.argsort(-scores)); ties keep their input order.
 */
int ron_np_sort_nms(const int32_t* classes, const float* scores, const float* bboxes,
                    const int32_t* n_valid, int n, int n_in, int top_k, float nms_threshold,
                    void* workspace, int64_t workspace_bytes,
                    ron_detections* out, ron_detections* sorted_out, void* stream);
int64_t ron_np_sort_nms_workspace_bytes(int n, int n_in);

/* RONNet.bboxes_decode (nets/ron_vgg_320.py:188-195 -> ssd_common.py:448-498): one layer,
 * loc [N,H,W,A,4] raw -> out [N,H,W,A,4] (ymin,xmin,ymax,xmax). */
int ron_bboxes_decode_layer(const float* loc, int n, int feat_h, int feat_w, int num_anchors,
                            const float* anchor_y, const float* anchor_x,
                            const float* anchor_h, const float* anchor_w,
                            const float prior_scaling[4], float* out, void* stream);

/* slim.softmax over the last axis (nets/ron_vgg_320.py:572,574): x [rows, c] -> y [rows, c].
 * With pick >= 0 only channel `pick` is written: y [rows, 1] (objness_pred, :576). */
int ron_softmax_last(const float* x, int64_t rows, int c, int pick, float* y, void* stream);

/* RONNet.bboxes_filter_min (nets/ron_vgg_320.py:196-233) as an operator of its own: per list the rows with
 * w = xmax - xmin > minsize and h = ymax - ymin > minsize, in their order (tf.boolean_mask), zeros behind them
 * (tfe_tensors.pad_axis, tf_extended/tensors.py:59-86).
 *   scores [num_lists, rows], bboxes [num_lists, rows, 4] (ymin, xmin, ymax, xmax)  ->  out_scores [num_lists, out_rows],
 *   out_bboxes [num_lists, out_rows, 4] (out_rows >= rows; every row written), counts [num_lists] = rows that passed.
 * The reference returns max(count, top_k) rows of a list: the caller slices (ops.bboxes_filter_min).  (ron_post_tfe applies the
 * same filter inside detected_bboxes, ron_tfe_cfg.min_size.) */
int ron_bboxes_filter_min(const float* scores, const float* bboxes, int num_lists, int rows, float minsize,
                          float* out_scores, float* out_bboxes, int out_rows, int32_t* counts, void* stream);

/* Detection records for the multi-GPU exchange: one float32 tensor [n, capacity + 1, 7] per rank, rows
 * 0..capacity-1 = (class, score, ymin, xmin, ymax, xmax, anchor_index), zero padded past `count`; row `capacity` = the
 * count replicated.  The only thing that crosses xGMI: one RCCL all-gather of these. */
int ron_pack_records(const ron_detections* det, int n, float* records, void* stream);
/* ... and that all-gather (SURVEY.md 8e names ncclAllGather): every rank's `records` [n, capacity + 1, 7] into `gathered`
 * [world, n, capacity + 1, 7] on every rank, enqueued on `stream`.  nccl_comm: an ncclComm_t (rccl.h) the caller created with one
 * rank per GPU; RCCL is looked up at the first call (the host's own librccl, or one already loaded, or librccl.so.1) and is not a
 * link-time dependency of this library: RON_ERR_UNSUPPORTED when the process has none.  `records` may be the slice of `gathered`
 * that belongs to this rank (in place).  The reference has no inference data parallelism (eval_ron_network.py:93-94: batch 1, one
 * device); the Python host does the same exchange through torch.distributed (ron_tensorflow_amd/parallel.py). */
int ron_gather_records(const float* records, int n, int capacity, float* gathered, void* nccl_comm, void* stream);

/* ------------------------------------------------------------------------------------------
 * TF evaluation variant of the post-processing (what eval_ron_network.py:226-236 runs):
 *   tf_ssd_bboxes_select (ssd_common.py:504-589) -> tfe.bboxes_clip (bboxes.py:105-144)
 *   -> RONNet.bboxes_filter_min (ron_vgg_320.py:196-233) -> tfe.bboxes_sort (bboxes.py:60-101)
 *   -> tfe.bboxes_nms_batch (bboxes.py:173-234, :262-302).
 * Output: per class c = 1..C-1, keep_top_k rows, zero padded (tfe.pad_axis, tensors.py:59-86):
 *   scores [N, C-1, keep_top_k], bboxes [N, C-1, keep_top_k, 4].
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float objectness_thres;
  float select_threshold;
  float nms_threshold;
  int32_t top_k;            /* select_top_k (200), <= RON_MAX_TOPK */
  int32_t keep_top_k;       /* 100 */
  int32_t nms_mode;         /* 0 = 'min' (bboxes.py:207-208), 1 = 'union' (:205-206) */
  int32_t clip;             /* clipping_bbox given? */
  float clipping_bbox[4];
  float min_size;           /* bboxes_filter_min minsize (0.03); < 0 disables (SSD) */
  float prior_scaling[4];
  uint32_t input_flags;
} ron_tfe_cfg;

int64_t ron_post_tfe_workspace_bytes(const ron_heads* heads, int n);
int ron_post_tfe(const ron_heads* heads, int n, const ron_tfe_cfg* cfg,
                 void* workspace, int64_t workspace_bytes,
                 float* scores, float* bboxes, void* stream);

/* ------------------------------------------------------------------------------------------
 * ron_eval.py variant of the post-processing (the reference's per-image harness, ron_eval.py:111-206, :369-392, :466-477):
 *   flaten_predict: score[c] = objectness * class probability, label = argmax over all classes, kept when label > 0 and
 *   objectness > objectness_thres -> tfe.bboxes_clip(bbox_img) -> filter_boxes (sides > min_size, centre inside the image)
 *   -> tf_bboxes_nms: score > select_threshold, all classes together, greedy in score order, at most keep_top_k kept,
 *   overlap 'union' (what main() passes) or 'min' -> tfe.bboxes_resize(bbox_img).
 *   nms_mode | 2: tf_bboxes_nms_by_class_v1 instead (ron_eval.py:282-366, the variant behind the commented call of :474): a kept
 *   box suppresses boxes of its own label only; the first keep_top_k kept rows in score order are returned.
 *   nms_mode | 4: tf_bboxes_nms_by_class (ron_eval.py:212-280, the other variant of that commented call): every score COLUMN
 *   (objectness * probability of class c, the background column included) is a list of its own - rows with score[c] >
 *   select_threshold, sorted by score[c], greedy with at most keep_top_k picks; a row some list kept is returned with the largest of
 *   its kept scores and that score's class (lowest class among equals), rows in the flattened anchor order (NOT score order), at most
 *   num_classes * keep_top_k of them: `out` needs that capacity, and the workspace ron_post_eval_workspace_bytes_mode() gives.
 * min_sizes: device [n], filter_boxes' min_size of every image (max(1e-4, 0.03 * sqrt(h * w / (320 * 320)))).
 * Output: ron_detections (classes = labels), capacity >= keep_top_k, kept rows in score order, zero padded.
 * The 1024 highest scores that pass the filters are the NMS candidates (the reference considers all of them; with its
 * thresholds, 0.95 / 0.6, a few dozen pass).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float objectness_thres;   /* 0.95 */
  float select_threshold;   /* 0.6  */
  float nms_threshold;      /* 0.4  */
  int32_t keep_top_k;       /* nms_topk = 20 */
  int32_t nms_mode;         /* 1 = 'union', 0 = 'min'; + 2 = by class (tf_bboxes_nms_by_class_v1); + 4 = tf_bboxes_nms_by_class */
  float bbox_img[4];
  float prior_scaling[4];
  uint32_t input_flags;
} ron_eval_cfg;
int64_t ron_post_eval_workspace_bytes(const ron_heads* heads, int n);                      /* nms_mode 0 .. 3 */
int64_t ron_post_eval_workspace_bytes_mode(const ron_heads* heads, int n, int nms_mode);   /* any nms_mode */
int ron_post_eval(const ron_heads* heads, int n, const float* min_sizes, const ron_eval_cfg* cfg,
                  void* workspace, int64_t workspace_bytes, ron_detections* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation preprocessing: preprocess_for_eval with Resize.WARP_RESIZE
 * (preprocessing/ssd_vgg_preprocessing.py:358-425, tf_image.py:269-282): uint8 RGB -> float, minus the channel means,
 * TF1 bilinear resize (align_corners=False, no half-pixel centres) to [out_h, out_w].
 *   packed  : the n images' HWC uint8 bytes back to back (device)
 *   offsets : [n] byte offset of image i in `packed` (device, int64)
 *   hw      : [n, 2] height, width of image i (device)
 *   means   : [3] host floats (123, 117, 104)
 *   out     : [n, out_h, out_w, 3] float32 (device) -- the `images` argument of ron_forward
 * ---------------------------------------------------------------------------------------- */
int ron_preprocess_eval(const uint8_t* packed, const int64_t* offsets, const int32_t* hw, int n,
                        int out_h, int out_w, const float* means, float* out, void* stream);
/* The other resize modes of preprocess_for_eval (CENTRAL_CROP: tf_image.resize_image_bboxes_with_crop_or_pad,
 * tf_image.py:141-254; PAD_AND_RESIZE: ssd_vgg_preprocessing.py:392-405) as one geometry table (device, int32 [n, 8]):
 *   {crop_y, crop_x, crop_h, crop_w, pad_y, pad_x, resized_h, resized_w}: the crop window of the whitened image is resized to
 *   resized_h x resized_w (TF1 bilinear) and placed at (pad_y, pad_x) of the output; the rest of the output is 0. */
int ron_preprocess_eval_geom(const uint8_t* packed, const int64_t* offsets, const int32_t* hw, const int32_t* geom, int n,
                             int out_h, int out_w, const float* means, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Evaluation bookkeeping: tfe.bboxes_matching_batch (tf_extended/bboxes.py:316-450) on the dense output of
 * ron_post_tfe.  List (i, c) holds class label c + 1 of image i:
 *   scores [N, L, K] (unused by the matching itself, kept for the reference's argument list), bboxes [N, L, K, 4],
 *   glabels [N, G] (0 = padding), gbboxes [N, G, 4], gdifficults [N, G] (non-zero = difficult), G <= RON_MAX_GT
 *   -> n_gbboxes [N, L] (non-difficult ground truth of the class), tp / fp [N, L, K] (0 / 1).
 * ---------------------------------------------------------------------------------------- */
int ron_bboxes_matching(const float* scores, const float* bboxes, int n, int num_lists, int k,
                        const int32_t* glabels, const float* gbboxes, const uint8_t* gdifficults, int g,
                        float matching_threshold, int32_t* n_gbboxes, uint8_t* tp, uint8_t* fp, void* stream);

/* ------------------------------------------------------------------------------------------
 * Conv stack.  Replaces RONNet.net / ron_net / ron_net_reducedfc
 * (nets/ron_vgg_320.py:136-154, :434-508, :510-580) with slim semantics of ron_arg_scope
 * (:595-629).  Weights enter by TF variable name (SURVEY.md 8b "weight contract").
 * ---------------------------------------------------------------------------------------- */
typedef struct ron_ctx ron_ctx;

typedef struct {
  int32_t variant;          /* ron_variant: which body `net` builds                          */
  int32_t dtype;            /* ron_dtype: arithmetic type of the conv stack                  */
  int32_t img_h, img_w;     /* 320, 320                                                      */
  int32_t num_classes;      /* 21                                                            */
  int32_t max_batch;        /* workspace is sized for this many images per call              */
  int32_t device;           /* HIP device ordinal                                            */
  uint32_t flags;           /* RON_CFG_*                                                     */
} ron_config;
/* Do not materialise block1..block3 (conv1_2, conv2_2, conv3_3 at full resolution): their 2x2 max-pool is
 * fused into the conv epilogue.  ron_end_point_copy then fails for those names. */
#define RON_CFG_FUSE_POOLS 1u
/* Run the head branches of the three coarse scales (block7/6/5: small grids) on internal side streams beside the
 * main chain; ron_forward / ron_detect fork after each reference map and join before returning control to `stream`. */
#define RON_CFG_MULTI_STREAM 2u
/* With RON_CFG_FUSE_POOLS (bf16 / f16) conv1_1 + conv1_2 + pool1 run as ONE kernel that goes from the fp32 image to pool1
 * (neither 64-channel full-resolution map touches HBM).  This flag keeps them as separate launches (A/B tests). */
#define RON_CFG_NO_STEM2 4u
/* The small, mutually independent head convolutions of the coarse scales (block7 / block6 and the 1x1 / skinny ones of
 * block5) run as grouped launches, several convolutions per launch (33 head launches -> 16); this flag keeps one launch
 * per convolution (same results up to the order of the fp32 partial sums: the split of K differs; tests compare the two). */
#define RON_CFG_NO_GROUPS 8u
/* Small maps with large filters spend a good part of their MACs on the zero halo (fc6: 7x7 on 10 x 10, 31 %; conv6 of SSD-512;
 * the 3x3 heads of the 5 x 5 / 10 x 10 scales).  By default such launches order their GEMM rows by output row first (output row,
 * image, column) and every tile skips the filter rows that fall outside the image for all of its rows, walking the others from the
 * filter's centre row (only products with zeros go; the fp32 sums are taken in another order, so the last bit may differ).
 * This flag keeps the image-major order and the full K loop (tests compare the two). */
#define RON_CFG_NO_HALO_SKIP 16u
/* Which grouped plan the RON heads run: by default contexts with max_batch <= 12 launch the heads one launch per dependency
 * level (7 launches, the large convolutions grouped too: at such batches every launch is latency-bound); 13..23: mixed-width
 * level groups with the large layers on their own; >= 24: small convolutions packed into the partial rounds of 256 x 256-tile
 * launches (csrc/graph.cpp, plan_groups).  RON_CFG_LEVEL_GROUPS forces the level plan, RON_CFG_BATCH_GROUPS the plan max_batch
 * would select without the level rule (A/B tests; same results up to the order of the fp32 partial sums). */
#define RON_CFG_LEVEL_GROUPS 32u
#define RON_CFG_BATCH_GROUPS 64u

int ron_create(ron_ctx** out, const ron_config* cfg);
int ron_destroy(ron_ctx* ctx);
/* A second execution slot over the weights of `src` (finalized): own activations, head buffers, scratch and streams,
 * so that several batches can be in flight on different streams (a serving loop keeps the GPU's 256 CUs busy across the
 * partial last round of workgroups every launch ends with).  Destroy the slots before the context that owns the weights. */
int ron_clone(ron_ctx* src, ron_ctx** out);

/* Number of variables the graph expects and the i-th name/shape ("ron_320_vgg/conv1/conv1_1/weights",
 * HWIO for conv, [kh,kw,Cout,Cin] for deconv, as TF stores them). */
int ron_num_variables(const ron_ctx* ctx);
int ron_variable_info(const ron_ctx* ctx, int i, const char** name, int64_t shape[4], int* ndim);
/* Host fp32 data for one variable (copied). */
int ron_load_weight(ron_ctx* ctx, const char* tf_name, const float* host_ptr,
                    const int64_t* shape, int ndim);
/* Fold BatchNorm (eps 1e-5), fuse parallel branches, repack to the kernel layout, cast, upload. */
int ron_finalize_weights(ron_ctx* ctx);

/* Filled by ron_forward: device fp32 tensors owned by the CALLER (ron_heads.cls/obj/loc must
 * point at writable buffers of the right size; anchor_* are filled in from the ctx). */
int ron_heads_describe(const ron_ctx* ctx, ron_heads* heads);   /* shapes + ctx anchor pointers */
int ron_forward(ron_ctx* ctx, const float* d_images, int n, ron_heads* d_out, void* stream);

/* Copy an end_point (nets/ron_vgg_320.py:455-483: block1..block7, plus every internal
 * activation by layer name) as dense fp32 NHWC into `d_out`; shape returned in nhwc. */
int ron_end_point_shape(const ron_ctx* ctx, const char* name, int n, int64_t nhwc[4]);
int ron_end_point_copy(ron_ctx* ctx, const char* name, int n, float* d_out, void* stream);

/* Fused: forward + np_methods post-processing, the graded path (config 2/3 of BASELINE.json). */
int ron_detect(ron_ctx* ctx, const float* d_images, int n, const ron_post_cfg* cfg,
               ron_detections* out, void* stream);

/* Algorithmic work of one image through the conv stack (2*MAC of conv + deconv). */
double ron_flops_per_image(const ron_ctx* ctx);

/* Per-launch timing with HIP events on the caller's stream (what bench.py's roofline uses; the reference's
 * only timing is wall-clock prints, eval_ron_network.py:353,363-366).  ron_profile_enable(ctx, n) makes the next n
 * ron_forward / ron_detect calls record one event per launch on the launch's stream (n = 0: off); ron_profile_get synchronises on the recorded events and
 * returns, for launch i (the last index is the post-processing stage of ron_detect), its name, whether it is
 * the implicit-GEMM conv kernel, its algorithmic FLOPs per image, the accumulated time and launch count, and its
 * algorithmic HBM bytes (activations in + out per image; packed weights once per launch).  A grouped launch (several small
 * head convolutions in one launch) is reported on its first member as "group[first+N]", the other members as "(name)" with
 * no work of their own.  *name points into the context and stays valid until ron_destroy. */
int ron_profile_enable(ron_ctx* ctx, int enable);
int ron_profile_num_ops(const ron_ctx* ctx);
/* Grouped launches in the plan of this context: RON-320 10 (max_batch >= 24), 7 (13..23), 6 (the level plan, <= 12 or
 * RON_CFG_LEVEL_GROUPS); SSD-512 5; 0 with RON_CFG_NO_GROUPS / RON_CFG_MULTI_STREAM. */
int ron_num_grouped_launches(const ron_ctx* ctx);
int ron_profile_get(ron_ctx* ctx, int i, const char** name, int* is_conv, double* flops_per_image,
                    double* total_ms, int* launches, double* act_bytes_per_image, double* weight_bytes);
int ron_profile_reset(ron_ctx* ctx);

/* ------------------------------------------------------------------------------------------
 * Single operators (used by the parity tests to pin each kernel against the oracle).
 * conv2d NHWC: x [n,h,w,cin] fp32 (device), w HWIO fp32 [kh,kw,cin,cout] and bias [cout] or NULL (HOST
 * pointers: they are packed on the host like ron_finalize_weights does), residual [n,ho,wo,cout] (device)
 * or NULL: y = act(conv + bias) ; if residual: y = relu(y + residual).  These two calls allocate scratch
 * and synchronise the stream (test / tooling use, not for graph capture).
 * `dtype` selects the arithmetic (operands rounded to bf16/f16, fp32 accumulate).
 * transpose != 0: slim.conv2d_transpose with kernel = stride (weights [kh,kw,cout,cin]).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t n, h, w, cin, cout;
  int32_t kh, kw, stride, dilation;
  int32_t relu;
  int32_t transpose;
  int32_t dtype;            /* ron_dtype */
  int32_t tile_cfg;         /* -1 = by shape (what the graph does), else one of the ron_conv_num_tile_cfgs() selectable
                               configurations (csrc/conv_mfma.h kCfg*: 0-3 and 7 row-gather tiles, 4-6 halo-patch N tiles,
                               8 the resident-weight 3x3 kernel for 64-channel maps);
                               anything else, or one that does not cover this conv, is RON_ERR_INVALID             */
  int32_t in_cstride;       /* tooling: input laid out as a channel slice: elements per pixel (0 = cin) ...   */
  int32_t in_coff;          /* ... and first channel of the slice                                              */
  int32_t pool;             /* fuse a 2x2 stride-2 max-pool into the epilogue: y is [n, h/2, w/2, cout]        */
  int32_t splitk;           /* split-K factor: -1 = by grid size, 1 = off                                      */
  int32_t center_from;      /* > 0: output channels >= center_from have weights in the centre tap only (the caller's w is
                             * zero elsewhere): their column tiles run that tap's K steps alone.  0 = none              */
} ron_conv_desc;
int ron_conv2d_nhwc(const ron_conv_desc* d, const float* x, const float* w, const float* bias,
                    const float* residual, float* y, void* stream);
/* How ron_conv2d_nhwc would run `d`, without running it (tests, tools): out = {tile configuration (csrc/conv_mfma.h kCfg*),
 * split-K factor, tile order (0 = column tiles fastest inside an XCD's run, 1 = row tiles fastest, P >= 2 = panels of P column tiles
 * walked row by row), K order (1 = taps innermost)}.  Allocates and frees the operands like ron_conv2d_nhwc does. */
int ron_conv_plan(const ron_conv_desc* d, int32_t out[4]);
/* Two fp32 head tensors from one convolution over a shared input - the class and the box convolution of an SSD feature layer
 * (nets/ssd_vgg_300.py:403-431), which the SSD-512 graph runs as one launch: w HWIO [kh,kw,cin,cout], its first `split_first`
 * output channels go to y_first [n,h,w,split_first], the other cout - split_first to y_second [n,h,w,cout - split_first]
 * (both fp32, device).  d: a plain stride-1 convolution (no transpose / pool); tile_cfg and splitk select the launch as in
 * ron_conv2d_nhwc. */
int ron_conv2d_heads_nhwc(const ron_conv_desc* d, int split_first, const float* x, const float* w, const float* bias,
                          float* y_first, float* y_second, void* stream);
int ron_maxpool2x2_nhwc(const float* x, int n, int h, int w, int c, int dtype, float* y, void* stream);
/* Tooling: time the conv kernel alone on random data (ms per launch, HIP events, default stream, synchronises). */
int ron_conv2d_bench(const ron_conv_desc* d, int warmup, int iters, float* ms_per_launch);
int ron_conv_num_tile_cfgs(void);

#ifdef __cplusplus
}
#endif
#endif /* RON_HIP_H_ */
