/*
 * y3.h -- C ABI of the MI355X-native YOLOv3 inference path (liby3hip.so).
 *
 * The reference (ronen-halevy/yolo-v3-tf2) has no FFI of its own: its operator surface is
 * three Python callables and one Keras Layer that hand tensors to the TensorFlow runtime.
 * This header declares the entry points a binding for that surface calls instead; each one
 * cites the reference interface it replaces.  The ctypes binding the package itself uses is
 * yolo-v3-tf2_amd/_lib.py; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns y3_status (0 = OK, <0 = error) and never
 *     throws or aborts: each entry point that can reach a host allocation is a function-try-block, so a failed allocation
 *     comes back as Y3_ERR_OOM (any other C++ exception as Y3_ERR_INTERNAL) with a message, not as a terminated host process.
 *     y3_last_error() returns a thread-local message for the last failure (a fixed buffer: reporting allocates nothing).
 *   - "dev" pointers are device (HBM) addresses owned by the caller (e.g. torch `data_ptr()`);
 *     "host" pointers are ordinary host memory, copied during the call.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises the device unless stated.  A y3_net is bound to the device that was current
 *     at creation and is not thread-safe.  Calls on a net (plan, weights, forward, detect, measure) run on the
 *     net's device whatever the caller's current device is and leave the caller's current device as they found it.
 *   - tensors are NHWC fp32 unless stated; boxes are normalised (xmin,ymin,xmax,ymax).
 */
#ifndef Y3_H
#define Y3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int y3_status;
enum {
    Y3_OK = 0,
    Y3_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    Y3_ERR_HIP = -2,       /* HIP runtime error */
    Y3_ERR_OOM = -3,       /* device allocation failed */
    Y3_ERR_STATE = -4,     /* call order (weights missing, plan missing, ...) */
    Y3_ERR_NODEVICE = -5,  /* no usable GPU */
    Y3_ERR_COMM = -6,      /* RCCL error (or librccl not loadable) */
    Y3_ERR_INTERNAL = -7   /* a C++ exception other than std::bad_alloc (reported as Y3_ERR_OOM) was stopped at the boundary */
};

/* activation storage / MFMA input type of the conv stack */
enum { Y3_DTYPE_F32 = 0, Y3_DTYPE_BF16 = 1, Y3_DTYPE_F32X3 = 2, Y3_DTYPE_F32X2 = 3 };

int y3_version(void);
const char *y3_last_error(void);
/* number of visible HIP devices (0 when none); never fails */
int y3_device_count(void);
/* 1 when tile id `tile` of the conv kernel family of `dtype` (Y3_DTYPE_*) exists in this library, else 0.  Tile ids are stable
 * (tuning tables refer to them); the ids of tiles that were measured and lost in rounds 1-4 (timing-only ablations, the
 * stream-K schedule, residual prefetch, the pipelined bf16 tile, bf16 tap-row reuse and the four-wave 256x256 bf16 tile;
 * DESIGN.md section 4, records under profiles/) are retired and answer 0.  Every id that answers 1 is named by a packaged
 * tuning table or chosen by the library's heuristic for some conv (tests/test_abi.py). */
int y3_tile_built(int dtype, int tile);

/* ------------------------------------------------------------------------------------------
 * Network = the fused conv program.
 * Replaces: the Keras Model built by ParseModel.build_model (reference: core/parse_model.py:279-314)
 * and executed by model(inputs) / model.predict (reference: inference.py:109,125,161).
 * One y3_conv_desc is one launch: Conv2D [+BatchNormalization] [+LeakyReLU(0.1)] [+Add] with the
 * optional UpSampling2D(2)+Concatenate feeding a 1x1 conv read in place
 * (reference: core/parse_model.py:13-56,59-75,102-160).
 * ---------------------------------------------------------------------------------------- */
typedef struct y3_net y3_net;

typedef struct {
    int32_t size;          /* kernel size: 1 or 3 */
    int32_t stride;        /* 1, or 2 (3x3 only: top/left zero pad then 'valid') */
    int32_t cin;           /* total input channels */
    int32_t cout;
    int32_t bn;            /* 1: BatchNormalization(eps) after the conv, no bias; 0: bias */
    int32_t leaky;         /* 1: LeakyReLU(alpha=0.1) */
    int32_t src0;          /* tensor id of input channels [0, c0) */
    int32_t src0_upsample; /* 1: src0 is half resolution and read through nearest x2 up-sampling */
    int32_t c0;            /* channels taken from src0 (== cin when src1 < 0) */
    int32_t src1;          /* -1, or tensor id of input channels [c0, cin) */
    int32_t residual;      /* -1, or tensor id added after the activation (shortcut) */
    int32_t dst;           /* tensor id written */
    int32_t in_div;        /* input spatial size  = image_size / in_div  (after up-sampling) */
    int32_t out_div;       /* output spatial size = image_size / out_div */
} y3_conv_desc;

/* auxiliary ops that the lowering could not fold into a conv (not used by YOLOv3 itself) */
enum { Y3_AUX_ADD = 0, Y3_AUX_UPSAMPLE2X = 1, Y3_AUX_CONCAT = 2 };
typedef struct {
    int32_t kind;
    int32_t src0;
    int32_t src1; /* -1 for upsample */
    int32_t dst;
} y3_aux_desc;

typedef struct {
    int32_t channels;
    int32_t div; /* spatial size = image_size / div */
} y3_tensor_desc;

/* op_kinds[i] == 0: take the next y3_conv_desc; == 1: take the next y3_aux_desc (execution order).
 * tensors[0..n_tensors) describes every tensor id; `input_tensor` is the image, `outputs[3]` the head
 * outputs in model order (coarsest grid first), each with 3*(5+nclasses) channels. */
y3_status y3_net_create(const y3_tensor_desc *tensors, int n_tensors, const int32_t *op_kinds, int n_ops,
                        const y3_conv_desc *convs, int n_convs, const y3_aux_desc *aux, int n_aux,
                        int input_tensor, const int32_t outputs[3], int nclasses, y3_net **out);
void y3_net_destroy(y3_net *net);

/* Weights of conv `conv_slot` (index into the `convs` array given at creation), host pointers.
 * w is HWIO [size,size,cin,cout] (the Keras Conv2D kernel layout, reference: convert.py:61-68).
 * bn convs: gamma/beta/mean/var [cout], eps (Keras default 1e-3); bias convs: bias [cout], others NULL.
 * Replaces model.load_weights(...) (reference: inference.py:102). */
y3_status y3_net_set_conv_weights(y3_net *net, int conv_slot, const float *w, const float *gamma,
                                  const float *beta, const float *mean, const float *var, const float *bias,
                                  float eps);

/* Tuning/testing knobs (no reference counterpart).
 * y3_net_set_tile: force the block tile of one conv (index into the kernel's tile table; -1 = heuristic).
 * A tile that cannot serve the conv (shape, or a bf16-only tile on a conv that writes an fp32 net output) is refused here with
 * Y3_ERR_INVALID and a message, not by the forward.
 * y3_net_keep_activations(1) before y3_net_plan: no buffer reuse, so y3_net_read_tensor can read any
 * intermediate after a forward. */
y3_status y3_net_set_tile(y3_net *net, int conv_slot, int tile);
y3_status y3_net_set_tile_bf16(y3_net *net, int conv_slot, int tile);
y3_status y3_net_set_tile_x3(y3_net *net, int conv_slot, int tile);
y3_status y3_net_set_tile_x2(y3_net *net, int conv_slot, int tile);   /* same tile table as _x3; a subset is built */
/* Run a forward as `lanes` (1..4) equal sub-batches: the first on the caller's stream itself, the others on forked
 * internal streams joined back into it: the tail of one sub-batch's conv kernel overlaps the next kernel of another.
 * Results are unchanged (images are independent).  Falls back to fewer lanes when the batch is not divisible. */
y3_status y3_net_set_lanes(y3_net *net, int lanes);
/* Placement of the fp32 conv tiles on the 8 XCDs (each has a private 4 MB L2).  1 (default): per conv, the XCDs form an
 * (8/gn) x gn grid over the (pixel-tile, channel-tile) matrix, gn chosen so that an XCD's slice of the weights stays in
 * its L2 (the 256->512 / 512->1024 3x3 weights are 4.7 / 18.9 MB); 0: every XCD takes a contiguous run of tiles.
 * Results are bit-identical in both modes (same per-tile arithmetic). */
y3_status y3_net_set_xcd_mode(y3_net *net, int mode);
/* K order of the fp32 3x3 convs.  channels > 0 (a multiple of 32): for every chunk of that many input channels all 9 taps,
 * then the next chunk, so that the 9 reads of a pixel (one per tap) fall close together in time and hit in L2 (the tap-
 * major order re-fetched the operands of the 13x13 layers ~18x from beyond L2).  The same products in another summation
 * order: results differ from the tap-major ones in the last bits.  0: tap-major; -1 (default): chosen per conv. */
y3_status y3_net_set_k_chunk(y3_net *net, int channels);
y3_status y3_net_keep_activations(y3_net *net, int keep);
/* 1 (default): when the program starts with conv0 (3x3/1, 3 -> 32) feeding only conv1 (3x3/2, 32 -> 64) -- the Darknet-53
 * stem, reference config/models/yolov3/backbone.yaml layers 1-2 -- and the plan is fp32 or bf16 without keep_activations,
 * the two run as ONE kernel that keeps conv0's output (the largest tensor of the network, 1.4 GB at 64 x 416^2) in LDS; the
 * 1x1 conv that follows (64 -> 32, backbone.yaml layer 3) is computed by the same kernel from conv1's tile.
 * 2: conv0 + conv1 in one kernel, the 1x1 conv as its own launch (bf16 plans: bit-identical to 1).
 * 0: one launch per conv.  fp32 plans: results agree to fp32 rounding (conv0's summation order differs between the two
 * kernels).  bf16 plans: the fused kernel forms conv0's products on the bf16 matrix cores from operands split x = hi + lo
 * (hi*hi + hi*lo + lo*hi, fp32 accumulation: ~2^-16 relative error per product, NOT fp32 arithmetic) and rounds the result to
 * bf16 where the pipeline stores it; against the one-launch-per-conv form a fraction of a percent of conv0's bf16 values round
 * the other way (bounded by tests/test_gpu_parity.py::test_fused_stem_bf16_matches_oracle_and_the_two_launch_form). */
y3_status y3_net_set_stem_fusion(y3_net *net, int on);
/* Measurement aid (bench.py): the shader clock the chip holds under this network's load.  Runs `forwards` forwards back to
 * back (grids_dev as for y3_net_forward); in the last one, thread 0 of the middle workgroup of the conv with the most FLOPs (fp32
 * plans: an MFMA conv launch; bf16 plans: the fused stem kernel) reads s_memtime and s_memrealtime at its entry and after
 * its epilogue: MHz = d(memtime) / d(memrealtime) x 100 (MI355X_MICROARCH.md, DVFS give-back item 6).  Synchronises the
 * stream.  No product launch carries stamps (the kernels test a null pointer).  The three measure_sclk calls temporarily
 * change the net's lane count and stamp fields: never run them concurrently with a forward on the same net. */
y3_status y3_net_measure_sclk(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                              float *mhz_out, void *stream);
/* The same measurement on the launch of conv slot `conv` (tools/sclk_per_layer.py: the clock differs from layer to layer
 * with the power each one draws).  Y3_ERR_STATE when that conv's kernel carries no stamps in this plan. */
y3_status y3_net_measure_sclk_conv(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                                   int conv, float *mhz_out, void *stream);
/* ... and on every conv launch of ONE forward (the last of `forwards`): mhz_out[n_convs] (0 where the conv's kernel
 * carries no stamps or runs inside another launch), and -- when not NULL -- start_us / end_us [n_convs]: when the
 * first workgroup of each launch began and when its clock-stamped (middle) workgroup ended on the chip's 100 MHz
 * real-time counter, relative to the earliest stamp: the timeline of the conv stack with no host event in it; the
 * difference of consecutive starts is a launch's duration in the real forward (bench.py: time-weighted clock). */
y3_status y3_net_measure_sclk_all(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], int forwards,
                                  float *mhz_out, double *start_us, double *end_us, void *stream);
/* Before y3_net_plan: run the first n_convs convs chunk_images images at a time, then the rest of the network on the
 * whole batch.  The first layers' activations are the largest tensors of the network (1.4 GB for 64 images at 416x416);
 * in chunks they are still in the Infinity Cache when the next conv reads them.  Results are unchanged (images are
 * independent).  0, 0 switches it off. */
y3_status y3_net_set_early_chunk(y3_net *net, int n_convs, int chunk_images);

/* Allocate the activation arena for batches up to `max_batch` of image_size x image_size inputs
 * and select the conv arithmetic type (Y3_DTYPE_*).  May be called again to re-plan.
 * Y3_DTYPE_BF16: intermediate activations and weights are bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulate and
 * epilogue); the image batch stays fp32 (the Cin = 3 first layer reads it directly) and the head grids stay fp32.
 * Y3_DTYPE_F32X3: fp32-accurate arithmetic on the bf16 matrix cores -- every value is held as three bf16 planes
 * (x = hi + mid + lo exactly) and each product uses its six leading partial products with fp32 accumulation; same
 * image / head-grid conventions, same parity bar as Y3_DTYPE_F32.
 * Y3_DTYPE_F32X2: the same idea on the fp16 matrix cores with two planes per value, x = h + l' * 2^-11
 * (h = fp16(x), l' = fp16((x - h) * 2^11)), three partial products per product (h*h, h*l', l'*h) in two fp32
 * accumulators: representation error 2^-22 |x| (fp32: 2^-24), half the MFMAs of F32X3.  Values must stay inside the
 * fp16 range: BN-scaled weights are checked (|w| < 65504: y3_net_plan / y3_net_forward refuse the mode otherwise),
 * activations are not checked.
 * In the three non-fp32 modes an output tensor that another op of the net reads again, or that a residual conv writes,
 * is kept in the arena in the mode's format and converted into the caller's fp32 buffer at the end of the forward. */
y3_status y3_net_plan(y3_net *net, int max_batch, int image_size, int dtype);

/* images_dev [B,S,S,3] fp32 -> grids_dev[3], each [B,g,g,3*(5+nc)] fp32 (== [B,g,g,3,5+nc]).
 * Replaces model(inputs) (reference: inference.py:109). */
y3_status y3_net_forward(y3_net *net, const float *images_dev, int batch, float *const grids_dev[3], void *stream);

/* debugging / tests: copy tensor `tensor_id` of the last forward to dst_dev as fp32 (element count returned
 * through *n_elems; dst_dev may be NULL to query the size) */
y3_status y3_net_read_tensor(y3_net *net, int tensor_id, int batch, float *dst_dev, size_t *n_elems, void *stream);

/* conv FLOPs (2*MAC) of one image at the planned size */
double y3_net_flops_per_image(const y3_net *net);

/* Time every conv launch of one forward (hipEvents on `stream`); ms_out[n_convs]. Synchronises. */
y3_status y3_net_profile_convs(y3_net *net, const float *images_dev, int batch, float *ms_out, int n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Image input stage (the step right before the path): decode_image(channels=3, dtype=float32) + tf.image.resize
 * (reference: inference.py:157-158).  image_dev: [height,width,channels] uint8 (is_uint8=1; converted with * 1/255) or
 * float32 (is_uint8=0); channels 3 or 4 (alpha dropped).  is_uint8=2 is the tfrecords source's order of operations
 * (reference: core/load_tfrecords.py:46-48): uint8 taken as 0..255, resized, then divided by 255.  Writes the bilinear (half-pixel centres, no antialias)
 * resize to image_size x image_size into batch_dev[slot] of an NHWC fp32 batch [*,image_size,image_size,3].
 * ---------------------------------------------------------------------------------------- */
y3_status y3_preprocess_image(const void *image_dev, int is_uint8, int height, int width, int channels,
                              float *batch_dev, int slot, int image_size, void *stream);

/* ------------------------------------------------------------------------------------------
 * yolo_decode   (reference: core/yolo_decode_layer.py:15-36)
 * grids_dev[s]: [B,g_s,g_s,3,5+nc]; anchors_host: [3][3][2] normalised (w,h), scale s uses anchors[s].
 * Outputs [B,N,4], [B,N,1], [B,N,nc] with N = 3*sum g_s^2, scales concatenated in input order.
 * ---------------------------------------------------------------------------------------- */
y3_status y3_yolo_decode(const float *const grids_dev[3], const int32_t grid_sizes[3], int batch, int nclasses,
                         const float *anchors_host, float *bboxes_dev, float *conf_dev, float *probs_dev,
                         void *stream);

/* decode fused with the class arg-max / score of yolo_nms (reference: core/yolo_nms.py:18-24):
 * writes bboxes [B,N,4], class_indices [B,N] int64, scores [B,N]; class probabilities are not stored. */
y3_status y3_yolo_decode_scores(const float *const grids_dev[3], const int32_t grid_sizes[3], int batch,
                                int nclasses, const float *anchors_host, float *bboxes_dev,
                                int64_t *class_idx_dev, float *scores_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * yolo_nms / YoloNmsLayer.call   (reference: core/yolo_nms.py:16-34, core/yolo_nms_layer.py:26-29)
 * ---------------------------------------------------------------------------------------- */
/* class_indices = argmax(probs), scores = conf * max(probs)   (core/yolo_nms.py:18-24) */
y3_status y3_class_scores(const float *conf_dev, const float *probs_dev, int batch, int n, int nclasses,
                          int64_t *class_idx_dev, float *scores_dev, void *stream);

/* bytes of scratch y3_nms_padded needs for (batch, n) */
size_t y3_nms_workspace_bytes(int batch, int n);

/* tf.image.non_max_suppression_padded(boxes[B,N,4], scores[B,N], max_output_size, iou_threshold,
 * score_threshold, pad_to_max_output_size=True)   (core/yolo_nms.py:26-33)
 * -> selected_idx [B,max_output_size] int32 (zero padded), num_valid [B] int32. */
y3_status y3_nms_padded(const float *bboxes_dev, const float *scores_dev, int batch, int n, int max_output_size,
                        float iou_threshold, float score_threshold, int32_t *selected_idx_dev,
                        int32_t *num_valid_dev, void *workspace_dev, size_t workspace_bytes, void *stream);

/* Inference.gather_valid_detections_results for a whole batch (reference: inference.py:21-28), packed for the
 * multi-GPU exchange: per image max_out rows of {box[4], score, class (int32), index (int32)} = 7 x 4 bytes
 * (rows >= num_valid zeroed), i.e. packed_dev is [B,max_out,7] of 32-bit words. */
y3_status y3_pack_detections(const float *bboxes_dev, const int64_t *class_idx_dev, const float *scores_dev,
                             const int32_t *selected_idx_dev, const int32_t *num_valid_dev, int batch, int n,
                             int max_out, void *packed_dev, void *stream);

/* Conv program + decode in one call: images -> (bboxes [B,N,4], class_indices [B,N] int64, scores [B,N]), i.e.
 * model(inputs) -> yolo_decode -> argmax / score (reference: inference.py:109-117 up to the NMS; core/yolo_decode_layer.py:15-36,
 * core/yolo_nms.py:18-24).  Where the graph allows it (every head a 1x1 conv + bias writing its grid; fp32 and bf16 plans) the
 * three head convs decode their own output tiles while these are still on chip, and the [B,g,g,3*(5+nc)] grids are neither
 * written nor read back; otherwise it is y3_net_forward into net-owned scratch + y3_yolo_decode_scores.  Either way the
 * results are bit-identical to that composed route.  Outputs are caller-owned device buffers; bboxes 16-byte aligned. */
y3_status y3_net_forward_decode(y3_net *net, const float *images_dev, int batch, const float *anchors_host, float *bboxes_dev,
                                int64_t *class_idx_dev, float *scores_dev, void *stream);
/* ------------------------------------------------------------------------------------------
 * The whole path in one call: Model(inputs, nms_output).predict(batch) followed by the per-image gather
 * (reference: inference.py:109-117, 125-128, 21-28) = y3_net_forward -> y3_yolo_decode_scores -> y3_nms_padded ->
 * y3_pack_detections on net-owned scratch.  images_dev [batch,S,S,3] fp32; anchors_host [3][3][2];
 * packed_dev [batch,max_boxes,7] 32-bit words {xmin,ymin,xmax,ymax,score,class(int32),index(int32)}, rows >= num_valid
 * zeroed; num_valid_dev [batch] int32.  max_boxes in [1,1024].  Everything is enqueued on `stream`: the scratch (grids,
 * decoded tensors, NMS workspace) is allocated by y3_net_plan for max_batch images, so the call allocates nothing and
 * may be the first thing captured into a HIP graph.
 * ---------------------------------------------------------------------------------------- */
y3_status y3_net_detect(y3_net *net, const float *images_dev, int batch, const float *anchors_host, int max_boxes,
                        float iou_threshold, float score_threshold, void *packed_dev, int32_t *num_valid_dev,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU exchange (no reference counterpart: the reference is single-device).
 * One process per GPU; images are sharded by rank and are independent end to end, so the only collective of the path
 * is the all-gather of the packed final detections (north_star: "RCCL all-gather of the final box list over xGMI").
 *   y3_comm_get_unique_id   rank 0 draws an id (Y3_COMM_ID_BYTES bytes, host) and hands it to the other ranks by any
 *                           out-of-band means (the Python host uses the torch.distributed store / broadcast)
 *   y3_comm_init_rank       every rank, on its own current device: joins the communicator (ncclCommInitRank)
 *   y3_allgather_results    packed_dev [batch,max_boxes,7] 32-bit words + num_valid_dev [batch] of this rank ->
 *                           packed_all_dev [world*batch,max_boxes,7], num_valid_all_dev [world*batch] in rank order;
 *                           equal batch on every rank; both gathers form ONE RCCL group enqueued on `stream`
 *                           (capturable into the same HIP graph as y3_net_detect).
 * ---------------------------------------------------------------------------------------- */
typedef struct y3_comm y3_comm;
#define Y3_COMM_ID_BYTES 128
y3_status y3_comm_get_unique_id(void *id_out_host);
y3_status y3_comm_init_rank(const void *id_host, int world_size, int rank, y3_comm **out);
void y3_comm_destroy(y3_comm *comm);
y3_status y3_comm_info(const y3_comm *comm, int32_t *world_size, int32_t *rank);
y3_status y3_allgather_results(y3_comm *comm, const void *packed_dev, const int32_t *num_valid_dev, int batch,
                               int max_boxes, void *packed_all_dev, int32_t *num_valid_all_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * TFRecord framing checksum (host): CRC-32C (Castagnoli) of a host buffer, unmasked.  The tfrecords input source
 * (reference: core/load_tfrecords.py:97-99, tf.data.TFRecordDataset) verifies it per record.
 * ---------------------------------------------------------------------------------------- */
uint32_t y3_crc32c(const void *data_host, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* Y3_H */
