/*
 * mtadgat.h -- C ABI of the MI355X-native MTAD-GAT per-window forward path.
 *
 * The reference (ML4ITS/mtad-gat-pytorch) is pure Python on PyTorch and has no
 * FFI of its own; the interface this library replaces is the Python call
 *
 *     predictions, recons = MTAD_GAT.forward(x)          reference mtad_gat.py:64-79
 *
 * and, stage by stage, the nn.Module.forward() methods it is made of
 * (reference modules.py, cited per entry point below).  The Python module
 * `mtad-gat-pytorch_amd/mtad_gat.py` mirrors the reference class on top of this
 * ABI through ctypes; any other host language binds the
 * same symbols.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a
 *     negative mtadgat_status; mtadgat_last_error() gives the message
 *     (thread-local).  Nothing throws across the boundary.
 *   - all tensors are float32, row-major contiguous, in the reference's own
 *     shapes -- with one exception: mtadgat_forward_xbf16 takes its input
 *     windows as bfloat16 (BASELINE config 2's "bf16 inference"); its outputs
 *     are float32 like everyone else's.  `*_dev` pointers are device (HBM)
 *     pointers owned by the caller, `*_host` pointers are host pointers.
 *   - every launch goes to the HIP stream the caller passes (a hipStream_t cast
 *     to void*; NULL = the null stream) and nothing synchronises the device.
 *   - the library keeps no per-call state besides the packed weights held by
 *     the handle; one handle may be used from one thread at a time.
 *   - gfx950 only.  There is no CPU implementation behind this ABI.
 */
#ifndef MTADGAT_H
#define MTADGAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTADGAT_ABI_VERSION 1
#define MTADGAT_MAX_LAYERS 8

typedef enum mtadgat_status {
    MTADGAT_OK = 0,
    MTADGAT_ERR_INVALID = -1,      /* bad argument / inconsistent shapes         */
    MTADGAT_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels cover       */
    MTADGAT_ERR_HIP = -3,          /* a HIP runtime call failed                  */
    MTADGAT_ERR_NOWEIGHTS = -4,    /* forward before mtadgat_load_weights        */
    MTADGAT_ERR_WORKSPACE = -5     /* workspace pointer NULL or too small        */
} mtadgat_status;

/* Model hyper-parameters: the constructor arguments of reference
 * MTAD_GAT.__init__ (mtad_gat.py:37-54) after the defaulting that
 * modules.py:36-63 / :137-164 apply. */
typedef struct mtadgat_config {
    int32_t n_features;       /* F  (mtad_gat.py:39)                                   */
    int32_t window_size;      /* W  (mtad_gat.py:40)                                   */
    int32_t out_dim;          /* mtad_gat.py:41                                        */
    int32_t kernel_size;      /* odd; ConvLayer pads (k-1)/2, modules.py:14            */
    int32_t use_gatv2;        /* 1: GATv2 (modules.py:74-77), 0: GAT (modules.py:80-83)*/
    int32_t feat_embed;       /* rows of feature_gat.lin.weight  (doubled already for v2, modules.py:47-50) */
    int32_t time_embed;       /* rows of temporal_gat.lin.weight (modules.py:148-151)  */
    int32_t gru_n_layers;     /* modules.py:233                                        */
    int32_t gru_hid_dim;
    int32_t forecast_n_linear;/* number of nn.Linear in Forecasting_Model = n_layers+1 (modules.py:297-301) */
    int32_t forecast_hid_dim;
    int32_t recon_n_layers;   /* modules.py:253                                        */
    int32_t recon_hid_dim;
    float   alpha;            /* LeakyReLU negative slope (modules.py:62,163)          */
} mtadgat_config;

/* The reference's parameters (its state_dict, mtad_gat.py:56-62), as host
 * pointers to float32 arrays in the reference's own shapes. */
typedef struct mtadgat_params {
    const float* conv_weight;      /* conv.conv.weight   (F, F, k)                     */
    const float* conv_bias;        /* conv.conv.bias     (F)                           */
    const float* feat_lin_weight;  /* feature_gat.lin.weight  v2 (E_f, 2W)  v1 (E_f, W)*/
    const float* feat_lin_bias;    /* feature_gat.lin.bias    (E_f)                    */
    const float* feat_a;           /* feature_gat.a           v2 (E_f,1)   v1 (2E_f,1) */
    const float* feat_bias;        /* feature_gat.bias        (F, F)                   */
    const float* temp_lin_weight;  /* temporal_gat.lin.weight v2 (E_t, 2F)  v1 (E_t, F)*/
    const float* temp_lin_bias;    /* temporal_gat.lin.bias   (E_t)                    */
    const float* temp_a;           /* temporal_gat.a          v2 (E_t,1)   v1 (2E_t,1) */
    const float* temp_bias;        /* temporal_gat.bias       (W, W)                   */
    /* gru.gru.{weight_ih,weight_hh,bias_ih,bias_hh}_l<i>; gate order r|z|n            */
    const float* gru_w_ih[MTADGAT_MAX_LAYERS];   /* (3H, 3F) for l0, (3H, H) after     */
    const float* gru_w_hh[MTADGAT_MAX_LAYERS];   /* (3H, H)                            */
    const float* gru_b_ih[MTADGAT_MAX_LAYERS];   /* (3H)                               */
    const float* gru_b_hh[MTADGAT_MAX_LAYERS];   /* (3H)                               */
    /* forecasting_model.layers.<i>.{weight,bias}                                      */
    const float* fc_weight[MTADGAT_MAX_LAYERS];
    const float* fc_bias[MTADGAT_MAX_LAYERS];
    /* recon_model.decoder.rnn.*_l<i>                                                  */
    const float* rec_w_ih[MTADGAT_MAX_LAYERS];   /* (3Hr, H) for l0, (3Hr, Hr) after   */
    const float* rec_w_hh[MTADGAT_MAX_LAYERS];
    const float* rec_b_ih[MTADGAT_MAX_LAYERS];
    const float* rec_b_hh[MTADGAT_MAX_LAYERS];
    const float* rec_fc_weight;    /* recon_model.fc.weight (out, Hr)                  */
    const float* rec_fc_bias;      /* recon_model.fc.bias   (out)                      */
} mtadgat_params;

typedef struct mtadgat_handle_s* mtadgat_handle;

int         mtadgat_abi_version(void);
const char* mtadgat_last_error(void);

/* Replaces MTAD_GAT.__init__ (mtad_gat.py:37-62): validates the configuration. */
int mtadgat_create(const mtadgat_config* cfg, mtadgat_handle* out);
int mtadgat_destroy(mtadgat_handle h);

/* Replaces load_state_dict / the in-place parameter update seen by forward()
 * (training.py:127, :248; utils.py:189): re-packs the parameters into the
 * kernels' tile format on the host and uploads them on `stream`.  Call again
 * whenever a parameter changed. */
int mtadgat_load_weights(mtadgat_handle h, const mtadgat_params* params_host, void* stream);

/* The same for parameters that are already in device memory -- the training loop's optimizer.step() followed by a
 * forward (training.py:127 -> :110): `flat_dev` holds all parameters back to back in the field order of
 * mtadgat_params (the order of the flat gradient buffer, mtadgat_grad_offsets / mtadgat_grad_floats), and the tile
 * image is rebuilt from it by kernels on `stream`; nothing travels to the host and nothing synchronises (the column order of
 * the folded GATv2 projection follows the signs of the attention vectors `a`: derived on the device since round 5, the kernels
 * read the sign-group boundaries from the image).  Requires one
 * earlier mtadgat_load_weights on this device and precision mode 0 or 2 (the fp32 image; the split-operand packs of mode 2
 * are re-derived from it on the device); mode 1 returns MTADGAT_ERR_UNSUPPORTED (callers then use
 * mtadgat_load_weights).  The bf16 weight streams are not maintained: mtadgat_bf16_ready turns 0. */
int mtadgat_update_weights_device(mtadgat_handle h, const float* flat_dev, int64_t n_floats, void* stream);

/* Bit-exact 64-bit checksum of a list of device tensors of 32-bit elements (the model's parameters), written to
 * *out_dev on `stream`: one launch.  The Python module compares it between calls to notice in-place parameter edits
 * that autograd's version counters do not see (p.data.mul_(), nn.init.*_(p.data)). */
int mtadgat_params_fingerprint(const void* const* tensors_dev, const int64_t* n_elements, int n_tensors, uint64_t* out_dev, void* stream);

/* Host-only self check of the gather table behind mtadgat_update_weights_device (needs no GPU): packs `params_host`
 * with the host packer, derives the table, and returns the number of image positions the table would fill with a value
 * different from the host packer's (0 = consistent; negative = error).  *n_gathered: positions the table covers. */
int64_t mtadgat_selfcheck_gather_table(mtadgat_handle h, const mtadgat_params* params_host, int64_t* n_gathered);

/* Diagnostic: copies the packed weight image (mtadgat_packed_floats floats) to host memory after synchronising
 * `stream` -- the tests compare the device-side re-pack with the host packer through it. */
int64_t mtadgat_packed_floats(mtadgat_handle h);
/* (offset, length) pairs in floats of the image regions that are derived on the device from other regions of the image
 * (the split-bf16 packs); returns their number (at most max_pairs are written) */
int mtadgat_derived_regions(mtadgat_handle h, int64_t* out_pairs, int max_pairs);
int mtadgat_read_packed(mtadgat_handle h, float* dst_host, int64_t n_floats, void* stream);

/* Arithmetic of the inference entry points (forward / forward_series / stage calls):
 *   0 (default of a new handle)  fp32 operands on the exact fp32 MFMA: <= 1e-5 of the reference's float32 forward
 *   2            fp32 results, the products of the large-batch kernels (k_gru above 16 384 windows, the attention
 *                projection above 4 096) formed from split 16-bit operands: every fp32 operand is the exact sum of three
 *                bf16 pieces (six MFMA terms with fp32 accumulation per product) or, where its range is bounded by
 *                construction -- recurrent state, attention outputs, weights scaled by a per-layer power of two --, of
 *                two fp16 pieces (three terms): within ~2e-7 of mode 0, same 1e-5 gate, 2.7-5x less matrix-pipe time
 *                on the pipe that runs beside the VALU (the Python module's default).
 *                The split weight packs are derived on the device by the first launch that reads them after a load
 *                (round 6; rounds 2-5: at every load); switching needs no reload.
 *   1            bf16 MFMA operands (weights packed to bf16 once per load_weights, activations rounded on the
 *                way into the matrix unit), fp32 accumulation, fp32 recurrent state / gates / softmax:
 *                <= 2e-2 of the fp32 reference on outputs of scale ~1 (BASELINE configs "bf16 inference").
 *                From 4 096 windows per chunk the convolution and the two attention layers run on mode 2's two-fp16-piece
 *                kernels instead (faster than their bf16 builds and closer to fp32); the recurrences and heads stay bf16.
 * The training entry points (mtadgat_forward_train / mtadgat_backward) compute in fp32 -- modes 0 and 2 -- and REFUSE mode 1
 * (MTADGAT_ERR_UNSUPPORTED): the bf16-operand recurrences of rounds 2-5 were slower than the fp32 step at every batch size and
 * were removed in round 6. */
int mtadgat_set_precision(mtadgat_handle h, int mode);
/* The bf16 weight streams are packed by mtadgat_load_weights only while mode 1 is selected (select first, or load
 * again after switching); 1 when they are present. */
int mtadgat_bf16_ready(mtadgat_handle h);

/* Testing / measurement hook (not needed for normal use).  "gru_kernel": which kernel runs the large-batch recurrences
 * (GRULayer.forward modules.py:235-238, the decoder modules.py:276-283) in precision mode 2:
 *   0 automatic (default), 1 the tile-major kernel at every batch size, 2 the chunk-major kernel wherever it applies,
 *   3 the hidden-tile-split kernel on split operands wherever it applies.
 * "gat_kernel": which kernel runs the fused attention layers (modules.py:65-95, :166-193) in precision mode 2:
 *   0 automatic (default: from 4096 windows per chunk the fp16-piece build k_gath of the row-split kernel when the convolution's
 *   outputs are below 2^15, else k_gat), 1 k_gat at every batch size, 3 k_gath at every batch size (2 was round 4's
 *   column-sliced kernel, removed: it lost to k_gath on every shipped shape).
 * "conv_kernel": the convolution of the fused front end (modules.py:18-22) in precision mode 2: 0 automatic (the
 *   window-per-workgroup kernel on fp16 pieces from 4096 windows per chunk), 1 k_conv_lds (fp32 MFMA), 2 k_conv_win at any size.
 * "conv_fused": 0 automatic -- in precision mode 2, wherever k_conv_win and k_gath both apply (from 4096 windows per chunk), the
 *   workgroup that runs the temporal attention layer on a window computes that window's convolution itself (mtad_gat.py:67-70 is
 *   one dataflow): no convolution launch, h_cat[:, :F] written once and not read back by that layer; the fp16 range guard is then
 *   per window.  1: always two launches.
 * "conv_shared": stride-1 series scoring in precision mode 2: 0 automatic (k_conv_win reads each window out of the series where it
 *   applies, the shared-row convolution of k_conv_lds otherwise), 1 the shared-row convolution wherever it applies.
 * ("series_band", the shared temporal pair scores of rounds 4-5, is gone: the option is refused.)
 * "rowgemm_kernel": the data-gradient products d X = d Y W of mtadgat_backward: 0 automatic (three bf16 pieces per operand from
 *   4096 rows in precision mode 2), 1 fp32 MFMA, 2 the split-bf16 build always.
 * "gemm_lds" (process-wide, not per handle): the split-bf16 row GEMMs and the wide models' convolution on launches of >= 131 072
 *   rows (Linear / Conv1d layers: modules.py:18-22, :76-81, :176-181; the data gradients of training.py:126): 0 automatic -- a
 *   workgroup of four waves owns 256 rows and shares each chunk's weight words through LDS; 1 the one-wave kernels everywhere
 *   (results are bit-identical either way).
 * "lanes": 0 automatic: mtadgat_forward / _forward_series walk calls of 8 193 .. 16 384 (two halves; 8 192 + the rest when the rest
 *   is at most 1 536 windows) and of more than 32 768 windows per chunk (whole 32 768-window pieces, then the rest) in
 *   pieces that alternate between `stream` and a second stream owned by the handle (each with its own half of the workspace;
 *   `stream` waits for the second lane before the call's work on it counts as complete, so the caller's ordering rules do not
 *   change); results = those of the call on each piece.  Models on the un-fused (wide) attention path alternate the CHUNKS of a call
 *   between the two streams (first piece: half a chunk).  1: everything on `stream`.
 * "gath_dbg": measurement hooks of k_gath (knock-outs: results invalid; sensitivity probes: results unchanged), csrc/mtadgat_kernels.h.
 * "wgrad_kernel": the weight-gradient GEMMs of mtadgat_backward (training.py:126): 0 automatic (three bf16 pieces per operand on
 *   the 16-bit matrix pipe in precision mode 2), 1 fp32 MFMA, 2 the split-bf16 build in every mode. */
int mtadgat_set_option(mtadgat_handle h, const char* name, int value);
/* Diagnostics for bench.py: the largest convolution output of the last forward() that used workspace `ws` (its last
 * chunk; synchronises `stream`).  Below 2^15 the large-batch kernels used two fp16 pieces per operand, otherwise three
 * bf16 pieces for the convolution's channels (device-side range guard). */
int mtadgat_last_conv_max(mtadgat_handle h, const void* ws_dev, int64_t batch, float* out_host, void* stream);

/* Bytes of device scratch forward() needs for a batch of `batch` windows
 * (intermediates of at most mtadgat_chunk_windows() windows are live at once). */
size_t  mtadgat_workspace_bytes(mtadgat_handle h, int64_t batch);
int64_t mtadgat_chunk_windows(mtadgat_handle h);
int     mtadgat_set_chunk_windows(mtadgat_handle h, int64_t windows);

/* Replaces MTAD_GAT.forward (mtad_gat.py:64-79), eval mode.
 *   x_dev      (batch, W, F)          in, not modified
 *   preds_dev  (batch, out_dim)       out   (may be NULL: skip both heads if recons_dev is NULL too)
 *   recons_dev (batch, W, out_dim)    out
 *   hend_dev   (batch, gru_hid_dim)   out, optional (NULL to skip): h_end of mtad_gat.py:74 */
int mtadgat_forward(mtadgat_handle h, const float* x_dev, int64_t batch,
                    float* preds_dev, float* recons_dev, float* hend_dev,
                    void* workspace_dev, size_t workspace_bytes, void* stream);

/* mtadgat_forward with the input given as bfloat16 (batch, W, F): the convolution reads it directly, no fp32 copy
 * of x is made (BASELINE "bf16 inference"; outputs stay float32).  MTADGAT_ERR_UNSUPPORTED for n_features beyond
 * the LDS-staged convolution. */
int mtadgat_forward_xbf16(mtadgat_handle h, const void* x_bf16_dev, int64_t batch, float* preds_dev, float* recons_dev,
                          float* hend_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Same as mtadgat_forward, with the windows gathered on the GPU from a device-resident series instead of
 * materialised by the caller: window w = series rows [s_w, s_w + W), s_w = starts_dev[w] when starts_dev is
 * not NULL, else start0 + w * stride.  Replaces SlidingWindowDataset.__getitem__ + default collate feeding
 * forward() (utils.py:107-120, prediction.py:43-44, 51-55): consecutive windows share W-1 rows, so the input
 * read from HBM shrinks ~W-fold.
 *   series_dev (n_rows, F) float32;  starts_dev (batch) int64 or NULL;  every window must lie inside the series.
 *   recons_last_dev (batch, out_dim), optional: recons[:, -1, :] only (what Predictor.get_score keeps,
 *   prediction.py:63); recons_dev may then be NULL and the full (batch, W, out_dim) tensor is never written. */
int mtadgat_forward_series(mtadgat_handle h, const float* series_dev, int64_t n_rows,
                           const int64_t* starts_dev, int64_t start0, int64_t stride, int64_t batch,
                           float* preds_dev, float* recons_dev, float* recons_last_dev,
                           void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- stage entry points (the reference's sub-module forward() calls) ---------
 * Same workspace / stream contract; each is what forward() runs for that stage. */

/* ConvLayer.forward, modules.py:18-22: x (batch,W,F) -> y (batch,W,F). */
int mtadgat_conv(mtadgat_handle h, const float* x_dev, int64_t batch, float* y_dev,
                 void* workspace_dev, size_t workspace_bytes, void* stream);

/* FeatureAttentionLayer.forward (modules.py:65-95) when which == 0,
 * TemporalAttentionLayer.forward (modules.py:166-193) when which == 1:
 * xc (batch,W,F) -> h (batch,W,F). */
int mtadgat_gat(mtadgat_handle h, int which, const float* xc_dev, int64_t batch, float* h_dev,
                void* workspace_dev, size_t workspace_bytes, void* stream);

/* GRULayer.forward, modules.py:235-238: h_cat (batch,W,3F) -> h_end (batch,H)
 * (the h[-1] the reference keeps; its out[-1] is discarded by mtad_gat.py:73). */
int mtadgat_gru(mtadgat_handle h, const float* hcat_dev, int64_t batch, float* hend_dev,
                void* workspace_dev, size_t workspace_bytes, void* stream);

/* Forecasting_Model.forward (modules.py:307-311) and ReconstructionModel.forward
 * (modules.py:276-283): h_end (batch,H) -> preds (batch,out), recons (batch,W,out). */
int mtadgat_heads(mtadgat_handle h, const float* hend_dev, int64_t batch,
                  float* preds_dev, float* recons_dev,
                  void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- training step --------------------------------------------------------------------------------
 * Replaces what autograd does for the reference around MTAD_GAT.forward in Trainer.fit
 * (training.py:106-127: preds, recons = model(x); loss.backward()): a forward that keeps the
 * activations the backward needs in a caller-owned "tape" and applies dropout inside the kernels
 * (attention matrices modules.py:90 / :189, forecasting layers modules.py:310), and the backward
 * that turns d loss / d preds, d loss / d recons into the gradients of all parameters.
 *
 * Dropout is counter based: the mask of window `window0 + w` depends only on (seed, site, that global
 * window index, element), so chunking, sharding or recomputing a batch reproduces the same masks.
 * dropout_p = 0 gives the deterministic (eval-mode) function, e.g. for gradients in eval().
 *
 * Gradients are ACCUMULATED (+=) into `grads_dev`, a flat float32 buffer of mtadgat_grad_floats()
 * entries holding the reference's parameters' gradients in their own shapes, in the field order of
 * mtadgat_params (offsets: mtadgat_grad_offsets); zero it before the first chunk of a step.
 * Every configuration the forward accepts has a HIP backward (mtadgat_backward_supported: GATv2 and GAT (v1), any number of stacked
 * GRU / decoder layers; attention layers with more than 128 nodes or features -- up to 512 -- run the kernels of
 * csrc/mtadgat_bwdw.hip with every matrix through memory; the GATv2 score backward of ALL layers is that file's one-pass k_bw_pair
 * behind a projection row GEMM since round 6).  `batch` windows are processed as one
 * chunk: tape and workspace grow linearly with it (~0.85 + 0.85 MB per window at W=100, F=55). */
int     mtadgat_backward_supported(mtadgat_handle h);
size_t  mtadgat_tape_bytes(mtadgat_handle h, int64_t batch);
size_t  mtadgat_backward_workspace_bytes(mtadgat_handle h, int64_t batch);
int64_t mtadgat_grad_floats(mtadgat_handle h);
/* offsets (floats) of conv w,b | feature lin w,b,a,bias | temporal lin w,b,a,bias | gru w_ih,w_hh,b_ih,b_hh |
 * fc (w,b) x forecast_n_linear | decoder w_ih,w_hh,b_ih,b_hh | recon fc w,b; returns the count */
int     mtadgat_grad_offsets(mtadgat_handle h, int64_t* offsets_out, int max_n);
int mtadgat_forward_train(mtadgat_handle h, const float* x_dev, int64_t batch, int64_t window0, float dropout_p,
                          uint64_t seed, float* preds_dev, float* recons_dev, void* tape_dev, size_t tape_bytes,
                          void* stream);
int mtadgat_backward(mtadgat_handle h, const float* x_dev, int64_t batch, int64_t window0, float dropout_p, uint64_t seed,
                     const float* d_preds_dev, const float* d_recons_dev, const void* tape_dev, size_t tape_bytes,
                     float* grads_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* Gradient with respect to the input windows (the reference's autograd provides it when x.requires_grad: mtad_gat.py:64-79 under
 * training.py:126; none of its callers asks).  Call right after mtadgat_backward of the same chunk, on the same stream, with the
 * same workspace: the convolution's pre-activation gradients are still in it.  dx_dev: (batch, W, F) float32, overwritten. */
int mtadgat_backward_input(mtadgat_handle h, int64_t batch, const void* workspace_dev, size_t workspace_bytes, float* dx_dev, void* stream);
/* Diagnostics for the tests: the keep-masks (1 / 0) the kernels apply -- mask_feat (batch, F, F), mask_temp
 * (batch, W, W), mask_fc (forecast_n_linear - 1, batch, forecast_hid_dim), any may be NULL -- and the offsets
 * (floats) of the tape / backward-workspace regions (order: see mtadgat_capi.cpp). */
int mtadgat_dropout_masks(mtadgat_handle h, int64_t batch, int64_t window0, float dropout_p, uint64_t seed,
                          float* mask_feat_dev, float* mask_temp_dev, float* mask_fc_dev, void* stream);
/* ... and of nn.GRU's dropout between stacked layers (reference modules.py:233, :253; training only): mask_gru
 * (gru_n_layers - 1, batch, W, gru_hid_dim), mask_rec (recon_n_layers - 1, batch, W, recon_hid_dim); either may be NULL. */
int mtadgat_dropout_masks_rnn(mtadgat_handle h, int64_t batch, int64_t window0, float dropout_p, uint64_t seed, float* mask_gru,
                              float* mask_rec, void* stream);
int mtadgat_train_layout(mtadgat_handle h, int64_t batch, int64_t* offsets_out, int max_n);

/* ---- anomaly-score post-processing on the device (callers' data path) ------
 * The arithmetic of Predictor.get_score (prediction.py:72-91) and of the threshold evaluation
 * (eval_methods.py: find_epsilon :189-236, adjust_predicts :6-55 + calc_point2point :58-72 for one threshold --
 * epsilon_eval -- or a whole sweep -- bf_search :117-158).  Device arrays in, small host tables out (these calls
 * synchronise the stream); the scalar bookkeeping on top is mtad-gat-pytorch_amd/evaluation.py.  Status 0 / -1
 * (bad argument) / -3 (HIP) / -5 (more anomaly segments than max_seg). */
int mtadgat_eval_scores(const float* preds_dev, const float* recons_dev, const float* actual_dev, int64_t n, int d,
                        int64_t ld_actual, const int* dims_dev, float gamma, float* per_dim_dev, float* global_dev, void* stream);
int mtadgat_eval_moments(const float* e_dev, int64_t n, double* scratch_dev, double* out_host, void* stream);
int mtadgat_eval_epsilon_table(const float* e_dev, int64_t n, const double* eps_host, int nz, int halo, double* scratch_dev,
                               double* out_host, void* stream);
int mtadgat_eval_point_adjust(const float* score_dev, const unsigned char* label_dev, int64_t n, const double* thr_host,
                              int n_thr, int compare_f32, int max_seg, double* scratch_dev, double* out_host, void* stream);

/* Per-kernel launch timing for bench.py's roofline leg: when enabled, forward()
 * brackets each kernel family with hipEvents on `stream`; mtadgat_profile_read
 * synchronises those events and returns accumulated milliseconds + launch counts
 * since the last read.  names: "conv","proj","attend","gru","fc","recon". */
#define MTADGAT_PROFILE_SLOTS 6
int mtadgat_profile_enable(mtadgat_handle h, int on);
int mtadgat_profile_read(mtadgat_handle h, double ms[MTADGAT_PROFILE_SLOTS],
                         int64_t launches[MTADGAT_PROFILE_SLOTS]);
const char* mtadgat_profile_name(int slot);

#ifdef __cplusplus
}
#endif
#endif /* MTADGAT_H */
