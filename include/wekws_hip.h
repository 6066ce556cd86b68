/*
 * wekws_hip.h -- C ABI of libwekws_hip.so, the MI355X (gfx950) implementation of the
 * WeKws keyword-spotting inference hot path.
 *
 * The reference has no C ABI of its own; the path sits behind two call boundaries
 * (paths relative to the reference tree):
 *   - Python   wekws/model/kws_model.py:65-76   KWSModel.forward(x, in_cache) -> (y, out_cache)
 *              wekws/model/kws_model.py:78-90   KWSModel.forward_softmax
 *   - C++      runtime/core/kws/keyword_spotting.h:26-55   wekws::KeywordSpotting::{ctor,Reset,Forward}
 *              runtime/core/frontend/fbank.h:138-198       wenet::Fbank::Compute
 * Every entry point below names the reference interface it replaces.  INTEGRATION.md shows
 * the ctypes / C++ stubs a reference maintainer would add to bind them.
 *
 * Conventions
 *   - plain C, no torch / STL types; all tensors are caller-owned DEVICE pointers (float32,
 *     contiguous unless a stride is given); the library owns only its device copy of the
 *     weights and scratch workspaces (one per model and calling stream, grown on demand).
 *   - every call is asynchronous on the hipStream_t passed as `void* stream` (NULL = the default stream of the
 *     object's device); no hidden synchronisation except when a scratch buffer has to grow (wekws_hip_reserve).
 *   - an object lives on the device it was created for; calls make that device current for their duration and
 *     restore the caller's current device before returning.
 *   - return value: 0 on success, a negative WEKWS_HIP_E* code on failure; the message is
 *     available from wekws_hip_last_error() (thread-local).  Nothing throws across the ABI.
 *   - a model is immutable after create: wekws_hip_forward is re-entrant across streams as
 *     long as each call has its own x / y / cache buffers.  Calls that need scratch memory
 *     (inputs longer than one LDS tile, the GRU) take it from the workspace of the stream they are
 *     issued on; growing a workspace synchronises that stream once.
 */
#ifndef WEKWS_HIP_H_
#define WEKWS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WEKWS_HIP_ABI_VERSION 2 /* 2: wekws_hip_forward_status; failures of a stream surface on its next call */

/* frames one kernel launch keeps resident in LDS; longer inputs are processed as a sequence of
 * tiles that hand the causal left context over through the streaming cache */
#define WEKWS_HIP_TILE_FRAMES 112

enum wekws_hip_error {
  WEKWS_HIP_OK = 0,
  WEKWS_HIP_EINVAL = -1,      /* bad argument / unsupported configuration */
  WEKWS_HIP_ENOMEM = -2,      /* device allocation failed */
  WEKWS_HIP_EDEVICE = -3,     /* HIP runtime error (no device, launch failure ...) */
  WEKWS_HIP_EUNSUPPORTED = -4 /* valid reference config that this build has no kernel for.  Since ABI 2 no MODEL
                                 configuration returns it: shapes beyond the specialised kernels (more than 256 channels, kernel
                                 sizes above 8 / 5, GRU hidden sizes above 128 or pooled heads on a GRU, ...) run on the
                                 any-shape exact-f32 path (csrc/generic.hip.h; wekws_hip_effective_precision reports F32).
                                 That path is one launch per layer -- and for a GRU two launches per layer AND TIME STEP (2 T L
                                 launches per call: a correctness path; do not capture long GRU calls of such shapes into a graph).
                                 Left: fbank frame lengths outside 65 .. 512 samples. */
};

/* backbone.type of the reference model config (wekws/model/kws_model.py:126-170) */
enum wekws_hip_backbone {
  WEKWS_HIP_BACKBONE_DS_TCN = 0, /* type: tcn, ds: true   (wekws/model/tcn.py:91-119)  */
  WEKWS_HIP_BACKBONE_TCN = 1,    /* type: tcn, ds: false  (wekws/model/tcn.py:67-88)   */
  WEKWS_HIP_BACKBONE_MDTC = 2,   /* type: mdtc            (wekws/model/mdtc.py:201-276) */
  WEKWS_HIP_BACKBONE_GRU = 3,    /* type: gru             (wekws/model/kws_model.py:128-133) */
  WEKWS_HIP_BACKBONE_FSMN = 4    /* type: fsmn            (wekws/model/fsmn.py:394-495) */
};

/* classifier of the reference config (wekws/model/kws_model.py:175-199) */
enum wekws_hip_head {
  WEKWS_HIP_HEAD_LINEAR = 0,  /* LinearClassifier, per frame   (classifier.py:54-67) */
  WEKWS_HIP_HEAD_GLOBAL = 1,  /* GlobalClassifier, mean over T (classifier.py:19-28) */
  WEKWS_HIP_HEAD_LAST = 2,    /* LastClassifier, last frame    (classifier.py:31-40) */
  WEKWS_HIP_HEAD_IDENTITY = 3 /* classifier.type: identity     (kws_model.py:188-189) */
};

enum wekws_hip_activation {
  WEKWS_HIP_ACT_IDENTITY = 0,
  WEKWS_HIP_ACT_SIGMOID = 1, /* kws_model.py:196-199 */
  WEKWS_HIP_ACT_SOFTMAX = 2  /* the model *is* forward_softmax: what wekws/bin/export_onnx.py:46-48 exports for CTC
                                recipes (forward := forward_softmax before tracing).  Per-frame heads only; set by the
                                exported-file reader, never by a training config.  wekws_hip_forward then applies the
                                softmax whatever its `softmax` argument says. */
};

/* How the 1x1 / dense convolutions and Linear layers of the conv backbones are multiplied.  All modes read and
 * write float32 and accumulate in float32; F32 and F16X3 meet the 1e-4 posterior bar against the reference, F16 is the
 * opt-in reduced-precision mode of BASELINE.json's "fp16 weights + fp16 MFMA pointwise conv" configuration. */
enum wekws_hip_precision {
  WEKWS_HIP_PRECISION_DEFAULT = 0, /* library's choice: F16X3 */
  WEKWS_HIP_PRECISION_F32 = 1,     /* exact-f32 matrix instructions: each product rounded once like the reference's fp32
                                      math (conv backbones and GRU: MFMA kernels; FSMN, and every shape without a specialised
                                      kernel: the any-shape path of csrc/generic.hip.h -- v_mfma_f32_16x16x4_f32 for the
                                      matrix products, v_fma_f32 for the depthwise taps and the GRU cell) */
  WEKWS_HIP_PRECISION_F16X3 = 2,   /* operands split into fp16 hi + lo, three fp16 matrix products per term, with BLOCK
                                      FLOATING POINT: every weight matrix and every operand tile carries an exact
                                      power-of-two scale chosen from its magnitude, so the accuracy is fp32-level (~2^-22
                                      of the tile maximum) at ANY operand scale -- no overflow at 65504, no loss below
                                      fp16's normal range (csrc/conv_stack_f16.hip.h); ~5x the f32 matrix rate */
  WEKWS_HIP_PRECISION_F16 = 3      /* weights and activations rounded to fp16 where they enter a pointwise-conv / input
                                      Linear product, ONE matrix product per term, fp32 accumulate; depthwise taps,
                                      biases, residuals and the classifier stay fp32.  Posterior error vs the fp32
                                      reference ~1e-3 (tests state the bound).  Honoured by the 16-wave kernels
                                      (DS-TCN hidden 256 with a keyword head, MDTC hidden 64); every other shape runs
                                      F16X3, i.e. more accurate than asked. */
};

/*
 * Model descriptor = the reference's configs['model'] dict (kws_model.py:97-214) as plain ints.
 * The weights travel separately as ONE float32 blob with inference-time constants already
 * folded by the host packer (wekws_amd/pack.py): GlobalCMVN into the first Linear, every
 * eval-mode BatchNorm1d into the preceding conv, Dropout dropped.  Blob layout, in order
 * (row-major, all float32):
 *   preprocessing       Wpre[hdim][idim], bpre[hdim]
 *                       (preproc_relu = 0 encodes NoSubsampling / CMVN-only as a diagonal Wpre)
 *   DS_TCN  per block   wd[hdim][ksize], bd[hdim], Wp[hdim][hdim], bp[hdim]
 *   TCN     per block   W[hdim][hdim][ksize], b[hdim]
 *   MDTC    per block   wd[hdim][ksize], bd[hdim], W1[hdim][hdim], b1[hdim], W2[hdim][hdim], b2[hdim]
 *                       (block order: preprocessor, then stack 0 block 0 ... as mdtc.py:247-275)
 *   GRU     per layer   W_ih[3H][H], W_hh[3H][H], b_ih[3H], b_hh[3H]   (gate order r,z,n)
 *   head LINEAR         Wc[odim][hdim], bc[odim]
 *   head GLOBAL / LAST  W1[head_hidden][hdim], b1[head_hidden], W2[odim][head_hidden], b2[odim]
 *   head IDENTITY       (nothing; odim == hdim)
 *
 * FSMN (fsmn.py:394-495; always preprocessing none + identity head, fsmn_ctc.yaml:36-56) has no preprocessing /
 * head sections; its blob is
 *   in_linear1 W[aux0][idim], b[aux0]  (CMVN folded) ; in_linear2 W[hdim][aux0], b[hdim]
 *   per layer  Wproj[proj][hdim] (no bias), taps[proj][lorder + rorder], Waff[hdim][proj], baff[hdim]
 *              taps = [conv_left taps, +1 on the last (the block's identity path, fsmn.py:236-237) | conv_right taps]
 *   out_linear1 W[aux1][hdim], b[aux1] ; out_linear2 W[odim][aux1], b[odim]
 * with the descriptor slots read as: hdim = linear_dim, num_layers = fsmn layers, num_stack = proj_dim,
 * kernel_size = left_order, stack_size = right_order, aux[0] = input_affine_dim, aux[1] = output_affine_dim.
 * Memory blocks run with stride 1, as the reference builds them whatever left_stride / right_stride say
 * (fsmn.py:381-383).  The cache is 4-D: (B, proj_dim, left_order - 1 + right_order, layers), layer index innermost.
 */
typedef struct wekws_hip_desc {
  int32_t abi_version;  /* WEKWS_HIP_ABI_VERSION */
  int32_t backbone;     /* enum wekws_hip_backbone */
  int32_t idim;         /* input_dim  (feature dim, e.g. 40) */
  int32_t hdim;         /* hidden_dim (channels of the backbone) */
  int32_t odim;         /* output_dim (keywords / classes / tokens) */
  int32_t num_layers;   /* tcn / gru / fsmn: num_layers; mdtc: unused (0) */
  int32_t num_stack;    /* mdtc: num_stack; fsmn: proj_dim; else 0 */
  int32_t stack_size;   /* mdtc: stack_size; fsmn: right_order; else 0 */
  int32_t kernel_size;  /* tcn / mdtc conv kernel size; fsmn: left_order; gru: 0 */
  int32_t preproc_relu; /* 1: LinearSubsampling1 (Linear+ReLU, subsampling.py:39-61); 0: no ReLU */
  int32_t head;         /* enum wekws_hip_head */
  int32_t head_hidden;  /* GLOBAL / LAST: width of the MLP (64 in kws_model.py:181-186), else 0 */
  int32_t activation;   /* enum wekws_hip_activation */
  int32_t precision;    /* enum wekws_hip_precision */
  int32_t aux[2];       /* fsmn: input_affine_dim, output_affine_dim; every other backbone: must be 0 */
} wekws_hip_desc;

typedef struct wekws_hip_model wekws_hip_model;

/* Thread-local description of the last failure on the calling thread ("" if none). */
const char* wekws_hip_last_error(void);

/* ABI version the library was built with (== WEKWS_HIP_ABI_VERSION of its header). */
int wekws_hip_abi_version(void);

/* Number of float32 elements the weight blob for `desc` must hold; 0 if desc is invalid. */
size_t wekws_hip_blob_elems(const wekws_hip_desc* desc);

/*
 * Replaces: init_model(configs) + load_state_dict + .to(device)   (kws_model.py:97-214,
 * wekws/utils/checkpoint.py:23-36) and KeywordSpotting::KeywordSpotting(model_path)
 * (runtime/core/kws/keyword_spotting.cc:28-45).
 * host_blob: `n_elems` float32 on the HOST, layout above.  device: HIP device ordinal.
 */
int wekws_hip_create(const wekws_hip_desc* desc, const float* host_blob, size_t n_elems,
                     int device, wekws_hip_model** out);

void wekws_hip_destroy(wekws_hip_model* m);

/* Streaming-cache geometry, as the exporter publishes it in the ONNX metadata
 * (wekws/bin/export_onnx.py:55-77: cache_dim, cache_len).  Conv backbones: cache is
 * (B, cache_dim = hdim, cache_len = sum of paddings); GRU: (num_layers, B, hdim) and
 * cache_len = 0 (the reference cannot export GRU); FSMN: (B, cache_dim = proj_dim,
 * cache_len = left_order - 1 + right_order, num_layers) as export_onnx.py:57-60 shapes it. */
int wekws_hip_cache_dim(const wekws_hip_model* m);
int wekws_hip_cache_len(const wekws_hip_model* m);
/* float32 elements of the cache tensor for a batch of B streams */
size_t wekws_hip_cache_elems(const wekws_hip_model* m, int B);
/* float32 elements of y for (B, T): B*T*odim for per-frame heads, B*odim for GLOBAL/LAST */
size_t wekws_hip_output_elems(const wekws_hip_model* m, int B, int T);

/*
 * Kernel selection.  A model picks its kernels from its shape (DESIGN.md 3.1 "which kernel runs"); these options override
 * the choice -- for A/B measurements and for the tests that keep every kernel family parity-green.  They change speed,
 * never results beyond rounding (all families meet the same parity bar).  No environment variable is read anywhere.
 */
enum wekws_hip_option {
  WEKWS_HIP_OPT_W16 = 0,        /* DS-TCN hidden 256: 1 (default) the 16-wave kernel, 0 the generic 8-wave kernel */
  WEKWS_HIP_OPT_MDTC16 = 1,     /* MDTC hidden 64: 1 (default) the 16-wave kernel, 0 the generic 8-wave kernel */
  WEKWS_HIP_OPT_STREAM = 2,     /* chunks of <= 16 frames: 1 (default) the kernels with the LDS-resident cache, 0 the batch kernels */
  WEKWS_HIP_OPT_MM = 3,         /* DS-TCN hidden 256: the all-matrix-core kernel -- -1 (default) for CTC-sized heads only, 0 never, 1 whenever eligible */
  WEKWS_HIP_OPT_HEAD_SLICES = 4,/* workgroups sharing a CTC-sized last layer on small calls: -1 (default) automatic, 0 / 1 none, n exactly n */
  WEKWS_HIP_OPT_G16 = 5,        /* calls without an incoming cache: 1 (default) the register-resident kernels -- DS-TCN hidden 256: ds256_g16.hip.h (split fp16 / fp16) and ds256_g32.hip.h (precision F32), MDTC hidden 64: mdtc64_g4.hip.h --, 0 the LDS-tile kernels, 2 like 1 but one workgroup per utterance instead of persistent ones, 3 like 1 but calls WITH an incoming cache (later chunks of 17 .. 112 frames) keep the LDS-tile kernels instead of the register-resident kernels' context variants (measurement aids) */
  WEKWS_HIP_OPT_GRU_PIPE = 7,   /* GRU: 1 (default) the layer wavefront -- one launch, the stages of all layers running at the same time on different CUs (gru_pipe.hip.h) -- up to eight rounds of stream tiles per resident slot (B <= 128 x CUs / (2 x layers)), the layer-major kernels (gru_f16.hip.h) beyond; 2 the wavefront always; 0 never; bit-identical results */
  WEKWS_HIP_OPT_ENVELOPE = 6    /* weights outside the split-fp16 envelope (wekws_hip_weight_spread_log2): 1 (default) run the exact-f32 kernels, 0 keep the split-fp16 kernels (to MEASURE where the envelope ends; accuracy is then not promised) */
};
int wekws_hip_set_option(wekws_hip_model* m, int option, int value);

/*
 * The arithmetic a model's calls REALLY run in (an enum wekws_hip_precision, never DEFAULT; negative error code for a NULL
 * model): desc.precision is a request, and not every backbone has a kernel for every mode -- F16 is honoured by the
 * 16-wave DS-TCN / MDTC kernels only (FSMN and every other shape run F16X3: more accurate than asked), a GRU without an fp16
 * kernel for its shape runs F32, every model on the any-shape path (csrc/generic.hip.h: shapes beyond the specialised
 * kernels; FSMN with precision F32 or with weights outside the split-fp16 envelope) runs F32 whatever was asked.  A caller
 * that needs exact-f32 reference rounding (a parity baseline) checks this instead of trusting the request.  Reflects the
 * kernel-selection options as they are set now.
 */
int wekws_hip_effective_precision(const wekws_hip_model* m);
/*
 * The envelope of the split-fp16 kernels.  Block floating point gives every weight matrix ONE power-of-two scale, so an
 * element keeps 22 significand bits only within 16 binades of the matrix maximum; a model whose matrices spread the
 * magnitudes of their rows (or of their K columns) over more than 2^WEKWS_HIP_F16X3_ENVELOPE_LOG2 would lose fp32-level
 * accuracy in the small rows (measured against the live reference: tests/golden/make_hetero_golden.py).  wekws_hip_create
 * measures the spread; a DEFAULT / F16X3 model beyond the envelope runs the exact-f32 kernels instead
 * (wekws_hip_effective_precision reports F32); an FSMN beyond it runs the any-shape exact-f32 path (csrc/generic.hip.h).
 *   wekws_hip_weight_spread_log2   the largest such spread of the model, in binades (-1 for a NULL model)
 */
#define WEKWS_HIP_F16X3_ENVELOPE_LOG2 20
float wekws_hip_weight_spread_log2(const wekws_hip_model* m);

/*
 * Scratch memory.  Inputs longer than one LDS tile (WEKWS_HIP_TILE_FRAMES frames; FSMN: 64) and every GRU call take
 * scratch from a grow-only buffer owned by (model, stream).  Growing it allocates, and frees the previous buffer behind
 * ONE synchronisation of that stream -- the only synchronisation the library ever performs on a caller's stream.  To keep
 * calls free of it (latency-critical loops; required before capturing a stream into a HIP graph, where a call that
 * would have to grow fails with WEKWS_HIP_EINVAL instead), size the buffer up front:
 *   wekws_hip_workspace_bytes  bytes a forward of exactly (B, T) needs (0: none).  NOT monotonic: a GRU chunk of <= 16
 *                              frames is spread over more workgroups and needs more scratch than a longer one
 *   wekws_hip_reserve          make the stream's buffer large enough for every call of at most B streams x at most T
 *                              frames now (the maximum of the above over the shapes where the launch geometry changes)
 *   wekws_hip_release          free the stream's buffer (e.g. before destroying the stream)
 */
size_t wekws_hip_workspace_bytes(const wekws_hip_model* m, int B, int T);
int wekws_hip_reserve(wekws_hip_model* m, int B, int T, void* stream);
int wekws_hip_release(wekws_hip_model* m, void* stream);
/*
 * Device-side health of the forwards issued on `stream` so far.  The GRU wavefront (WEKWS_HIP_OPT_GRU_PIPE) hands sequences
 * between workgroups inside one launch; every wait in it is bounded (~0.1 s), and a wait that gives up (a bug, or a device
 * whose other tenants keep the launch's later workgroups off the CUs for that long) ends the launch with garbage in the
 * outputs instead of hanging the GPU, and leaves a code in a word of pinned HOST memory owned by (model, stream).
 *   - the NEXT wekws_hip_forward on that (model, stream) reads the word (no synchronisation), launches nothing, clears it and
 *     returns WEKWS_HIP_EDEVICE: a caller that only ever calls forward -- the reference's scripts with the one-line import
 *     change -- cannot keep consuming garbage silently; wekws_hip_release returns it and wekws_hip_destroy leaves it in
 *     wekws_hip_last_error() if nobody asked before;
 *   - this call SYNCHRONISES the stream (for every model and stream, with or without such launches), then does the same
 *     check: WEKWS_HIP_EDEVICE (wekws_hip_last_error names the stage) if any forward since the last report gave up, else OK.
 * Nothing in the reference corresponds to it (ORT's Run is synchronous and throws, keyword_spotting.cc:77-79); call it
 * where results are consumed -- the C++ runtime does after every Forward (its one synchronisation), KWSModel.check() in Python.
 */
int wekws_hip_forward_status(wekws_hip_model* m, void* stream);

/*
 * Replaces: KWSModel.forward(x, in_cache) -> (y, out_cache)   (kws_model.py:65-76) and the body
 * of KeywordSpotting::Forward (keyword_spotting.cc:63-94, one ORT Run with the carried cache).
 *   x          (B, T, idim) device, contiguous
 *   in_cache   device, wekws_hip_cache_elems(m,B) floats, or NULL = the reference's empty-cache
 *              sentinel torch.zeros(0,0,0) (zero left context, tcn.py:49-52; zero h0 for GRU)
 *   y          device: (B, T, odim) for LINEAR / IDENTITY heads, (B, odim) for GLOBAL / LAST
 *   out_cache  device, same geometry as in_cache, or NULL if the caller does not need it;
 *              must not alias in_cache
 *   softmax    0: forward; 1: forward_softmax (softmax over the last axis, kws_model.py:89)
 */
int wekws_hip_forward(wekws_hip_model* m, const float* x, int B, int T, const float* in_cache,
                      float* y, float* out_cache, int softmax, void* stream);

/* -------------------------------------------------------------------------------------------
 * Fbank front-end  --  replaces wenet::Fbank::Compute (runtime/core/frontend/fbank.h:138-198)
 * with the framing rule of FeaturePipeline::AcceptWaveform (feature_pipeline.cc:30-47).
 * ------------------------------------------------------------------------------------------*/
enum wekws_hip_window {
  WEKWS_HIP_WINDOW_HAMMING = 0, /* the C++ runtime (fbank.h:90-96) */
  WEKWS_HIP_WINDOW_POVEY = 1    /* torchaudio.compliance.kaldi default (processor.py:196-202) */
};

typedef struct wekws_hip_fbank_cfg {
  int32_t num_bins;     /* 40 (or 80); 1..128, every triangular filter must cover an FFT bin (else EINVAL: the reference's
                           constructor CHECK-fails, fbank.h:81) */
  int32_t sample_rate;  /* 16000 */
  int32_t frame_length; /* samples, 400; 65..512 (the reference's 128- / 256- / 512-point FFT cases, fbank.h:43; else EUNSUPPORTED) */
  int32_t frame_shift;  /* samples, 160 */
  int32_t window;       /* enum wekws_hip_window */
  int32_t reserved[3];
} wekws_hip_fbank_cfg;

typedef struct wekws_hip_fbank wekws_hip_fbank;

int wekws_hip_fbank_create(const wekws_hip_fbank_cfg* cfg, int device, wekws_hip_fbank** out);
void wekws_hip_fbank_destroy(wekws_hip_fbank* f);
/* frames produced for nsamp samples: 1 + (nsamp - frame_length) / frame_shift, 0 if too short
 * (fbank.h:141-142) */
int wekws_hip_fbank_num_frames(const wekws_hip_fbank* f, int nsamp);
/*
 * pcm    (B, nsamp) device float32 in int16 scale (NOT divided by 32768; wav.h:98-102)
 * feats  (B, num_frames, num_bins) device float32
 */
int wekws_hip_fbank_compute(wekws_hip_fbank* f, const float* pcm, int B, int nsamp, float* feats,
                            void* stream);
/*
 * The same extractor on int16 PCM in device memory  --  the input of
 * FeaturePipeline::AcceptWaveform(const std::vector<int16_t>&) (feature_pipeline.cc:49-55): samples are widened to float
 * in registers (no /32768), so a caller that owns int16 audio uploads and reads 2 bytes per sample.  Bit-identical
 * features to wekws_hip_fbank_compute on the widened samples.
 *   pcm    (B, nsamp) device int16
 */
int wekws_hip_fbank_compute_i16(wekws_hip_fbank* f, const int16_t* pcm, int B, int nsamp, float* feats,
                                void* stream);

/*
 * Context expansion + frame skip  --  replaces context_expansion / frame_skip of the data pipeline
 * (wekws/dataset/init_dataset.py:24-68, wekws/dataset/processor.py:267-311; fsmn_ctc.yaml:21-25 uses left 2,
 * right 2, skip 3: 80-d fbank -> the 400-d FSMN input at a third of the frame rate), fused into one gather:
 *   out[b][i][(lag + left) * F + f] = feats[b][max(i * skip + lag, 0)][f],   lag = -left .. right,
 *   i = 0 .. wekws_hip_splice_frames(T, right, skip) - 1
 * (the left margin replicates frame 0, the last `right` frames are dropped, then every skip-th frame is kept).
 */
/* ceil((T - right) / skip) for T >= right.  T < right (an utterance shorter than its right context): the reference's cut
 * feats_ctx[:, :T - right] is a negative slice that keeps 2 T - right frames (0 if that is <= 0), whose right-hand blocks are
 * torch.roll's wrap-around, frame (i * skip + lag) mod T -- reproduced bit for bit: ceil(max(2 T - right, 0) / skip). */
int wekws_hip_splice_frames(int T, int right, int skip);
/*
 * feats  (B, T, F) device float32
 * out    (B, wekws_hip_splice_frames(T, right, skip), (left + right + 1) * F) device float32
 * left >= 1 and left >= T: WEKWS_HIP_EINVAL -- the reference's left-margin loop (init_dataset.py:45-48) raises IndexError there.
 */
int wekws_hip_splice(const float* feats, int B, int T, int F, int left, int right, int skip, float* out,
                     void* stream);

/* -------------------------------------------------------------------------------------------
 * MFCC tail  --  what torchaudio.compliance.kaldi.mfcc does after its fbank call (the reference's MDTC recipes,
 * wekws/dataset/processor.py:160-169: num_ceps 80 of num_mel_bins 80):
 *   out = (logmel @ DCT) * lifter,  DCT = DCT-II 'ortho' (N x N) with column 0 := sqrt(1/N), first num_ceps columns;
 *   lifter_i = 1 + 0.5 Q sin(pi i / Q), Q = cepstral_lifter (22 in the reference's call; 0 disables it).
 * The log-mel rows come from wekws_hip_fbank_compute with WEKWS_HIP_WINDOW_POVEY.  torchaudio itself is not installable
 * here; parity is pinned against independent third-party implementations of the same algorithm (Hugging Face transformers'
 * numpy port of kaldi.fbank, scipy's DCT: tests/golden/kaldi_golden.npz) and a restatement of torchaudio's published code.
 *   logmel (rows, num_bins) device float32;  out (rows, num_ceps) device float32;  num_ceps <= num_bins <= 128
 * ------------------------------------------------------------------------------------------*/
int wekws_hip_dct_lifter(const float* logmel, int64_t rows, int num_bins, int num_ceps, float cepstral_lifter, float* out,
                         void* stream);

/*
 * Row softmax + top-k  --  the first beam prune of the CTC prefix beam search: `logits.softmax(2)` followed by
 * `probs.topk(score_beam_size)` per frame (wekws/bin/stream_kws_ctc.py:487-488, wekws/model/loss.py:236-238), fused so
 * that the (frames x vocabulary) posterior matrix is neither written nor copied to the host.
 *   logits (rows, K) device float32;  k in 1..8 (the reference uses 3)
 *   probs  (rows, k) device float32: the k largest softmax posteriors of each row, descending
 *   idx    (rows, k) device int32:   their column indices (equal values: lower index first; -1 if K < k)
 */
int wekws_hip_softmax_topk(const float* logits, int64_t rows, int K, int k, float* probs, int32_t* idx,
                           void* stream);

/* -------------------------------------------------------------------------------------------
 * DET scoring  --  replaces the arithmetic of wekws/bin/compute_det.py:79-106 on the score lists that
 * wekws/bin/score.py:128-137 writes (one per utterance and keyword: the per-frame posteriors scores[b][0:len][k]),
 * so that an evaluation loop moves B*K maxima (+ B*n_thr counts) to the host instead of the (B, T, K) matrix.
 * Bit-exact: comparisons only.
 * ------------------------------------------------------------------------------------------*/
/*
 * Max pooling over time  --  compute_det.py:82-85 `score = max(score_list)` (a keyword utterance is a false reject at
 * threshold th iff max < th).
 *   scores     (B, T, K) device float32 (posteriors of wekws_hip_forward, per-frame heads)
 *   lengths    (B) device int32 valid frames per utterance (score.py:131 `logits[i][:feats_lengths[i]]`), clamped to
 *              [0, T]; NULL = T for every utterance
 *   max_out    (B, K) device float32; -inf for an empty utterance
 *   argmax_out (B, K) device int32 first frame that attains the maximum (list.index(max(list))), -1 if empty; or NULL
 */
int wekws_hip_score_maxpool(const float* scores, int B, int T, int K, const int32_t* lengths, float* max_out,
                            int32_t* argmax_out, void* stream);
/*
 * False-alarm counts of filler utterances  --  compute_det.py:88-96: per utterance and threshold,
 *   i = 0; while i < len: if score[i] >= th: n += 1; i += window_shift  else: i += 1
 *   keyword      column of `scores` (0 <= keyword < K)
 *   thresholds   (n_thr) device float64, the exact doubles the reference's `threshold += step` loop visits
 *   alarms       (B, n_thr) device int32
 * Scores are widened to double for the comparison, as Python does with the parsed floats.
 */
int wekws_hip_det_false_alarms(const float* scores, int B, int T, int K, int keyword, const int32_t* lengths,
                               const double* thresholds, int n_thr, int window_shift, int32_t* alarms, void* stream);
/*
 * The same scan on the scores AS THE REFERENCE'S TEXT FILE CARRIES THEM: score.py:134-135 writes '{:.6f}' and
 * compute_det.py parses that back, so every score is rounded to six decimals before it meets a threshold -- 0.4999997
 * is "0.500000" and counts as >= 0.5.  Use this one (and round the maxima of wekws_hip_score_maxpool the same way on the
 * host: rounding is monotonic, so max and rounding commute) to reproduce a stats file bit for bit; use the plain one for
 * the float32 posteriors themselves.
 */
int wekws_hip_det_false_alarms_text(const float* scores, int B, int T, int K, int keyword, const int32_t* lengths,
                                    const double* thresholds, int n_thr, int window_shift, int32_t* alarms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WEKWS_HIP_H_ */
