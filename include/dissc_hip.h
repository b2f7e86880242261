/*
 * dissc_hip.h -- C ABI of libdissc_hip.so, the MI355X (gfx950) implementation of
 * the DISSC inference hot path.
 *
 * The reference (gallilmaimon/DISSC) has no FFI layer: the hot path sits behind
 * four nn.Module call signatures.  Each entry point below
 * names the reference interface it replaces; the dissc_amd Python package re-creates those
 * Python signatures on top of this ABI through ctypes.
 *
 * Conventions
 *   - every data pointer passed to a *_forward / kernel entry point is a DEVICE
 *     pointer owned by the caller (PyTorch's allocator in practice); pointers
 *     passed to *_create are HOST pointers (weights are copied and re-packed
 *     once, the library owns the device copy);
 *   - no allocation and no synchronisation inside *_forward: work is enqueued
 *     on the given hipStream_t (passed as void*; NULL = the null stream);
 *   - return 0 on success, a negative DISSC_E* code on failure; nothing throws
 *     across the ABI; dissc_last_error() returns a thread-local message;
 *   - one handle per process/GPU, handles are independent; calls on ONE handle must be stream-ordered
 *     (a handle owns helper streams/events and the caller's workspace is its scratch); the only
 *     process-wide state is the tuning table behind dissc_set_option.
 */
#ifndef DISSC_HIP_H
#define DISSC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DISSC_OK 0
#define DISSC_EINVAL (-1)   /* bad argument / unsupported configuration */
#define DISSC_ENOMEM (-2)   /* workspace too small / device allocation failed */
#define DISSC_EHIP (-3)     /* a HIP runtime call failed */
#define DISSC_ENOTFOUND (-4) /* a required weight tensor is missing */
#define DISSC_ENOTSUP (-5)  /* an optional dependency is absent (RCCL for the dissc_comm_* / dissc_allgather_waves entries) */
#define DISSC_ECOMM (-6)    /* an RCCL call failed */

/* A named host tensor (fp32, contiguous, PyTorch layout). */
typedef struct {
  const char* name;   /* state_dict key after weight-norm folding, e.g. "ups.0.weight" */
  const float* data;  /* host pointer */
  int64_t shape[4];
  int32_t ndim;
} DisscTensor;

const char* dissc_last_error(void);
/* ABI version of this header: bumped when a signature changes. */
int dissc_abi_version(void);
/* Number of HIP devices visible / name of device `dev` (for diagnostics). */
int dissc_device_count(void);
int dissc_device_name(int dev, char* buf, size_t buflen);

/* ------------------------------------------------------------------------- *
 * HiFi-GAN unit/F0/speaker-conditioned generator.
 * Replaces: CodeGenerator.forward + Generator.forward + ResBlock1.forward,
 *           reference sr/models.py:179-225, :98-114, :34-41 (called from
 *           generate(), reference sr/inference.py:67-76).
 * ------------------------------------------------------------------------- */
#define DISSC_MAX_UPS 8
#define DISSC_MAX_RK 4

typedef struct {
  int32_t model_in_dim;             /* 257 = 128 code | 1 f0 | 128 spkr  (config "model_in_dim") */
  int32_t upsample_initial_channel; /* 512 */
  int32_t num_upsamples;            /* 5 */
  int32_t upsample_rates[DISSC_MAX_UPS];        /* 5,4,4,2,2 */
  int32_t upsample_kernel_sizes[DISSC_MAX_UPS]; /* 11,8,8,4,4 */
  int32_t num_kernels;              /* 3 */
  int32_t resblock_kernel_sizes[DISSC_MAX_RK];      /* 3,7,11 */
  int32_t resblock_dilations[DISSC_MAX_RK][3];      /* 1,3,5 each */
  int32_t num_embeddings;           /* 100 */
  int32_t embedding_dim;            /* 128 */
  int32_t num_speakers;             /* 200 rows in spkr.weight (reference sr/models.py:133) */
  int32_t has_f0;                   /* config "f0" */
  int32_t has_spkr;                 /* config "multispkr" */
} DisscGenConfig;

typedef struct dissc_gen* dissc_gen_t;

/* weights: folded (weight-norm removed) tensors named like the reference's
 * state_dict after remove_weight_norm(): "conv_pre.weight/.bias",
 * "ups.{i}.weight/.bias" ([Cin,Cout,k]), "resblocks.{n}.convs{1,2}.{m}.weight/.bias",
 * "conv_post.weight/.bias", "dict.weight", "spkr.weight". */
int dissc_gen_create(const DisscGenConfig* cfg, const DisscTensor* weights, size_t n_weights,
                     dissc_gen_t* out);
/* the same with the handle's arithmetic given explicitly instead of read from the process-wide "precision" option:
 * -1 = that option, 0 = exact fp32, 1 = split-bf16 (opt-in, see dissc_set_option) */
int dissc_gen_create_ex(const DisscGenConfig* cfg, const DisscTensor* weights, size_t n_weights, int precision,
                        dissc_gen_t* out);
void dissc_gen_destroy(dissc_gen_t g);
/* samples produced per input frame (= prod(upsample_rates) = 320) */
int dissc_gen_hop(dissc_gen_t g);
size_t dissc_gen_workspace_bytes(dissc_gen_t g, int B, int Tmax);
/* FLOPs (2*MAC) of one forward for `frames` total valid frames (roofline accounting) */
double dissc_gen_flops(dissc_gen_t g, int64_t frames);
/* 2 * multiply-adds the matrix pipe EXECUTES for `frames` code frames: equal to dissc_gen_flops unless ResBlock convs
 * run in the Toom-Cook F(4,3) transform domain (csrc/conv_wino.hip; option "wino", default on for the C >= 64 stages),
 * which do 6 ceil(k / 3) / 4 products per output and channel pair instead of k. */
double dissc_gen_flops_executed(dissc_gen_t g, int64_t frames);
/*
 * code    i64 [B,Tmax]   unit ids in [0,num_embeddings)
 * f0      f32 [B,Tmax]   one value per frame (reference shape [B,1,T])
 * spkr    i64 [B]        speaker ids in [0,num_speakers)
 * lengths i32 [B]        valid frames per utterance (NULL = all Tmax).  Every
 *                        layer treats positions >= length as the zero padding the
 *                        reference's B=1 run would see, so a ragged batch is
 *                        sample-exact with per-utterance runs.
 * wav_out f32 [B,hop*Tmax]  samples >= hop*length[b] are written as 0.
 */
int dissc_gen_forward(dissc_gen_t g, const int64_t* code, const float* f0, const int64_t* spkr,
                      const int32_t* lengths, int B, int Tmax, float* wav_out, void* workspace,
                      size_t workspace_bytes, void* stream);
/* Stand-alone conv entry points (unit-testable building blocks of the above).
 * x f32 [B,Cin,ldx] -> y f32 [B,Cout,ldo]; "same" zero padding at both ends of
 * each utterance's valid length; in_slope = leaky-ReLU slope applied to the
 * input on load (1.0 = none).  w: HOST pointer [Cout,Cin,k] (re-packed per call:
 * test/debug path only). */
int dissc_conv1d(const float* x, const float* w_host, const float* bias_host, float* y,
                 const int32_t* lengths, int B, int Cin, int Cout, int k, int dilation, int ldx,
                 int ldo, int Lmax, float in_slope, void* stream);
/* ConvTranspose1d(Cin,Cout,k,stride,padding=(k-stride)/2); w HOST [Cin,Cout,k];
 * y [B,Cout,ldo] with length stride*len. */
int dissc_conv_transpose1d(const float* x, const float* w_host, const float* bias_host, float* y,
                           const int32_t* lengths, int B, int Cin, int Cout, int k, int stride,
                           int ldx, int ldo, int Lmax, float in_slope, void* stream);

/* Stride-2 VALID conv (HuBERT's feature convs; reference data/encode.py:21-22,32 via fairseq ConvFeatureExtractionModel):
 * x f32 [B,Cin,ldx] -> y f32 [B,Cout,ldo] with (len - k) / 2 + 1 outputs per utterance, no padding; w HOST [Cout,Cin,k];
 * act 1 = exact-erf GELU on the output.  form 0 = the direct implicit GEMM, 1 = the polyphase Toom-Cook form (k = 3 only:
 * F(7,2) on the even samples + a 1-tap GEMM on the odd ones, 15 instead of 21 products per 7 outputs; conv_s2tc.hip).
 * lengths_in: device int32 [B] or NULL (= Lmax_in everywhere).  Synchronous; test / gate entry, weights packed per call. */
int dissc_conv1d_s2(const float* x, const float* w_host, const float* bias_host, float* y, const int32_t* lengths_in, int B,
                    int Cin, int Cout, int k, int ldx, int ldo, int Lmax_in, int act, int form, void* stream);
/* Diagnostics: average ms of `iters` launches of one C -> C, k = 3, stride 2 conv + GELU on B rows of L input samples. */
int dissc_conv_s2_bench(int B, int C, int L, int form, int iters, float* ms_out);

/* Tuning hook: the DEFAULTS of handles created LATER (also reachable through the environment variable
 * DISSC_OPTIONS="key=value,key=value" read by the Python binding).  Every handle -- generator, HuBERT, predictor, trainer --
 * takes a snapshot of all options when it is created and never reads the process-wide values again: forwards are
 * re-entrant per handle, two handles created under different options keep their own behaviour whatever is set in between
 * (tests/test_gpu_generator.py::test_options_are_frozen_into_the_handle).  The stand-alone entries without a handle
 * (dissc_conv1d, dissc_respair1d, the *_bench diagnostics) read the defaults at call time.
 * Options marked [experimental] select kernels whose gates failed; they are only in libraries built with
 * DISSC_EXPERIMENTAL=1 (dissc_get_option("experimental") == 1) and are refused or ignored otherwise.
 *   multistream (1)      generator: the ResBlocks of a stage run as concurrent chains on HIP streams (two side
 *                        streams per device, shared by all generator handles of the process)
 *   stream_prio (1)      ... and the longer chains get higher HIP stream priority
 *   par_ups (1)          ... and the phase groups of a ConvTranspose layer run concurrently on the same streams
 *   precision (0)        0 = exact fp32 MFMA everywhere (default, bench.py's headline); 1 = split-bf16 GENERATOR
 *                        ("bf16x3": hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32 accumulate; conv_bf3.hip,
 *                        resblock_bf3.hip) -- ~2^-17 product error, waveform RMS ~4e-6 vs the reference (bar 1e-4),
 *                        ~2x the fp32 rate.  Predictors and HuBERT feed integer decisions and always stay fp32.
 *   mfma32 (1)           use the 32x32x2 MFMA kernel for layers with >= 32 output rows
 *   conv_cfg_bm{16,32,64,128,256} / conv32_cfg_bm{32,64,128,256}
 *                        tile-shape id per GEMM-M class (tables in conv_mfma.hip / conv_mfma32.hip)
 *   small_grid (1)       launches with fewer than value x 256 workgroups step down to smaller conv tiles (0 = never)
 *   lin_tile (2)         1x1 convs: 16-channel chunks staged per barrier (2 or 4)
 *   cpb2 (0)             k <= value convs stage 32 channels per barrier
 *   pair_max_c (32)      widest fp32 stage whose residual pairs (conv_d -> conv_1 -> +x) run as ONE launch each
 *                        (respair.hip; 0 = every conv its own launch; results are bit-identical either way)
 *   pair_f23 (3)         read at dissc_gen_create, a bit mask: 1 = the k = 11 residual pairs of the 32-channel stage, 2 = those of the
 *                        16-channel stage run as register-only Toom-Cook F(2,3) (respair_f23.hip / respair16_f23.hip: 8 products per
 *                        output instead of 11, no LDS exchange; forward 1.5 % faster; not bit-identical to the direct pairs, error no
 *                        larger); 4 / 8 = their k = 3 pairs too (2 products per output instead of 3: per launch -12 % at C = 32, 0 at
 *                        C = 16, nothing in the forward -- off); 0 = the direct pairs of respair.hip
 *   wino8 (1)            read at dissc_gen_create: 1 = the ResBlock convs selected by wino8_mask run on conv_wino8.hip's
 *                        8-wave workgroups (eight Toom-Cook points: F(6,3), 8 ceil(k / 3) / 6 products per output, or F(5,4) with
 *                        4-tap sub-filters, 8 ceil(k / 4) / 5 -- forward 4 % faster than with F(4,3) everywhere, per-layer rounding
 *                        1.4-1.7x the F(4,3) form's); 0 = the F(4,3) form everywhere; 2 = dissc_conv1d uses it too (tests).
 *                        wino8_mask (0770770771 octal): one bit per SHAPE, 9 cls + 3 ki + di with cls 0 / 1 / 2 for C = 64 / 128 / >= 256,
 *                        ki 0 / 1 / 2 for k = 3 / 7 / 11, di 0 / 1 / 2 for dilation 1 / 3 / 5 (three octal digits per class).
 *                        wino8_r4 (1): the shapes of wino8_r4_mask (0770770010, same layout) run as F(5,4); 2 = dissc_conv1d too; 0 =
 *                        never.  wino8_c64_wide (3): C = 64 instances -- 1 = 64 x 128 tiles, 0 = 64 x 64, 2 = 64 x 64
 *                        built for two workgroups per CU, 3 = 1 for k = 7 as F(6,3) and 2 for everything else (k = 11; k = 7 as F(5,4))
 *   ragged_enum (1)      conv_mfma32_kernel on a ragged batch enumerates only the (time tile, utterance) pairs that exist (the
 *                        empty workgroups all sit at the end of the dispatch order); 0 = tile x utterance grid with early exits
 *   pair_wino (0)        [experimental] read at dissc_gen_create: 0 = off (default: the whole-forward gain is 0.2 %); 1 = the residual pairs this measured faster for (C = 32, k = 11, d = 1 / 3;
 *                        the first pair of the C = 64, k = 3 chain) run as ONE launch with both convs in the Toom-Cook
 *                        transform domain and the intermediate in LDS (respair_wino.hip); 2 = every shape with an instance
 *                        (C = 32: k = 7 / 11; C = 64: k = 3, first pair of a chain); 0 = none
 *   pairw_chv (2)        respair_wino.hip: 2 = one 12-wave workgroup per CU, 1 = two 6-wave workgroups with half the tile
 *   bf3_pairs (-1)       split-bf16 fused ResBlocks as three launches of one residual pair each: -1 = for >= 64
 *                        channels only, 0 = never, 1 = always
 *   fused_variant (0)    split-bf16 fused ResBlocks: 0 = 512-column windows, 1 = 1024
 *   wino (1)             read at dissc_gen_create: 1 = ResBlock convs with C >= wino_min_c (64; at C = 64 those with k >=
 *                        wino_c64_kmin = 3) run in the Toom-Cook F(4,3) transform domain (conv_wino.hip), 0 = all direct,
 *                        2 = dissc_conv1d uses it too (tests)
 *   wino_sv (1)          conv_wino.hip, C >= 128: the 12 waves of a workgroup share the input transform (one barrier per
 *                        8 channels) instead of every wave forming its own tile; bit-identical, faster (0 = private tiles)
 *   enc_tc (0)           [experimental] read at dissc_hubert_create: 1 = the k = 3, stride-2 feature convs (conv1..conv4) run in polyphase
 *                        Toom-Cook form (conv_s2tc.hip: 15 / 7 instead of 3 MFMA products per output; opt-in: its gate failed --
 *                        6 % faster per launch, 2.35x the direct form's rounding error); 0 = direct implicit GEMM (default).
 *                        s2tc_xmode (0): 0 = row tiles pinned to XCDs, 1 = the row tiles of a time tile share an XCD
 *   attn_fused (1)       HuBERT attention as one fused kernel (0: batched GEMM -> softmax -> batched GEMM)
 *   hubert_split (1)     dissc_hubert_forward runs a batch of >= 16 utterances as 2-4 parts on streams of their own, which
 *                        fill the partly filled last workgroup rounds of each other's launches (0 never, 1 unless the
 *                        whole batch fills whole rounds by itself, N >= 2 always N parts); the units do not depend on it
 *   mfast (0)            M-fastest block order for convs with many M tiles
 *   xcd_order (11)       XCD-aware workgroup order of the 32x32x2 implicit-GEMM launches with >= 2 M tiles (bit 0: 1x1 convs =
 *                        HuBERT's linears, bit 1: stride-2 convs = its feature extractor, bit 2: every other instance): a 1-D
 *                        grid whose ids are dealt in sweeps over groups of M tiles (weight slabs that fit one XCD's L2 together),
 *                        the M tiles of one input window on consecutive slots of one XCD.  Same tiles, same arithmetic, bit-
 *                        identical results; 21 % less fabric traffic on the encoder.  xcd_mg (0 = automatic): M tiles per sweep.
 *                        Bit 3: the fused attention's workgroups in XCD order (the query tiles of one (utterance, head) on one
 *                        XCD: its K / V rows cross the fabric once)
 * Unknown keys return DISSC_EINVAL. */
int dissc_set_option(const char* key, int value);
/* Read back any option's DEFAULT (so that a wrapper can set an option around the creation of one handle and restore it), plus
 * "experimental" (1: the library was built with DISSC_EXPERIMENTAL=1), "graph_hits", "graph_captures". */
int dissc_get_option(const char* key, int* value);

/* Diagnostics (not on the product path): average milliseconds of `iters` launches of
 * one Conv1d layer shape on synthetic device data; `epi` 0 plain, 1 +residual,
 * 2..4 MRF accumulate modes; `flags` = 0x8000 | cfg<<8 | bm_class<<16 overrides the
 * tile shape for this process (0 = keep). */
int dissc_conv_bench(int B, int Cin, int Cout, int k, int dilation, int L, int epi, int iters,
                     int flags, float* ms_out);

/* Diagnostics / tests (not on the product path): ONE residual pair of a ResBlock1,
 *   y = x + conv_1(lrelu(conv_d(lrelu(x))))          (reference sr/models.py:34-41; `epi` 1 = that, 2..4 = the MRF modes on `acc`)
 * on device data x / y / acc f32 [B,C,ld] with host weights [C,C,k] + biases [C], through a chosen implementation:
 * mode 0 = two direct conv launches, 1 = the fused direct pair (C = 16 / 32), 2 = two transform-domain launches
 * (conv_wino), 3 = the fused transform-domain pair (respair_wino: C = 32 with k = 7 / 11, C = 64 with k = 3).
 * y must not alias x.  dissc_respair1d synchronises the stream; dissc_pair_bench times `iters` launches on synthetic data. */
int dissc_respair1d(const float* x, const float* w1_host, const float* b1_host, const float* w2_host, const float* b2_host,
                    float* y, float* acc, const int32_t* lengths, int B, int C, int k, int dilation, int ld, int Lmax,
                    float slope, int epi, float mrf_div, int mode, void* stream);
int dissc_pair_bench(int B, int C, int k, int dilation, int L, int epi, int iters, int mode, float* ms_out);

/* ------------------------------------------------------------------------- *
 * Length / pitch predictors and infer.py's integer sample logic.
 * Replaces: LenPredictor.forward (reference model/len_predictor.py:35-52),
 *   PitchPredictor / PitchPredictorBase .infer_freq (model/pitch_predictor.py:72-104,
 *   145-176), dedup_seq (dataset/utils.py:14-16), len_carryover_correction
 *   (infer.py:158-172), torch.repeat_interleave (infer.py:32).
 * The reference runs B=1 per (utterance x target); these take a ragged batch
 * (lengths i32 [B], NULL = all Lmax) and are sample-exact with B=1 runs.
 * ------------------------------------------------------------------------- */
typedef struct dissc_pred* dissc_pred_t;
/* kind: 0 = LenPredictor, 1 = PitchPredictor ("new", positional encoding, BN on cnn2),
 * 2 = PitchPredictorBase.  weights: the state_dict tensors by name; every eval-mode
 * BatchNorm is passed pre-folded as "<conv>.bn_scale" / "<conv>.bn_shift"
 * (alpha = weight/sqrt(var+eps), beta = bias - mean*alpha). */
int dissc_pred_create(int kind, const DisscTensor* weights, size_t n_weights, dissc_pred_t* out);
void dissc_pred_destroy(dissc_pred_t p);
size_t dissc_pred_workspace_bytes(dissc_pred_t p, int B, int Lmax);
/* LenPredictor.norm_mean / norm_std (reference infer.py:72: len_norm_stats.pth) */
int dissc_len_set_norm(dissc_pred_t p, float mean, float std);
/* seq i64 [B,Lmax] dedup'd units, spk i64 [B] -> out f32 [B,ldo] frames per unit */
int dissc_len_forward(dissc_pred_t p, const int64_t* seq, const int64_t* spk, const int32_t* lengths,
                      int B, int Lmax, float* out, int ldo, void* workspace, size_t workspace_bytes,
                      void* stream);
/* seq i64 [B,Tmax] frame-rate units -> out f32 [B,ldo]: (class_logit > 0) * f0; norm != 0 keeps
 * the speaker-normalised value, else f0*id2std[spk] + id2mean[spk] (device arrays of n_stats
 * entries; ids are clamped into them -- the Python wrapper raises IndexError first, like
 * nn.Embedding does in the reference). */
int dissc_pitch_forward(dissc_pred_t p, const int64_t* seq, const int64_t* spk,
                        const int32_t* lengths, int B, int Tmax, int norm, const float* id2mean,
                        const float* id2std, int n_stats, float* out, int ldo, void* workspace,
                        size_t workspace_bytes, void* stream);
/* Per-speaker F0 statistics over VOICED (non-zero) frames, fp64 (replaces reference
 * data/data_utils.py:33-46 calculate_pitch_stats, the step between data/encode.py and infer.py).
 * f0 f64 [offsets[n_speakers]]: the frames grouped by speaker, speaker s = [offsets[s], offsets[s+1]);
 * -> mean, std (population, ddof = 0) f64 [n_speakers], count of voiced frames i64 [n_speakers].
 * A speaker without voiced frames yields NaN like numpy.  Deterministic (fixed reduction tree). */
int dissc_pitch_stats(const double* f0, const int64_t* offsets, int n_speakers, double* mean_out,
                      double* std_out, int64_t* count_out, void* stream);
/* run-length encode: units i64 [B,Tmax] -> vals i64 [B,Tmax], counts i32 [B,Tmax], n i32 [B] */
int dissc_dedup(const int64_t* units, const int32_t* lengths, int B, int Tmax, int64_t* vals,
                int32_t* counts, int32_t* n_out, void* stream);
/* error-diffusion rounding of predicted lengths: lens f32 [B,ld] (n[b] valid) ->
 * lens_int i32 [B,ld], totals i32 [B] = frames after expansion */
int dissc_len_carryover(const float* lens, const int32_t* n, int B, int ld, int32_t* lens_int,
                        int32_t* totals, void* stream);
/* repeat_interleave: vals i64 [B,ld_in], lens_int i32 [B,ld_in] -> out i64 [B,ld_out] */
int dissc_expand(const int64_t* vals, const int32_t* lens_int, const int32_t* n, int B, int ld_in,
                 int64_t* out, int ld_out, void* stream);

/* ------------------------------------------------------------------------- *
 * HuBERT-base unit encoder + k-means quantiser.
 * Replaces: textless SpeechEncoder.__call__ as used at reference data/encode.py:21-22,32
 *   = fairseq HubertModel.extract_features(source, mask=False, output_layer=n_layers)
 *   + KMeansQuantizer.predict.  Those libraries are un-vendored third parties (fairseq @
 *   dd106d95, textlesslib HEAD; reference README.md:31-34); the algorithm is restated in
 *   oracle/hubert_ref.py and pinned to HF transformers.HubertModel + sklearn KMeans.
 * weights: fairseq checkpoint names ("feature_extractor.conv_layers.{i}.0.weight",
 *   "feature_extractor.conv_layers.0.2.weight/bias", "layer_norm.*", "post_extract_proj.*",
 *   "encoder.pos_conv.0.weight" (weight-norm already folded, [768,48,128]) / ".bias",
 *   "encoder.layer_norm.*", "encoder.layers.{i}.self_attn.{q,k,v,out}_proj.*",
 *   ".self_attn_layer_norm.*", ".fc1.*", ".fc2.*", ".final_layer_norm.*").
 * centers: host f32 [n_centers,768] k-means centroids (NULL = dense features only).
 * ------------------------------------------------------------------------- */
typedef struct dissc_hubert* dissc_hubert_t;
int dissc_hubert_create(int n_layers, const DisscTensor* weights, size_t n_weights,
                        const float* centers, int n_centers, dissc_hubert_t* out);
void dissc_hubert_destroy(dissc_hubert_t m);
/* frames produced for n_samples input samples: floor((n-400)/320)+1 for n >= 400 */
int dissc_hubert_frames(int n_samples);
size_t dissc_hubert_workspace_bytes(dissc_hubert_t m, int B, int Nmax);
/* wav f32 [B,Nmax] (16 kHz, un-normalised like hubert-base), n_samples i32 [B] (NULL = Nmax)
 * -> dense_out f32 [B,768,ldT] channels-first, ldT = frames(Nmax) rounded up to 4 (may be NULL)
 *    units_out i64 [B,frames(Nmax)] (may be NULL).  Each utterance is exact w.r.t. a B=1 run.
 * A model handle is SINGLE-STREAM: the part streams and fork/join events of the split forward belong to the handle, so
 * two forwards of one handle must not be in flight at once (create one handle per concurrent stream).  On error, `stream`
 * has already been made to wait for every part stream that received work: the workspace may be reused once it drains. */
int dissc_hubert_forward(dissc_hubert_t m, const float* wav, const int32_t* n_samples, int B, int Nmax,
                         float* dense_out, int64_t* units_out, void* workspace,
                         size_t workspace_bytes, void* stream);
/* The quantiser's predict() on its own -- the integer step of the path (reference data/encode.py:21-22: textless
 * KMeansQuantizer -> sklearn predict), the very launch dissc_hubert_forward makes on its own features:
 *   units[t] = the FIRST k minimising  cnorm[k] - 2 <dense[t], centers[k]>,
 * every dot product ONE fp32 fma chain acc = fmaf(x[d], c[d], acc) over d = 0..D-1 from 0, k scanned upwards with a strict '<'
 * from +inf (a NaN score never wins; a row of NaNs gets unit 0) -- specified to the bit, so that identical inputs give identical
 * indices on any implementation (oracle/host_ref.c: oracle_kmeans_assign_f32), exact ties included.
 * dense f32 [T,D] rows, centers f32 [K,D], cnorm f32 [K] (NULL = computed here as the chain fmaf(c[d], c[d], s); then the call
 * allocates and synchronises), units int64 [T]: device pointers. */
int dissc_kmeans_assign(const float* dense, const float* centers, const float* cnorm, int T, int K, int D, int64_t* units,
                        void* stream);

/* Diagnostics: sustained fp32 v_mfma_f32_16x16x4_f32 rate (TFLOP/s) of this GPU at its
 * real clocks -- the practical ceiling the conv kernels are compared with. */
int dissc_mfma_peak(int iters, float* tflops);
/* Diagnostics: y[i] = erf(x[i]) as the GELU epilogues evaluate it (branch-free, < 1 ulp; HuBERT's exact-erf GELU,
 * arch per HF:154-213,371-445 [3P]); x, y: device f32 [n]. */
int dissc_erf_check(const float* x, float* y, int n, void* stream);
/* Diagnostics: ms[0] a pure-MFMA kernel alone, ms[1] a pure-fp32-VALU kernel alone, ms[2] both
 * at once on two streams (do the two pipes overlap on this part?). */
int dissc_pipe_overlap(int mfma_iters, int valu_iters, float* ms);

/* ------------------------------------------------------------------------- *
 * Waveform post-processing.
 * Replaces: generate(), reference sr/inference.py:73-75 ((y*32768).astype(int16),
 * C truncation + wrap) and librosa.util.normalize at :206/:250 (x / max|x|).
 * wav f32 [B,ld] (in place), lengths in SAMPLES per utterance.
 * ------------------------------------------------------------------------- */
int dissc_wav_postprocess(float* wav, const int32_t* n_samples, int B, int ld, void* stream);

/* ------------------------------------------------------------------------- *
 * Waveform exchange buffer (the payload of the path's all-gather).
 * Replaces: the per-worker file writes of the reference's Pool(8) (sr/inference.py:205-207,
 * 249-251,288-292,351-354).  Every rank packs the waveforms it decoded into ONE flat f32 buffer of
 * the same size on every rank (ragged: rows back to back, nothing padded to the longest waveform):
 *   floats [0,4)            int32 bits: {rows used, 0, data floats used lo, hi}
 *   floats [4, 4+4*n_cap)   table, one entry per row: int32 {job id (-1 = unused), sample count,
 *                           offset lo, offset hi} -- offset in floats from the start of the data region,
 *                           a multiple of 4 (the exclusive prefix sum of the sample counts rounded up to 4)
 *   floats [4+4*n_cap, ..)  data region: row r at its offset, zero-filled up to the next multiple of 4
 * The header and table are written by the host (a few KB, one H2D copy); dissc_pack_rows moves the B
 * utterances of one generator batch (wav f32 [B, ld_wav], n_samples i32 [B], offsets i64 [B], all device;
 * n_max = the largest sample count of the batch, sizes the launch) into the data region, 16 B per lane.
 * data must be 16-byte aligned.
 * ------------------------------------------------------------------------- */
int dissc_pack_rows(const float* wav, long long ld_wav, const int32_t* n_samples, const long long* offsets,
                    int B, int n_max, float* data, void* stream);

/* ------------------------------------------------------------------------- *
 * The path's one collective: the all-gather of the waveform exchange buffers over RCCL (xGMI inside a node).
 * Replaces: the reference's Pool(8) result hand-back (sr/inference.py:288-292,351-354: every worker writes its own
 * files, the parent joins) -- north_star's "single RCCL all-gather of decoded waveforms".  With these four entries a
 * maintainer who binds the C ABI alone (no torch.distributed) can run the multi-GPU path: pack with dissc_pack_rows,
 * gather with dissc_allgather_waves, unpack on the host (layout above).
 * librccl.so is NOT a link dependency of this library: the entries resolve ncclGetUniqueId / ncclCommInitRank /
 * ncclAllGather / ncclCommDestroy at first use from the RCCL the process already holds (PyTorch's, when it was imported
 * first), else from $DISSC_RCCL_LIB, librccl.so.1, /opt/rocm/lib/librccl.so.1 -- DISSC_ENOTSUP when none loads.
 *   dissc_comm_unique_id   rank 0 makes the 128-byte id (DISSC_COMM_ID_BYTES) and hands it to the other ranks out of band
 *   dissc_comm_create      collective over the nranks processes, each on its own current HIP device (RCCL: one GPU per rank);
 *                          *comm is an ncclComm_t
 *   dissc_allgather_waves  nccl_comm: an ncclComm_t -- from dissc_comm_create or any other owner (e.g. the one PyTorch's
 *                          ProcessGroupNCCL holds); send: n_floats f32 on this rank's device, recv: nranks * n_floats f32, rank r's
 *                          block at r * n_floats (send may alias its own block: in place); enqueued on `stream`, returns at once
 *                          (ordering and completion are the stream's); n_floats must be the same on every rank
 *   dissc_comm_destroy     after the stream has drained
 * ------------------------------------------------------------------------- */
#define DISSC_COMM_ID_BYTES 128
int dissc_comm_unique_id(void* id_out);
int dissc_comm_create(const void* id, int nranks, int rank, void** comm_out);
int dissc_comm_destroy(void* comm);
int dissc_allgather_waves(void* nccl_comm, const float* send, size_t n_floats, float* recv, void* stream);

/* ------------------------------------------------------------------------- *
 * YAAPT F0 tracker, device front end.
 * Replaces the numerically heavy stages of amfm_decompy.pYAAPT.yaapt as the reference calls it
 *   (get_yaapt_f0, reference sr/dataset.py:27-43; eval.py:26-33; inside textless' SpeechEncoder for
 *   data/encode.py:32): band-pass FIR of the signal and of its square, the per-frame spectra behind NLFER and
 *   the spectral-harmonics-correlation candidates, and the NCCF candidates.  amfm_decompy is an un-vendored
 *   third party that is absent offline: the algorithm is restated in oracle/yaapt_ref.py, PARITY UNPINNED.
 *   The short sequential stages (median smoothing, dynamic programming) are host logic in dissc_amd/f0.py.
 * fir: HOST f32 [n_taps] band-pass coefficients (scipy.signal.firwin in the caller, like amfm_decompy).
 * ------------------------------------------------------------------------- */
typedef struct {
  int32_t fs;            /* 16000 */
  int32_t frame_len;     /* samples: frame_length 20 ms -> 320 */
  int32_t frame_hop;     /* frame_space 5 ms -> 80 */
  int32_t tda_len;       /* tda_frame_length 25 ms -> 400 */
  int32_t nfft;          /* fft_length 8192 */
  float f0_min, f0_max;  /* 60, 400 */
  int32_t shc_numharms;  /* 3 */
  float shc_window_hz, shc_pwidth_hz, shc_thresh1, shc_thresh2, f0_double, f0_half; /* 40, 50, 5, 1.25, 150, 150 */
  float nccf_thresh1, nccf_thresh2; /* 0.25 (reference override), 0.9 */
  int32_t nccf_pwidth;   /* 5 */
} DisscYaaptConfig;
typedef struct dissc_yaapt* dissc_yaapt_t;
int dissc_yaapt_create(const float* fir, int n_taps, const DisscYaaptConfig* cfg, dissc_yaapt_t* out);
void dissc_yaapt_destroy(dissc_yaapt_t y);
/* frames of the spectral stages / of the time-domain stage for n_samples input samples; SHC bins per frame */
int dissc_yaapt_frames(dissc_yaapt_t y, int n_samples);
int dissc_yaapt_tda_frames(dissc_yaapt_t y, int n_samples);
int dissc_yaapt_shc_bins(dissc_yaapt_t y);
size_t dissc_yaapt_workspace_bytes(dissc_yaapt_t y, int B, int Nmax);
/* wav f32 [B,Nmax] (already zero-padded by frame_len/2 at both ends, as the reference does), n_samples i32 [B]
 * (NULL = Nmax), F = dissc_yaapt_frames(Nmax) ->
 *   filt, nlfilt f32 [B,Nmax]   band-passed signal / squared signal
 *   energy f32 [B,F]            NLFER band sums (un-normalised; 0 beyond an utterance's frames)
 *   cand_pitch, cand_merit f32 [B,F,4]   SHC peak candidates per frame (pitch 0 / merit 1 = no candidate)
 *   shc_out f32 [B,F,shc_bins]  optional (NULL): the SHC itself, for tests */
int dissc_yaapt_spectral(dissc_yaapt_t y, const float* wav, const int32_t* n_samples, int B, int Nmax, float* filt,
                         float* nlfilt, float* energy, float* cand_pitch, float* cand_merit, float* shc_out,
                         void* workspace, size_t workspace_bytes, void* stream);
/* NCCF candidates of `sig` (filt or nlfilt): frame f = sig[f*hop : f*hop + tda_len], lags lag_min[b,f] <= lag <
 * lag_max[b,f] (i32 [B,F], from the spectral track) -> pitch, merit f32 [B,F,3]; phi_out f32 [B,F,tda_len]
 * optional (NULL).  Frames beyond dissc_yaapt_tda_frames(n_samples[b]) give pitch 0 / merit 0.001. */
int dissc_yaapt_nccf(dissc_yaapt_t y, const float* sig, const int32_t* n_samples, const int32_t* lag_min,
                     const int32_t* lag_max, int B, int Nmax, int F, float* pitch, float* merit, float* phi_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The sequential stages of the tracker, one workgroup per utterance, fp64 (pYAAPT's spec_track, refine and both
 * dynamic-programming passes as restated in oracle/yaapt_ref.py [3P-unverified]).  All pointers are device pointers.
 *   dissc_yaapt_spec_track: energy f32 [B,F], cand_pitch / cand_merit f32 [B,F,4] (dissc_yaapt_spectral's outputs),
 *     n_frames / n_tda i32 [B] (dissc_yaapt_frames / _tda_frames of each utterance) ->
 *     en_norm f64 [B,F] (NLFER / its mean), vuv u8 [B,F], spec f64 [B,F] (smoothed spectral F0 track),
 *     spec_std f64 [B], lag_min / lag_max i32 [B,F] (the NCCF search range of every time-domain frame)
 *   dissc_yaapt_final_track: the NCCF candidates of the band-passed signal (tp1, tm1) and of its square (tp2, tm2),
 *     f32 [B,F,3] each, plus the outputs above -> f0 f32 [B,F] (0 = unvoiced, 0 beyond n_tda[b]) */
typedef struct {
  double nlfer_thresh1, nlfer_thresh2; /* 0.75, 0.1 */
  double dp5_k1;                       /* 11 */
  double merit_boost, merit_pivot, merit_extra; /* 0.20, 0.99, 0.4 */
  double dp_w1, dp_w2, dp_w3, dp_w4;   /* 0.15, 0.5, 0.1, 0.9 */
  double spec_pitch_min_std;           /* 0.05 */
  double f0_min, f0_max;               /* 60, 400 */
  int32_t median_value;                /* 7 (odd, 3..9) */
  int32_t nccf_pwidth;                 /* 5 */
  int32_t fs;                          /* 16000 */
  int32_t reserved;
} DisscYaaptTrackConfig;
size_t dissc_yaapt_track_workspace_bytes(int B, int F);
int dissc_yaapt_spec_track(const DisscYaaptTrackConfig* cfg, const float* energy, const float* cand_pitch,
                           const float* cand_merit, const int32_t* n_frames, const int32_t* n_tda, int B, int F,
                           double* en_norm, uint8_t* vuv, double* spec, double* spec_std, int32_t* lag_min,
                           int32_t* lag_max, void* workspace, size_t workspace_bytes, void* stream);
int dissc_yaapt_final_track(const DisscYaaptTrackConfig* cfg, const float* tp1, const float* tm1, const float* tp2,
                            const float* tm2, const double* en_norm, const uint8_t* vuv, const double* spec,
                            const double* spec_std, const int32_t* n_tda, int B, int F, float* f0, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Band-limited sinc resampling.
 * Replaces: resampy.resample(data, sr, 16000) at reference data/preprocess.py:22 (and sr/dataset.py:226) --
 * an un-vendored third party; algorithm (Kaiser-windowed sinc table, linear table interpolation) restated in
 * oracle/preprocess_ref.py, PARITY UNPINNED.  All arrays fp64 on the device: x [n_orig] -> y [n_out],
 * ratio = sr_new / sr_orig, win / delta [nwin] = the (already ratio-scaled) filter table and its differences,
 * num_table = table entries per zero crossing (2^precision).
 * ------------------------------------------------------------------------- */
int dissc_resample(const double* x, int n_orig, double* y, int n_out, double ratio, const double* win,
                   const double* delta, int nwin, int num_table, void* stream);

/* ------------------------------------------------------------------------- *
 * Predictor training.
 * Replaces one iteration of the training loops: model.train() forward, LenSumLoss / PitchLoss, loss.backward(),
 *   torch.optim.Adam.step() -- reference train_len_predictor.py:57-68, train_f0_predictor.py:58-66,
 *   loss/len_loss.py:16-30, loss/pitch_loss.py:6-27, model/len_predictor.py:35-52, model/pitch_predictor.py:72-94,145-166.
 * kind: 0 LenPredictor, 1 PitchPredictor ("new"), 2 PitchPredictorBase.  tensors: the model's state_dict by name
 *   (HOST fp32, incl. the BatchNorm running statistics and "pe.pe"; num_batches_tracked is kept by the caller).
 * The random masks of train() mode are inputs: keep f32 [B,L] (1 = token embedding kept; NULL = all kept),
 *   pe_mult f32 [B,L,32] (PositionalEncoding dropout multipliers 0 or 1/(1-p); kind 1 only; NULL = none).
 * One step: seq i64 [B,L] (pad token = n_tokens), spk i64 [B], target f32 [B,L] (pad_value marks padding) -- device
 *   pointers; loss_out: device f32[1]; parameters, Adam moments and running statistics live in the handle.
 * ------------------------------------------------------------------------- */
typedef struct dissc_trainer* dissc_trainer_t;
int dissc_train_create(int kind, const DisscTensor* tensors, size_t n, dissc_trainer_t* out);
void dissc_train_destroy(dissc_trainer_t t);
int dissc_train_set_len_norm(dissc_trainer_t t, float mean, float std);          /* LenPredictor.norm_mean / norm_std */
int dissc_train_set_pitch_stats(dissc_trainer_t t, const float* id2mean, const float* id2std, int n);  /* host arrays */
size_t dissc_train_workspace_bytes(dissc_trainer_t t, int B, int L);
int dissc_train_step(dissc_trainer_t t, const int64_t* seq, const int64_t* spk, const float* target, const float* keep,
                     const float* pe_mult, int B, int L, float pad_value, float lr, float* loss_out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* state access: tensors in a fixed order (trainable ones first); which = 0 value, 1 gradient of the last step */
int dissc_train_num_tensors(dissc_trainer_t t);
const char* dissc_train_tensor_name(dissc_trainer_t t, int i);
long long dissc_train_tensor_numel(dissc_trainer_t t, int i);
long long dissc_train_steps(dissc_trainer_t t);
int dissc_train_read(dissc_trainer_t t, int i, int which, float* host_out, void* stream);
/* diagnostics: an activation buffer of the last step ([B][C][ld], ld = L rounded up to 4): which 0 conv output,
 * 1 activation, 2 gradient w.r.t. the activation, 3 gradient w.r.t. the conv output; layer -1 = the embedding */
int dissc_train_debug_read(dissc_trainer_t t, int layer, int which, float* host_out, size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISSC_HIP_H */
