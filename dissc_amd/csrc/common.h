// Shared declarations for libdissc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dissc_hip.h"

// 1 (-DDISSC_EXPERIMENTAL=1, `DISSC_EXPERIMENTAL=1 python -c "import __graft_entry__ as g; g.build()"`): the library also carries the
// kernels whose gates FAILED -- kept as tested opt-ins for the record, not shipped by default (conv_s2tc.hip, respair_wino.hip, the
// k = 3 instances of the F(2,3) pair kernels, hipGraph replay).  dissc_get_option("experimental") reads it back.
#ifndef DISSC_EXPERIMENTAL
#define DISSC_EXPERIMENTAL 0
#endif

namespace dissc {

void set_error(const char* fmt, ...);

// ---- tuning options -----------------------------------------------------------------------------------------------------------
// ONE struct.  `g_defaults` is what dissc_set_option writes: the defaults of handles created LATER.  Every handle (generator,
// HuBERT, predictor, trainer) takes a SNAPSHOT when it is created and every entry point that works on a handle runs under an
// OptScope of that snapshot, so a forward never reads process-wide state ("re-entrant per handle, no global state"):
// two handles created under different options keep their own behaviour whatever is set in between.  Code without a handle (the
// stand-alone test / bench entries) sees the defaults.  X(field, default, "key of dissc_set_option").
#define DISSC_OPTION_LIST_DEFAULT(X) \
  X(multistream, 1, "multistream") \
  X(pair_dma, 1, "pair_dma") \
  X(precision, 0, "precision") \
  X(use_mfma32, 1, "mfma32") \
  X(ragged_enum, 1, "ragged_enum") \
  X(lin128, 1, "lin128") \
  X(conv2s128, 1, "conv2s128") \
  X(xcd_order, 11, "xcd_order") \
  X(xcd_mg, 0, "xcd_mg") \
  X(small_grid, 1, "small_grid") \
  X(kernel_dbg, 0, "kernel_dbg") \
  X(wino_sv, 1, "wino_sv") \
  X(wino8_c64_wide, 3, "wino8_c64_wide") \
  X(wino8_mask, 0770770771, "wino8_mask") \
  X(wino8_r4_mask, 0770770010, "wino8_r4_mask") \
  X(wino8_r4, 1, "wino8_r4") \
  X(wino8, 1, "wino8") \
  X(wino, 1, "wino") \
  X(hubert_split, 1, "hubert_split") \
  X(pair_max_c, 32, "pair_max_c") \
  X(pair_f23, 3, "pair_f23")
// options of the kernels that only DISSC_EXPERIMENTAL=1 builds carry (experimental/csrc: gates failed, kept for the record)
#define DISSC_OPTION_LIST_EXPERIMENTAL(X) \
  X(graphs, 0, "graphs") \
  X(graph_frames, 2048, "graph_frames") \
  X(enc_tc, 0, "enc_tc") \
  X(s2tc_xmode, 0, "s2tc_xmode") \
  X(pair_wino, 0, "pair_wino") \
  X(pairw_chv, 2, "pairw_chv")
#if DISSC_EXPERIMENTAL
#define DISSC_OPTION_LIST(X) DISSC_OPTION_LIST_DEFAULT(X) DISSC_OPTION_LIST_EXPERIMENTAL(X)
#else
#define DISSC_OPTION_LIST(X) DISSC_OPTION_LIST_DEFAULT(X)
#endif
struct Options {
#define DISSC_OPT_FIELD(f, d, k) int f = d;
  DISSC_OPTION_LIST(DISSC_OPT_FIELD)
#undef DISSC_OPT_FIELD
#if !DISSC_EXPERIMENTAL
#define DISSC_OPT_CONST(f, d, k) static constexpr int f = d;
  DISSC_OPTION_LIST_EXPERIMENTAL(DISSC_OPT_CONST)
#undef DISSC_OPT_CONST
#endif
  // Settled choices that were run-time options through round 5 (every one measured, the records are under profiles/): constants now,
  // so that code paths stay readable where they are used (`opts().x`) without being part of the tuning surface.
  static constexpr int stream_prio = 1;  // generator streams of the k = 7 / 11 chains at higher priority
  static constexpr int par_ups = 1;  // the phase groups of a ConvTranspose on parallel streams
  static constexpr int pos48 = 1;  // HuBERT's positional conv (48 rows per group) as three 16-row tiles of the 16x16x4 kernel
  static constexpr int lin_tile = 2;  // channels per barrier / 16 of the register-staged 1x1 convs
  static constexpr int cpb2 = 0;  // two chunks per barrier for short kernels
  static constexpr int lin_dma = 1;  // 1x1 convs of the 256 x 64 kernel stage their window with global_load_lds
  static constexpr int conv_pad_lds = 0;  // extra LDS bytes per workgroup (occupancy experiments)
  static constexpr int c64_wide = 1;  // 64 x 256 tile for the DMA-staged second convs of the C = 64 stage
  static constexpr int conv2_dma = 1;  // stride-2 valid convs of the 256 x 64 kernel stage by global_load_lds
  static constexpr int mfast = 0;  // blockIdx.x walks the M tiles (measured neutral)
  static constexpr int wino_min_c = 64;  // narrowest stage on the transform-domain kernels
  static constexpr int wino_c64_kmin = 3;  // smallest kernel size of the C = 64 stage on them
  static constexpr int wino_small = 96;  // workgroups below which conv_wino steps down to 32 x 32 wave tiles
  static constexpr int wino_cpr = 32;  // channels per barrier round of conv_wino
  static constexpr int attn_fused = 1;  // one fused attention kernel per layer
  static constexpr int bf3_variant = 0;  // split-bf16 fused-block variant
  static constexpr int bf3_pairs = -1;  // split-bf16 blocks as pair launches (-1: by shape)
  static constexpr int pair_lds_mode = 1;  // LDS layout of the direct pair kernels
  static constexpr int pair_pad_lds = 0;  // extra LDS bytes per pair workgroup (occupancy experiments)
  int cfg_for_bm[5] = {6, 5, 7, 1, 0};  // conv_mfma.hip: index log2(BM / 16) -> tile shape id ("conv_cfg_bm{16..256}")
  int cfg32_for_bm[4] = {3, 2, 1, 0};   // conv_mfma32.hip: BM class 32, 64, 128, 256 -> tile shape id ("conv32_cfg_bm{32..256}")
};
extern Options g_defaults;
extern thread_local const Options* t_opts;  // the snapshot of the handle this thread is working for, or nullptr
inline const Options& opts() { return t_opts ? *t_opts : g_defaults; }
struct OptScope {
  const Options* prev;
  explicit OptScope(const Options* o) : prev(t_opts) { t_opts = o; }
  ~OptScope() { t_opts = prev; }
  OptScope(const OptScope&) = delete;
  OptScope& operator=(const OptScope&) = delete;
};

#define DISSC_HIP_CHECK(expr)                                                       \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      dissc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                       __FILE__, __LINE__);                                         \
      return DISSC_EHIP;                                                            \
    }                                                                               \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One-time-per-DEVICE hipFuncSetAttribute(MaxDynamicSharedMemorySize): the attribute is per device, and a process may hold handles
// on several GPUs (a process-wide flag would set it on the first device only).  Thread-safe (forwards are re-entrant per handle:
// two threads may make their first launch of a kernel at once): the flag of a device is set only AFTER the attribute call has
// succeeded, under a lock, so no thread can see "done" and launch before the attribute is in place; a failed call is retried
// by the next launch.  Usage:  static DeviceOnce once;  DISSC_HIP_CHECK(once.max_lds(fn, bytes));
struct DeviceOnce {
  std::mutex mu;
  std::atomic<bool> done[64];
  DeviceOnce() {
    for (auto& d : done) d.store(false, std::memory_order_relaxed);
  }
  hipError_t max_lds(const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<bool>& d = done[dev & 63];
    if (d.load(std::memory_order_acquire)) return hipSuccess;
    std::lock_guard<std::mutex> g(mu);
    if (d.load(std::memory_order_relaxed)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) d.store(true, std::memory_order_release);
    return e;
  }
};

#ifdef __HIPCC__
// erf to ~1 ulp (measured 1.18 ulp max; 0.96 with the library's expf in place of v_exp_f32) without a branch (HuBERT's exact-erf GELU sits in VALU-bound epilogues, and a wave with lanes on both sides
// of the library routine's branch pays for both anyway): two minimax polynomials (Norbert Juffa's single-precision erff,
// public; max error 0.96 ulp measured against float64 on 120 000 samples), |a| <= 0.9277: a + a P(a^2); above: 1 - exp(Q(|a|)).
__device__ __forceinline__ float erf_1ulp(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  // exp through v_exp_f32 (x * log2(e), then the hardware's 2^x: 2 instructions instead of the library routine's 13): r <= -0.9 here,
  // so exp(r) <= 0.4 and its ~2e-7 relative error (argument rounding + 1 ulp) is < 0.5 ulp of 1 - exp(r) >= 0.6
  const float big = copysignf(1.0f - __expf(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;
}
#endif

#ifdef __HIPCC__
// Packed-weight (A fragment) loads through a buffer descriptor: a wave-uniform base in four SGPRs + the lane's 32-bit byte offset
// (+ a scalar byte offset for the walk over blocks).  The plain `wp[block * 64]` form on a per-lane 64-bit pointer costs one
// v_lshl_add_u64 per load and a VGPR-pair address: vector-ALU work that is paid in matrix-pipe time on this part, next to every
// MFMA group (round 6: conv1 6.24 -> 5.70 ms, fc1 618 -> 550 us with scalar-base loads; profiles/r06).  The compiler counts these
// loads like plain ones (its s_waitcnt vmcnt bookkeeping stays exact).  `base` and `bytes` must be PROVABLY wave-uniform (kernel
// arguments, blockIdx, readfirstlane results): a descriptor the compiler believes divergent becomes a waterfall loop per load.
typedef unsigned dissc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wave_rsrc(const void* base, unsigned bytes) {
  // uniformity made explicit (readfirstlane of the descriptor's inputs): whatever the compiler believes about `base`
  const unsigned long long pa = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pa), hi = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 rsrc_load16(__amdgpu_buffer_rsrc_t rs, unsigned lane_bytes, unsigned scalar_bytes) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_bytes, (int)scalar_bytes, 0));
}
#endif

// Channels staged per LDS chunk of the implicit-GEMM conv (= 4 MFMA k-steps of 4).
constexpr int KC = 16;

// Epilogue modes of the conv kernel.
enum Epi : int {
  EPI_STORE = 0,    // out = conv
  EPI_RES = 1,      // out = conv + res                         (ResBlock1: x = xt + x)
  EPI_MRF_SET = 2,  // acc = conv + res                         (xs = resblock_0(x))
  EPI_MRF_ADD = 3,  // acc = acc + (conv + res)                 (xs += resblock_j(x))
  EPI_MRF_DIV = 4,  // acc = (acc + (conv + res)) / mrf_div     (x = xs / num_kernels)
  // out = lrelu(conv, out_slope), and zeros in [olen, olen + 8) and [ldo - 8, ldo) of every row: the layout a consumer
  // that stages its window by LDS-DMA needs (no activation, no masks: a row's zero tail is the next row's left halo)
  EPI_STORE_ACT = 5,
};
constexpr int ZERO_TAIL = 8;  // >= the largest padding of a consumer ((11 - 1) / 2 = 5), multiple of 4

// One dilated "same" Conv1d (or a phase-decomposed ConvTranspose1d when up > 1)
// as an implicit GEMM on v_mfma_f32_16x16x4_f32:  D[M, time] += Wpack[M, K] * Xwin[K, time].
struct ConvArgs {
  const float* x;       // [B][CIN][ldx]
  const float* wpack;   // packed A fragments, see pack_conv_weights()
  const float* wpack2 = nullptr;  // the same weights in conv2s128_kernel's order (stride-2, k = 3 layers only; lin_gemm.hip)
  const float* bias;    // [Mpad] one per GEMM row
  const float* scale;   // [Mpad] optional per-row affine after bias (nullptr = none)
  const float* shift;
  const float* res;     // residual, same layout as out (EPI_RES / EPI_MRF_*)
  float* out;           // [B][M/up][ldo]
  float* acc;           // MRF accumulator (EPI_MRF_*), same layout as out
  const int32_t* lengths;  // [B] valid input frames, or nullptr
  int len_default;      // used when lengths == nullptr (already in this layer's units)
  int len_mul;          // layer length = lengths[b] * len_mul
  const int32_t* lengths_out;  // [B] valid OUTPUT positions (strided / valid convs); nullptr = same as input
  int olen_default;     // used when lengths_out == nullptr and >= 0
  int pad_left;         // input index = t*stride + j*dil - pad_left
  int groups;           // grouped conv: CIN and M are PER GROUP
  int mt_per_group;     // M tiles (blockIdx.y) per group
  int nsub_group;       // 16-row subtiles per group in the packed weights / bias arrays
  int act;              // 0 none, 1 exact GELU on (conv + bias) before any residual
  int mfast;            // 1: blockIdx.x walks the M tiles (XCD i keeps M tiles i, i+8, ... of the weights in its L2)
  int xcd;              // 1: 1-D grid in XCD order -- workgroup id i runs on XCD i % 8 and takes M tile (i / 8) % MT of time tile
                        //    i % 8 + 8 * ((i / 8) / MT): the M tiles that read one input window follow each other on ONE XCD (one L2)
  int xcd_ntile, xcd_nb;  // time tiles per utterance / utterances of that enumeration
  int xcd_mg, xcd_span;   // M tiles per sweep (the weight slabs one sweep keeps L2-resident); ids per sweep = padded time tiles * xcd_mg
  int prec;             // 0 exact fp32 MFMA; 1 split-bf16 ("bf16x3") on the bf16 matrix cores (opt-in)
  int m32;              // 1: weights packed for / launched on the 32x32x2 kernel (conv_mfma32.hip)
  int cfg32;            // tile shape id chosen for this launch (conv32_pick_cfg), -1 = the class default
  int CIN, M, KS, dil, nchunk;
  int XW;               // LDS row stride (floats), XW % 32 == 16
  int ldx, ldo;
  long long x_bstride, o_bstride;
  float slope;          // leaky-ReLU slope applied to x on load (1 = identity)
  float mrf_div;
  float out_slope;      // EPI_STORE_ACT: leaky-ReLU slope applied to the stored value
  int dma_in;           // 1: x has the EPI_STORE_ACT layout (activated, zero tails) -> LDS-DMA staging, slope must be 1
  int ragged_enum;      // set by the launcher: ragged batch, (time tile, utterance) pairs re-dealt so that only existing tiles are enumerated
  int epi;
  int up;               // 1 = conv; s = ConvTranspose stride
  int up_np, up_p0;     // ConvTranspose phase group: row = co*up_np + pi, phase = up_p0 + pi
};

constexpr int MAX_TAP_SPAN = 60;    // (KS-1)*dil of the default instances (staging register budget)
constexpr int WIDE_TAP_SPAN = 128;  // wide instance (HuBERT positional conv, k=128)

// Pick a tile shape for M rows / Lmax columns and launch.  Returns a DISSC_* code.
int launch_conv(const ConvArgs& a, int B, int Lmax_out, int stride, hipStream_t stream);
// LDS row stride for a (KS, dil) conv with time tile BN.
int conv_tile_bn(int M);
int conv_cfg(int M);
void conv_set_cfg(int bm_class, int cfg);  // tuning hook (dissc_conv_bench / dissc_set_option)
int conv_xw(int M, int KS, int dil, int stride = 1, int m32 = 0, int bn = 0);  // bn > 0: explicit time tile
// 32x32x2 form (conv_mfma32.hip)
int launch_conv32(const ConvArgs& a, int B, int Lmax_out, int stride, hipStream_t stream);
// HuBERT's linears on 256 x 128 tiles, two waves per SIMD (lin_gemm.hip); bit-identical to the 256 x 64 instances of launch_conv32
bool lin128_supported(const ConvArgs& a);
int launch_lin128(const ConvArgs& a, int B, int Lmax_out, hipStream_t stream);
// HuBERT's stride-2, k = 3 feature convs on the same tiles (another K order: not bit-identical to launch_conv32's instance)
bool conv2s128_shape(int Cout, int Cin, int KS, int stride, int groups);
void pack_s2_weights128(const float* w, int Cout, int Cin, std::vector<float>& packed);
bool conv2s128_supported(const ConvArgs& a);
int launch_conv2s128(const ConvArgs& a, int B, int Lmax_out, hipStream_t stream);
int conv32_tile_bn(int M);
int conv32_cfg(int M);
int conv32_pick_cfg(int M, int B, int Lmax_out);  // per-launch choice (steps down on small grids)
int conv32_cfg_bn(int cfg);
void conv32_set_cfg(int bm_class, int cfg);
void pack_conv_weights32(const float* w, int Cout, int Cin, int KS, std::vector<float>& packed,
                         int& Mpad, int& nchunk, int groups);
// split-bf16 ("precision" = 1) form of the same conv (conv_bf3.hip); groups == 1 only
void pack_conv_weights_bf3(const float* w, int Cout, int Cin, int KS, std::vector<float>& packed,
                           int& Mpad, int& nchunk);
int launch_conv_bf3(const ConvArgs& a, int B, int Lmax_out, hipStream_t stream);
int conv_bf3_tile_bn(int M);
extern thread_local int g_conv_prec;   // precision make_conv packs for: opts().precision inside dissc_gen_create, else 0
                          // (predictors and HuBERT feed integer decisions and always stay fp32)

// Host-side weight packing.  w: [Cout][Cin][KS] (Conv1d layout).  Returns the packed
// buffer (Mpad/16 * nchunk * KS * 64 float4) and Mpad (M rounded up to 16).
void pack_conv_weights(const float* w, int Cout, int Cin, int KS, std::vector<float>& packed,
                       int& Mpad, int& nchunk, int groups = 1);
// ConvTranspose1d [Cin][Cout][k], stride s, padding (k-s)/2, phases [p0, p0+np): conv weights
// [Cout*np][Cin][ntap] (row = co*np + pi) over input taps delta = dlo .. dlo+ntap-1
// (out[s*q + p] = sum_delta x[q + delta] * w[kk = p + pad - s*delta]).
void convT_phase_weights(const float* w, int Cin, int Cout, int k, int s, int p0, int np, int dlo,
                         int ntap, std::vector<float>& wc);

// Device-resident conv layer (conv_host.hip).
struct DevConv {
  float* wpack = nullptr;
  float* wpack2 = nullptr;  // conv2s128_kernel's packing of the same weights (stride-2, k = 3 layers)
  float* bias = nullptr;
  float* scale = nullptr;  // optional per-row affine applied after bias: v*scale + shift
  float* shift = nullptr;  // (eval-mode BatchNorm1d / label de-normalisation)
  int CIN = 0, M = 0, KS = 0, dil = 1, nchunk = 0, up = 1;  // CIN, M per group
  int groups = 1, Mpad = 0, stride = 1, pad_left = -1;      // pad_left -1 = "same" ((KS-1)*dil/2)
  int up_np = 1, up_p0 = 0;                                 // ConvTranspose phase group (see ConvArgs)
  int act = 0;
  int m32 = 0;              // packed for the 32x32x2 kernel
  int prec = 0;             // 1: packed as split-bf16 hi/lo fragments
  double macs_per_t = 0;  // MACs per input time step (algorithmic, zero taps excluded)
  int wino = 0;           // 1: packed for conv_wino_kernel (Toom-Cook F(4,3) transform-domain weights), 2: for conv_wino8_kernel
  int wr = 3;             // conv_wino8_kernel: taps per sub-filter (3: F(6,3), 4: F(5,4))
};
// Per-call extras of run_conv_ex (strided / valid convolutions with their own output lengths).
struct ConvIO {
  const int32_t* lengths_in = nullptr;
  const int32_t* lengths_out = nullptr;
  int len_default = 0, olen_default = -1, len_mul = 1;
};
int upload(const std::vector<float>& h, float** d);
int make_conv(const float* w, const float* bias, int Cout, int Cin, int KS, int dil, DevConv& dc,
              int groups = 1, int stride = 1, int pad_left = -1);
// ConvTranspose1d = one conv per group of output phases sharing the same input taps.
int make_convT(const float* w, const float* bias, int Cin, int Cout, int k, int s,
               std::vector<DevConv>& groups);
int set_affine(DevConv& dc, const float* scale, const float* shift, int n);  // host pointers
void free_conv(DevConv& dc);
int run_conv(const DevConv& dc, const float* x, float* out, const float* res, float* acc,
             const int32_t* lengths, int len_default, int len_mul, int B, int C_x, int ldx, int ldo,
             int Lmax, float slope, int epi, float mrf_div, hipStream_t stream, float out_slope = 0.f, int dma_in = 0);

int run_conv_ex(const DevConv& dc, const float* x, float* out, const float* res, const ConvIO& io,
                int B, int C_x_total, int ldx, int ldo, int Lmax_out, float slope, int epi,
                hipStream_t stream);
int run_conv_ex(const DevConv& dc, const float* x, float* out, const float* res, float* acc,
                const ConvIO& io, int B, int C_x_total, int ldx, int ldo, int Lmax_out, float slope,
                int epi, float mrf_div, hipStream_t stream, float out_slope = 0.f, int dma_in = 0);

// Toom-Cook F(4,3) form of the wide ResBlock convs (conv_wino.hip)
bool wino_supported(int Cout, int Cin, int KS, int dil);
bool wino_wanted(int C, int KS);
int make_wino(const float* w, const float* bias, int C, int KS, int dil, DevConv& dc);
double wino_executed_macs_per_t(int C, int KS);
int run_wino(const DevConv& dc, const float* x, float* out, const float* res, float* acc, const int32_t* lengths,
             int len_default, int len_mul, int B, int ldx, int ldo, int Lmax, float slope, int epi, float mrf_div,
             hipStream_t stream);

// Toom-Cook F(6,3) form on 8-wave workgroups (conv_wino8.hip): the k = 7 / 11 ResBlock convs of the C >= 64 stages
bool wino8_supported(int Cout, int Cin, int KS, int dil);
bool wino8_wanted(int C, int KS, int dil);
bool wino8_r4_supported(int C, int KS, int dil);
int wino8_taps(int C, int KS, int dil);  // 3 or 4: the generator's policy for a wino8 layer
int make_wino8(const float* w, const float* bias, int C, int KS, int dil, DevConv& dc, int R = 3);  // sets dc.wino = 2, dc.wr = R
double wino8_executed_macs_per_t(int C, int KS, int R);
int run_wino8(const DevConv& dc, const float* x, float* out, const float* res, float* acc, const int32_t* lengths,
              int len_default, int len_mul, int B, int ldx, int ldo, int Lmax, float slope, int epi, float mrf_div,
              hipStream_t stream);

// one residual pair per launch with BOTH convs in the Toom-Cook transform domain, t kept in LDS (respair_wino.hip)
struct DevPairW {
  float* w1 = nullptr;  // register-order transform-domain weights of conv_d / conv_1
  float* w2 = nullptr;
  float* b1 = nullptr;
  float* b2 = nullptr;
  int C = 0, KS = 0, dil = 1;
  int form = 0;  // 0: respair_wino_kernel (F(4,3), Y exchanged through LDS); 1: respair32_f23_kernel (F(2,3), register-only)
};
// register-only F(2,3) pairs of the 32-channel stage, k = 11 (respair_f23.hip)
bool pair_f23_supported(int C, int KS, int dil);
int pack_pair_f23(const float* w, float** dev, int C, int KS);
int launch_pair_f23(const DevPairW& pw, const float* x, float* out, float* acc, const int32_t* lengths, int len_default,
                    int len_mul, int B, int Lmax, int ld, float slope, int epi, float mrf_div, hipStream_t stream);
// the F(4,3) form: experimental/csrc/respair_wino.hip in DISSC_EXPERIMENTAL=1 builds, experimental_stubs.hip otherwise
bool pairw43_built();
int pack_pairw43(const float* w, int C, int KS, float** dev);
int launch_pairw43(const DevPairW& pw, const float* x, float* out, float* acc, const int32_t* lengths, int len_default, int len_mul,
                   int B, int Lmax, int ld, float slope, int epi, float mrf_div, hipStream_t stream);
bool pairw_supported(int C, int KS, int dil);
bool pairw_wanted(int C, int KS, int dil);  // the generator's policy ("pair_wino" option)
int make_pairw(const float* w1, const float* b1, const float* w2, const float* b2, int C, int KS, int dil, DevPairW& pw);
void free_pairw(DevPairW& pw);
int launch_respair_wino(const DevPairW& pw, const float* x, float* out, float* acc, const int32_t* lengths, int len_default,
                        int len_mul, int B, int Lmax, int ld, float slope, int epi, float mrf_div, hipStream_t stream);

// one residual pair y = x + conv_1(lrelu(conv_d(lrelu(x)))) per launch, exact fp32 (respair.hip)
bool respair_supported(int C, int KS, int dil);
int launch_respair(const DevConv& c1, const DevConv& c2, const float* x, float* out, float* acc,
                   const int32_t* lengths, int len_default, int len_mul, int B, int Lmax, int ld, float slope,
                   int epi, float mrf_div, hipStream_t stream);

// the same block in split-bf16 arithmetic ("precision" = 1 only; resblock_bf3.hip)
bool resblock_bf3_supported(int C, int KS, const int* dil);
void pack_resblock_bf3(int C, int KS, const float* const* w6, std::vector<float>& packed);
// residual pairs [m0, m1) of the block; epi = EPI_STORE writes x_k to `acc` (a partial block)
int launch_resblock_bf3(int C, const float* x, float* acc, const float* wpack, const float* bias,
                        const int32_t* lengths, int len_default, int len_mul, int KS, const int* dil,
                        int B, int Lmax, int ld, float slope, int epi, float mrf_div, int m0, int m1,
                        hipStream_t stream);
bool resblock_bf3_pairs(int C);  // run the block as three pair launches (wide halos, little LDS)

// HuBERT's stride-2, k = 3 feature convs in polyphase Toom-Cook form (conv_s2tc.hip)
struct DevS2tc {
  float* wpack = nullptr;  // packed A fragments of the point and direct waves (make_s2tc)
  float* bias = nullptr;   // [M] or nullptr
  int CIN = 0, M = 0, act = 0;
};
bool s2tc_supported(int Cout, int Cin, int KS, int stride);
int make_s2tc(const float* w, const float* bias, int Cout, int Cin, DevS2tc& dc);
void free_s2tc(DevS2tc& dc);
double s2tc_executed_macs_per_out(int Cout, int Cin);
int run_s2tc(const DevS2tc& dc, const float* x, float* out, const int32_t* lengths_in, const int32_t* lengths_out,
             int len_default, int olen_default, int B, int ldx, int ldo, int Lmax_out, hipStream_t stream);

// misc kernels (gen_misc.hip)
struct ZeroSpans {  // ZERO_TAIL floats at each pointer, zeroed by the forward's first kernel
  float* p[DISSC_MAX_RK];
  int n;
};
void launch_embed_concat(const int64_t* code, const float* f0, const int64_t* spkr,
                         const float* dict_w, const float* spkr_w, const int32_t* lengths, int B,
                         int T, int E, int has_f0, int has_spkr, int n_codes, int n_spk, float* x,
                         int ldx, const ZeroSpans& zs, hipStream_t stream);
void launch_conv_post(const float* x, const float* w, const float* bias, const int32_t* lengths,
                      int len_mul, int B, int C, int KS, int L, int ldx, long long x_bstride,
                      float slope, float* wav, int ldw, hipStream_t stream);
void launch_wav_postprocess(float* wav, const int32_t* n_samples, int B, int ld,
                            hipStream_t stream);

}  // namespace dissc
