// dissc_gen_*: the unit/F0/speaker-conditioned HiFi-GAN generator on one MI355X.
// Orchestrates ~100 launches of conv_mfma_kernel per forward on the caller's stream;
// all activations live in the caller-provided workspace (no allocation here).
//
// Workspace = 4 buffers of B * max_i(C_i * ld_i) floats:
//   X   stage input (ConvTranspose output), read by the 3 resblocks
//   TMP conv1 output of the current (conv1, conv2) pair
//   XK  running x of the current resblock
//   ACC MRF accumulator; becomes the next stage's ConvTranspose input
// Activations are channels-first [B][C][ld] fp32 (time contiguous => coalesced tile
// loads and MFMA-row stores), ld = stage length rounded up to 4.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <exception>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include <algorithm>

#include "common.h"

namespace dissc {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

Options g_defaults;
thread_local const Options* t_opts = nullptr;

}  // namespace dissc

using namespace dissc;

// option "stream_prio" (Options::stream_prio, default 1): "stream_prio" option: prioritise the longer ResBlock chains
// option "graphs" (Options::graphs, default 0): "graphs" option: replay small forwards from a captured hipGraph.  OFF by default:
                                  // measured on ROCm 7.2 / MI355X, hipGraphLaunch of the ~85-node three-branch graph costs
                                  // 1.3-1.8 ms MORE per forward than the plain three-stream launches (tools/graph_ab.py)
// option "graph_frames" (Options::graph_frames, default 2048): "graph_frames" option: largest B * Tmax that is graphed
[[maybe_unused]] static int g_graph_hits = 0, g_graph_captures = 0;  // diagnostics (dissc_get_option)
// option "pair_dma" (Options::pair_dma, default 1): "pair_dma" option: wide residual pairs hand their intermediate over in the EPI_STORE_ACT layout
// option "multistream" (Options::multistream, default 1): "multistream" option: concurrent ResBlock chains (read at create)
// option "par_ups" (Options::par_ups, default 1): "par_ups" option: ConvTranspose phase groups on concurrent streams

// The side streams of the concurrent ResBlock chains are shared by every generator handle of a
// device (created on first use, kept for the life of the process): HIP multiplexes streams onto a
// few hardware queues (4 by default), and a second handle with streams of its own would push the
// chains of both onto shared queues and serialise them (measured: +12 % per step for whichever
// handle was created second).  Work of different handles on a shared stream is ordered by the same
// events as before, so sharing changes no result.
static hipStream_t shared_aux_stream(int chain, int prio) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int>, hipStream_t> pool;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(dev, chain, prio);
  auto it = pool.find(key);
  if (it != pool.end()) return it->second;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) return nullptr;
  pool[key] = st;
  return st;
}

struct dissc_gen {
  Options opt;  // this handle's snapshot of the tuning options (common.h): every entry point that works on the handle runs under it
  DisscGenConfig cfg;
  int hop = 1;
  DevConv conv_pre;
  std::vector<std::vector<DevConv>> ups;  // per stage: one conv per phase group
  std::vector<DevConv> rb1, rb2;  // [stage*nk*3 + j*3 + m]
  std::vector<DevPairW> pw;       // same index: the pair as ONE transform-domain launch (respair_wino.hip), w1 == nullptr if not
  std::vector<float*> fused_w, fused_b;  // [stage*nk + j]: 6 packed convs / biases of a split-bf16 fused ResBlock
  std::vector<char> fused_bf3;           // ... (resblock_bf3.hip; precision = 1 only)
  float* post_w = nullptr;
  float* post_b = nullptr;
  int post_C = 0, post_KS = 0;
  float* dict_w = nullptr;
  float* spkr_w = nullptr;
  std::vector<int> stage_C, stage_mul;  // channels / length multiplier after ups[i]
  // The num_kernels ResBlocks of a stage are independent until the MRF accumulate: they run as
  // concurrent chains (chain 0 on the caller's stream, the others on these) so bandwidth-bound
  // small-kernel layers overlap with matrix-bound large-kernel ones.
  hipStream_t aux[DISSC_MAX_RK] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_x = nullptr, ev_fin[DISSC_MAX_RK] = {nullptr, nullptr, nullptr, nullptr};
  // Opt-in (option "graphs"): small forwards (the reference's one-utterance-at-a-time mode: ~85 launches of a few us each
  // over three streams, 11 % of the period idle) captured once per (shapes, buffers) into a hipGraph and replayed.
  struct GraphEntry {
    const void *code, *f0, *spkr, *lengths, *out, *ws;
    int B, T;
    hipGraphExec_t exec;
    unsigned long long stamp;
  };
  std::vector<GraphEntry> graphs;
  hipStream_t cap_stream = nullptr;
  unsigned long long graph_clock = 0;
  int graph_failures = 0;  // captures that did not produce a graph: after two, this handle stops trying
  ~dissc_gen() {
    for (auto& e : graphs) (void)hipGraphExecDestroy(e.exec);
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    free_conv(conv_pre);
    for (auto& v : ups) for (auto& c : v) free_conv(c);
    for (auto& c : rb1) free_conv(c);
    for (auto& c : rb2) free_conv(c);
    for (auto& c : pw) free_pairw(c);
    for (float* p : fused_w) if (p) (void)hipFree(p);
    for (float* p : fused_b) if (p) (void)hipFree(p);
    if (post_w) (void)hipFree(post_w);
    if (post_b) (void)hipFree(post_b);
    for (int j = 0; j < DISSC_MAX_RK; ++j)  // aux[] streams belong to the process-wide pool
      if (ev_fin[j]) (void)hipEventDestroy(ev_fin[j]);
    if (ev_x) (void)hipEventDestroy(ev_x);
    if (dict_w) (void)hipFree(dict_w);
    if (spkr_w) (void)hipFree(spkr_w);
  }
};

static inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

extern "C" {

const char* dissc_last_error(void) { return g_err; }
int dissc_abi_version(void) { return 3; }

int dissc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int dissc_device_name(int dev, char* buf, size_t buflen) {
  hipDeviceProp_t p;
  DISSC_HIP_CHECK(hipGetDeviceProperties(&p, dev));
  snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return DISSC_OK;
}

int dissc_gen_create(const DisscGenConfig* cfg, const DisscTensor* weights, size_t n_weights,
                     dissc_gen_t* out) {
  return dissc_gen_create_ex(cfg, weights, n_weights, -1, out);
}

int dissc_gen_create_ex(const DisscGenConfig* cfg, const DisscTensor* weights, size_t n_weights, int precision,
                        dissc_gen_t* out) {
  if (precision < -1 || precision > 1) {
    set_error("dissc_gen_create_ex: precision %d (use -1 = process option, 0 = fp32, 1 = split-bf16)", precision);
    return DISSC_EINVAL;
  }
  const int prec = precision < 0 ? g_defaults.precision : precision;  // this handle's arithmetic, fixed from here on
  if (!cfg || !weights || !out) {
    set_error("dissc_gen_create: null argument");
    return DISSC_EINVAL;
  }
  if (cfg->num_upsamples < 1 || cfg->num_upsamples > DISSC_MAX_UPS || cfg->num_kernels < 1 ||
      cfg->num_kernels > DISSC_MAX_RK) {
    set_error("dissc_gen_create: unsupported num_upsamples=%d / num_kernels=%d", cfg->num_upsamples,
              cfg->num_kernels);
    return DISSC_EINVAL;
  }
  std::map<std::string, const DisscTensor*> byname;
  for (size_t i = 0; i < n_weights; ++i) byname[weights[i].name] = &weights[i];
  auto get = [&](const std::string& name, std::initializer_list<int64_t> shape,
                 const float** p) -> int {
    auto it = byname.find(name);
    if (it == byname.end()) {
      set_error("dissc_gen_create: missing tensor '%s'", name.c_str());
      return DISSC_ENOTFOUND;
    }
    const DisscTensor* t = it->second;
    size_t d = 0;
    bool ok = (size_t)t->ndim == shape.size();
    for (int64_t s : shape) {
      if (ok && t->shape[d] != s) ok = false;
      ++d;
    }
    if (!ok) {
      set_error("dissc_gen_create: tensor '%s' has the wrong shape", name.c_str());
      return DISSC_EINVAL;
    }
    *p = t->data;
    return DISSC_OK;
  };

  dissc_gen* g = new dissc_gen();
  g->opt = g_defaults;          // frozen here: later dissc_set_option calls do not reach this handle
  g->opt.precision = prec;
  OptScope opt_scope(&g->opt);
  g->cfg = *cfg;
  struct PrecScope {  // only the generator's layers may be packed for split-bf16
    explicit PrecScope(int p) { g_conv_prec = p; }
    ~PrecScope() { g_conv_prec = 0; }
  } prec_scope(prec);
  int rc = DISSC_OK;
  const float *w = nullptr, *b = nullptr;
  const int c0 = cfg->upsample_initial_channel;
  const int in_dim = cfg->model_in_dim;
  const int E = cfg->embedding_dim;
  auto fail = [&](int code) {
    delete g;
    return code;
  };
  // Packing a layer's weights (direct fragments, Toom-Cook transforms in double) is host work of ~60 ms for this model when done
  // one layer after the other: the layers are independent, so their make_* calls are collected as tasks and run on a few host
  // threads at the end (run_pack_tasks; DISSC_PACK_THREADS, default min(8, cores); 1 = in place).  Each task writes one
  // pre-sized slot of the handle; a worker carries the creating thread's device, options snapshot and packing precision.
  std::vector<std::function<int()>> tasks;
  auto run_pack_tasks = [&]() -> int {
    // default: min(8, the CPUs this process may really use / ranks on the node) -- hardware_concurrency() knows neither the cgroup
    // CPU quota (the GPU boxes: 16 of 256 logical CPUs) nor that 8 ranks create handles at once
    int nthr = 8;
    {
      double cpus = (double)std::thread::hardware_concurrency();
      if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long per = 0;
        if (fscanf(f, "%31s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
          const double quota = atof(q) / (double)per;
          if (quota >= 1.0 && (cpus <= 0 || quota < cpus)) cpus = quota;
        }
        fclose(f);
      }
      int ranks = 1;
      if (const char* e = getenv("LOCAL_WORLD_SIZE")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= 1024) ranks = (int)v;
      }
      if (cpus >= 1.0 && nthr > (int)(cpus / ranks)) nthr = (int)(cpus / ranks) > 1 ? (int)(cpus / ranks) : 1;
    }
    if (const char* e = getenv("DISSC_PACK_THREADS")) {  // explicit override, validated
      char* end = nullptr;
      const long v = strtol(e, &end, 10);
      if (end != e && *end == 0 && v >= 1 && v <= 256) nthr = (int)v;
    }
    if (nthr > (int)tasks.size()) nthr = (int)tasks.size();
    if (nthr <= 1) {
      for (auto& t : tasks)
        if (int r = t()) return r;
      return DISSC_OK;
    }
    int dev = 0;
    DISSC_HIP_CHECK(hipGetDevice(&dev));
    std::atomic<size_t> next{0};
    std::atomic<int> first_rc{DISSC_OK};
    std::mutex err_mu;
    std::string err_msg;
    auto worker = [&]() {
      if (hipSetDevice(dev) != hipSuccess) {
        first_rc = DISSC_EHIP;
        return;
      }
      OptScope scope(&g->opt);
      g_conv_prec = prec;
      for (size_t i = next++; i < tasks.size() && first_rc == DISSC_OK; i = next++) {
        int r;
        try {
          r = tasks[i]();
        } catch (const std::exception& e) {  // (an exception leaving a std::thread would terminate the process)
          set_error("dissc_gen_create: packing a layer failed: %s", e.what());
          r = DISSC_ENOMEM;
        } catch (...) {
          set_error("dissc_gen_create: packing a layer failed (unknown exception)");
          r = DISSC_ENOMEM;
        }
        if (r != DISSC_OK) {
          std::lock_guard<std::mutex> lk(err_mu);
          if (first_rc == DISSC_OK) {
            first_rc = r;
            err_msg = g_err;  // the worker's thread-local message
          }
        }
      }
      g_conv_prec = 0;
    };
    std::vector<std::thread> pool;
    try {
      for (int t = 0; t + 1 < nthr; ++t) pool.emplace_back(worker);
    } catch (const std::exception&) {  // no more threads to be had: whoever started shares the tasks with this thread
    }
    worker();  // the creating thread packs too (and alone, if no thread could be started)
    for (auto& t : pool) t.join();
    g_conv_prec = prec;  // (worker() cleared this thread's copy; the bf3 tail below and PrecScope's exit expect it set)
    if (first_rc != DISSC_OK) set_error("%s", err_msg.empty() ? "dissc_gen_create: a packing thread failed" : err_msg.c_str());
    return first_rc;
  };
  if (in_dim != E + (cfg->has_f0 ? 1 : 0) + (cfg->has_spkr ? E : 0)) {
    set_error("dissc_gen_create: model_in_dim %d != embedding_dim %d (+1 f0) (+%d spkr)", in_dim, E, E);
    return fail(DISSC_EINVAL);
  }
  if ((rc = get("conv_pre.weight", {c0, in_dim, 7}, &w))) return fail(rc);
  if ((rc = get("conv_pre.bias", {c0}, &b))) return fail(rc);
  tasks.push_back([=]() { return make_conv(w, b, c0, in_dim, 7, 1, g->conv_pre); });

  int ch = c0, mul = 1;
  g->ups.resize(cfg->num_upsamples);
  const int nk = cfg->num_kernels;
  g->rb1.resize((size_t)cfg->num_upsamples * nk * 3);
  g->rb2.resize((size_t)cfg->num_upsamples * nk * 3);
  g->pw.resize((size_t)cfg->num_upsamples * nk * 3);
  g->fused_w.assign((size_t)cfg->num_upsamples * nk, nullptr);
  g->fused_b.assign((size_t)cfg->num_upsamples * nk, nullptr);
  g->fused_bf3.assign((size_t)cfg->num_upsamples * nk, 0);
  for (int i = 0; i < cfg->num_upsamples; ++i) {
    const int s = cfg->upsample_rates[i], k = cfg->upsample_kernel_sizes[i];
    if (s < 1 || k < s || (k - s) % 2 != 0) {  // L_out = s * L_in needs k - s even
      set_error("dissc_gen_create: upsample k=%d s=%d unsupported", k, s);
      return fail(DISSC_EINVAL);
    }
    const int cout = ch / 2;
    char name[96];
    snprintf(name, sizeof(name), "ups.%d.weight", i);
    if ((rc = get(name, {ch, cout, k}, &w))) return fail(rc);
    snprintf(name, sizeof(name), "ups.%d.bias", i);
    if ((rc = get(name, {cout}, &b))) return fail(rc);
    tasks.push_back([=]() { return make_convT(w, b, ch, cout, k, s, g->ups[i]); });
    ch = cout;
    mul *= s;
    g->stage_C.push_back(ch);
    g->stage_mul.push_back(mul);
    for (int j = 0; j < nk; ++j) {
      const int rk = cfg->resblock_kernel_sizes[j];
      if (rk % 2 != 1) {
        set_error("dissc_gen_create: even resblock kernel size %d unsupported", rk);
        return fail(DISSC_EINVAL);
      }
      const bool bf3 = prec == 1 && resblock_bf3_supported(ch, rk, cfg->resblock_dilations[j]);
      std::vector<float> fw, fb;
      const float* w6[6];
      for (int m = 0; m < 3; ++m) {
        const int d = cfg->resblock_dilations[j][m];
        const size_t idx = ((size_t)i * nk + j) * 3 + m;
        snprintf(name, sizeof(name), "resblocks.%d.convs1.%d.weight", i * nk + j, m);
        if ((rc = get(name, {ch, ch, rk}, &w))) return fail(rc);
        snprintf(name, sizeof(name), "resblocks.%d.convs1.%d.bias", i * nk + j, m);
        if ((rc = get(name, {ch}, &b))) return fail(rc);
        const bool wino = prec == 0 && wino_wanted(ch, rk) && wino_supported(ch, ch, rk, d);
        const bool w8 = wino && wino8_wanted(ch, rk, d);  // the eight-point forms on 8-wave workgroups (per shape: conv_wino8.hip)
        {
          const int taps1 = w8 ? wino8_taps(ch, rk, d) : 0;
          tasks.push_back([=]() {
            return w8 ? make_wino8(w, b, ch, rk, d, g->rb1[idx], taps1)
                      : wino ? make_wino(w, b, ch, rk, d, g->rb1[idx]) : make_conv(w, b, ch, ch, rk, d, g->rb1[idx]);
          });
        }
        const float* w1c = w;
        const float* b1c = b;
        if (bf3) {
          w6[2 * m] = w;
          fb.insert(fb.end(), b, b + ch);
        }
        snprintf(name, sizeof(name), "resblocks.%d.convs2.%d.weight", i * nk + j, m);
        if ((rc = get(name, {ch, ch, rk}, &w))) return fail(rc);
        snprintf(name, sizeof(name), "resblocks.%d.convs2.%d.bias", i * nk + j, m);
        if ((rc = get(name, {ch}, &b))) return fail(rc);
        const bool wino2 = wino;
        const bool w82 = wino && wino8_wanted(ch, rk, 1);
        {
          const int taps2 = w82 ? wino8_taps(ch, rk, 1) : 0;
          tasks.push_back([=]() {
            return w82 ? make_wino8(w, b, ch, rk, 1, g->rb2[idx], taps2)
                       : wino2 ? make_wino(w, b, ch, rk, 1, g->rb2[idx]) : make_conv(w, b, ch, ch, rk, 1, g->rb2[idx]);
          });
        }
        // the whole pair as one transform-domain launch (respair_wino.hip): C = 32, k = 7 / 11 and C = 64, k = 3
        // (C = 64: only a chain's FIRST pair -- the later ones update x_k in place, which a fused pair cannot)
        if (prec == 0 && opts().wino && pairw_wanted(ch, rk, d) && (ch > 32 ? (wino && m == 0) : ch <= opts().pair_max_c))
          tasks.push_back([=]() { return make_pairw(w1c, b1c, w, b, ch, rk, d, g->pw[idx]); });
        if (bf3) {
          w6[2 * m + 1] = w;
          fb.insert(fb.end(), b, b + ch);
        }
      }
      if (bf3) {
        pack_resblock_bf3(ch, rk, w6, fw);
        g->fused_bf3[(size_t)i * nk + j] = 1;
      }
      if (bf3) {
        if ((rc = upload(fw, &g->fused_w[(size_t)i * nk + j]))) return fail(rc);
        if ((rc = upload(fb, &g->fused_b[(size_t)i * nk + j]))) return fail(rc);
      }
    }
  }
  g->hop = mul;
  if ((rc = get("conv_post.weight", {1, ch, 7}, &w))) return fail(rc);
  if ((rc = get("conv_post.bias", {1}, &b))) return fail(rc);
  g->post_C = ch;
  g->post_KS = 7;
  if ((rc = upload(std::vector<float>(w, w + ch * 7), &g->post_w))) return fail(rc);
  if ((rc = upload(std::vector<float>(b, b + 1), &g->post_b))) return fail(rc);
  if ((rc = get("dict.weight", {cfg->num_embeddings, E}, &w))) return fail(rc);
  if ((rc = upload(std::vector<float>(w, w + (size_t)cfg->num_embeddings * E), &g->dict_w)))
    return fail(rc);
  if (cfg->has_spkr) {
    if ((rc = get("spkr.weight", {cfg->num_speakers, E}, &w))) return fail(rc);
    if ((rc = upload(std::vector<float>(w, w + (size_t)cfg->num_speakers * E), &g->spkr_w)))
      return fail(rc);
  }
  if ((rc = run_pack_tasks())) return fail(rc);
  if (opts().multistream && nk > 1) {
    if (hipEventCreateWithFlags(&g->ev_x, hipEventDisableTiming) != hipSuccess) return fail(DISSC_EHIP);
    for (int j = 0; j < nk; ++j) {
      if (hipEventCreateWithFlags(&g->ev_fin[j], hipEventDisableTiming) != hipSuccess) return fail(DISSC_EHIP);
      // later chains have larger kernels (k = 3 < 7 < 11): the longest one is the stage's critical
      // path, so it gets the highest stream priority and the short chains fill in around it
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent, hi = most urgent (numerically lower)
      int prio = lo;
      if (opts().stream_prio && nk > 1) prio = lo + (hi - lo) * j / (nk - 1);
      if (j > 0 && !(g->aux[j] = shared_aux_stream(j, prio))) return fail(DISSC_EHIP);
    }
  }
  *out = g;
  return DISSC_OK;
}

void dissc_gen_destroy(dissc_gen_t g) { delete g; }

int dissc_gen_hop(dissc_gen_t g) { return g ? g->hop : 0; }

constexpr size_t BUF_GUARD = 64;  // floats in front of the chains' scratch buffers (their last ZERO_TAIL are kept zero)
static size_t gen_buf_floats(const dissc_gen* g, int B, int Tmax) {
  // rows may carry a zero tail (EPI_STORE_ACT layout): ZERO_TAIL extra columns
  size_t per = (size_t)g->cfg.upsample_initial_channel * round_up(Tmax + ZERO_TAIL, 4);
  per = std::max(per, (size_t)g->cfg.model_in_dim * round_up(Tmax + ZERO_TAIL, 4));
  for (size_t i = 0; i < g->stage_C.size(); ++i)
    per = std::max(per, (size_t)g->stage_C[i] * round_up((size_t)Tmax * g->stage_mul[i] + ZERO_TAIL, 4));
  return round_up(per * B + BUF_GUARD + 512, 64);  // + slack: LDS-DMA windows run up to ~130 floats past a row
}

size_t dissc_gen_workspace_bytes(dissc_gen_t g, int B, int Tmax) {
  if (!g || B <= 0 || Tmax <= 0) return 0;
  OptScope opt_scope(&g->opt);
  return (size_t)(2 + 2 * g->cfg.num_kernels) * gen_buf_floats(g, B, Tmax) * sizeof(float) + 256;
}

double dissc_gen_flops(dissc_gen_t g, int64_t frames) {
  if (!g) return 0;
  double macs = g->conv_pre.macs_per_t;
  int mul = 1;
  const int nk = g->cfg.num_kernels;
  for (int i = 0; i < g->cfg.num_upsamples; ++i) {
    for (auto& c : g->ups[i]) macs += c.macs_per_t * mul;
    mul = g->stage_mul[i];
    for (int j = 0; j < nk * 3; ++j)
      macs += (g->rb1[(size_t)i * nk * 3 + j].macs_per_t + g->rb2[(size_t)i * nk * 3 + j].macs_per_t) * mul;
  }
  macs += (double)g->post_C * g->post_KS * mul;
  return 2.0 * macs * (double)frames;
}

// multiply-adds the matrix pipe actually executes: the layers that run in the Toom-Cook transform domain (conv_wino.hip)
// do 6 ceil(k / 3) / 4 products per output and channel pair instead of k
double dissc_gen_flops_executed(dissc_gen_t g, int64_t frames) {
  if (!g) return 0;
  double macs = g->conv_pre.macs_per_t;
  int mul = 1;
  const int nk = g->cfg.num_kernels;
  auto ex = [](const DevConv& c) {
    return c.wino == 2 ? wino8_executed_macs_per_t(c.M, c.KS, c.wr) : c.wino ? wino_executed_macs_per_t(c.M, c.KS) : c.macs_per_t;
  };
  for (int i = 0; i < g->cfg.num_upsamples; ++i) {
    for (auto& c : g->ups[i]) macs += c.macs_per_t * mul;
    mul = g->stage_mul[i];
    for (int j = 0; j < nk * 3; ++j) {
      const size_t idx = (size_t)i * nk * 3 + j;
      if (g->pw[idx].w1)  // the pair runs as one transform-domain launch (respair_wino.hip)
        macs += (g->pw[idx].form == 1 ? 2.0 * g->pw[idx].C * g->pw[idx].C * 2.0 * ((g->pw[idx].KS + 2) / 3)   // F(2,3): 4 products per 2 outputs and sub-filter
                                      : 2.0 * wino_executed_macs_per_t(g->pw[idx].C, g->pw[idx].KS)) * mul;
      else
        macs += (ex(g->rb1[idx]) + ex(g->rb2[idx])) * mul;
    }
  }
  macs += (double)g->post_C * g->post_KS * mul;
  return 2.0 * macs * (double)frames;
}

static int gen_forward_body(dissc_gen_t g, const int64_t* code, const float* f0, const int64_t* spkr,
                            const int32_t* lengths, int B, int Tmax, float* wav_out, void* workspace,
                            hipStream_t stream);

int dissc_gen_forward(dissc_gen_t g, const int64_t* code, const float* f0, const int64_t* spkr,
                      const int32_t* lengths, int B, int Tmax, float* wav_out, void* workspace,
                      size_t workspace_bytes, void* stream_) {
  if (!g || !code || !wav_out || !workspace || (g->cfg.has_f0 && !f0) ||
      (g->cfg.has_spkr && !spkr)) {
    set_error("dissc_gen_forward: null argument");
    return DISSC_EINVAL;
  }
  OptScope opt_scope(&g->opt);  // the forward reads this handle's options only
  if (B <= 0 || Tmax <= 0) {
    set_error("dissc_gen_forward: B=%d Tmax=%d", B, Tmax);
    return DISSC_EINVAL;
  }
  if ((long long)Tmax * g->hop > 2000000000LL) {
    set_error("dissc_gen_forward: Tmax=%d too long", Tmax);
    return DISSC_EINVAL;
  }
  if (workspace_bytes < dissc_gen_workspace_bytes(g, B, Tmax)) {
    set_error("dissc_gen_forward: workspace %zu < %zu bytes", workspace_bytes,
              dissc_gen_workspace_bytes(g, B, Tmax));
    return DISSC_ENOMEM;
  }
  hipStream_t stream = (hipStream_t)stream_;
#if !DISSC_EXPERIMENTAL  // hipGraph replay failed its gate (slower than plain launches on ROCm 7.2): DISSC_EXPERIMENTAL=1 builds only
  return gen_forward_body(g, code, f0, spkr, lengths, B, Tmax, wav_out, workspace, stream);
#else
  if (!opts().graphs || g->graph_failures >= 2 || (long long)B * Tmax > opts().graph_frames)
    return gen_forward_body(g, code, f0, spkr, lengths, B, Tmax, wav_out, workspace, stream);
  // ---- small forward: replay (or first capture) its hipGraph ----
  for (auto& e : g->graphs)
    if (e.code == code && e.f0 == f0 && e.spkr == spkr && e.lengths == lengths && e.out == wav_out && e.ws == workspace &&
        e.B == B && e.T == Tmax) {
      e.stamp = ++g->graph_clock;
      ++g_graph_hits;
      DISSC_HIP_CHECK(hipGraphLaunch(e.exec, stream));
      return DISSC_OK;
    }
  if (!g->cap_stream && hipStreamCreateWithFlags(&g->cap_stream, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    return gen_forward_body(g, code, f0, spkr, lengths, B, Tmax, wav_out, workspace, stream);
  }
  // capture on the handle's own stream (the caller's may be the legacy default stream, which cannot capture); the side
  // streams join the capture through the events the body records and waits on, and all rejoin before it ends
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int rc = DISSC_EHIP;
  if (hipStreamBeginCapture(g->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
    rc = gen_forward_body(g, code, f0, spkr, lengths, B, Tmax, wav_out, workspace, g->cap_stream);
    const hipError_t e = hipStreamEndCapture(g->cap_stream, &graph);
    if (rc == DISSC_OK && (e != hipSuccess || !graph)) rc = DISSC_EHIP;
    if (rc == DISSC_OK && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = DISSC_EHIP;
    if (graph) (void)hipGraphDestroy(graph);
  }
  if (rc != DISSC_OK) {  // anything the capture cannot express: run it the plain way
    (void)hipGetLastError();
    ++g->graph_failures;
    return gen_forward_body(g, code, f0, spkr, lengths, B, Tmax, wav_out, workspace, stream);
  }
  // keep the eight most recently used graphs (the handle's options are frozen: a captured graph never goes stale)
  if (g->graphs.size() >= 8) {
    size_t old = 0;
    for (size_t i = 1; i < g->graphs.size(); ++i)
      if (g->graphs[i].stamp < g->graphs[old].stamp) old = i;
    (void)hipGraphExecDestroy(g->graphs[old].exec);
    g->graphs.erase(g->graphs.begin() + old);
  }
  ++g_graph_captures;
  g->graphs.push_back({code, f0, spkr, lengths, wav_out, workspace, B, Tmax, exec, ++g->graph_clock});
  DISSC_HIP_CHECK(hipGraphLaunch(exec, stream));
  return DISSC_OK;
#endif
}

static int gen_forward_body(dissc_gen_t g, const int64_t* code, const float* f0, const int64_t* spkr,
                            const int32_t* lengths, int B, int Tmax, float* wav_out, void* workspace,
                            hipStream_t stream) {
  const size_t nbuf = gen_buf_floats(g, B, Tmax);
  float* base = (float*)round_up((size_t)workspace, 256);
  float* X = base;
  float* ACC = base + nbuf;
  float* TMPj[DISSC_MAX_RK];
  float* XKj[DISSC_MAX_RK];
  ZeroSpans zs;
  for (int j = 0; j < g->cfg.num_kernels; ++j) {
    TMPj[j] = base + (size_t)(2 + 2 * j) * nbuf + BUF_GUARD;
    XKj[j] = base + (size_t)(3 + 2 * j) * nbuf;
    // the floats right in front of a scratch buffer are the left halo of its first row when a conv stages it by LDS-DMA:
    // zeroed by the forward's first kernel (embed_concat, on `stream` before the chains fork)
    zs.p[j] = TMPj[j] - ZERO_TAIL;
  }
  zs.n = g->cfg.num_kernels;
  float* TMP = TMPj[0];
  const bool multi = g->ev_x != nullptr;
  const DisscGenConfig& c = g->cfg;
  int rc;

  // 1. conditioning -> TMP [B, in_dim, ld0]
  const int ld0 = (int)round_up(Tmax, 4);
  launch_embed_concat(code, f0, spkr, g->dict_w, g->spkr_w, lengths, B, Tmax, c.embedding_dim,
                      c.has_f0, c.has_spkr, c.num_embeddings, c.num_speakers, TMP, ld0, zs, stream);
  // 2. conv_pre -> ACC [B, c0, ld0]
  if ((rc = run_conv(g->conv_pre, TMP, ACC, nullptr, nullptr, lengths, Tmax, 1, B, c.model_in_dim,
                     ld0, ld0, Tmax, 1.0f, EPI_STORE, 1.f, stream)))
    return rc;
  int ch = c.upsample_initial_channel, mul = 1, ld = ld0;
  const int nk = c.num_kernels;
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int s = c.upsample_rates[i];
    const int ch_out = ch / 2, mul_out = mul * s;
    const int ld_out = (int)round_up((size_t)Tmax * mul_out, 4);
    // lrelu(0.1) -> ConvTranspose: ACC [B,ch,ld] -> X [B,ch_out,ld_out]
    // (phase groups write disjoint output phases: with side streams they run concurrently, which
    // fills the CUs better than two or three small grids one after the other)
    const int ngrp = (int)g->ups[i].size();
    const bool par_ups = multi && opts().par_ups && ngrp > 1 && ngrp <= nk;
    if (par_ups) DISSC_HIP_CHECK(hipEventRecord(g->ev_x, stream));  // ACC is complete
    for (int gi = 0; gi < ngrp; ++gi) {
      hipStream_t sg = (par_ups && gi > 0) ? g->aux[gi] : stream;
      if (par_ups && gi > 0) DISSC_HIP_CHECK(hipStreamWaitEvent(sg, g->ev_x, 0));
      if ((rc = run_conv(g->ups[i][gi], ACC, X, nullptr, nullptr, lengths, Tmax * mul, mul, B, ch, ld,
                         ld_out, Tmax * mul, 0.1f, EPI_STORE, 1.f, sg)))
        return rc;
      if (par_ups && gi > 0) {
        DISSC_HIP_CHECK(hipEventRecord(g->ev_fin[gi], sg));
        DISSC_HIP_CHECK(hipStreamWaitEvent(stream, g->ev_fin[gi], 0));
      }
    }
    ch = ch_out; mul = mul_out; ld = ld_out;
    const int L = Tmax * mul;
    if (multi) DISSC_HIP_CHECK(hipEventRecord(g->ev_x, stream));  // X is ready
    for (int j = 0; j < nk; ++j) {
      hipStream_t sj = (multi && j > 0) ? g->aux[j] : stream;
      if (multi && j > 0) DISSC_HIP_CHECK(hipStreamWaitEvent(sj, g->ev_x, 0));
      if (g->fused_w[(size_t)i * nk + j]) {  // narrow stage: the whole ResBlock in one launch
        const int epi = (j == 0) ? (nk == 1 ? EPI_MRF_DIV : EPI_MRF_SET)
                                 : (j == nk - 1 ? EPI_MRF_DIV : EPI_MRF_ADD);
        const bool pairs = g->fused_bf3[(size_t)i * nk + j] && resblock_bf3_pairs(ch);
        // the MRF update is inside the launch: a whole-block launch has to wait for the previous chain
        if (multi && j > 0 && !pairs) DISSC_HIP_CHECK(hipStreamWaitEvent(sj, g->ev_fin[j - 1], 0));
        const float* fw = g->fused_w[(size_t)i * nk + j];
        const float* fb = g->fused_b[(size_t)i * nk + j];
        const int rk = c.resblock_kernel_sizes[j];
        const int* dl = c.resblock_dilations[j];
        if (!pairs) {
          rc = launch_resblock_bf3(ch, X, ACC, fw, fb, lengths, L, mul, rk, dl, B, L, ld, 0.1f, epi, (float)nk, 0, 3, sj);
        } else {  // X -> XK -> TMP -> MRF update of ACC, one launch per residual pair
          float* xk = multi ? XKj[j] : XKj[0];
          float* tm = multi ? TMPj[j] : TMPj[0];
          rc = launch_resblock_bf3(ch, X, xk, fw, fb, lengths, L, mul, rk, dl, B, L, ld, 0.1f, EPI_STORE, 1.f, 0, 1, sj);
          if (!rc)
            rc = launch_resblock_bf3(ch, xk, tm, fw, fb, lengths, L, mul, rk, dl, B, L, ld, 0.1f, EPI_STORE, 1.f, 1, 2, sj);
          if (!rc && multi && j > 0) DISSC_HIP_CHECK(hipStreamWaitEvent(sj, g->ev_fin[j - 1], 0));  // only the last pair
          if (!rc)
            rc = launch_resblock_bf3(ch, tm, ACC, fw, fb, lengths, L, mul, rk, dl, B, L, ld, 0.1f, epi, (float)nk, 2, 3, sj);
        }
        if (rc) return rc;
        if (multi) DISSC_HIP_CHECK(hipEventRecord(g->ev_fin[j], sj));
        continue;
      }
      float* TMPc = multi ? TMPj[j] : TMPj[0];
      float* XKc = multi ? XKj[j] : XKj[0];
      {
        // narrow stages in exact fp32: each residual pair is ONE launch (respair.hip), ping-ponging
        // X -> XK -> TMP -> MRF update of ACC (a pair cannot run in place: neighbours read its halo)
        const size_t i0 = ((size_t)i * nk + j) * 3;
        bool pairs = !g->rb1[i0].prec && g->rb1[i0].m32 == (ch >= 32 ? 1 : 0);
        for (int m = 0; m < 3 && pairs; ++m)
          pairs = respair_supported(ch, g->rb1[i0 + m].KS, g->rb1[i0 + m].dil);
        if (pairs) {
          const float* src[3] = {X, XKc, TMPc};
          float* dst[3] = {XKc, TMPc, nullptr};
          for (int m = 0; m < 3; ++m) {
            int epi = EPI_RES;
            if (m == 2) {
              epi = (j == 0) ? (nk == 1 ? EPI_MRF_DIV : EPI_MRF_SET) : (j == nk - 1 ? EPI_MRF_DIV : EPI_MRF_ADD);
              if (multi && j > 0) DISSC_HIP_CHECK(hipStreamWaitEvent(sj, g->ev_fin[j - 1], 0));
            }
            if (g->pw[i0 + m].w1)
              rc = launch_respair_wino(g->pw[i0 + m], src[m], dst[m], ACC, lengths, L, mul, B, L, ld, 0.1f, epi, (float)nk, sj);
            else
              rc = launch_respair(g->rb1[i0 + m], g->rb2[i0 + m], src[m], dst[m], ACC, lengths, L, mul, B, L, ld, 0.1f, epi,
                                  (float)nk, sj);
            if (rc) return rc;
          }
          if (multi) DISSC_HIP_CHECK(hipEventRecord(g->ev_fin[j], sj));
          continue;
        }
      }
      for (int m = 0; m < 3; ++m) {
        const size_t idx = ((size_t)i * nk + j) * 3 + m;
        const float* xin = (m == 0) ? X : XKc;
        if (m == 0 && g->pw[idx].w1) {
          // the chain's first pair as ONE transform-domain launch (respair_wino.hip): X -> XK, t never leaves LDS
          if ((rc = launch_respair_wino(g->pw[idx], xin, XKc, ACC, lengths, L, mul, B, L, ld, 0.1f, EPI_RES, (float)nk, sj)))
            return rc;
          continue;
        }
        if (g->rb1[idx].wino && g->rb2[idx].wino) {
          // Toom-Cook F(4,3) form (conv_wino.hip): t = conv_d(lrelu(x)); x = x + conv_1(lrelu(t)) / MRF update
          // per conv: the eight-point forms on 8 waves (conv_wino8.hip) or F(4,3) on 12 (conv_wino.hip)
          auto runw = g->rb1[idx].wino == 2 ? run_wino8 : run_wino;
          auto runw2 = g->rb2[idx].wino == 2 ? run_wino8 : run_wino;
          if ((rc = runw(g->rb1[idx], xin, TMPc, nullptr, nullptr, lengths, L, mul, B, ld, ld, L, 0.1f, EPI_STORE, 1.f, sj)))
            return rc;
          int epiw = EPI_RES;
          if (m == 2) {
            epiw = (j == 0) ? (nk == 1 ? EPI_MRF_DIV : EPI_MRF_SET) : (j == nk - 1 ? EPI_MRF_DIV : EPI_MRF_ADD);
            if (multi && j > 0) DISSC_HIP_CHECK(hipStreamWaitEvent(sj, g->ev_fin[j - 1], 0));
          }
          if ((rc = runw2(g->rb2[idx], TMPc, XKc, xin, ACC, lengths, L, mul, B, ld, ld, L, 0.1f, epiw, (float)nk, sj)))
            return rc;
          continue;
        }
        // wide stages: the first conv stores lrelu(t) with zero tails (EPI_STORE_ACT) so that the second one -- the only
        // reader of t -- stages its windows by LDS-DMA: no staging registers, no masks, one more wave per SIMD
        const bool dma2 = opts().pair_dma && g->rb1[idx].m32 && g->rb2[idx].m32 && !g->rb1[idx].prec && !g->rb2[idx].prec &&
                          ch % KC == 0;
        const int ldt = dma2 ? (int)round_up((size_t)L + ZERO_TAIL, 4) : ld;
        if ((rc = run_conv(g->rb1[idx], xin, TMPc, nullptr, nullptr, lengths, L, mul, B, ch, ld, ldt,
                           L, 0.1f, dma2 ? EPI_STORE_ACT : EPI_STORE, 1.f, sj, 0.1f, 0)))
          return rc;
        int epi = EPI_RES;
        if (m == 2) {
          epi = (j == 0) ? (nk == 1 ? EPI_MRF_DIV : EPI_MRF_SET)
                         : (j == nk - 1 ? EPI_MRF_DIV : EPI_MRF_ADD);
          // xs = r0; xs += r1; x = (xs + r2)/3: only the chains' LAST convs are ordered
          if (multi && j > 0) DISSC_HIP_CHECK(hipStreamWaitEvent(sj, g->ev_fin[j - 1], 0));
        }
        if ((rc = run_conv(g->rb2[idx], TMPc, XKc, xin, ACC, lengths, L, mul, B, ch, ldt, ld, L, dma2 ? 1.0f : 0.1f,
                           epi, (float)nk, sj, 0.f, dma2 ? 1 : 0)))
          return rc;
      }
      if (multi) DISSC_HIP_CHECK(hipEventRecord(g->ev_fin[j], sj));
    }
    if (multi) DISSC_HIP_CHECK(hipStreamWaitEvent(stream, g->ev_fin[nk - 1], 0));  // stage output complete
  }
  // tail: lrelu(0.01) -> conv_post -> tanh
  launch_conv_post(ACC, g->post_w, g->post_b, lengths, mul, B, ch, g->post_KS, Tmax * mul, ld,
                   (long long)ch * ld, 0.01f, wav_out, Tmax * mul, stream);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int dissc_wav_postprocess(float* wav, const int32_t* n_samples, int B, int ld, void* stream) {
  if (!wav || !n_samples || B <= 0) {
    set_error("dissc_wav_postprocess: bad argument");
    return DISSC_EINVAL;
  }
  launch_wav_postprocess(wav, n_samples, B, ld, (hipStream_t)stream);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

// ---- stand-alone conv entry points (weights packed per call: tests only) ----------------
static int conv_once(DevConv& dc, const float* x, float* y, const int32_t* lengths, int B,
                     int ldx, int ldo, int Lmax, float in_slope, hipStream_t stream) {
  int rc = run_conv(dc, x, y, nullptr, nullptr, lengths, Lmax, 1, B, dc.CIN, ldx, ldo, Lmax,
                    in_slope, EPI_STORE, 1.f, stream);
  hipError_t e = hipStreamSynchronize(stream);
  free_conv(dc);
  if (rc) return rc;
  DISSC_HIP_CHECK(e);
  return DISSC_OK;
}

int dissc_conv1d(const float* x, const float* w_host, const float* bias_host, float* y,
                 const int32_t* lengths, int B, int Cin, int Cout, int k, int dilation, int ldx,
                 int ldo, int Lmax, float in_slope, void* stream) {
  if (!x || !w_host || !y || k % 2 != 1 || dilation < 1) {
    set_error("dissc_conv1d: bad argument");
    return DISSC_EINVAL;
  }
  DevConv dc;
  const bool use8 = opts().wino8 >= 2 && wino8_supported(Cout, Cin, k, dilation);  // "wino8" = 2: the stand-alone entry uses it (tests)
  if (use8 || (opts().wino >= 2 && wino_supported(Cout, Cin, k, dilation))) {  // "wino" = 2: likewise for the F(4,3) form
    const int taps = opts().wino8_r4 >= 2 && wino8_r4_supported(Cout, k, dilation) ? 4 : 3;  // "wino8_r4" = 2: F(5,4) (tests)
    int rc = use8 ? make_wino8(w_host, bias_host, Cout, k, dilation, dc, taps) : make_wino(w_host, bias_host, Cout, k, dilation, dc);
    if (rc) return rc;
    rc = (use8 ? run_wino8 : run_wino)(dc, x, y, nullptr, nullptr, lengths, Lmax, 1, B, ldx, ldo, Lmax, in_slope, EPI_STORE, 1.f,
                                       (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    free_conv(dc);
    if (rc) return rc;
    DISSC_HIP_CHECK(e);
    return DISSC_OK;
  }
  int rc = make_conv(w_host, bias_host, Cout, Cin, k, dilation, dc);
  if (rc) return rc;
  return conv_once(dc, x, y, lengths, B, ldx, ldo, Lmax, in_slope, (hipStream_t)stream);
}

// Diagnostics / tests: ONE residual pair y = x + conv_1(lrelu(conv_d(lrelu(x)))) (or its MRF modes) on device data with
// host weights, through a chosen implementation: mode 0 = two direct conv launches, 1 = the fused direct pair
// (respair.hip), 2 = two conv_wino launches, 3 = the fused transform-domain pair (respair_wino.hip).  Synchronous.
static int pair_run(int mode, const DevConv& c1, const DevConv& c2, const DevPairW& pw, const float* x, float* tmp, float* y,
                    float* acc, const int32_t* lengths, int B, int C, int ld, int Lmax, float slope, int epi, float mrf_div,
                    hipStream_t st) {
  int rc;
  switch (mode) {
    case 0:
      if ((rc = run_conv(c1, x, tmp, nullptr, nullptr, lengths, Lmax, 1, B, C, ld, ld, Lmax, slope, EPI_STORE, 1.f, st))) return rc;
      return run_conv(c2, tmp, y, x, acc, lengths, Lmax, 1, B, C, ld, ld, Lmax, slope, epi, mrf_div, st);
    case 1:
      return launch_respair(c1, c2, x, y, acc, lengths, Lmax, 1, B, Lmax, ld, slope, epi, mrf_div, st);
    case 2:
      if ((rc = run_wino(c1, x, tmp, nullptr, nullptr, lengths, Lmax, 1, B, ld, ld, Lmax, slope, EPI_STORE, 1.f, st))) return rc;
      return run_wino(c2, tmp, y, x, acc, lengths, Lmax, 1, B, ld, ld, Lmax, slope, epi, mrf_div, st);
    case 3:
      return launch_respair_wino(pw, x, y, acc, lengths, Lmax, 1, B, Lmax, ld, slope, epi, mrf_div, st);
    default:
      set_error("pair mode %d", mode);
      return DISSC_EINVAL;
  }
}

static int pair_make(int mode, const float* w1, const float* b1, const float* w2, const float* b2, int C, int k, int d,
                     DevConv& c1, DevConv& c2, DevPairW& pw) {
  int rc;
  if (mode == 2) {
    if (!wino_supported(C, C, k, d) && !(C == 32 && (k == 3 || k == 7 || k == 11) && (d == 1 || d == 3 || d == 5))) {
      set_error("pair mode 2: no conv_wino instance for C = %d, k = %d, d = %d", C, k, d);
      return DISSC_EINVAL;
    }
    if ((rc = make_wino(w1, b1, C, k, d, c1))) return rc;
    return make_wino(w2, b2, C, k, 1, c2);
  }
  if (mode == 3) return make_pairw(w1, b1, w2, b2, C, k, d, pw);
  if ((rc = make_conv(w1, b1, C, C, k, d, c1))) return rc;
  return make_conv(w2, b2, C, C, k, 1, c2);
}

int dissc_respair1d(const float* x, const float* w1_host, const float* b1_host, const float* w2_host, const float* b2_host,
                    float* y, float* acc, const int32_t* lengths, int B, int C, int k, int dilation, int ld, int Lmax,
                    float slope, int epi, float mrf_div, int mode, void* stream) {
  if (!x || !w1_host || !w2_host || !y || k % 2 != 1 || dilation < 1 || B <= 0 || epi < EPI_RES || epi > EPI_MRF_DIV ||
      (epi != EPI_RES && !acc)) {
    set_error("dissc_respair1d: bad argument");
    return DISSC_EINVAL;
  }
  DevConv c1, c2;
  DevPairW pw;
  float* tmp = nullptr;
  int rc = pair_make(mode, w1_host, b1_host, w2_host, b2_host, C, k, dilation, c1, c2, pw);
  if (!rc && (mode == 0 || mode == 2) && hipMalloc((void**)&tmp, (size_t)B * C * ld * sizeof(float)) != hipSuccess) {
    set_error("dissc_respair1d: hipMalloc");
    rc = DISSC_ENOMEM;
  }
  if (!rc) rc = pair_run(mode, c1, c2, pw, x, tmp, y, acc, lengths, B, C, ld, Lmax, slope, epi, mrf_div, (hipStream_t)stream);
  const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
  if (tmp) (void)hipFree(tmp);
  free_conv(c1);
  free_conv(c2);
  free_pairw(pw);
  if (rc) return rc;
  DISSC_HIP_CHECK(e);
  return DISSC_OK;
}

// Diagnostics: average ms of `iters` launches of one residual pair (modes as dissc_respair1d) on synthetic data.
int dissc_pair_bench(int B, int C, int k, int dilation, int L, int epi, int iters, int mode, float* ms_out) {
  if (!ms_out || B <= 0 || L <= 0 || iters <= 0 || epi < EPI_RES || epi > EPI_MRF_DIV) {
    set_error("dissc_pair_bench: bad argument");
    return DISSC_EINVAL;
  }
  std::vector<float> w1((size_t)C * C * k), w2((size_t)C * C * k), bias(C, 0.1f);
  uint32_t s = 12345u;
  auto rnd = [&]() {
    s = s * 1664525u + 1013904223u;
    return (s >> 8) / 16777216.0f - 0.5f;
  };
  for (auto& v : w1) v = rnd() * 0.05f;
  for (auto& v : w2) v = rnd() * 0.05f;
  DevConv c1, c2;
  DevPairW pw;
  int rc = pair_make(mode, w1.data(), bias.data(), w2.data(), bias.data(), C, k, dilation, c1, c2, pw);
  const int ld = (L + 3) / 4 * 4;
  const size_t n = (size_t)B * C * ld;
  float *x = nullptr, *y = nullptr, *t = nullptr, *a = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipSuccess;
  float ms = 0.f;
  if (!rc) {
    std::vector<float> hx(n);
    for (auto& v : hx) v = rnd() * 2.f;
    if (hipMalloc((void**)&x, n * 4) != hipSuccess || hipMalloc((void**)&y, n * 4) != hipSuccess ||
        hipMalloc((void**)&t, n * 4) != hipSuccess || hipMalloc((void**)&a, n * 4) != hipSuccess ||
        hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemset(a, 0, n * 4) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
      set_error("dissc_pair_bench: allocation failed");
      rc = DISSC_ENOMEM;
    }
  }
  auto once = [&]() { return pair_run(mode, c1, c2, pw, x, t, y, a, nullptr, B, C, ld, L, 0.1f, epi, 3.f, nullptr); };
  for (int it = 0; it < 2 && !rc; ++it) rc = once();
  if (!rc) {
    (void)hipEventRecord(e0, nullptr);
    for (int it = 0; it < iters && !rc; ++it) rc = once();
    (void)hipEventRecord(e1, nullptr);
    e = hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  for (float* q : {x, y, t, a})
    if (q) (void)hipFree(q);
  free_conv(c1);
  free_conv(c2);
  free_pairw(pw);
  if (rc) return rc;
  DISSC_HIP_CHECK(e);
  return DISSC_OK;
}

int dissc_conv_transpose1d(const float* x, const float* w_host, const float* bias_host, float* y,
                           const int32_t* lengths, int B, int Cin, int Cout, int k, int stride,
                           int ldx, int ldo, int Lmax, float in_slope, void* stream) {
  if (!x || !w_host || !y || stride < 1 || (k - stride) % 2 != 0) {
    set_error("dissc_conv_transpose1d: bad argument");
    return DISSC_EINVAL;
  }
  std::vector<DevConv> grp;
  int rc = make_convT(w_host, bias_host, Cin, Cout, k, stride, grp);
  for (auto& dc : grp) {
    int r2 = rc ? rc : conv_once(dc, x, y, lengths, B, ldx, ldo, Lmax, in_slope, (hipStream_t)stream);
    if (rc) free_conv(dc);
    rc = rc ? rc : r2;
  }
  return rc;
}

// The DEFAULTS (what handles created later start from), whatever handle the calling thread may be working for
int dissc_get_option(const char* key, int* value) {
  if (!key || !value) return DISSC_EINVAL;
  if (strcmp(key, "graph_hits") == 0) { *value = g_graph_hits; return DISSC_OK; }
  if (strcmp(key, "graph_captures") == 0) { *value = g_graph_captures; return DISSC_OK; }
  if (strcmp(key, "experimental") == 0) { *value = DISSC_EXPERIMENTAL; return DISSC_OK; }
#define DISSC_OPT_GET(f, d, k) if (strcmp(key, k) == 0) { *value = g_defaults.f; return DISSC_OK; }
  DISSC_OPTION_LIST(DISSC_OPT_GET)
#undef DISSC_OPT_GET
  set_error("dissc_get_option: unknown key %s", key);
  return DISSC_EINVAL;
}

int dissc_set_option(const char* key, int value) {
  if (!key) return DISSC_EINVAL;
  if (strncmp(key, "conv_cfg_bm", 11) == 0) {  // "conv_cfg_bm16|32|64|128|256" -> tile config id
    const int bm = atoi(key + 11);
    int cls = 0;
    while ((16 << cls) < bm) ++cls;
    conv_set_cfg(cls, value);
    return DISSC_OK;
  }
  if (strncmp(key, "conv32_cfg_bm", 13) == 0) {
    const int bm = atoi(key + 13);
    int cls = 0;
    while ((32 << cls) < bm) ++cls;
    conv32_set_cfg(cls, value);
    return DISSC_OK;
  }
  if (strcmp(key, "pairw_chv") == 0) value = value == 2 ? 2 : 1;
#define DISSC_OPT_SET(f, d, k) if (strcmp(key, k) == 0) { g_defaults.f = value; return DISSC_OK; }
  DISSC_OPTION_LIST(DISSC_OPT_SET)
#undef DISSC_OPT_SET
  set_error("dissc_set_option: unknown key %s", key);
  return DISSC_EINVAL;
}

// Diagnostics: average ms of `iters` launches of one conv layer on synthetic data.
int dissc_conv_bench(int B, int Cin, int Cout, int k, int dilation, int L, int epi, int iters,
                     int flags, float* ms_out) {
  if (!ms_out || B <= 0 || L <= 0 || iters <= 0) {
    set_error("dissc_conv_bench: bad argument");
    return DISSC_EINVAL;
  }
  std::vector<float> w((size_t)Cout * Cin * k), bias(Cout, 0.1f);
  uint32_t s = 12345u;
  for (auto& v : w) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) / 16777216.0f - 0.5f) * 0.05f;
  }
  // flags bit 5 (0x20): zero-filled operands (the chip clocks to its power budget: how much of a launch's time is the data's
  // switching activity, not the schedule?  never a number to quote)
  if (flags & 0x20) std::fill(w.begin(), w.end(), 0.f);
  DevConv dc;
  g_conv_prec = opts().precision;  // diagnostics follow the "precision" option like the generator does
  const bool w8 = (flags & 4) && wino8_supported(Cout, Cin, k, dilation);
  const bool wino = w8 || ((flags & 2) && wino_supported(Cout, Cin, k, dilation));
  int rc = w8 ? make_wino8(w.data(), bias.data(), Cout, k, dilation, dc, (flags & 8) && wino8_r4_supported(Cout, k, dilation) ? 4 : 3)
              : wino ? make_wino(w.data(), bias.data(), Cout, k, dilation, dc)
                     : make_conv(w.data(), bias.data(), Cout, Cin, k, dilation, dc);
  g_conv_prec = 0;
  if (rc) return rc;
  const int ld = (L + 3) / 4 * 4;
  const size_t nx = (size_t)B * Cin * ld, no = (size_t)B * Cout * ld;
  float *x = nullptr, *y = nullptr, *r = nullptr, *a = nullptr;
  std::vector<float> hx(nx);
  for (auto& v : hx) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) / 16777216.0f - 0.5f) * 2.f;
  }
  if (flags & 0x20) std::fill(hx.begin(), hx.end(), 0.f);
  DISSC_HIP_CHECK(hipMalloc((void**)&x, nx * 4));
  DISSC_HIP_CHECK(hipMalloc((void**)&y, no * 4));
  DISSC_HIP_CHECK(hipMalloc((void**)&r, no * 4));
  DISSC_HIP_CHECK(hipMalloc((void**)&a, no * 4));
  DISSC_HIP_CHECK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
  DISSC_HIP_CHECK(hipMemset(r, 0, no * 4));
  DISSC_HIP_CHECK(hipMemset(a, 0, no * 4));
  hipEvent_t e0, e1;
  DISSC_HIP_CHECK(hipEventCreate(&e0));
  DISSC_HIP_CHECK(hipEventCreate(&e1));
  const int saved_cls = (flags >> 16) & 0xf;
  if (flags & 0x8000) conv_set_cfg(saved_cls, (flags >> 8) & 0x3f);
  auto once = [&]() {
    return w8 ? run_wino8(dc, x, y, r, a, nullptr, L, 1, B, ld, ld, L, (flags & 1) ? 1.0f : 0.1f, epi, 3.f, nullptr)
         : wino ? run_wino(dc, x, y, r, a, nullptr, L, 1, B, ld, ld, L, (flags & 1) ? 1.0f : 0.1f, epi, 3.f, nullptr)
                : run_conv(dc, x, y, r, a, nullptr, L, 1, B, Cin, ld, ld, L, (flags & 1) ? 1.0f : 0.1f, epi, 3.f, nullptr);
  };
  for (int it = 0; it < 2 && !rc; ++it) rc = once();
  DISSC_HIP_CHECK(hipEventRecord(e0, nullptr));
  for (int it = 0; it < iters && !rc; ++it) rc = once();
  DISSC_HIP_CHECK(hipEventRecord(e1, nullptr));
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (const char* tlp = getenv("DISSC_TIMELINE")) {  // diagnostics: the head of the accumulator buffer (a kernel's timeline stamps)
    std::vector<char> hb(std::min<size_t>(no * 4, (size_t)4096 * 64));
    if (hipMemcpy(hb.data(), a, hb.size(), hipMemcpyDeviceToHost) == hipSuccess)
      if (FILE* f = fopen(tlp, "wb")) {
        fwrite(hb.data(), 1, hb.size(), f);
        fclose(f);
      }
  }
  (void)hipFree(x); (void)hipFree(y); (void)hipFree(r); (void)hipFree(a);
  free_conv(dc);
  if (rc) return rc;
  DISSC_HIP_CHECK(e);
  return DISSC_OK;
}

}  // extern "C"
