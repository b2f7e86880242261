// dissc_pred_*: DISSC's length and pitch predictors (8 / 13-conv 1-D CNNs over unit sequences)
// plus infer.py's integer sample logic (dedup, carry-over rounding, expand) on the GPU.
//
// Replaces: LenPredictor.forward (reference model/len_predictor.py:35-52), PitchPredictor /
// PitchPredictorBase .forward/.infer_freq/.calc_freq (model/pitch_predictor.py:72-104,145-176),
// dedup_seq (dataset/utils.py:14-16), len_carryover_correction (infer.py:158-172) and
// torch.repeat_interleave at infer.py:32.
//
// The reference runs B=1 per (utterance x target speaker); here a ragged batch of sequences
// goes through the same conv_mfma_kernel as the generator (M=128 tiles), with every layer
// treating positions >= length as the zero padding a B=1 run would see.  Eval-mode BatchNorm is
// a per-row affine in the conv epilogue (x*alpha+beta, as PyTorch evaluates it), LeakyReLU(0.01)
// is applied on the next layer's load.
#include <string.h>

#include <map>
#include <string>

#include "common.h"

namespace dissc {

// x[b][c][t]: c < E: tok_emb[seq[b,t]][c];  c >= E: spk_emb[spk[b]][c-E] (+ pe[t][c-E])
__global__ void pred_embed_kernel(const int64_t* __restrict__ seq, const int64_t* __restrict__ spk,
                                  const int32_t* __restrict__ lengths, const float* __restrict__ tok,
                                  const float* __restrict__ spe, const float* __restrict__ pe, int T,
                                  int E, int n_tok_rows, int n_spk_rows, float* __restrict__ x,
                                  int ldx) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int len = lengths ? lengths[b] : T;
  if (t >= len) return;
  float v;
  if (c < E) {
    long long id = seq[(size_t)b * T + t];
    id = id < 0 ? 0 : (id >= n_tok_rows ? n_tok_rows - 1 : id);
    v = tok[(size_t)id * E + c];
  } else {
    long long id = spk[b];
    id = id < 0 ? 0 : (id >= n_spk_rows ? n_spk_rows - 1 : id);
    v = spe[(size_t)id * E + (c - E)];
    if (pe) v = v + pe[(size_t)t * E + (c - E)];
  }
  x[((size_t)b * 2 * E + c) * ldx + t] = v;
}

// out[b,t] = (cls > 0) * reg,  cls/reg = 1x1 convs over lrelu(h[0:128]) / lrelu(h[128:256])
__global__ void pitch_head_kernel(const float* __restrict__ h, const float* __restrict__ wc,
                                  const float* __restrict__ wr, float bc, float br,
                                  const int32_t* __restrict__ lengths, const int64_t* __restrict__ spk,
                                  const float* __restrict__ id2mean, const float* __restrict__ id2std,
                                  int n_stats, int norm, int T, int ld, int C, float slope,
                                  float* __restrict__ out, int ldo) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int len = lengths ? lengths[b] : T;
  if (t >= T) return;
  if (t >= len) {
    out[(size_t)b * ldo + t] = 0.f;
    return;
  }
  const float* hb = h + (size_t)b * 2 * C * ld + t;
  float cls = 0.f, reg = 0.f;
  for (int c = 0; c < C; ++c) {
    float a = hb[(size_t)c * ld];
    float r = hb[(size_t)(C + c) * ld];
    a = a > 0.f ? a : a * slope;
    r = r > 0.f ? r : r * slope;
    cls = fmaf(wc[c], a, cls);
    reg = fmaf(wr[c], r, reg);
  }
  cls += bc;
  reg += br;
  if (!norm) {
    long long s = spk[b];  // ids are validated on the host; clamp like the embedding lookups do
    s = s < 0 ? 0 : (s >= n_stats ? n_stats - 1 : s);
    reg = id2mean[s] + reg * id2std[s];
  }
  out[(size_t)b * ldo + t] = (cls > 0.f ? 1.f : 0.f) * reg;
}

// inclusive prefix sum of one int per thread over a 256-thread workgroup (wave shuffles, then the
// four wave totals through LDS); *total receives the workgroup sum
constexpr int SEQ_NT = 256;
__device__ __forceinline__ int block_scan_incl(int v, int* red, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(v, off);
    if (lane >= off) v += y;
  }
  __syncthreads();  // red[] may still be read from the previous call
  if (lane == 63) red[wave] = v;
  __syncthreads();
  int add = 0;
#pragma unroll
  for (int w = 0; w < SEQ_NT / 64; ++w) add += (w < wave) ? red[w] : 0;
  *total = red[0] + red[1] + red[2] + red[3];
  return v + add;
}

// run-length encode each row (integer, exact): one workgroup per sequence.  A frame is a run head
// when it differs from its predecessor; a prefix sum over the head flags gives every run its output
// slot, the run start goes to counts[] first and becomes the run length in a second sweep.
__global__ void __launch_bounds__(SEQ_NT) dedup_kernel(const int64_t* __restrict__ units,
                                                       const int32_t* __restrict__ lengths, int B, int T,
                                                       int64_t* __restrict__ vals, int32_t* __restrict__ counts,
                                                       int32_t* __restrict__ n_out) {
  __shared__ int red[SEQ_NT / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = lengths ? lengths[b] : T;
  const int64_t* u = units + (size_t)b * T;
  int64_t* v = vals + (size_t)b * T;
  int32_t* c = counts + (size_t)b * T;
  int n = 0;
  for (int t0 = 0; t0 < len; t0 += SEQ_NT) {
    const int t = t0 + tid;
    const int64_t x = t < len ? u[t] : 0;
    const int head = (t < len && (t == 0 || u[t - 1] != x)) ? 1 : 0;
    int total;
    const int pos = block_scan_incl(head, red, &total);
    if (head) {
      v[n + pos - 1] = x;
      c[n + pos - 1] = t;  // run start, turned into the run length below
    }
    n += total;
  }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += SEQ_NT) {
    const int i = i0 + tid;
    int start = 0, next = 0;
    if (i < n) {
      start = c[i];
      next = i + 1 < n ? c[i + 1] : len;
    }
    __syncthreads();  // every start of this sweep is read before any is overwritten
    if (i < n) c[i] = next - start;
  }
  if (tid == 0) n_out[b] = n;
}

// error-diffusion rounding, sequential fp32 running sum exactly like the reference's loop
__global__ void carryover_kernel(const float* __restrict__ lens, const int32_t* __restrict__ n_in,
                                 int B, int ld, int32_t* __restrict__ out, int32_t* __restrict__ totals) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* l = lens + (size_t)b * ld;
  int32_t* o = out + (size_t)b * ld;
  const int n = n_in[b];
  float total = 0.f;
  int sum = 0;
  for (int i = 0; i < n; ++i) {
    const float x = l[i];
    const float r = rintf(fmaxf(x, 1.f));  // torch.round = half-to-even; clamp(min=1)
    const float a = x - r;
    total = total + a;
    int adj = 0;
    if (total >= 1.f) {
      adj = 1;
      total = total - 1.f;
    } else if (total <= -1.f) {
      adj = -1;
      total = total + 1.f;
    }
    const int v = (int)r + adj;
    o[i] = v;
    sum += v > 0 ? v : 0;
  }
  totals[b] = sum;
}

// run-length decode (repeat_interleave): one workgroup per sequence; a prefix sum of the run
// lengths gives every run its first output position, each thread then writes its own run
__global__ void __launch_bounds__(SEQ_NT) expand_kernel(const int64_t* __restrict__ vals,
                                                        const int32_t* __restrict__ lens,
                                                        const int32_t* __restrict__ n_in, int B, int ld_in,
                                                        int64_t* __restrict__ out, int ld_out) {
  __shared__ int red[SEQ_NT / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = n_in[b];
  int64_t* o = out + (size_t)b * ld_out;
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += SEQ_NT) {
    const int i = i0 + tid;
    const int r = (i < n) ? max(lens[(size_t)b * ld_in + i], 0) : 0;
    int total;
    const int end = base + block_scan_incl(r, red, &total);
    if (r > 0) {
      const int64_t v = vals[(size_t)b * ld_in + i];
      for (int p = end - r; p < end && p < ld_out; ++p) o[p] = v;
    }
    base += total;
  }
}

}  // namespace dissc

using namespace dissc;

struct dissc_pred {
  Options opt;  // this handle's snapshot of the tuning options (common.h)
  int kind = 0;  // 0 len, 1 pitch "new", 2 pitch "base"
  int E = 32, C = 128;
  int n_tok_rows = 0, n_spk_rows = 0, pe_len = 0;
  float* tok = nullptr;
  float* spe = nullptr;
  float* pe = nullptr;
  std::vector<DevConv> body;  // cnn1, cnn11.. (and cnn2 for pitch)
  DevConv head;               // len: cnn2 (M=1); pitch: [cnn_class1; cnn_reg1] stacked (M=256)
  float* wc = nullptr;        // pitch: cnn_class2 / cnn_reg2 weights
  float* wr = nullptr;
  float bc = 0.f, br = 0.f;
  ~dissc_pred() {
    for (auto& c : body) free_conv(c);
    free_conv(head);
    for (float* p : {tok, spe, pe, wc, wr})
      if (p) (void)hipFree(p);
  }
};

static inline size_t rup(size_t x, size_t m) { return (x + m - 1) / m * m; }

extern "C" {

int dissc_pred_create(int kind, const DisscTensor* weights, size_t n_weights, dissc_pred_t* out) {
  if (!weights || !out || kind < 0 || kind > 2) {
    set_error("dissc_pred_create: bad argument");
    return DISSC_EINVAL;
  }
  std::map<std::string, const DisscTensor*> by;
  for (size_t i = 0; i < n_weights; ++i) by[weights[i].name] = &weights[i];
  auto find = [&](const std::string& name) -> const DisscTensor* {
    auto it = by.find(name);
    return it == by.end() ? nullptr : it->second;
  };
  auto numel = [](const DisscTensor* t) {
    size_t n = 1;
    for (int d = 0; d < t->ndim; ++d) n *= (size_t)t->shape[d];
    return n;
  };
  dissc_pred* p = new dissc_pred();
  p->opt = g_defaults;  // frozen here
  OptScope opt_scope(&p->opt);
  p->kind = kind;
  int rc = DISSC_OK;
  auto fail = [&](int code) {
    delete p;
    return code;
  };
  auto need = [&](const std::string& name, const DisscTensor** t) -> int {
    *t = find(name);
    if (!*t) {
      set_error("dissc_pred_create: missing tensor '%s'", name.c_str());
      return DISSC_ENOTFOUND;
    }
    return DISSC_OK;
  };
  const DisscTensor *t = nullptr, *tb = nullptr;
  if ((rc = need("token_emb.weight", &t))) return fail(rc);
  p->E = (int)t->shape[1];
  p->n_tok_rows = (int)t->shape[0];
  if ((rc = upload(std::vector<float>(t->data, t->data + numel(t)), &p->tok))) return fail(rc);
  if ((rc = need("spk_emb.weight", &t))) return fail(rc);
  p->n_spk_rows = (int)t->shape[0];
  if ((int)t->shape[1] != p->E) {
    set_error("dissc_pred_create: embedding sizes differ");
    return fail(DISSC_EINVAL);
  }
  if ((rc = upload(std::vector<float>(t->data, t->data + numel(t)), &p->spe))) return fail(rc);
  if (kind == 1) {
    if ((rc = need("pe.pe", &t))) return fail(rc);
    p->pe_len = (int)t->shape[t->ndim - 2];
    if ((rc = upload(std::vector<float>(t->data, t->data + numel(t)), &p->pe))) return fail(rc);
  }
  // conv stack; "<name>.bn_scale/.bn_shift" (optional) = eval BatchNorm folded to alpha/beta
  auto conv = [&](const std::string& name, DevConv& dc) -> int {
    const DisscTensor *w, *b;
    int r;
    if ((r = need(name + ".weight", &w))) return r;
    if ((r = need(name + ".bias", &b))) return r;
    if (w->ndim != 3) {
      set_error("dissc_pred_create: '%s.weight' must be [Cout,Cin,k]", name.c_str());
      return DISSC_EINVAL;
    }
    if ((r = make_conv(w->data, b->data, (int)w->shape[0], (int)w->shape[1], (int)w->shape[2], 1, dc)))
      return r;
    const DisscTensor* sc = find(name + ".bn_scale");
    const DisscTensor* sh = find(name + ".bn_shift");
    if (sc || sh) return set_affine(dc, sc ? sc->data : nullptr, sh ? sh->data : nullptr, (int)w->shape[0]);
    return DISSC_OK;
  };
  std::vector<std::string> names = {"cnn1", "cnn11", "cnn12", "cnn13", "cnn14", "cnn15", "cnn16"};
  if (kind != 0) {
    names.push_back("cnn17");
    names.push_back("cnn2");
  }
  p->body.resize(names.size());
  for (size_t i = 0; i < names.size(); ++i)
    if ((rc = conv(names[i], p->body[i]))) return fail(rc);
  p->C = p->body[0].M;
  if (kind == 0) {
    if ((rc = conv("cnn2", p->head))) return fail(rc);
  } else {
    // stack cnn_class1 / cnn_reg1 into one 256-row conv
    const DisscTensor *w1, *b1, *w2, *b2;
    if ((rc = need("cnn_class1.weight", &w1)) || (rc = need("cnn_class1.bias", &b1)) ||
        (rc = need("cnn_reg1.weight", &w2)) || (rc = need("cnn_reg1.bias", &b2)))
      return fail(rc);
    const int C = p->C, k = (int)w1->shape[2];
    std::vector<float> w((size_t)2 * C * C * k), b(2 * C);
    memcpy(w.data(), w1->data, (size_t)C * C * k * 4);
    memcpy(w.data() + (size_t)C * C * k, w2->data, (size_t)C * C * k * 4);
    memcpy(b.data(), b1->data, C * 4);
    memcpy(b.data() + C, b2->data, C * 4);
    if ((rc = make_conv(w.data(), b.data(), 2 * C, C, k, 1, p->head))) return fail(rc);
    const DisscTensor* s1 = find("cnn_class1.bn_scale");
    const DisscTensor* h1 = find("cnn_class1.bn_shift");
    const DisscTensor* s2 = find("cnn_reg1.bn_scale");
    const DisscTensor* h2 = find("cnn_reg1.bn_shift");
    if (s1 || s2) {
      std::vector<float> sc(2 * C, 1.f), sh(2 * C, 0.f);
      for (int i = 0; i < C; ++i) {
        if (s1) sc[i] = s1->data[i];
        if (h1) sh[i] = h1->data[i];
        if (s2) sc[C + i] = s2->data[i];
        if (h2) sh[C + i] = h2->data[i];
      }
      if ((rc = set_affine(p->head, sc.data(), sh.data(), 2 * C))) return fail(rc);
    }
    if ((rc = need("cnn_class2.weight", &t)) || (rc = need("cnn_class2.bias", &tb))) return fail(rc);
    if ((rc = upload(std::vector<float>(t->data, t->data + C), &p->wc))) return fail(rc);
    p->bc = tb->data[0];
    if ((rc = need("cnn_reg2.weight", &t)) || (rc = need("cnn_reg2.bias", &tb))) return fail(rc);
    if ((rc = upload(std::vector<float>(t->data, t->data + C), &p->wr))) return fail(rc);
    p->br = tb->data[0];
  }
  *out = p;
  return DISSC_OK;
}

void dissc_pred_destroy(dissc_pred_t p) { delete p; }

size_t dissc_pred_workspace_bytes(dissc_pred_t p, int B, int Lmax) {
  if (!p || B <= 0 || Lmax <= 0) return 0;
  const size_t ld = rup(Lmax, 4);
  // input [2E] + two ping-pong [C] + heads [2C]
  return ((size_t)B * ld * (2 * p->E + 2 * p->C + 2 * p->C)) * sizeof(float) + 1024;
}

static int pred_body(dissc_pred* p, const int64_t* seq, const int64_t* spk, const int32_t* lengths,
                     int B, int L, void* ws, size_t ws_bytes, hipStream_t stream, float** last,
                     float** spare, int* ld_out) {
  if (ws_bytes < dissc_pred_workspace_bytes(p, B, L)) {
    set_error("dissc_pred: workspace %zu < %zu bytes", ws_bytes, dissc_pred_workspace_bytes(p, B, L));
    return DISSC_ENOMEM;
  }
  const int ld = (int)rup(L, 4);
  float* x0 = (float*)rup((size_t)ws, 256);
  float* a = x0 + (size_t)B * 2 * p->E * ld;
  float* b = a + (size_t)B * p->C * ld;
  dim3 grid((L + 127) / 128, 2 * p->E, B);
  hipLaunchKernelGGL(pred_embed_kernel, grid, dim3(128), 0, stream, seq, spk, lengths, p->tok, p->spe,
                     p->pe, L, p->E, p->n_tok_rows, p->n_spk_rows, x0, ld);
  const float* in = x0;
  int cin = 2 * p->E;
  float slope = 1.0f;  // raw embeddings into cnn1, LeakyReLU(0.01) afterwards
  for (size_t i = 0; i < p->body.size(); ++i) {
    float* o = (i & 1) ? b : a;
    int rc = run_conv(p->body[i], in, o, nullptr, nullptr, lengths, L, 1, B, cin, ld, ld, L, slope,
                      EPI_STORE, 1.f, stream);
    if (rc) return rc;
    in = o;
    cin = p->C;
    slope = 0.01f;
  }
  *last = const_cast<float*>(in);
  *spare = (in == a) ? b : a;
  *ld_out = ld;
  return DISSC_OK;
}

int dissc_len_forward(dissc_pred_t p, const int64_t* seq, const int64_t* spk, const int32_t* lengths,
                      int B, int Lmax, float* out, int ldo, void* ws, size_t ws_bytes, void* stream_) {
  if (!p || p->kind != 0 || !seq || !spk || !out || !ws || B <= 0 || Lmax <= 0 || (ldo & 3) ||
      ldo < Lmax) {
    set_error("dissc_len_forward: bad argument");
    return DISSC_EINVAL;
  }
  OptScope opt_scope(&p->opt);
  hipStream_t stream = (hipStream_t)stream_;
  float *h, *spare;
  int ld;
  int rc = pred_body(p, seq, spk, lengths, B, Lmax, ws, ws_bytes, stream, &h, &spare, &ld);
  if (rc) return rc;
  // cnn2 (128 -> 1) with the label de-normalisation (*std + mean) as its affine epilogue
  rc = run_conv(p->head, h, out, nullptr, nullptr, lengths, Lmax, 1, B, p->C, ld, ldo, Lmax, 0.01f,
                EPI_STORE, 1.f, stream);
  if (rc) return rc;
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int dissc_len_set_norm(dissc_pred_t p, float mean, float std) {
  if (!p || p->kind != 0) return DISSC_EINVAL;
  if (p->head.scale) (void)hipFree(p->head.scale);
  if (p->head.shift) (void)hipFree(p->head.shift);
  p->head.scale = p->head.shift = nullptr;
  return set_affine(p->head, &std, &mean, 1);
}

int dissc_pitch_forward(dissc_pred_t p, const int64_t* seq, const int64_t* spk,
                        const int32_t* lengths, int B, int Tmax, int norm, const float* id2mean,
                        const float* id2std, int n_stats, float* out, int ldo, void* ws,
                        size_t ws_bytes, void* stream_) {
  if (!p || p->kind == 0 || !seq || !spk || !out || !ws || B <= 0 || Tmax <= 0 || ldo < Tmax ||
      (!norm && (!id2mean || !id2std || n_stats <= 0))) {
    set_error("dissc_pitch_forward: bad argument");
    return DISSC_EINVAL;
  }
  OptScope opt_scope(&p->opt);
  if (p->kind == 1 && Tmax > p->pe_len) {
    // the reference fails the same way: PositionalEncoding max_len (model/pitch_predictor.py:7,37)
    set_error("dissc_pitch_forward: %d frames exceed the positional encoding (%d)", Tmax, p->pe_len);
    return DISSC_EINVAL;
  }
  hipStream_t stream = (hipStream_t)stream_;
  float *h, *spare;
  int ld;
  int rc = pred_body(p, seq, spk, lengths, B, Tmax, ws, ws_bytes, stream, &h, &spare, &ld);
  if (rc) return rc;
  float* heads = (float*)rup((size_t)ws, 256) + (size_t)B * ld * (2 * p->E + 2 * p->C);
  rc = run_conv(p->head, h, heads, nullptr, nullptr, lengths, Tmax, 1, B, p->C, ld, ld, Tmax, 0.01f,
                EPI_STORE, 1.f, stream);
  if (rc) return rc;
  dim3 grid((Tmax + 127) / 128, B);
  hipLaunchKernelGGL(pitch_head_kernel, grid, dim3(128), 0, stream, heads, p->wc, p->wr, p->bc, p->br,
                     lengths, spk, id2mean, id2std, n_stats, norm, Tmax, ld, p->C, 0.01f, out, ldo);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int dissc_dedup(const int64_t* units, const int32_t* lengths, int B, int Tmax, int64_t* vals,
                int32_t* counts, int32_t* n_out, void* stream) {
  if (!units || !vals || !counts || !n_out || B <= 0 || Tmax <= 0) {
    set_error("dissc_dedup: bad argument");
    return DISSC_EINVAL;
  }
  hipLaunchKernelGGL(dedup_kernel, dim3(B), dim3(SEQ_NT), 0, (hipStream_t)stream, units, lengths,
                     B, Tmax, vals, counts, n_out);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int dissc_len_carryover(const float* lens, const int32_t* n, int B, int ld, int32_t* lens_int,
                        int32_t* totals, void* stream) {
  if (!lens || !n || !lens_int || !totals || B <= 0) {
    set_error("dissc_len_carryover: bad argument");
    return DISSC_EINVAL;
  }
  hipLaunchKernelGGL(carryover_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, lens, n, B,
                     ld, lens_int, totals);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int dissc_expand(const int64_t* vals, const int32_t* lens_int, const int32_t* n, int B, int ld_in,
                 int64_t* out, int ld_out, void* stream) {
  if (!vals || !lens_int || !n || !out || B <= 0) {
    set_error("dissc_expand: bad argument");
    return DISSC_EINVAL;
  }
  hipLaunchKernelGGL(expand_kernel, dim3(B), dim3(SEQ_NT), 0, (hipStream_t)stream, vals, lens_int,
                     n, B, ld_in, out, ld_out);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

}  // extern "C"
