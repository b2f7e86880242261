// Host-side helpers shared by the generator and the predictors: device copies of packed
// conv weights and the launch wrapper around conv_mfma_kernel.
#include <string.h>

#include "common.h"

namespace dissc {

// option "ragged_enum" (Options::ragged_enum, default 1): "ragged_enum" option (conv_mfma32.hip)
thread_local int g_conv_prec = 0;  // (per thread: handles may be built concurrently) what make_conv packs for; dissc_gen_create raises it to opts().precision for its own layers

int upload(const std::vector<float>& h, float** d) {
  DISSC_HIP_CHECK(hipMalloc((void**)d, h.size() * sizeof(float)));
  DISSC_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return DISSC_OK;
}

// w: [Cout][Cin/groups][KS] (PyTorch layout); Cout, Cin are the TOTAL channel counts.
int make_conv(const float* w, const float* bias, int Cout, int Cin, int KS, int dil, DevConv& dc,
              int groups, int stride, int pad_left) {
  std::vector<float> packed;
  int Mpad, nchunk;
  const int Mg = Cout / groups, Cg = Cin / groups;
  // 64-cycle MFMAs wherever a 32-row tile is not mostly padding; 48 rows per group (HuBERT's positional conv) are three
  // 16-row tiles of the 16x16x4 kernel instead of two 32-row tiles of which a quarter is padding ("pos48" option)
  dc.m32 = (opts().use_mfma32 && Mg >= 32 && !(opts().pos48 && Mg == 48 && groups > 1)) ? 1 : 0;
  // split-bf16 only where conv_mfma32.hip has an instance for it
  const bool lin_big = (KS == 1 && Mg >= 256 && (Cg + KC - 1) / KC >= 8);
  dc.prec = (g_conv_prec == 1 && dc.m32 && stride == 1 && groups == 1 && (KS - 1) * dil <= MAX_TAP_SPAN && !lin_big)
                ? 1 : 0;
  if (dc.prec) pack_conv_weights_bf3(w, Cout, Cg, KS, packed, Mpad, nchunk);
  else if (dc.m32) pack_conv_weights32(w, Cout, Cg, KS, packed, Mpad, nchunk, groups);
  else pack_conv_weights(w, Cout, Cg, KS, packed, Mpad, nchunk, groups);
  std::vector<float> b((size_t)Mpad * groups, 0.f);
  if (bias)
    for (int g = 0; g < groups; ++g) memcpy(b.data() + (size_t)g * Mpad, bias + (size_t)g * Mg, Mg * sizeof(float));
  dc.CIN = Cg; dc.M = Mg; dc.KS = KS; dc.dil = dil; dc.nchunk = nchunk; dc.up = 1;
  dc.groups = groups; dc.Mpad = Mpad; dc.stride = stride; dc.pad_left = pad_left;
  dc.macs_per_t = (double)Cout * Cg * KS;
  int rc = upload(packed, &dc.wpack);
  if (rc) return rc;
  if (dc.m32 && !dc.prec && dil == 1 && conv2s128_shape(Cout, Cin, KS, stride, groups) && opts().conv2s128) {
    std::vector<float> p2;  // the second packing costs 4 Cout Cin KS bytes (3 MB per HuBERT feature conv)
    pack_s2_weights128(w, Cout, Cin, p2);
    if ((rc = upload(p2, &dc.wpack2))) return rc;
  }
  return upload(b, &dc.bias);
}

int make_convT(const float* w, const float* bias, int Cin, int Cout, int k, int s,
               std::vector<DevConv>& groups) {
  const int pad = (k - s) / 2;
  groups.clear();
  // input taps of phase p: delta in [dlo, dhi] with 0 <= p + pad - s*delta < k
  auto taps = [&](int p, int& dlo, int& dhi) {
    dlo = 1 << 30;
    dhi = -(1 << 30);
    for (int kk = (p + pad) % s; kk < k; kk += s) {
      const int delta = (p + pad - kk) / s;  // exact
      dlo = delta < dlo ? delta : dlo;
      dhi = delta > dhi ? delta : dhi;
    }
  };
  // Narrow layers (few output rows) are store/latency bound: one launch over the union of the
  // taps (some zero weights) beats several small ones; wide layers skip the zero taps instead.
  const bool single = Cout < 64;
  int p0 = 0;
  while (p0 < s) {
    int dlo, dhi;
    taps(p0, dlo, dhi);
    int np = 1;
    while (p0 + np < s) {
      int l2, h2;
      taps(p0 + np, l2, h2);
      if (single) {
        dlo = l2 < dlo ? l2 : dlo;
        dhi = h2 > dhi ? h2 : dhi;
      } else if (l2 != dlo || h2 != dhi) {
        break;
      }
      ++np;
    }
    const int ntap = dhi - dlo + 1;
    std::vector<float> wc;
    convT_phase_weights(w, Cin, Cout, k, s, p0, np, dlo, ntap, wc);
    std::vector<float> bc((size_t)Cout * np);
    for (int co = 0; co < Cout; ++co)
      for (int pi = 0; pi < np; ++pi) bc[co * np + pi] = bias ? bias[co] : 0.f;
    groups.emplace_back();
    DevConv& dc = groups.back();
    int rc = make_conv(wc.data(), bc.data(), Cout * np, Cin, ntap, 1, dc, 1, 1, -dlo);
    if (rc) return rc;
    dc.up = s;
    dc.up_np = np;
    dc.up_p0 = p0;
    dc.macs_per_t = groups.size() == 1 ? (double)Cin * Cout * k : 0.0;  // algorithmic MACs counted once
    p0 += np;
  }
  return DISSC_OK;
}

int set_affine(DevConv& dc, const float* scale, const float* shift, int n) {
  // same [groups][Mpad] indexing as the bias
  std::vector<float> sc((size_t)dc.Mpad * dc.groups, 1.f), sh((size_t)dc.Mpad * dc.groups, 0.f);
  for (int i = 0; i < n && i < dc.M * dc.groups; ++i) {
    const size_t k = (size_t)(i / dc.M) * dc.Mpad + (i % dc.M);
    sc[k] = scale ? scale[i] : 1.f;
    sh[k] = shift ? shift[i] : 0.f;
  }
  int rc = upload(sc, &dc.scale);
  if (rc) return rc;
  return upload(sh, &dc.shift);
}

void free_conv(DevConv& dc) {
  if (dc.wpack) (void)hipFree(dc.wpack);
  if (dc.wpack2) (void)hipFree(dc.wpack2);
  if (dc.bias) (void)hipFree(dc.bias);
  if (dc.scale) (void)hipFree(dc.scale);
  if (dc.shift) (void)hipFree(dc.shift);
  dc.wpack = dc.wpack2 = dc.bias = dc.scale = dc.shift = nullptr;
}


int run_conv_ex(const DevConv& dc, const float* x, float* out, const float* res, float* acc,
                const ConvIO& io, int B, int C_x_total, int ldx, int ldo, int Lmax_out, float slope,
                int epi, float mrf_div, hipStream_t stream, float out_slope, int dma_in) {
  ConvArgs a;
  a.out_slope = out_slope; a.dma_in = dma_in;
  a.x = x; a.wpack = dc.wpack; a.wpack2 = dc.wpack2; a.bias = dc.bias; a.scale = dc.scale; a.shift = dc.shift; a.res = res;
  a.out = out; a.acc = acc;
  a.lengths = io.lengths_in; a.len_default = io.len_default; a.len_mul = io.len_mul;
  a.lengths_out = io.lengths_out; a.olen_default = io.olen_default;
  a.CIN = dc.CIN; a.M = dc.M; a.KS = dc.KS; a.dil = dc.dil; a.nchunk = dc.nchunk;
  a.pad_left = dc.pad_left >= 0 ? dc.pad_left : ((dc.KS - 1) * dc.dil) / 2;
  a.mfast = 0;
  a.xcd = 0; a.xcd_ntile = 0; a.xcd_nb = 0; a.xcd_mg = 0; a.xcd_span = 0;
  a.ragged_enum = 0;
  a.groups = dc.groups; a.nsub_group = dc.Mpad / (dc.m32 ? 32 : 16); a.act = dc.act; a.m32 = dc.m32; a.prec = dc.prec;
  a.cfg32 = -1;
  const int span = (dc.KS - 1) * dc.dil;
  const bool lin_big = (dc.KS == 1 && dc.M >= 256 && dc.nchunk >= 8);
  if (dc.m32 && !dc.prec && dc.stride == 1 && dc.groups == 1 && span <= MAX_TAP_SPAN && !lin_big)
    a.cfg32 = conv32_pick_cfg(dc.M, B, Lmax_out);  // general path only: the special instances keep their tile
  a.XW = conv_xw(dc.M, dc.KS, dc.dil, dc.stride, dc.m32, a.cfg32 >= 0 ? conv32_cfg_bn(a.cfg32) : 0);
  a.ldx = ldx; a.ldo = ldo;
  a.x_bstride = (long long)C_x_total * ldx;
  a.o_bstride = (long long)(dc.M * dc.groups / dc.up_np) * ldo;
  a.slope = slope; a.mrf_div = mrf_div; a.epi = epi; a.up = dc.up; a.up_np = dc.up_np; a.up_p0 = dc.up_p0;
  return launch_conv(a, B, Lmax_out, dc.stride, stream);
}

int run_conv_ex(const DevConv& dc, const float* x, float* out, const float* res, const ConvIO& io,
                int B, int C_x_total, int ldx, int ldo, int Lmax_out, float slope, int epi,
                hipStream_t stream) {
  return run_conv_ex(dc, x, out, res, nullptr, io, B, C_x_total, ldx, ldo, Lmax_out, slope, epi, 1.f,
                     stream);
}

int run_conv(const DevConv& dc, const float* x, float* out, const float* res, float* acc,
             const int32_t* lengths, int len_default, int len_mul, int B, int C_x, int ldx, int ldo,
             int Lmax, float slope, int epi, float mrf_div, hipStream_t stream, float out_slope, int dma_in) {
  ConvIO io;
  io.lengths_in = lengths; io.len_default = len_default; io.len_mul = len_mul;
  return run_conv_ex(dc, x, out, res, acc, io, B, C_x, ldx, ldo, Lmax, slope, epi, mrf_div, stream, out_slope, dma_in);
}

}  // namespace dissc
