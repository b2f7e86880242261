// Host-side helpers shared by the generator and the predictors: device copies of packed
// conv weights and the launch wrapper around conv_mfma_kernel.
#include <string.h>

#include "common.h"

namespace dissc {

int upload(const std::vector<float>& h, float** d) {
  DISSC_HIP_CHECK(hipMalloc((void**)d, h.size() * sizeof(float)));
  DISSC_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return DISSC_OK;
}

int make_conv(const float* w, const float* bias, int Cout, int Cin, int KS, int dil,
                     DevConv& dc) {
  std::vector<float> packed;
  int Mpad, nchunk;
  pack_conv_weights(w, Cout, Cin, KS, packed, Mpad, nchunk);
  std::vector<float> b(Mpad, 0.f);
  if (bias) memcpy(b.data(), bias, Cout * sizeof(float));
  dc.CIN = Cin; dc.M = Cout; dc.KS = KS; dc.dil = dil; dc.nchunk = nchunk; dc.up = 1;
  dc.macs_per_t = (double)Cout * Cin * KS;
  int rc = upload(packed, &dc.wpack);
  if (rc) return rc;
  return upload(b, &dc.bias);
}

int make_convT(const float* w, const float* bias, int Cin, int Cout, int k, int s,
                      DevConv& dc) {
  std::vector<float> w3;
  convT_to_conv(w, Cin, Cout, k, s, w3);
  std::vector<float> b3((size_t)Cout * s);
  for (int co = 0; co < Cout; ++co)
    for (int p = 0; p < s; ++p) b3[co * s + p] = bias ? bias[co] : 0.f;
  int rc = make_conv(w3.data(), b3.data(), Cout * s, Cin, 3, 1, dc);
  dc.up = s;
  dc.macs_per_t = (double)Cin * Cout * k;  // per INPUT step: every (ci,co,kk) used once
  return rc;
}

int set_affine(DevConv& dc, const float* scale, const float* shift, int n) {
  const int bm_pad = 256;  // generous: rows beyond M are never stored
  std::vector<float> sc((size_t)(dc.M + bm_pad), 1.f), sh((size_t)(dc.M + bm_pad), 0.f);
  for (int i = 0; i < n && i < dc.M; ++i) {
    sc[i] = scale ? scale[i] : 1.f;
    sh[i] = shift ? shift[i] : 0.f;
  }
  int rc = upload(sc, &dc.scale);
  if (rc) return rc;
  return upload(sh, &dc.shift);
}

void free_conv(DevConv& dc) {
  if (dc.wpack) (void)hipFree(dc.wpack);
  if (dc.bias) (void)hipFree(dc.bias);
  if (dc.scale) (void)hipFree(dc.scale);
  if (dc.shift) (void)hipFree(dc.shift);
  dc.wpack = dc.bias = dc.scale = dc.shift = nullptr;
}


int run_conv(const DevConv& dc, const float* x, float* out, const float* res, float* acc,
                    const int32_t* lengths, int len_default, int len_mul, int B, int C_x, int ldx,
                    int ldo, int Lmax, float slope, int epi, float mrf_div, hipStream_t stream) {
  ConvArgs a;
  a.x = x; a.wpack = dc.wpack; a.bias = dc.bias; a.scale = dc.scale; a.shift = dc.shift; a.res = res; a.out = out; a.acc = acc;
  a.lengths = lengths; a.len_default = len_default; a.len_mul = len_mul;
  a.CIN = dc.CIN; a.M = dc.M; a.KS = dc.KS; a.dil = dc.dil; a.nchunk = dc.nchunk;
  a.XW = conv_xw(dc.M, dc.KS, dc.dil);
  a.ldx = ldx; a.ldo = ldo;
  a.x_bstride = (long long)C_x * ldx;
  a.o_bstride = (long long)(dc.M / dc.up) * ldo;
  a.slope = slope; a.mrf_div = mrf_div; a.epi = epi; a.up = dc.up;
  return launch_conv(a, B, Lmax, stream);
}


}  // namespace dissc
