// Host side of the one-launch residual pairs in the Toom-Cook transform domain: which shapes take which form, packing, dispatch.
//   form 1: the register-only F(2,3) pairs (respair_f23.hip, respair16_f23.hip) -- the default for k = 11 at C = 32 / 16;
//   form 0: the F(4,3) pair kernel with the Y exchange through LDS (experimental/csrc/respair_wino.hip: its gate FAILED in round 4,
//           it is in DISSC_EXPERIMENTAL=1 builds only; experimental_stubs.hip answers for it otherwise).
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "respair_f23.h"

namespace dissc {

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// shapes that have an instance: a point's weights must fit 64 registers per lane (C^2 NS / 64 <= 64)
bool pairw_supported(int C, int KS, int dil) {
  if (!pairw43_built()) return false;  // (the kernel is in DISSC_EXPERIMENTAL=1 builds only)
  if (!(dil == 1 || dil == 3 || dil == 5)) return false;
  if (C == 32) return KS == 7 || KS == 11;
  if (C == 64) return KS == 3;
  return false;
}

// ... and the ones the generator uses it for with "pair_wino" = 1: where it measured faster than what it replaces
// (tools/pair_gate.py, B = 32 x 10 s, one MI355X: C = 32, k = 11: 895 / 988 us against 1 042 / 1 047 for the direct fused
// pair at d = 1 / 3, break-even at d = 5 and slower with the MRF epilogue that pair always has; C = 64, k = 3, d = 1:
// 624 against 658 for two conv_wino launches, break-even at d = 3 / 5; C = 32, k = 7: 884-963 against 732: never).
// The verdict's gates (C = 32, k = 11, d = 1 pair <= 720 us where the direct pair takes 921; C = 64, k = 3 pair <= 520 us)
// were NOT met: knock-outs (tools/pair_ko.py) put a k = 11 tile at MFMAs 476 + input transforms 100 + the two A^T
// exchanges 125 + skeleton (staging, barriers, weight loads, launch) 197 us, nothing overlapping -- one workgroup fills
// the CU and fp32 VALU work shares the MFMA datapath.  "pair_wino" = 2 takes every supported shape (tests).
bool pairw_wanted(int C, int KS, int dil) {
  if (opts().pair_f23 && pair_f23_supported(C, KS, dil)) return true;  // (make_pairw then builds the register-only F(2,3) form)
  if (!opts().pair_wino || !pairw_supported(C, KS, dil)) return false;
  if (opts().pair_wino >= 2) return true;
  if (C == 32) return KS == 11 && dil <= 3;
  return C == 64 && KS == 3 && dil == 1;
}

// w: [C][C][KS] -> U[p][co][ci][j] = sum_i G[p][i] w[co][ci][j + NS i] in the order the kernel's lanes hold them:
// [point][mi][tap j][8-channel sub-chunk][lane][k-step e] = U_p[32 mi + (lane & 31)][8 ksub + 2 e + (lane >> 5)][j]

int make_pairw(const float* w1, const float* b1, const float* w2, const float* b2, int C, int KS, int dil, DevPairW& pw) {
  pw.form = (opts().pair_f23 && pair_f23_supported(C, KS, dil)) ? 1 : 0;
  if (!pw.form && !pairw_supported(C, KS, dil)) {
    set_error("make_pairw: no instance for C = %d, k = %d, dilation %d", C, KS, dil);
    return DISSC_EINVAL;
  }
  pw.C = C; pw.KS = KS; pw.dil = dil;
  int rc = pw.form ? pack_pair_f23(w1, &pw.w1, C, KS) : pack_pairw43(w1, C, KS, &pw.w1);
  if (!rc) rc = pw.form ? pack_pair_f23(w2, &pw.w2, C, KS) : pack_pairw43(w2, C, KS, &pw.w2);
  std::vector<float> bb(C, 0.f);
  if (b1) memcpy(bb.data(), b1, C * sizeof(float));
  if (!rc) rc = upload(bb, &pw.b1);
  std::fill(bb.begin(), bb.end(), 0.f);
  if (b2) memcpy(bb.data(), b2, C * sizeof(float));
  if (!rc) rc = upload(bb, &pw.b2);
  return rc;
}

void free_pairw(DevPairW& pw) {
  for (float** q : {&pw.w1, &pw.w2, &pw.b1, &pw.b2}) {
    if (*q) (void)hipFree(*q);
    *q = nullptr;
  }
}

// option "pairw_chv" (Options::pairw_chv, default 2): "pairw_chv" option: column halves per workgroup of respair_wino_kernel (2: one 12-wave workgroup per CU;
                      // 1: two 6-wave workgroups with half the tile each -- measured 5-30 % slower, kept for the tests)


int launch_respair_wino(const DevPairW& pw, const float* x, float* out, float* acc, const int32_t* lengths, int len_default,
                        int len_mul, int B, int Lmax, int ld, float slope, int epi, float mrf_div, hipStream_t stream) {
  auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
  if (!pw.w1 || epi == EPI_STORE || x == out || ld < 4 || ld % 4 || misaligned(x) || misaligned(out) || misaligned(acc) ||
      B <= 0 || Lmax <= 0) {
    set_error("launch_respair_wino: bad argument (C=%d k=%d d=%d ld=%d epi=%d)", pw.C, pw.KS, pw.dil, ld, epi);
    return DISSC_EINVAL;
  }
  if (pw.form == 1) return launch_pair_f23(pw, x, out, acc, lengths, len_default, len_mul, B, Lmax, ld, slope, epi, mrf_div, stream);
  if (pairw43_built())
    return launch_pairw43(pw, x, out, acc, lengths, len_default, len_mul, B, Lmax, ld, slope, epi, mrf_div, stream);
  set_error("launch_respair_wino: no instance (the F(4,3) pair kernel is only in DISSC_EXPERIMENTAL=1 builds)");
  return DISSC_EINVAL;
}

}  // namespace dissc
