// What DISSC_EXPERIMENTAL=1 builds get from experimental/csrc/*.hip (kernels whose gates failed: kept buildable and tested there,
// not carried by the default library): in the default build these entry points say so.
#include "common.h"

#if !DISSC_EXPERIMENTAL
namespace dissc {

// experimental/csrc/conv_s2tc.hip: HuBERT's stride-2 feature convs in polyphase Toom-Cook form (round 5 gate failed)
bool s2tc_supported(int, int, int, int) { return false; }
int make_s2tc(const float*, const float*, int, int, DevS2tc&) {
  set_error("the polyphase Toom-Cook feature convs (experimental/csrc/conv_s2tc.hip) are only in DISSC_EXPERIMENTAL=1 builds");
  return DISSC_EINVAL;
}
void free_s2tc(DevS2tc&) {}
double s2tc_executed_macs_per_out(int Cout, int Cin) { return (double)Cout * Cin * 3.0; }
int run_s2tc(const DevS2tc&, const float*, float*, const int32_t*, const int32_t*, int, int, int, int, int, int, hipStream_t) {
  set_error("run_s2tc: only in DISSC_EXPERIMENTAL=1 builds");
  return DISSC_EINVAL;
}

// experimental/csrc/respair_wino.hip: the F(4,3) residual-pair kernel (round 4 gate failed)
bool pairw43_built() { return false; }
int pack_pairw43(const float*, int, int, float**) {
  set_error("make_pairw: the F(4,3) pair kernel is only in DISSC_EXPERIMENTAL=1 builds");
  return DISSC_EINVAL;
}
int launch_pairw43(const DevPairW&, const float*, float*, float*, const int32_t*, int, int, int, int, int, float, int, float, hipStream_t) {
  set_error("launch_respair_wino: the F(4,3) pair kernel is only in DISSC_EXPERIMENTAL=1 builds");
  return DISSC_EINVAL;
}

}  // namespace dissc
#endif
