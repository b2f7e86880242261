// attn_fused_kernel: HuBERT's self-attention (12 heads x 64, exact softmax, fp32) as ONE launch per layer -- S = K^T Q on the matrix
// pipe, online softmax in registers, O += V P^T on the matrix pipe, nothing but Q / K / V in and O out crosses HBM.
// Replaces the attention inside fairseq's TransformerSentenceEncoderLayer as reached from data/encode.py:21-22,32 (HF restatement:
// modeling_hubert.py HubertAttention; oracle/hubert_ref.py).
// A translation unit of its own because it is compiled with `-mllvm -amdgpu-mfma-vgpr-form` (__graft_entry__.py EXTRA_FLAGS): the
// kernel does VALU work on its accumulators between MFMAs (the softmax on S, the rescaling of O); with the accumulators in AGPRs the
// compiler moved them out and back through 256 v_accvgpr_read / _write per 256 MFMAs, and on this chip VALU instructions are paid in
// matrix-pipe time.  VGPR-form MFMAs need no moves (178 registers, two workgroups per CU as before).
#include "common.h"

namespace dissc {

// ---- fused exact attention for one (utterance, head, 128-query tile) --------------------------
// qkv is channels-first [B][3D][ld] (rows: Q | K | V, head h owns rows h*64..h*64+63 of each,
// Q already scaled by 1/sqrt(64)).  Per wave: AT_NQ groups of 16 queries.  S^T[key][query] = K^T Q is formed on
// the matrix pipe with rows = keys, so a lane's accumulator registers are keys of ONE query: the online-softmax
// reductions are in-lane plus two cross-group shuffles, and the probabilities are already in MFMA B-operand position
// for O[d][query] += V[d][key] P^T[key][query] -- no LDS round trip for P, no HBM round trip for S.
// Key order inside a 64-key tile (round 6): row r16 of the 16 x 16 sub-tile s is key 4 r16 + s (not 16 s + r16).  Lane (l15, g)
// then needs, as A operands of k-step ks for the four sub-tiles, K[d = 4 ks + g][4 l15 .. 4 l15 + 3] -- ONE ds_read_b128 of the
// row-major [d][key] tile feeding 8 MFMAs -- and holds S for keys 16 g + 4 r + s in register r of sub-tile s, so the PV product's
// k-step (s, st) (B operand: register st of sub-tile s = key 16 g + 4 st + s) takes V[d = 16 i + l15][16 g + 4 st .. + 3] -- again one
// ds_read_b128 per 8 MFMAs -- from the same row-major layout, which is also the layout in HBM: the tiles are stored with plain
// 16-byte writes.  (The first form read 128 scalars per tile through ds_read2_b32 with a v_add_u32 of address arithmetic each and
// stored V transposed: 600 VALU instructions per 256 MFMAs, and on this chip VALU instructions are paid in matrix-pipe time --
// 107 TFLOP/s at MFMA busy 0.69.)
constexpr int AT_NQ = 2;    // groups of 16 queries per wave
constexpr int AT_Q = 64 * AT_NQ;  // queries per block (4 waves x AT_NQ x 16)
constexpr int AT_LDQQ = AT_Q + 16;  // row stride of the Q tile (% 32 == 16: conflict-free fragment reads)
constexpr int AT_K = 64;    // keys per LDS tile
constexpr int AT_LDK = 68;  // row stride of the K / V tiles: 16-byte aligned rows; 16 lanes x 16 bytes of one row (K) or of 16
                            // consecutive rows (V: 68 l mod 64 = 4 l) touch every bank once

__global__ void __launch_bounds__(256) attn_fused_kernel(const float* __restrict__ qkv,
                                                         const int32_t* __restrict__ lens, int D,
                                                         int hd, int ld, float* __restrict__ out, int nqt, int H, int B) {
  __shared__ __attribute__((aligned(16))) float Qs[64 * AT_LDQQ];
  __shared__ __attribute__((aligned(16))) float Ks[64 * AT_LDK];
  __shared__ __attribute__((aligned(16))) float Vs[64 * AT_LDK];
  // Grid: nqt >= 0: the 3-D grid (query tile, head, utterance).  nqt < 0 (option "xcd_order" bit 3): a 1-D grid in XCD order -- workgroup
  // ids go round-robin over the 8 XCDs, so id -> (XCD = id & 7, slot = id >> 3); the -nqt query tiles of one (utterance, head) take
  // consecutive slots of ONE XCD and share its L2 for that head's K / V rows (the 3-D grid sent them to -nqt different XCDs: K and V
  // crossed the fabric once per query tile, 0.54 GB per launch against 0.20 algorithmic at 32 x 499 frames).
  int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
  if (nqt < 0) {
    const int n = -nqt, slot = blockIdx.x >> 3;
    const int p = (slot / n) * 8 + (blockIdx.x & 7);
    if (p >= H * B) return;
    qt = slot % n;
    b = p / H;
    h = p - b * H;
  }
  const int T = lens[b];
  const int q0 = qt * AT_Q;
  if (q0 >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const float* qb = qkv + ((size_t)b * 3 * D + (size_t)h * hd) * ld;
  const float* kb = qb + (size_t)D * ld;
  const float* vb = qb + (size_t)2 * D * ld;
  // Q tile -> LDS (zero beyond T)
  for (int e = tid; e < 64 * (AT_Q / 4); e += 256) {
    const int d = e / (AT_Q / 4), c = (e - d * (AT_Q / 4)) * 4;
    int col = q0 + c;
    col = col > ld - 4 ? ld - 4 : col;
    f32x4 v = *reinterpret_cast<const f32x4*>(qb + (size_t)d * ld + col);
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) v[e2] = (q0 + c + e2 < T) ? v[e2] : 0.f;
    *reinterpret_cast<f32x4*>(Qs + d * AT_LDQQ + c) = v;
  }
  __syncthreads();
  // this wave's AT_NQ groups of 16 queries: k-step ks -> Q[d = 4*ks + g][query = (wave*AT_NQ + j)*16 + l15].
  // The groups share every K / V fragment read and give the matrix pipe independent accumulator chains.
  float qf[AT_NQ][16];
#pragma unroll
  for (int j = 0; j < AT_NQ; ++j)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qf[j][ks] = Qs[(4 * ks + g) * AT_LDQQ + (wave * AT_NQ + j) * 16 + l15];
  f32x4 o[AT_NQ][4];
  float m[AT_NQ], l[AT_NQ];
#pragma unroll
  for (int j = 0; j < AT_NQ; ++j) {
    m[j] = -INFINITY;
    l[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // K / V tiles of 64 keys x 64 dimensions: the NEXT tile is fetched into registers (16 B per lane) while the current one is on the
  // matrix pipe, and written to LDS between the two barriers.  Loads: one buffer descriptor per operand over the head's 64 rows
  // (wave-uniform base), one per-lane byte offset for all four slots (slot i = rows + 16 i: a scalar offset), the tile's first key a
  // scalar offset too -- no per-lane 64-bit address arithmetic.  Columns past the row's end read the next row (or, past the last row,
  // the descriptor's bounds check returns 0): every such key is >= T and masked below.
  const __amdgpu_buffer_rsrc_t krs = wave_rsrc(kb, (unsigned)(64u * (unsigned)ld * 4u));
  const __amdgpu_buffer_rsrc_t vrs = wave_rsrc(vb, (unsigned)(64u * (unsigned)ld * 4u));
  const int sd = tid >> 4, sc = (tid & 15) * 4;               // this thread's staging slot: row sd + 16 i, keys sc .. sc + 3
  const unsigned ld_off = ((unsigned)sd * (unsigned)ld + (unsigned)sc) * 4u;
  const unsigned row16 = 16u * (unsigned)ld * 4u;
  f32x4 kr[4], vr[4];
  auto tile_load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kr[i] = rsrc_load16(krs, ld_off, (unsigned)i * row16 + (unsigned)k0 * 4u);
      vr[i] = rsrc_load16(vrs, ld_off, (unsigned)i * row16 + (unsigned)k0 * 4u);
    }
  };
  auto tile_store = [&](int k0) {
    if (k0 + AT_K > T) {  // uniform: only the last tile has keys beyond T.  V must be 0 there: p = 0, and 0 * garbage may be NaN
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const bool ok = k0 + sc + e2 < T;
          kr[i][e2] = ok ? kr[i][e2] : 0.f;
          vr[i][e2] = ok ? vr[i][e2] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(Ks + (sd + 16 * i) * AT_LDK + sc) = kr[i];
      *reinterpret_cast<f32x4*>(Vs + (sd + 16 * i) * AT_LDK + sc) = vr[i];
    }
  };
  const float* kfp = Ks + g * AT_LDK + 4 * l15;    // + 4 ks rows
  const float* vfp = Vs + l15 * AT_LDK + 16 * g;   // + 16 i rows + 4 st
  tile_load(0);
  for (int k0 = 0; k0 < T; k0 += AT_K) {
    __syncthreads();  // previous tile fully consumed
    tile_store(k0);
    __syncthreads();
    if (k0 + AT_K < T) tile_load(k0 + AT_K);
    __builtin_amdgcn_sched_barrier(0);
    // S^T for all 64 keys of the tile: sT[j][s][r] = S[query l15 of group j][key k0 + 16 g + 4 r + s]
    f32x4 sT[AT_NQ][4];
#pragma unroll
    for (int j = 0; j < AT_NQ; ++j)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) sT[j][s4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const f32x4 kf = *reinterpret_cast<const f32x4*>(kfp + 4 * ks * AT_LDK);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < AT_NQ; ++j)
          sT[j][s4] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s4], qf[j][ks], sT[j][s4], 0, 0, 0);
    }
    if (k0 + AT_K > T) {  // uniform: only the last tile has keys beyond T
#pragma unroll
      for (int j = 0; j < AT_NQ; ++j)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + 16 * g + 4 * r + s4 >= T) sT[j][s4][r] = -INFINITY;
    }
    // one online-softmax step per 64 keys (exp through v_exp_f32: e^x = 2^(x log2 e))
    constexpr float LOG2E = 1.44269504088896340736f;
#pragma unroll
    for (int j = 0; j < AT_NQ; ++j) {
      float mx = -INFINITY;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sT[j][s4][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mn = fmaxf(m[j], mx);  // finite: the tile has at least one valid key
      const float alpha = __expf(m[j] - mn);
      const float nb = -mn * LOG2E;  // e^(s - mn) = 2^(s log2 e - mn log2 e): one v_fma + v_exp per score (-inf stays -inf -> 0)
      float ps = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sT[j][s4][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[j][s4][r], LOG2E, nb));
          ps += sT[j][s4][r];
        }
      ps += __shfl_xor(ps, 16);
      ps += __shfl_xor(ps, 32);
      l[j] = l[j] * alpha + ps;
      m[j] = mn;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[j][i][0] *= alpha; o[j][i][1] *= alpha; o[j][i][2] *= alpha; o[j][i][3] *= alpha;
      }
    }
    // O += V P^T: k-step (s, st) covers keys 16 g + 4 st + s (k index g) = register st of sub-tile s of the same lane
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(vfp + 16 * i * AT_LDK + 4 * st);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int j = 0; j < AT_NQ; ++j)
            o[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s4], sT[j][s4][st], o[j][i], 0, 0, 0);
      }
  }
  // o[j][i][r] = O[d = i*16 + 4g + r][query = (wave*AT_NQ + j)*16 + l15] (unnormalised)
#pragma unroll
  for (int j = 0; j < AT_NQ; ++j) {
    const int q = q0 + (wave * AT_NQ + j) * 16 + l15;
    if (q < T) {
      const float inv = 1.f / l[j];
      float* ob = out + ((size_t)b * D + (size_t)h * hd) * ld + q;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(size_t)(i * 16 + 4 * g + r) * ld] = o[j][i][r] * inv;
    }
  }
}

// option "xcd_order" bit 3 (Options::xcd_order): the 1-D grid in XCD order (see the kernel)
int launch_attn_fused(const float* qkv, const int32_t* lens, int D, int hd, int ld, float* out, int T, int H, int B, hipStream_t st) {
  if (hd != 64 || T <= 0 || B <= 0 || H <= 0) {
    set_error("launch_attn_fused: head dimension %d (64 only), T %d, H %d, B %d", hd, T, H, B);
    return DISSC_EINVAL;
  }
  const int nqt = (T + AT_Q - 1) / AT_Q;
  if (opts().xcd_order & 8)
    hipLaunchKernelGGL(attn_fused_kernel, dim3((unsigned)((H * B + 7) / 8 * 8 * nqt)), dim3(256), 0, st, qkv, lens, D, hd, ld, out, -nqt, H,
                       B);
  else
    hipLaunchKernelGGL(attn_fused_kernel, dim3(nqt, H, B), dim3(256), 0, st, qkv, lens, D, hd, ld, out, nqt, H, B);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

}  // namespace dissc
