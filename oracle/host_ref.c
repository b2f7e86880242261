/* Plain-C restatement of the integer / byte-exact host logic of the path (TEST INFRASTRUCTURE:
 * linked only by tests/ through ctypes; built by __graft_entry__.build_oracle with gcc).
 *
 *   oracle_dedup            <- dedup_seq, reference dataset/utils.py:14-16
 *   oracle_len_carryover    <- len_carryover_correction, reference infer.py:158-172
 *   oracle_wav_postprocess  <- generate() int16 cast + librosa.util.normalize,
 *                              reference sr/inference.py:73-75,206
 *   oracle_kmeans_assign    <- argmin_k ||x-c_k||^2 in float64 (sklearn KMeans.predict semantics,
 *                              lowest index on ties); third-party, see oracle/hubert_ref.py
 *   oracle_kmeans_assign_f32 / oracle_kmeans_cnorm_f32
 *                           <- the SAME step in the fp32 arithmetic the device specifies to the bit
 *                              (dissc_amd/csrc/hubert.hip kmeans_assign_kernel): the expression sklearn's
 *                              dense predict evaluates, ||c||^2 - 2 x.c, each dot product one fma chain
 *                              in index order, first minimum wins, NaN never wins
 * Pinned by tests/test_oracle_golden.py against tests/golden/pred.npz (reference outputs). */
#include <math.h>
#include <stdint.h>

int oracle_dedup(const int64_t* seq, int n, int64_t* vals, int32_t* counts) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (m > 0 && vals[m - 1] == seq[i]) {
      counts[m - 1] += 1;
    } else {
      vals[m] = seq[i];
      counts[m] = 1;
      ++m;
    }
  }
  return m;
}

/* lens: predicted frames per unit (fp32).  out[i] = round_half_even(max(lens[i],1)) + carry. */
void oracle_len_carryover(const float* lens, int n, int32_t* out) {
  volatile float total = 0.0f; /* volatile: keep every fp32 rounding step, as torch does */
  for (int i = 0; i < n; ++i) {
    const float x = lens[i];
    const float c = x < 1.0f ? 1.0f : x;
    const float r = nearbyintf(c); /* default rounding mode = ties to even = torch.round */
    const float a = x - r;
    total = total + a;
    int adj = 0;
    if (total >= 1.0f) {
      adj = 1;
      total = total - 1.0f;
    } else if (total <= -1.0f) {
      adj = -1;
      total = total + 1.0f;
    }
    out[i] = (int32_t)r + adj;
  }
}

/* y in [-1,1] -> float32 samples the reference writes */
void oracle_wav_postprocess(const float* y, int n, float* out) {
  float peak = 0.0f;
  for (int i = 0; i < n; ++i) {
    const float s = y[i] * 32768.0f;
    int v = (int)truncf(s);                 /* astype('int16'): C truncation ... */
    v = ((v + 32768) & 65535) - 32768;      /* ... with two's-complement wrap */
    out[i] = (float)v;
    if (fabsf(out[i]) > peak) peak = fabsf(out[i]);
  }
  if (peak < 1.17549435e-38f) return;
  for (int i = 0; i < n; ++i) out[i] = out[i] / peak;
}

void oracle_kmeans_assign(const float* x, int T, int D, const float* centers, int K, int64_t* units) {
  for (int t = 0; t < T; ++t) {
    double best = INFINITY;
    int64_t arg = 0;
    for (int k = 0; k < K; ++k) {
      double d = 0;
      for (int j = 0; j < D; ++j) {
        const double e = (double)x[(long)t * D + j] - (double)centers[(long)k * D + j];
        d += e * e;
      }
      if (d < best) {
        best = d;
        arg = k;
      }
    }
    units[t] = arg;
  }
}

/* cnorm[k] = fma chain s = fmaf(c[d], c[d], s), d = 0..D-1 */
void oracle_kmeans_cnorm_f32(const float* centers, int K, int D, float* cnorm) {
  for (int k = 0; k < K; ++k) {
    float s = 0.0f;
    for (int j = 0; j < D; ++j) s = fmaf(centers[(long)k * D + j], centers[(long)k * D + j], s);
    cnorm[k] = s;
  }
}

/* units[t] = first k minimising cnorm[k] - 2 * chain(x_t, c_k); x [T][D], centers [K][D] */
void oracle_kmeans_assign_f32(const float* x, int T, int D, const float* centers, const float* cnorm, int K,
                              int64_t* units) {
  for (int t = 0; t < T; ++t) {
    float best = INFINITY;
    int64_t arg = 0;
    for (int k = 0; k < K; ++k) {
      float acc = 0.0f;
      for (int j = 0; j < D; ++j) acc = fmaf(x[(long)t * D + j], centers[(long)k * D + j], acc);
      const float s = cnorm[k] - 2.0f * acc;
      if (s < best) {
        best = s;
        arg = k;
      }
    }
    units[t] = arg;
  }
}
