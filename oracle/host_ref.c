/* Plain-C restatement of the integer / byte-exact host logic of the path (TEST INFRASTRUCTURE:
 * linked only by tests/ through ctypes; built by __graft_entry__.build_oracle with gcc).
 *
 *   oracle_dedup            <- dedup_seq, reference dataset/utils.py:14-16
 *   oracle_len_carryover    <- len_carryover_correction, reference infer.py:158-172
 *   oracle_wav_postprocess  <- generate() int16 cast + librosa.util.normalize,
 *                              reference sr/inference.py:73-75,206
 *   oracle_kmeans_assign    <- argmin_k ||x-c_k||^2 (sklearn KMeans.predict semantics,
 *                              lowest index on ties); third-party, see oracle/hubert_ref.py
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
