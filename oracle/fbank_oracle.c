/* CPU oracle for the fbank front-end  --  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C (float32) restatement of the reference C++ runtime front-end.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it (through oracle/fbank_oracle.py).
 * Parity status: PINNED -- tests/test_fbank_oracle.py checks it (a) against the reference itself compiled
 * from /root/reference into oracle/_ref/ (when present) and (b) against tests/golden/fbank_golden.npz,
 * recorded from that compiled reference.
 *
 * Follows (paths relative to the reference tree):
 *   runtime/core/frontend/fbank.h:33-97    tables: mel bank (float32), Hamming window (double -> float)
 *   runtime/core/frontend/fbank.h:138-198  per frame: DC removal, pre-emphasis 0.97, window, FFT, power,
 *                                          mel, log with FLT_EPSILON floor
 *   runtime/core/frontend/fft.cc:11-35     sine table by recurrence (float32)
 *   runtime/core/frontend/fft.cc:37-52     bit-reversal permutation
 *   runtime/core/frontend/fft.cc:59-119    in-place radix-2 decimation-in-time FFT
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI_D 3.14159265358979323846

static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

/* quarter-wave recurrence, then symmetry; tbl has n + n/4 entries (fft.cc:11-35) */
static void sine_table(int n, float* tbl) {
  const int n2 = n / 2, n4 = n / 4, n8 = n / 8;
  float t = (float)sin(PI_D / n);
  float dc = 2 * t * t;
  float ds = (float)sqrt(dc * (2 - dc));
  float c = 1, s = 0;
  t = 2 * dc;
  tbl[n4] = 1;
  tbl[0] = 0;
  for (int i = 1; i < n8; ++i) {
    c -= dc; dc += t * c;
    s += ds; ds -= t * s;
    tbl[i] = s;
    tbl[n4 - i] = c;
  }
  if (n8 != 0) tbl[n8] = (float)sqrt(0.5);
  for (int i = 0; i < n4; ++i) tbl[n2 - i] = tbl[i];
  for (int i = 0; i < n2 + n4; ++i) tbl[i + n2] = -tbl[i];
}

static void bit_reverse_table(int n, int* rev) {
  int j = 0;
  for (int i = 0; i < n; ++i) {
    rev[i] = j;
    int k = n / 2;
    while (k >= 1 && k <= j) { j -= k; k /= 2; }
    j += k;
  }
}

static void fft_forward(const int* rev, const float* tbl, float* re, float* im, int n) {
  const int n4 = n / 4;
  for (int i = 0; i < n; ++i) {
    int j = rev[i];
    if (i < j) { float t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
  }
  for (int half = 1; half < n; half *= 2) {
    const int span = 2 * half, step = n / span;
    for (int j = 0, h = 0; j < half; ++j, h += step) {
      const float c = tbl[h + n4], s = tbl[h];
      for (int i = j; i < n; i += span) {
        const int k = i + half;
        const float dx = s * im[k] + c * re[k];
        const float dy = c * im[k] - s * re[k];
        re[k] = re[i] - dx; re[i] += dx;
        im[k] = im[i] - dy; im[i] += dy;
      }
    }
  }
}

/* out: (num_frames, num_bins) row-major; returns num_frames (0 if the signal is shorter than one frame).
 * window: 0 = Hamming (the runtime), 1 = Povey (torchaudio/Kaldi default; parity unpinned). */
int wekws_oracle_fbank(const float* wave, int nsamp, int num_bins, int sample_rate, int frame_length, int frame_shift,
                       int window, float* out) {
  if (nsamp < frame_length) return 0;
  const int nframes = 1 + (nsamp - frame_length) / frame_shift;          /* fbank.h:141-142 */
  const int N = next_pow2(frame_length), NB = N / 2;
  float* tbl = (float*)malloc(sizeof(float) * (N + N / 4));
  int* rev = (int*)malloc(sizeof(int) * N);
  float* win = (float*)malloc(sizeof(float) * frame_length);
  float* re = (float*)malloc(sizeof(float) * N);
  float* im = (float*)malloc(sizeof(float) * N);
  float* power = (float*)malloc(sizeof(float) * NB);
  float* melw = (float*)calloc((size_t)num_bins * NB, sizeof(float));
  int* first = (int*)malloc(sizeof(int) * num_bins);
  int* count = (int*)malloc(sizeof(int) * num_bins);
  sine_table(N, tbl);
  bit_reverse_table(N, rev);
  {                                                                      /* fbank.h:90-96 */
    const double a = 2.0 * PI_D / (frame_length - 1);
    for (int i = 0; i < frame_length; ++i) {
      double w = 0.54 - 0.46 * cos(a * (double)i);
      if (window == 1) w = pow(0.5 - 0.5 * cos(a * (double)i), 0.85);
      win[i] = (float)w;
    }
  }
  {                                                                      /* fbank.h:51-88 */
    const float bin_width = (float)sample_rate / N;
    const float lo = mel_scale(20.0f), hi = mel_scale((float)(sample_rate / 2));
    const float delta = (hi - lo) / (num_bins + 1);
    for (int b = 0; b < num_bins; ++b) {
      const float left = lo + b * delta, center = lo + (b + 1) * delta, right = lo + (b + 2) * delta;
      int fi = -1, li = -1;
      for (int i = 0; i < NB; ++i) {
        const float m = mel_scale(bin_width * i);
        if (m > left && m < right) {
          melw[(size_t)b * NB + i] = (m <= center) ? (m - left) / (center - left) : (right - m) / (right - center);
          if (fi < 0) fi = i;
          li = i;
        }
      }
      first[b] = fi < 0 ? 0 : fi;
      count[b] = fi < 0 ? 0 : li + 1 - fi;
    }
  }
  for (int f = 0; f < nframes; ++f) {
    const float* src = wave + (size_t)f * frame_shift;
    float mean = 0.0f;
    for (int j = 0; j < frame_length; ++j) mean += src[j];                /* fbank.h:155-160 */
    mean /= frame_length;
    for (int j = 0; j < frame_length; ++j) re[j] = src[j] - mean;
    for (int j = frame_length - 1; j > 0; --j) re[j] -= 0.97f * re[j - 1];  /* fbank.h:122-127 */
    re[0] -= 0.97f * re[0];
    for (int j = 0; j < frame_length; ++j) re[j] *= win[j];               /* fbank.h:130-135 */
    for (int j = frame_length; j < N; ++j) re[j] = 0.0f;
    memset(im, 0, sizeof(float) * N);
    fft_forward(rev, tbl, re, im, N);
    for (int j = 0; j < NB; ++j) power[j] = re[j] * re[j] + im[j] * im[j];  /* fbank.h:173-175 */
    for (int b = 0; b < num_bins; ++b) {                                  /* fbank.h:179-190 */
      float e = 0.0f;
      const float* w = melw + (size_t)b * NB + first[b];
      for (int k = 0; k < count[b]; ++k) e += w[k] * power[first[b] + k];
      if (e < FLT_EPSILON) e = FLT_EPSILON;
      out[(size_t)f * num_bins + b] = logf(e);
    }
  }
  free(tbl); free(rev); free(win); free(re); free(im); free(power); free(melw); free(first); free(count);
  return nframes;
}
