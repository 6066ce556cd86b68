// Thin C driver around the REFERENCE's own front-end, compiled from where it lies under /root/reference
// (never copied into this repo) into oracle/_ref/libref_fbank.so by oracle/Makefile.  TEST INFRASTRUCTURE.
// Calls wenet::FeaturePipeline::AcceptWaveform (runtime/core/frontend/feature_pipeline.cc:30-55), i.e. the
// reference's framing + wenet::Fbank::Compute, optionally in several uneven pushes to exercise the
// leftover-sample rule, and copies the queued frames out.
#include <vector>

#include "frontend/feature_pipeline.h"

extern "C" int ref_fbank(const float* wave, int nsamp, int num_bins, int sample_rate, int first_push, float* out) {
  wenet::FeaturePipelineConfig config(num_bins, sample_rate);
  wenet::FeaturePipeline pipe(config);
  int pushed = 0;
  if (first_push > 0 && first_push < nsamp) {
    pipe.AcceptWaveform(std::vector<float>(wave, wave + first_push));
    pushed = first_push;
  }
  pipe.AcceptWaveform(std::vector<float>(wave + pushed, wave + nsamp));
  pipe.set_input_finished();
  std::vector<float> frame;
  int n = 0;
  while (pipe.ReadOne(&frame)) {
    for (int j = 0; j < num_bins; ++j) out[n * num_bins + j] = frame[j];
    ++n;
  }
  return n;
}
