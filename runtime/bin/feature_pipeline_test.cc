// Test driver for wenet::FeaturePipeline (not part of the reference's tools): pushes a raw int16 PCM file through
// AcceptWaveform in pieces of the given sizes -- the framing / leftover rule of feature_pipeline.cc:30-47 under uneven
// pushes -- through either overload, and writes the frames it reads back as raw float32.
//   feature_pipeline_test <num_bins> <f32|i16|mixed> <pcm.raw> <out.f32> <push sizes...>   (the last size repeats)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "frontend/feature_pipeline.h"
#include "utils/check.h"

int main(int argc, char** argv) {
  if (argc < 6) WEKWS_FATAL() << "Usage: feature_pipeline_test num_bins f32|i16|mixed pcm.raw out.f32 push_size...";
  const int bins = std::atoi(argv[1]);
  const std::string mode = argv[2];
  std::vector<int16_t> pcm;
  {
    FILE* f = std::fopen(argv[3], "rb");
    WEKWS_CHECK(f) << "cannot read " << argv[3];
    int16_t buf[4096];
    size_t n;
    while ((n = std::fread(buf, sizeof(int16_t), 4096, f)) > 0) pcm.insert(pcm.end(), buf, buf + n);
    std::fclose(f);
  }
  std::vector<size_t> sizes;
  for (int i = 5; i < argc; ++i) sizes.push_back(std::strtoull(argv[i], nullptr, 10));
  wenet::FeaturePipeline fp(wenet::FeaturePipelineConfig(bins, 16000));
  size_t pos = 0, k = 0, pushes = 0;
  while (pos < pcm.size()) {
    size_t n = sizes[k < sizes.size() ? k : sizes.size() - 1];
    ++k;
    if (n > pcm.size() - pos) n = pcm.size() - pos;
    const bool as_i16 = mode == "i16" || (mode == "mixed" && pushes % 2 == 0);
    if (as_i16) {
      fp.AcceptWaveform(std::vector<int16_t>(pcm.begin() + pos, pcm.begin() + pos + n));
    } else {
      std::vector<float> f(n);
      for (size_t i = 0; i < n; ++i) f[i] = static_cast<float>(pcm[pos + i]);
      fp.AcceptWaveform(f);
    }
    pos += n;
    ++pushes;
  }
  fp.set_input_finished();
  FILE* out = std::fopen(argv[4], "wb");
  WEKWS_CHECK(out) << "cannot write " << argv[4];
  std::vector<float> frame;
  int frames = 0;
  while (fp.ReadOne(&frame)) {
    std::fwrite(frame.data(), sizeof(float), frame.size(), out);
    ++frames;
  }
  std::fclose(out);
  WEKWS_CHECK(frames == fp.num_frames());
  std::printf("%d frames from %zu pushes\n", frames, pushes);
  return 0;
}
