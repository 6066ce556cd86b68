// kws_main on MI355X: same command line and same output lines as the reference
// (runtime/core/bin/kws_main.cc:23-61): wav -> fbank (GPU) -> KeywordSpotting::Forward in chunks of batch_size
// frames with the carried streaming cache -> "frame <i> prob <p0> <p1> ...".
#include <iostream>
#include <string>
#include <vector>

#include "frontend/feature_pipeline.h"
#include "frontend/wav.h"
#include "kws/keyword_spotting.h"
#include "utils/check.h"

int main(int argc, char* argv[]) {
  if (argc != 5) {
    WEKWS_FATAL() << "Usage: kws_main fbank_dim(int) batch_size(int) kws_model_path test_wav_path";
  }
  const int num_bins = std::stoi(argv[1]);  // Fbank feature dim
  const int batch_size = std::stoi(argv[2]);
  const std::string model_path = argv[3];
  const std::string wav_path = argv[4];

  wenet::WavReader wav_reader(wav_path);
  WEKWS_CHECK(wav_reader.ok()) << "cannot read " << wav_path;
  wenet::FeaturePipelineConfig feature_config(num_bins, 16000);
  wenet::FeaturePipeline feature_pipeline(feature_config);
  std::vector<float> wav(wav_reader.data(), wav_reader.data() + wav_reader.num_samples());
  feature_pipeline.AcceptWaveform(wav);
  feature_pipeline.set_input_finished();

  wekws::KeywordSpotting spotter(model_path);
  WEKWS_CHECK(spotter.feature_dim() == num_bins) << "model expects " << spotter.feature_dim() << "-d features";

  int offset = 0;  // simulate streaming, detect batch by batch
  while (true) {
    std::vector<std::vector<float>> feats;
    const bool ok = feature_pipeline.Read(batch_size, &feats);
    std::vector<std::vector<float>> prob;
    spotter.Forward(feats, &prob);
    for (size_t i = 0; i < prob.size(); i++) {
      std::cout << "frame " << offset + i << " prob";
      for (size_t j = 0; j < prob[i].size(); j++) std::cout << " " << prob[i][j];
      std::cout << std::endl;
    }
    if (!ok) break;  // reached the end of the feature pipeline
    offset += prob.size();
  }
  return 0;
}
