// kws_main for the MI355X runtime.  Observable behaviour follows the reference tool
// (runtime/core/bin/kws_main.cc:23-61) so that scripts built around it keep working:
//   kws_main fbank_dim(int) batch_size(int) kws_model_path test_wav_path
//   -> one line per frame:  "frame <index> prob <p0> <p1> ..."   (6 significant digits, like operator<<(float))
// The wav file is decoded, its features come from the GPU fbank (wenet::FeaturePipeline here), and the spotter is fed
// `batch_size` frames at a time with its streaming cache carried from call to call.  model_path may be the exporter's
// .onnx, an ORT-format .ort, or a packed model (runtime/kws/model_file.h).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "frontend/feature_pipeline.h"
#include "frontend/wav.h"
#include "kws/keyword_spotting.h"
#include "utils/check.h"

namespace {

struct Options {
  int feature_dim = 0;
  int frames_per_call = 0;
  std::string model, audio;
};

Options ParseCommandLine(int argc, char** argv) {
  if (argc != 5) WEKWS_FATAL() << "Usage: kws_main fbank_dim(int) batch_size(int) kws_model_path test_wav_path";
  Options o;
  o.feature_dim = std::atoi(argv[1]);
  o.frames_per_call = std::atoi(argv[2]);
  o.model = argv[3];
  o.audio = argv[4];
  WEKWS_CHECK(o.feature_dim > 0 && o.frames_per_call > 0) << "fbank_dim and batch_size must be positive integers";
  return o;
}

using Frames = std::vector<std::vector<float>>;

// "frame <n> prob <p...>" for every row, numbered from first_index
void PrintRows(const Frames& rows, size_t first_index) {
  std::string line;
  char num[48];
  for (size_t r = 0; r < rows.size(); ++r) {
    line = "frame " + std::to_string(first_index + r) + " prob";
    for (float p : rows[r]) {
      std::snprintf(num, sizeof(num), " %g", static_cast<double>(p));
      line += num;
    }
    std::puts(line.c_str());
  }
  std::fflush(stdout);
}

}  // namespace

int main(int argc, char** argv) {
  const Options opt = ParseCommandLine(argc, argv);

  // whole file in, as the reference tool does; the pipeline hands frames out batch by batch below
  std::vector<float> samples;
  {
    wenet::WavReader reader(opt.audio);
    WEKWS_CHECK(reader.ok()) << "cannot read " << opt.audio;
    samples.assign(reader.data(), reader.data() + reader.num_samples());
  }
  wenet::FeaturePipeline features(wenet::FeaturePipelineConfig(opt.feature_dim, 16000));
  features.AcceptWaveform(samples);
  features.set_input_finished();

  wekws::KeywordSpotting spotter(opt.model);
  WEKWS_CHECK(spotter.feature_dim() == opt.feature_dim) << "model expects " << spotter.feature_dim() << "-d features";

  size_t emitted = 0;
  for (bool more = true; more;) {
    Frames chunk, scores;
    more = features.Read(opt.frames_per_call, &chunk);   // false once the pipeline has been drained
    spotter.Forward(chunk, &scores);
    PrintRows(scores, emitted);
    emitted += scores.size();
  }
  return 0;
}
