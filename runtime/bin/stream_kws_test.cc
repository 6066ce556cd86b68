// Threaded test driver (not one of the reference's tools): the producer / consumer arrangement wenet::FeaturePipeline exists for
// (runtime/core/frontend/feature_pipeline.h:48-54; the reference's stream_kws_main.cc:63-93 runs it against a microphone).
//   stream_kws_test fbank_dim batch_size model wav n_streams push_size...
// n_streams independent streams run at once, each with its own HIP stream, its own FeaturePipeline, its own KeywordSpotting and
// TWO host threads: a producer that pushes the wav's PCM in pieces of the given sizes (the last size repeats; a short sleep
// between pushes) and marks the input finished, and a consumer that blocks in Read(batch_size) and feeds Forward with the carried
// cache.  Prints stream 0's rows in kws_main's format -- they must equal the offline tool's -- and fails if any other stream
// printed anything else.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "frontend/feature_pipeline.h"
#include "frontend/wav.h"
#include "kws/keyword_spotting.h"
#include "utils/check.h"

using Frames = std::vector<std::vector<float>>;

static std::string FormatRows(const Frames& rows, size_t first_index) {
  std::string out;
  char num[48];
  for (size_t r = 0; r < rows.size(); ++r) {
    out += "frame " + std::to_string(first_index + r) + " prob";
    for (float p : rows[r]) {
      std::snprintf(num, sizeof(num), " %g", static_cast<double>(p));
      out += num;
    }
    out += "\n";
  }
  return out;
}

int main(int argc, char** argv) {
  if (argc < 7) WEKWS_FATAL() << "Usage: stream_kws_test fbank_dim batch_size model wav n_streams push_size...";
  const int dim = std::atoi(argv[1]), batch = std::atoi(argv[2]), nstreams = std::atoi(argv[5]);
  const std::string model = argv[3];
  std::vector<size_t> sizes;
  for (int i = 6; i < argc; ++i) sizes.push_back(std::strtoull(argv[i], nullptr, 10));
  WEKWS_CHECK(dim > 0 && batch > 0 && nstreams > 0 && !sizes.empty());
  std::vector<int16_t> pcm;
  {
    wenet::WavReader reader(argv[4]);
    WEKWS_CHECK(reader.ok()) << "cannot read " << argv[4];
    pcm.resize(reader.num_samples());
    for (int i = 0; i < reader.num_samples(); ++i) pcm[i] = static_cast<int16_t>(reader.data()[i]);
  }
  std::vector<std::string> printed(nstreams);
  std::vector<std::thread> threads;
  std::vector<hipStream_t> streams(nstreams);
  std::vector<std::unique_ptr<wenet::FeaturePipeline>> pipes(nstreams);
  std::vector<std::unique_ptr<wekws::KeywordSpotting>> spotters(nstreams);
  for (int s = 0; s < nstreams; ++s) {
    WEKWS_CHECK(hipStreamCreateWithFlags(&streams[s], hipStreamNonBlocking) == hipSuccess);
    pipes[s] = std::make_unique<wenet::FeaturePipeline>(wenet::FeaturePipelineConfig(dim, 16000), 0, streams[s]);
    spotters[s] = std::make_unique<wekws::KeywordSpotting>(model, 0, streams[s]);
  }
  for (int s = 0; s < nstreams; ++s) {
    threads.emplace_back([&, s] {                                                  // producer
      size_t pos = 0, k = size_t(s);                                               // (every stream cuts the audio differently)
      while (pos < pcm.size()) {
        size_t n = sizes[k % sizes.size()];
        ++k;
        if (n > pcm.size() - pos) n = pcm.size() - pos;
        pipes[s]->AcceptWaveform(std::vector<int16_t>(pcm.begin() + pos, pcm.begin() + pos + n));
        pos += n;
        std::this_thread::sleep_for(std::chrono::microseconds(200 + 37 * s));
      }
      pipes[s]->set_input_finished();
    });
    threads.emplace_back([&, s] {                                                  // consumer
      size_t emitted = 0;
      for (bool more = true; more;) {
        Frames chunk, scores;
        more = pipes[s]->Read(batch, &chunk);                                      // blocks until `batch` frames or the end
        spotters[s]->Forward(chunk, &scores);
        printed[s] += FormatRows(scores, emitted);
        emitted += scores.size();
      }
    });
  }
  for (auto& t : threads) t.join();
  for (int s = 1; s < nstreams; ++s)
    if (printed[s] != printed[0]) WEKWS_FATAL() << "stream " << s << " printed something else than stream 0";
  spotters.clear();
  pipes.clear();
  for (auto st : streams) (void)hipStreamDestroy(st);
  std::fputs(printed[0].c_str(), stdout);
  return 0;
}
