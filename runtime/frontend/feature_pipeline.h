// wenet::FeaturePipeline with the extractor on the GPU: same public interface and the same framing / leftover rule
// as the reference (runtime/core/frontend/feature_pipeline.h:56-114, feature_pipeline.cc:30-111), but
// AcceptWaveform runs wekws_hip_fbank_compute (one wave per frame) instead of the scalar wenet::Fbank::Compute.
// Thread model as in the reference: one producer thread calls AcceptWaveform / set_input_finished, one consumer
// thread calls Read / ReadOne (blocking while the queue is empty and the input is not finished).
#ifndef RUNTIME_FRONTEND_FEATURE_PIPELINE_H_
#define RUNTIME_FRONTEND_FEATURE_PIPELINE_H_

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <vector>

#include "wekws_hip.h"

namespace wenet {

struct FeaturePipelineConfig {
  int num_bins;
  int sample_rate;
  int frame_length;
  int frame_shift;
  FeaturePipelineConfig(int num_bins, int sample_rate) : num_bins(num_bins), sample_rate(sample_rate) {
    frame_length = sample_rate / 1000 * 25;  // 25 ms (feature_pipeline.h:34-39)
    frame_shift = sample_rate / 1000 * 10;   // 10 ms
  }
};

class FeaturePipeline {
 public:
  // device / stream: where the extractor runs (the reference's class has neither: CPU); stream = a hipStream_t, nullptr =
  // the device's default stream.  Uploads, the kernel and the read-back are issued on it; AcceptWaveform returns after
  // the frames have reached the host queue, like the reference's.
  explicit FeaturePipeline(const FeaturePipelineConfig& config, int device = 0, void* stream = nullptr);
  ~FeaturePipeline();
  FeaturePipeline(const FeaturePipeline&) = delete;
  FeaturePipeline& operator=(const FeaturePipeline&) = delete;

  void AcceptWaveform(const std::vector<float>& wav);    // samples in int16 scale, not normalised
  void AcceptWaveform(const std::vector<int16_t>& wav);
  int num_frames() const { return num_frames_; }
  int feature_dim() const { return config_.num_bins; }
  const FeaturePipelineConfig& config() const { return config_; }
  void set_input_finished();
  bool input_finished() const { return input_finished_; }
  bool ReadOne(std::vector<float>* feat);
  bool Read(int num_frames, std::vector<std::vector<float>>* feats);
  void Reset();
  bool IsLastFrame(int frame) const { return input_finished_ && (frame == num_frames_ - 1); }
  int NumQueuedFrames() const;

 private:
  const FeaturePipelineConfig config_;
  void Extract(const void* host_pcm, size_t bytes_per_sample, int n);   // upload + fbank + frames into the queue
  wekws_hip_fbank* fbank_ = nullptr;
  const int device_;
  void* const stream_;
  void* d_pcm_ = nullptr;                // float or int16 samples, as pushed
  float* d_feats_ = nullptr;
  size_t cap_pcm_bytes_ = 0, cap_frames_ = 0;
  std::vector<float> remained_wav_;      // leftover samples (feature_pipeline.cc:41-44); int16 values are exact in float
  bool remained_integral_ = true;        // every leftover sample came from an int16 push -> the next int16 push can
                                         // upload 2 bytes per sample
  int num_frames_ = 0;
  bool input_finished_ = false;
  mutable std::mutex mutex_;
  std::condition_variable cv_;
  std::deque<std::vector<float>> queue_;
};

}  // namespace wenet
#endif  // RUNTIME_FRONTEND_FEATURE_PIPELINE_H_
