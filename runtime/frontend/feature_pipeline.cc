#include "frontend/feature_pipeline.h"

#include <hip/hip_runtime_api.h>

#include "utils/check.h"
#include "utils/device_guard.h"

namespace wenet {

FeaturePipeline::FeaturePipeline(const FeaturePipelineConfig& config, int device, void* stream)
    : config_(config), device_(device), stream_(stream) {
  wekws_hip_fbank_cfg cfg{};
  cfg.num_bins = config.num_bins;
  cfg.sample_rate = config.sample_rate;
  cfg.frame_length = config.frame_length;
  cfg.frame_shift = config.frame_shift;
  cfg.window = WEKWS_HIP_WINDOW_HAMMING;  // the C++ runtime's window (fbank.h:90-96)
  WEKWS_CHECK(wekws_hip_fbank_create(&cfg, device_, &fbank_) == WEKWS_HIP_OK) << wekws_hip_last_error();
}

FeaturePipeline::~FeaturePipeline() {
  wekws::ScopedDevice dev(device_);
  if (d_pcm_) (void)hipFree(d_pcm_);
  if (d_feats_) (void)hipFree(d_feats_);
  wekws_hip_fbank_destroy(fbank_);
}

// n samples of `bytes_per_sample` (4: float in int16 scale, 2: int16) -> frames appended to the queue
void FeaturePipeline::Extract(const void* host_pcm, size_t bytes_per_sample, int n) {
  const int nf = wekws_hip_fbank_num_frames(fbank_, n);
  if (nf <= 0) return;
  hipStream_t st = static_cast<hipStream_t>(stream_);
  wekws::ScopedDevice dev(device_);   // current for the allocations, copies and the launch; caller's device restored
  WEKWS_CHECK(dev.ok()) << "hipSetDevice(" << device_ << ")";
  const size_t bytes = size_t(n) * bytes_per_sample;
  if (bytes > cap_pcm_bytes_) {
    if (d_pcm_) (void)hipFree(d_pcm_);
    WEKWS_CHECK(hipMalloc(&d_pcm_, bytes) == hipSuccess);
    cap_pcm_bytes_ = bytes;
  }
  if (static_cast<size_t>(nf) > cap_frames_) {
    if (d_feats_) (void)hipFree(d_feats_);
    WEKWS_CHECK(hipMalloc(reinterpret_cast<void**>(&d_feats_), size_t(nf) * config_.num_bins * sizeof(float)) == hipSuccess);
    cap_frames_ = nf;
  }
  WEKWS_CHECK(hipMemcpyAsync(d_pcm_, host_pcm, bytes, hipMemcpyHostToDevice, st) == hipSuccess);
  const int rc = bytes_per_sample == 2
                     ? wekws_hip_fbank_compute_i16(fbank_, static_cast<const int16_t*>(d_pcm_), 1, n, d_feats_, stream_)
                     : wekws_hip_fbank_compute(fbank_, static_cast<const float*>(d_pcm_), 1, n, d_feats_, stream_);
  WEKWS_CHECK(rc == WEKWS_HIP_OK) << wekws_hip_last_error();
  std::vector<float> host(size_t(nf) * config_.num_bins);
  WEKWS_CHECK(hipMemcpyAsync(host.data(), d_feats_, host.size() * sizeof(float), hipMemcpyDeviceToHost, st) == hipSuccess);
  WEKWS_CHECK(hipStreamSynchronize(st) == hipSuccess);
  std::lock_guard<std::mutex> lock(mutex_);
  for (int i = 0; i < nf; ++i)
    queue_.emplace_back(host.begin() + size_t(i) * config_.num_bins, host.begin() + size_t(i + 1) * config_.num_bins);
  num_frames_ += nf;
}

void FeaturePipeline::AcceptWaveform(const std::vector<float>& wav) {
  // feature_pipeline.cc:30-47: frames are cut from [leftover | new samples]; what does not fill a hop is kept
  std::vector<float> waves;
  waves.reserve(remained_wav_.size() + wav.size());
  waves.insert(waves.end(), remained_wav_.begin(), remained_wav_.end());
  waves.insert(waves.end(), wav.begin(), wav.end());
  const int n = static_cast<int>(waves.size());
  const int nf = wekws_hip_fbank_num_frames(fbank_, n);
  Extract(waves.data(), sizeof(float), n);
  const int consumed = config_.frame_shift * nf;  // feature_pipeline.cc:41-44
  remained_wav_.assign(waves.begin() + consumed, waves.end());
  remained_integral_ = false;
  cv_.notify_one();
}

void FeaturePipeline::AcceptWaveform(const std::vector<int16_t>& wav) {
  // feature_pipeline.cc:49-55 widens to float on the host (no /32768) and calls the float overload.  Here the samples
  // travel as int16 and are widened in the kernel's registers -- half the PCIe and HBM bytes, the same numbers -- as long
  // as the leftover is int16 too (it is, unless float pushes were mixed in).
  if (!remained_integral_) {
    std::vector<float> f(wav.size());
    for (size_t i = 0; i < wav.size(); ++i) f[i] = static_cast<float>(wav[i]);
    AcceptWaveform(f);
    return;
  }
  std::vector<int16_t> waves;
  waves.reserve(remained_wav_.size() + wav.size());
  for (float v : remained_wav_) waves.push_back(static_cast<int16_t>(v));   // exact: they were int16
  waves.insert(waves.end(), wav.begin(), wav.end());
  const int n = static_cast<int>(waves.size());
  const int nf = wekws_hip_fbank_num_frames(fbank_, n);
  Extract(waves.data(), sizeof(int16_t), n);
  const int consumed = config_.frame_shift * nf;
  remained_wav_.assign(waves.begin() + consumed, waves.end());
  cv_.notify_one();
}

void FeaturePipeline::set_input_finished() {
  WEKWS_CHECK(!input_finished_);
  {
    std::lock_guard<std::mutex> lock(mutex_);
    input_finished_ = true;
  }
  cv_.notify_all();
}

bool FeaturePipeline::ReadOne(std::vector<float>* feat) {
  std::unique_lock<std::mutex> lock(mutex_);
  cv_.wait(lock, [this] { return !queue_.empty() || input_finished_; });
  if (queue_.empty()) return false;  // finished and drained (feature_pipeline.cc:66-86)
  *feat = std::move(queue_.front());
  queue_.pop_front();
  return true;
}

bool FeaturePipeline::Read(int num_frames, std::vector<std::vector<float>>* feats) {
  feats->clear();
  std::vector<float> feat;
  while (static_cast<int>(feats->size()) < num_frames) {  // feature_pipeline.cc:88-104
    if (!ReadOne(&feat)) return false;
    feats->push_back(std::move(feat));
  }
  return true;
}

void FeaturePipeline::Reset() {  // feature_pipeline.cc:106-111
  std::lock_guard<std::mutex> lock(mutex_);
  input_finished_ = false;
  num_frames_ = 0;
  remained_wav_.clear();
  remained_integral_ = true;
  queue_.clear();
}

int FeaturePipeline::NumQueuedFrames() const {
  std::lock_guard<std::mutex> lock(mutex_);
  return static_cast<int>(queue_.size());
}

}  // namespace wenet
