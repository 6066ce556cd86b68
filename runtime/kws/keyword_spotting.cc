#include "kws/keyword_spotting.h"

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include <exception>

#include "kws/model_file.h"
#include "utils/check.h"
#include "utils/device_guard.h"

namespace wekws {


KeywordSpotting::KeywordSpotting(const std::string& model_path, int device, void* stream)
    : device_(device), stream_(stream) {
  // model_path: what the reference's constructor takes (the exporter's .onnx / an ORT-format .ort,
  // keyword_spotting.cc:28-45) or a packed file of wekws_amd.bin.export_packed -- kws/model_file.h
  wekws_hip_desc desc;
  std::vector<float> blob;
  try {
    ReadModelFile(model_path, &desc, &blob);
  } catch (const std::exception& e) {
    WEKWS_CHECK(false) << e.what();
  }
  WEKWS_CHECK(wekws_hip_create(&desc, blob.data(), blob.size(), device_, &model_) == WEKWS_HIP_OK)
      << wekws_hip_last_error();
  ScopedDevice dev(device_);   // the buffers below belong to device_, whatever the caller's current device is
  WEKWS_CHECK(dev.ok()) << "hipSetDevice(" << device_ << ")";
  WEKWS_CHECK(desc.head == WEKWS_HIP_HEAD_LINEAR || desc.head == WEKWS_HIP_HEAD_IDENTITY)
      << "the streaming runtime needs a per-frame head";
  idim_ = desc.idim;
  odim_ = desc.odim;
  const size_t ce = wekws_hip_cache_elems(model_, 1);  // cache_dim * cache_len (keyword_spotting.cc:33-40)
  for (float*& c : d_cache_) WEKWS_CHECK(hipMalloc(reinterpret_cast<void**>(&c), ce * sizeof(float)) == hipSuccess);
}

KeywordSpotting::~KeywordSpotting() {
  ScopedDevice dev(device_);
  if (d_x_) (void)hipFree(d_x_);
  if (d_y_) (void)hipFree(d_y_);
  for (float* c : d_cache_) if (c) (void)hipFree(c);
  wekws_hip_destroy(model_);
}

void KeywordSpotting::Reset() { have_cache_ = false; }

// (called with device_ current: Forward holds the guard)
void KeywordSpotting::EnsureCapacity(int frames) {
  if (frames <= cap_frames_) return;
  (void)hipStreamSynchronize(static_cast<hipStream_t>(stream_));   // (nothing of ours is in flight: Forward ends with a sync)
  if (d_x_) (void)hipFree(d_x_);
  if (d_y_) (void)hipFree(d_y_);
  WEKWS_CHECK(hipMalloc(reinterpret_cast<void**>(&d_x_), size_t(frames) * idim_ * sizeof(float)) == hipSuccess);
  WEKWS_CHECK(hipMalloc(reinterpret_cast<void**>(&d_y_), size_t(frames) * odim_ * sizeof(float)) == hipSuccess);
  cap_frames_ = frames;
}

void KeywordSpotting::Forward(const std::vector<std::vector<float>>& feats, std::vector<std::vector<float>>* prob) {
  prob->clear();
  if (feats.empty()) return;  // keyword_spotting.cc:59
  const int T = static_cast<int>(feats.size());
  // device_ becomes current BEFORE anything is allocated, freed or synchronised (EnsureCapacity does all three) and
  // the caller's device is restored on return: two instances on two GPUs can be driven alternately from one thread
  ScopedDevice dev(device_);
  WEKWS_CHECK(dev.ok()) << "hipSetDevice(" << device_ << ")";
  EnsureCapacity(T);
  h_x_.resize(size_t(T) * idim_);
  for (int t = 0; t < T; ++t) {  // keyword_spotting.cc:63-68: (1, T, dim) row-major
    WEKWS_CHECK(static_cast<int>(feats[t].size()) == idim_) << "frame " << t << " has " << feats[t].size() << " dims";
    std::memcpy(h_x_.data() + size_t(t) * idim_, feats[t].data(), idim_ * sizeof(float));
  }
  hipStream_t st = static_cast<hipStream_t>(stream_);
  WEKWS_CHECK(hipMemcpyAsync(d_x_, h_x_.data(), h_x_.size() * sizeof(float), hipMemcpyHostToDevice, st) == hipSuccess);
  WEKWS_CHECK(wekws_hip_forward(model_, d_x_, /*B=*/1, T, have_cache_ ? d_cache_[cur_] : nullptr, d_y_,
                                d_cache_[cur_ ^ 1], /*softmax=*/0, stream_) == WEKWS_HIP_OK)
      << wekws_hip_last_error();
  cur_ ^= 1;  // keyword_spotting.cc:82: r_cache becomes the next call's cache
  have_cache_ = true;
  h_y_.resize(size_t(T) * odim_);
  WEKWS_CHECK(hipMemcpyAsync(h_y_.data(), d_y_, h_y_.size() * sizeof(float), hipMemcpyDeviceToHost, st) == hipSuccess);
  // the caller reads *prob next: the one sync of a Forward -- inside the status query, which also says whether a bounded
  // device-side wait of the forward gave up (the reference's ORT Run would have thrown)
  WEKWS_CHECK(wekws_hip_forward_status(model_, stream_) == WEKWS_HIP_OK) << wekws_hip_last_error();
  prob->resize(T);  // keyword_spotting.cc:89-94
  for (int t = 0; t < T; ++t) (*prob)[t].assign(h_y_.begin() + size_t(t) * odim_, h_y_.begin() + size_t(t + 1) * odim_);
}

}  // namespace wekws
