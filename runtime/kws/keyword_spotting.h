// wekws::KeywordSpotting on MI355X: the reference class (runtime/core/kws/keyword_spotting.h:26-55) with its
// ONNX Runtime session replaced by libwekws_hip.so (include/wekws_hip.h).  Same four public members, so
// runtime/core/bin/kws_main.cc-style callers compile unchanged.  `model_path` names a packed-model file written by
// wekws_amd.pack.save_packed (magic, 16 x int32 descriptor, uint64 count, float32 folded weights).
// One instance per stream; not re-entrant (it carries the streaming cache), exactly like the reference.
#ifndef RUNTIME_KWS_KEYWORD_SPOTTING_H_
#define RUNTIME_KWS_KEYWORD_SPOTTING_H_

#include <string>
#include <vector>

#include "wekws_hip.h"

namespace wekws {

class KeywordSpotting {
 public:
  // device / stream: the GPU and the hipStream_t (nullptr = the device's default stream) every call of this instance
  // works on -- the reference's constructor has no such arguments (ORT on CPU); with the defaults the class is
  // source-compatible with it.  Instances on different streams / devices run concurrently.
  explicit KeywordSpotting(const std::string& model_path, int device = 0, void* stream = nullptr);
  ~KeywordSpotting();
  KeywordSpotting(const KeywordSpotting&) = delete;
  KeywordSpotting& operator=(const KeywordSpotting&) = delete;

  // Call reset if keyword is detected (keyword_spotting.cc:47-54 zero-fills the cache; here the next Forward
  // starts from the empty-cache sentinel, which is the same thing: tcn.py:49-52)
  void Reset();

  // keyword_spotting.h:32-35 configures ORT's CPU thread pools; nothing to configure on the GPU path
  static void InitEngineThreads(int /*num_threads*/) {}

  // feats: T x feature_dim; prob: cleared, then T x output_dim (keyword_spotting.cc:56-95)
  void Forward(const std::vector<std::vector<float>>& feats, std::vector<std::vector<float>>* prob);

  int feature_dim() const { return idim_; }
  int output_dim() const { return odim_; }

 private:
  void EnsureCapacity(int frames);

  wekws_hip_model* model_ = nullptr;
  const int device_;
  void* const stream_;
  int idim_ = 0, odim_ = 0;
  float* d_x_ = nullptr;
  float* d_y_ = nullptr;
  float* d_cache_[2] = {nullptr, nullptr};
  int cap_frames_ = 0;
  int cur_ = 0;
  bool have_cache_ = false;
  std::vector<float> h_x_, h_y_;
};

}  // namespace wekws
#endif  // RUNTIME_KWS_KEYWORD_SPOTTING_H_
