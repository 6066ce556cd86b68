// Model files of the reference's deployment flow, read without onnx / onnxruntime / protobuf / flatbuffers:
//   * the packed file written by `python -m wekws_amd.bin.export_packed` (magic "WEKWSHIP"),
//   * the `.onnx` written by wekws/bin/export_onnx.py:62-77 (protobuf ModelProto),
//   * its ORT-format conversion, e.g. runtime/android/app/src/main/assets/kws.ort (FlatBuffers, "ORTM"),
// so that `wekws::KeywordSpotting(model_path)` takes the same file the reference's constructor takes
// (runtime/core/kws/keyword_spotting.cc:28-45).  An exported graph is *recognised* (DS-TCN / TCN / MDTC / FSMN with a
// per-frame or pooled head), never executed: its constants are re-packed into the descriptor + folded float32 blob that
// wekws_hip_create consumes.  C++ twin of wekws_amd/utils/onnx_model.py + onnx_lower.py + pack.py (same checks, same
// blob, tests/test_runtime_cpp.py compares the two).  Everything here throws std::runtime_error with a message.
#ifndef RUNTIME_KWS_MODEL_FILE_H_
#define RUNTIME_KWS_MODEL_FILE_H_

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "wekws_hip.h"

namespace wekws {

struct ModelTensor {
  std::vector<int64_t> dims;
  std::vector<float> f;    // float32 payload (float / double tensors)
  std::vector<int64_t> i;  // integer payload (int32 / int64 / bool tensors)
  bool is_float = true;
  size_t size() const { return is_float ? f.size() : i.size(); }
};

struct ModelAttr {
  int64_t i = 0;
  float f = 0.f;
  std::string s;
  std::vector<int64_t> ints;
  ModelTensor t;
  bool has_tensor = false;
};

struct ModelNode {
  std::string op, name;
  std::vector<std::string> in, out;
  std::map<std::string, ModelAttr> attr;
};

struct ModelGraph {
  std::vector<ModelNode> nodes;               // topologically ordered
  std::map<std::string, ModelTensor> init;    // initializers + folded Constant nodes
  std::vector<std::string> inputs, outputs;   // inputs exclude initializers
  std::map<std::string, std::string> meta;    // metadata_props (cache_dim, cache_len)
};

ModelGraph ParseModelBytes(const std::string& bytes);  // .onnx or .ort image
void LowerGraph(const ModelGraph& g, wekws_hip_desc* desc, std::vector<float>* blob);
void ReadModelFile(const std::string& path, wekws_hip_desc* desc, std::vector<float>* blob);  // any of the three

}  // namespace wekws
#endif  // RUNTIME_KWS_MODEL_FILE_H_
