// model_convert exported_model(.onnx|.ort|packed) packed_out
// Host-only twin of `python -m wekws_amd.bin.export_packed --exported`: reads what the reference's exporter wrote
// (wekws/bin/export_onnx.py:62-77, or its ORT-format conversion) with runtime/kws/model_file.cc and writes the packed
// file (magic "WEKWSHIP" | 16 x int32 descriptor | uint64 n | n x float32).  Mostly a test hook: tests/test_runtime_cpp.py
// compares its output with the Python reader's.
#include <cstdint>
#include <cstdio>
#include <exception>
#include <vector>

#include "kws/model_file.h"

int main(int argc, char** argv) {
  if (argc != 3) {
    std::fprintf(stderr, "Usage: model_convert exported_model(.onnx|.ort) packed_out\n");
    return 1;
  }
  wekws_hip_desc desc;
  std::vector<float> blob;
  try {
    wekws::ReadModelFile(argv[1], &desc, &blob);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 2;
  }
  std::FILE* f = std::fopen(argv[2], "wb");
  if (!f) {
    std::fprintf(stderr, "cannot write %s\n", argv[2]);
    return 2;
  }
  const uint64_t n = blob.size();
  std::fwrite("WEKWSHIP", 1, 8, f);
  std::fwrite(&desc, sizeof(desc), 1, f);
  std::fwrite(&n, sizeof(n), 1, f);
  std::fwrite(blob.data(), sizeof(float), blob.size(), f);
  std::fclose(f);
  std::printf("backbone=%d idim=%d hdim=%d odim=%d head=%d activation=%d floats=%llu\n", desc.backbone, desc.idim,
              desc.hdim, desc.odim, desc.head, desc.activation, static_cast<unsigned long long>(n));
  return 0;
}
