// Minimal RIFF/WAVE reader: PCM 8 / 16 / 32-bit, any channel count (channel 0 is used), samples returned as floats
// in their integer scale WITHOUT normalisation -- the convention of the reference reader
// (runtime/core/frontend/wav.h:90-114) that the fbank front-end expects.
#ifndef RUNTIME_FRONTEND_WAV_H_
#define RUNTIME_FRONTEND_WAV_H_

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace wenet {

class WavReader {
 public:
  explicit WavReader(const std::string& path) { ok_ = Open(path); }
  bool ok() const { return ok_; }
  int num_channel() const { return channels_; }
  int sample_rate() const { return sample_rate_; }
  int bits_per_sample() const { return bits_; }
  int num_samples() const { return static_cast<int>(data_.size()); }
  const float* data() const { return data_.data(); }

 private:
  bool Open(const std::string& path) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char id[4];
    uint32_t sz = 0;
    bool good = std::fread(id, 1, 4, f) == 4 && !std::memcmp(id, "RIFF", 4) && std::fread(&sz, 4, 1, f) == 1 &&
                std::fread(id, 1, 4, f) == 4 && !std::memcmp(id, "WAVE", 4);
    bool have_fmt = false;
    while (good && std::fread(id, 1, 4, f) == 4 && std::fread(&sz, 4, 1, f) == 1) {
      if (!std::memcmp(id, "fmt ", 4)) {
        uint16_t fmt = 0, ch = 0, align = 0, bits = 0;
        uint32_t rate = 0, bps = 0;
        good = std::fread(&fmt, 2, 1, f) == 1 && std::fread(&ch, 2, 1, f) == 1 && std::fread(&rate, 4, 1, f) == 1 &&
               std::fread(&bps, 4, 1, f) == 1 && std::fread(&align, 2, 1, f) == 1 && std::fread(&bits, 2, 1, f) == 1 &&
               fmt == 1 && ch >= 1 && (bits == 8 || bits == 16 || bits == 32);
        channels_ = ch; sample_rate_ = rate; bits_ = bits; have_fmt = good;
        if (sz > 16) std::fseek(f, sz - 16, SEEK_CUR);
      } else if (!std::memcmp(id, "data", 4) && have_fmt) {
        const int bytes = bits_ / 8;
        const size_t n = sz / (size_t(bytes) * channels_);
        std::vector<uint8_t> raw(n * bytes * channels_);
        good = std::fread(raw.data(), 1, raw.size(), f) == raw.size();
        data_.resize(good ? n : 0);
        for (size_t i = 0; i < data_.size(); ++i) {
          const uint8_t* p = raw.data() + i * bytes * channels_;
          if (bits_ == 8) data_[i] = static_cast<float>(*reinterpret_cast<const int8_t*>(p));
          else if (bits_ == 16) { int16_t v; std::memcpy(&v, p, 2); data_[i] = static_cast<float>(v); }
          else { int32_t v; std::memcpy(&v, p, 4); data_[i] = static_cast<float>(v); }
        }
        std::fclose(f);
        return good;
      } else {
        std::fseek(f, sz + (sz & 1), SEEK_CUR);
      }
    }
    std::fclose(f);
    return false;
  }
  bool ok_ = false;
  int channels_ = 0, sample_rate_ = 0, bits_ = 0;
  std::vector<float> data_;
};

}  // namespace wenet
#endif  // RUNTIME_FRONTEND_WAV_H_
