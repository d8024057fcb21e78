// Internal: model context = named host tensors staged by idxtts_ctx_load_tensor, then packed
// into device-resident kernel layouts by finalize().
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "conv1d.h"

namespace idxtts {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

struct DeviceArena {   // owns hipMalloc'd blocks of a context
  std::vector<void*> blocks;
  ~DeviceArena() { for (void* p : blocks) (void)hipFree(p); }
  int upload(const float* host, size_t n, float** out) {
    void* d = nullptr;
    IDX_HIP(hipMalloc(&d, (n ? n : 1) * sizeof(float)));
    blocks.push_back(d);
    if (n) IDX_HIP(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    *out = static_cast<float*>(d);
    return 0;
  }
  int upload_bytes(const void* host, size_t bytes, void** out) {
    void* d = nullptr;
    IDX_HIP(hipMalloc(&d, bytes ? bytes : 1));
    blocks.push_back(d);
    if (bytes) IDX_HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    *out = d;
    return 0;
  }
};

struct ModelBase {
  virtual ~ModelBase() {}
  virtual bool accepts(const std::string& name) const = 0;
  virtual int finalize(std::map<std::string, HostTensor>& t, DeviceArena& arena) = 0;
};

}  // namespace idxtts

struct idxtts_ctx {
  std::map<std::string, idxtts::HostTensor> tensors;
  idxtts::DeviceArena arena;
  std::unique_ptr<idxtts::ModelBase> model;
  bool finalized = false;
};
