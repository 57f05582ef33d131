// Sanitizer driver for the C++17 host layer (built by tests/test_sanitizers.py with
// -fsanitize=address,undefined from the host sources themselves): everything of the layer that
// runs without a GPU -- status / tensor / buffer plumbing on the CPU allocator, the KV cache
// arena's layout for every layer, the tensor-parallel shard plan over the benchmark geometries,
// config validation, the int8 weight + scales blob contract of LayerParam::set_weight, and the
// model's refusal paths when no device memory can be had.
#include <cstdio>
#include <cstring>
#include <vector>

#include "base/alloc.h"
#include "base/buffer.h"
#include "base/kv_cache_manager.h"
#include "kx_host.h"
#include "model/llama.h"
#include "model/shard_plan.h"
#include "op/matmul.h"
#include "tensor/tensor.h"

static int fails = 0;
#define EXPECT(c)                                              \
  do {                                                         \
    if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } \
  } while (0)

int main() {
  auto cpu = base::CPUDeviceAllocatorFactory::get_instance();
  {  // tensors and buffers
    tensor::Tensor t(base::DataType::kDataTypeFp32, 3, 5, true, cpu);
    EXPECT(t.size() == 15 && t.byte_size() == 60 && t.dims_size() == 2);
    for (int i = 0; i < 15; ++i) t.index<float>(i) = (float)i;
    tensor::Tensor c = t.clone();
    EXPECT(c.index<float>(14) == 14.f && c.ptr<float>() != t.ptr<float>());
    t.reshape({5, 3});
    EXPECT(t.get_dim(0) == 5);
    std::vector<float> ext(8, 2.f);
    tensor::Tensor v(base::DataType::kDataTypeFp32, 8, false, nullptr, ext.data());
    EXPECT(v.ptr<float>() == ext.data());
    auto b1 = std::make_shared<base::Buffer>(64, cpu);
    auto b2 = std::make_shared<base::Buffer>(32, cpu);
    std::memset(b1->ptr(), 7, 64);
    b2->copy_from(*b1);  // the smaller size wins
    EXPECT(static_cast<unsigned char*>(b2->ptr())[31] == 7);
  }
  {  // KV cache arena: K block then V block, every layer's rows inside the arena
    base::KVCacheConfig c;
    c.max_seq_len = 33; c.layer_num = 5; c.kv_head_num = 3; c.kv_dim = 3 * 16;
    base::KVCacheManager mgr(base::DeviceType::kDeviceCPU, c);
    EXPECT(mgr.allocate());
    char* base_ptr = static_cast<char*>(mgr.get_raw_buffer_k());
    for (int l = 0; l < c.layer_num; ++l) {
      auto [k, v] = mgr.get_cache_ptr(l);
      float* kl = static_cast<float*>(k);
      float* vl = static_cast<float*>(v);
      kl[(size_t)c.max_seq_len * c.kv_dim - 1] = 1.f;  // last element of the layer: must be inside
      vl[(size_t)c.max_seq_len * c.kv_dim - 1] = 1.f;
      EXPECT(static_cast<char*>(v) + (size_t)c.max_seq_len * c.kv_dim * 4 <= base_ptr + mgr.total_bytes());
    }
  }
  {  // shard plans of the benchmark geometries
    struct G { int dim, hid, H, KVH, tp; } gs[] = {{4096, 14336, 32, 8, 8}, {3584, 18944, 28, 4, 4},
                                                   {8192, 28672, 64, 8, 8}, {2048, 8192, 32, 8, 2},
                                                   {768, 2048, 12, 12, 5}};
    for (const G& g : gs)
      for (int r = 0; r < g.tp; ++r) {
        model::ShardPlan p;
        const bool ok = model::make_shard_plan(g.dim, g.hid, g.H, g.KVH, g.tp, r, &p);
        EXPECT(ok == (g.H % g.tp == 0 && g.KVH % g.tp == 0 && g.hid % g.tp == 0));
        if (ok) EXPECT(p.wq.row0 + p.wq.rows <= g.dim && p.w2.col0 + p.w2.cols <= g.hid);
        int32_t out[28], res = 0;
        (void)kxm_shard_plan(g.dim, g.hid, g.H, g.KVH, g.tp, r, out, &res);
      }
  }
  {  // config validation
    model::TransformerConfig c;
    c.dim = 256; c.hidden_dim = 512; c.layer_num = 2; c.head_num = 4; c.kv_head_num = 2;
    c.vocab_size = 100; c.seq_len = 16; c.group_size = 64;
    EXPECT(c.finalize());
    c.tp_size = 4; c.tp_rank = 1;
    EXPECT(!c.finalize());  // two kv heads do not divide over four ranks
    c.tp_size = 1; c.tp_rank = 0; c.head_num = 3;
    EXPECT(!c.finalize());
  }
  {  // the int8 weight blob: [dim0*dim1 int8][dim0*dim1/gs fp32 scales], scales found right behind
    const int d0 = 4, d1 = 128, gs = 64;
    std::vector<unsigned char> blob((size_t)d0 * d1 + (size_t)d0 * d1 / gs * 4);  // exact size
    for (size_t i = 0; i < blob.size(); ++i) blob[i] = (unsigned char)i;
    op::MatmulLayer layer(base::DeviceType::kDeviceCPU, d0, d1, true, false);
    layer.set_group_size(gs);
    EXPECT(layer.set_weight(0, {d0, d1}, blob.data(), base::DeviceType::kDeviceCPU));
  }
  {  // no GPU here: creating a model must fail cleanly, with a message
    kxm_config c{};
    c.dim = 256; c.hidden_dim = 512; c.layer_num = 1; c.head_num = 4; c.kv_head_num = 2;
    c.vocab_size = 64; c.seq_len = 16; c.group_size = 64; c.tp_size = 1; c.rope_theta = 1e4f; c.rms_eps = 1e-5f;
    void* m = kxm_create(&c);
    if (m) kxm_destroy(m);  // a GPU is present after all
    else EXPECT(std::strlen(kxm_last_error()) > 0);
    c.head_num = 3;
    EXPECT(kxm_create(&c) == nullptr);
  }
  std::printf("host sanitizer driver: %d failure(s)\n", fails);
  return fails ? 1 : 0;
}
