/*
 * kx_hip.h -- C ABI of libkx_hip.so: the MI355X (gfx950) kernels and the thin HIP runtime
 * wrapper that the C++17 host layer (and any FFI) binds.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns 0 on success, a hipError_t / ncclResult_t value (>0) when the
 *     runtime failed, or a negative KX_E* code when an argument breaks a stated precondition.
 *     Nothing throws across this boundary.  kx_error_string() names any of them.
 *   - all device pointers are plain HIP device addresses; `stream` is a hipStream_t passed as
 *     void* (NULL = the default stream).  Every kernel entry is asynchronous on `stream`, never
 *     allocates, never synchronises.
 *   - matrices are row-major.  A linear layer's weight is W[K][M]: K = out features (dim0),
 *     M = in features (dim1), as in the reference (kuiper/source/op/matmul.cpp:6-18).
 *   - KV cache: one block per K and per V, each [layer][seq_len][kv_dim] fp32; entries take the
 *     base pointer of the whole block plus the layer index
 *     (kuiper/source/op/kernels/cpu/mha_kernel.cpp:10, cuda/mha_kernel.cu:115).
 *
 * Each entry cites the reference interface it stands in for (paths relative to the reference).
 */
#ifndef KX_HIP_H_
#define KX_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  KX_OK = 0,
  KX_EINVAL = -1,      /* bad pointer / size / alignment */
  KX_EUNSUPPORTED = -2, /* shape outside what the kernels are built for */
  KX_EWORKSPACE = -3,  /* workspace too small */
  KX_ELIMIT = -4,      /* input larger than the fixed capacity of an entry point that cannot allocate */
};

enum { KX_ROPE_INTERLEAVED = 0, KX_ROPE_HALF_SPLIT = 1 };

const char* kx_error_string(int code);

/* ---------------- runtime: replaces CUDADeviceAllocator / cudaStream use ----------------
 * kuiper/source/base/alloc_cu.cpp (allocate/release), alloc.cpp (memcpy/memset_zero),
 * kuiper/include/base/cuda_config.h:6-17 (stream pair), matmul.cpp:95-120 (events). */
int kx_device_count(int* count);
int kx_set_device(int device);
int kx_get_device(int* device);
int kx_device_name(char* buf, size_t buf_bytes);
int kx_malloc(void** ptr, size_t bytes);
int kx_free(void* ptr);
/* stream == NULL: the fill is complete when the call returns (the library's streams are
 * non-blocking: they do not wait for work queued on the null stream).  With a stream it is
 * ordered on that stream like a kernel. */
int kx_memset(void* ptr, int value, size_t bytes, void* stream);
int kx_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream); /* stream NULL: blocking */
int kx_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream);
int kx_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int kx_mem_info(size_t* free_bytes, size_t* total_bytes); /* device memory, hipMemGetInfo */
int kx_host_alloc_pinned(void** ptr, size_t bytes);
int kx_host_free_pinned(void* ptr);
int kx_stream_create(void** stream);
int kx_stream_destroy(void* stream);
int kx_stream_sync(void* stream);
int kx_device_sync(void);
int kx_event_create(void** event);
int kx_event_destroy(void* event);
int kx_event_record(void* event, void* stream);
int kx_event_sync(void* event);
int kx_event_elapsed_ms(void* start, void* stop, float* ms);
int kx_stream_wait_event(void* stream, void* event);
/* stream capture -> replayable graph (no counterpart in the reference, which relaunches eagerly) */
int kx_graph_begin_capture(void* stream);
int kx_graph_end_capture(void* stream, void** graph_exec);
int kx_graph_launch(void* graph_exec, void* stream);
int kx_graph_destroy(void* graph_exec);

/* ---------------- bounds-checking build (make BOUNDS=1; no reference counterpart) ----------------
 * A library built with -DKX_BOUNDS compares the address of every global load, store and LDS-DMA
 * transfer of its kernels with the kx_malloc block its operand points into.  A violation does not
 * trap: the access is redirected to the block's first byte and recorded.  kx_bounds_report copies
 * up to max_records records (and the total number of violations, which may be larger than what the
 * record holds) to the host after a device synchronise; kx_bounds_reset clears the record.  In a
 * normal build kx_bounds_enabled() is 0 and the other two return KX_EUNSUPPORTED.
 * file_id: 1 gemv_w8, 2 gemm_w8_rows, 3 gemm_w8, 4 gemm_f16, 5 attn_decode, 6 attn_prefill,
 * 7 support, 8 gemv_f32, 9 gemv_engine; operand: the index the kernel file gives the buffer; offset: byte offset
 * of the access from the block's start; extent: the block's size in bytes. */
typedef struct kx_bounds_record {
  int32_t file_id, line, operand, pad_;
  int64_t offset, extent;
} kx_bounds_record;
int kx_bounds_enabled(void);
int kx_bounds_report(kx_bounds_record* out, int32_t max_records, uint64_t* total);
int kx_bounds_reset(void);

/* ---------------- W8A32 group-quant GEMV (decode) ----------------
 * Replaces matmul_kernel_cu_qint8 / matmul_kernel_cu_fp32int8
 * (kuiper/source/op/kernels/cuda/matmul_kernel.cu:56-87,111-134), reached through
 * MatmulKernelQuant (kernels_interface.h:12-14):
 *   y[p] = sum_i x[i] * scales[(p*M+i)/group_size] * (float)w[p*M+i],  p in [0,K)
 * Preconditions (as the reference: M % 4 == 0): M % 4 == 0, (K*M) % group_size == 0.
 * Fast path needs M % group_size == 0, group_size % 16 == 0, M % 16 == 0 and 16-byte aligned
 * x/w; anything else runs a slower generic kernel with identical results. */
int kx_gemv_w8_f32(const float* x, const int8_t* w, const float* scales, int32_t group_size,
                   float* y, int32_t M, int32_t K, void* stream);

/* Same GEMV with fused epilogues used by the token loop (no reference counterpart: the
 * reference launches add_kernel_cu / swiglu_kernel_cu separately, llama3.cpp:751-787):
 *   bias     (nullable) [K]  y += bias            (matmul.cpp:76-79)
 *   residual (nullable) [K]  y += residual        (llama3.cpp:751-752,787); may alias y
 * Rows [0,K) may be routed to up to three destinations (q | k-cache row | v-cache row):
 * rows [0,k0) -> y0, [k0,k1) -> y1, [k1,K) -> y2 (fused Wq|Wk|Wv, llama3.cpp:656-669). */
int kx_gemv_w8_f32_ex(const float* x, const int8_t* w, const float* scales, int32_t group_size,
                      float* y0, float* y1, float* y2, int32_t k0, int32_t k1, const float* bias,
                      const float* residual, int32_t M, int32_t K, void* stream);

/* Fused gate/up + SwiGLU: w13 = [W1;W3] stacked on rows (2*H rows), out[h] =
 * silu(W1.x)[h] * (W3.x)[h]  (llama3.cpp:761-776 + swiglu_kernel.cu:4-22). */
int kx_gemv_w8_swiglu_f32(const float* x, const int8_t* w13, const float* scales13,
                          int32_t group_size, float* out, int32_t M, int32_t H, void* stream);

/* The general fused form behind the three entries above, plus the fusions the token loop uses:
 * RMSNorm of x while it is staged (rmsnorm_kernel.cu:62-108 + llama3.cpp:637-646,755-759) and,
 * for mode KX_GEMV_QKV_ROPE, RoPE on the finished q/k rotation pairs (rope_kernel.cu) with k and
 * v written into their KV-cache row: y1/y2 point at the layer's [seq_len][kv_dim] block and are
 * advanced by pos*kv_pos_stride elements (model.cpp:237-245).  `pos_dev` (device int32) is read
 * by the kernel when non-null, so a captured graph can be replayed as the position advances;
 * otherwise `pos` is used.  Unused pointer fields are NULL, unused ints 0.  Precondition for
 * KX_GEMV_QKV_ROPE: the sin/cos tables hold at least (position + 1) rows of head_size floats and the
 * cache blocks at least (position + 1) rows of kv_pos_stride elements. */
enum { KX_GEMV_PLAIN = 0, KX_GEMV_GATE_UP = 1, KX_GEMV_QKV_ROPE = 2 };
typedef struct kx_gemv_w8_desc {
  int32_t mode;         /* KX_GEMV_* */
  int32_t group_size;
  int32_t M, K;         /* K counts all stacked rows (2*H for KX_GEMV_GATE_UP) */
  const float* x;
  const int8_t* w;
  const float* scales;
  const float* norm_w;  /* nullable: fused input RMSNorm weight [M] */
  float norm_eps;
  float* y0;            /* rows [0,k0)  (all rows when y1/y2 are NULL) */
  float* y1;            /* rows [k0,k1) */
  float* y2;            /* rows [k1,K)  */
  int32_t k0, k1;
  const float* bias;     /* nullable [K] */
  const float* residual; /* nullable [K], KX_GEMV_PLAIN only; may alias y0 */
  /* KX_GEMV_QKV_ROPE */
  const float* sin_cache;
  const float* cos_cache;
  const int32_t* pos_dev;
  int32_t pos;
  int32_t rope_variant;  /* KX_ROPE_* */
  int32_t head_size;
  int32_t kv_pos_stride;
  int32_t kv_f16;        /* non-zero: y1/y2 are fp16 cache blocks (16-bit KV cache); values are rounded once */
} kx_gemv_w8_desc;
int kx_gemv_w8_fused_f32(const kx_gemv_w8_desc* desc, void* stream);

/* Two consecutive GEMVs of the token loop's chain in ONE persistent launch (no counterpart in the
 * reference, which launches every MatmulLayer::forward on its own, llama3.cpp:648-788): ops[0] is a
 * kx_gemv_w8_fused_f32 description whose input vector is in memory; ops[1] (optional) takes the
 * output vector of ops[0] as its input (x = NULL or ops[0].y0) with its own fused RMSNorm and
 * epilogue.  One workgroup per CU streams its rows of BOTH matrices through an LDS ring without
 * stopping at the seam; ops[0]'s outputs cross the seam as 8-byte {value, tag} words in `workspace`
 * (kx_gemv_w8_chain_workspace_bytes(rows of ops[0]'s output), 256-byte aligned) besides being
 * stored to ops[0].y0 as usual.  tag = (position + 1) * 128 + chain_id (0..127): give every seam of
 * a token its own chain_id, and call kx_gemv_w8_chain_reset whenever the position is set to
 * anything but its successor (and once after allocating the workspace).  The position is read from
 * pos_dev when non-null (graph replay), else `pos`.  Results equal kx_gemv_w8_fused_f32's formulas
 * (the 16 products of a lane summed, times the group scale; the row sum in another order).
 * group_size 64, M % 1024 == 0, K % 8 == 0, head sizes that are powers of two; KX_EUNSUPPORTED
 * otherwise (run the single launches).  The launch needs every CU of the device to itself.  Every
 * in-kernel wait is bounded (1 s); one that gave up is reported by kx_gemv_w8_chain_status. */
size_t kx_gemv_w8_chain_workspace_bytes(int32_t max_rows);
int kx_gemv_w8_chain_reset(void* workspace, size_t workspace_bytes, void* stream);
int kx_gemv_w8_chain_status(void* workspace, int32_t* gave_up);
int kx_gemv_w8_chain_f32(const kx_gemv_w8_desc* ops, int32_t nops, void* workspace,
                         size_t workspace_bytes, int32_t chain_id, const int32_t* pos_dev, int32_t pos,
                         void* stream);

/* ---------------- fp32 GEMV ----------------
 * Replaces matmul_kernel_cu / matmul_kernel_cu_fp32 (cuda/matmul_kernel.cu:6-54,89-109) and
 * matmul_kernel_cpu (cpu/matmul_kernel.cpp:5-41) behind MatmulKernel (kernels_interface.h:9-10):
 * y = (W[K][M] . x) * scale.  The CUDA kernel ignores `scale`; the CPU kernel applies it and
 * mha_kernel relies on that (cpu/mha_kernel.cpp:38), so it is applied here. */
int kx_gemv_f32(const float* x, const float* w, float* y, int32_t M, int32_t K, float scale,
                void* stream);

/* ---------------- prefill GEMMs (new capability behind the same registry) ----------------
 * Y[T][K] = X[T][M] . W^T with the W8 formula above per row.  MFMA path: activations are
 * rounded to fp16, int8 weights are exact in fp16, per-group partial sums are fp32 and scaled
 * in fp32.  T == 1 forwards to the GEMV. */
int kx_gemm_w8_f32(const float* X, const int8_t* w, const float* scales, int32_t group_size,
                   float* Y, int32_t T, int32_t M, int32_t K, void* stream);
/* General form: X rows at stride ldx; output columns [0,k0) -> y0 (row stride ld0), [k0,k1) -> y1,
 * [k1,K) -> y2 (fused Wq|Wk|Wv with k/v rows landing in consecutive KV-cache rows); `residual`
 * (nullable, row stride ldr) is added to the y0 columns and may alias y0. */
int kx_gemm_w8_f32_ex(const float* X, int64_t ldx, const int8_t* w, const float* scales,
                      int32_t group_size, float* y0, int64_t ld0, float* y1, int64_t ld1, float* y2,
                      int64_t ld2, int32_t k0, int32_t k1, const float* residual, int64_t ldr,
                      int32_t T, int32_t M, int32_t K, void* stream);
/* Same contract for long prompts, with a caller-provided workspace of
 * kx_gemm_w8_workspace_bytes(T, M, K) bytes (256-byte aligned): X and the scaled weights are
 * converted to fp16 once per call and the main loop is a plain fp16 MFMA GEMM.  Falls back to
 * kx_gemm_w8_f32_ex when the workspace is missing / too small or T < 512 (below that the
 * conversion passes cost more than they save). */
size_t kx_gemm_w8_workspace_bytes(int32_t T, int32_t M, int32_t K);
int kx_gemm_w8_f32_ws(const float* X, int64_t ldx, const int8_t* w, const float* scales,
                      int32_t group_size, float* y0, int64_t ld0, float* y1, int64_t ld1, float* y2,
                      int64_t ld2, int32_t k0, int32_t k1, const float* residual, int64_t ldr,
                      int32_t T, int32_t M, int32_t K, void* workspace, size_t workspace_bytes,
                      void* stream);
/* The long-prompt pipeline keeps activations in fp16 between its kernels (new capability; the
 * reference has no prompt batching).  X16 is fp16 [T][M], dense.  mode KX_GEMM_ROUTE_F32: fp32
 * outputs with the routing / residual contract of kx_gemm_w8_f32_ex.  mode KX_GEMM_SWIGLU_F16: w
 * is [gate rows | up rows] (K/2 each, K % 64 == 0) and h16 receives silu(gate)*up as fp16
 * [T][K/2] (swiglu_kernel_cu, cuda/swiglu_kernel.cu:4-22, fused into the epilogue).  The
 * workspace (kx_gemm_w8_f16act_workspace_bytes, 256-byte aligned) holds the fp16 weights. */
#define KX_GEMM_ROUTE_F32 0
#define KX_GEMM_SWIGLU_F16 1
/* OR-ed into `mode`: the workspace still holds the fp16 weights that an earlier call with the same
 * w / scales / mode / M / K left there, so the dequantisation pass is skipped.  A caller with the
 * memory for it keeps one workspace per matrix and pays the pass once instead of once per call. */
#define KX_GEMM_WEIGHTS_READY 0x100
size_t kx_gemm_w8_f16act_workspace_bytes(int32_t M, int32_t K);
int kx_gemm_w8_f16act_ws(const void* X16, const int8_t* w, const float* scales, int32_t group_size,
                         int32_t mode, void* h16, float* y0, int64_t ld0, float* y1, int64_t ld1,
                         float* y2, int64_t ld2, int32_t k0, int32_t k1, const float* residual,
                         int64_t ldr, int32_t T, int32_t M, int32_t K, void* workspace,
                         size_t workspace_bytes, void* stream);
/* The same with a second workspace (kx_gemm_split_workspace_bytes(); no reference counterpart)
 * through which the 256 x 256-tile kernel balances a last, partial round of tiles: the reduction
 * of each of those tiles is cut into 2 or 4 parts that run on otherwise idle CUs, the parts meet in
 * fp32 slabs and the workgroup that finishes last adds them in index order (results do not depend
 * on arrival order).  The first 4096 bytes (arrival tickets) must be ZERO before the first use and
 * are left zero by every call; calls that share it must not overlap.  NULL: never cut. */
size_t kx_gemm_split_workspace_bytes(void);
/* The Wq|Wk|Wv product of a prompt chunk with what follows it folded into the GEMM's epilogue
 * (attention_qkv + the RoPE layer, llama3.cpp:654-677; the reference runs them as separate
 * kernels, one row at a time): bias (K = q_dim + 2 kv_dim values, or NULL) added ahead of the
 * rotation as MatmulLayer does (matmul.cpp:76-79), q columns to q[t][ld_q], k and v columns to
 * the given rows (the KV-cache rows of positions pos0 .. pos0 + T - 1), and q and k rotated for
 * position pos0 + t with the sin/cos tables of kx_sin_cos_cache_f32.  head_size 64 or 128,
 * 16-byte aligned rows; KX_EUNSUPPORTED otherwise (run kx_gemm_w8_f16act_ws + kx_rope_bias_f32). */
int kx_gemm_w8_f16act_qkv_rope(const void* X16, const int8_t* w, const float* scales,
                               int32_t group_size, int32_t weights_ready, float* q, int64_t ld_q,
                               float* k, int64_t ld_k, float* v, int64_t ld_v, int32_t q_dim,
                               int32_t kv_dim, int32_t head_size, int32_t rope_variant,
                               const float* sin_cache, const float* cos_cache, int32_t pos0,
                               const float* bias, int32_t T, int32_t M, void* workspace,
                               size_t workspace_bytes, void* split_workspace,
                               size_t split_workspace_bytes, void* stream);
int kx_gemm_w8_f16act_split_ws(const void* X16, const int8_t* w, const float* scales,
                               int32_t group_size, int32_t mode, void* h16, float* y0, int64_t ld0,
                               float* y1, int64_t ld1, float* y2, int64_t ld2, int32_t k0, int32_t k1,
                               const float* residual, int64_t ldr, int32_t T, int32_t M, int32_t K,
                               void* workspace, size_t workspace_bytes, void* split_workspace,
                               size_t split_workspace_bytes, void* stream);
/* The same product for FEW rows (2 <= T <= 192; a short prompt, the reference feeds such rows one
 * launch each, demo/main.cpp:18-41): the int8 weights are streamed once, as by the decode GEMV,
 * and meet all T activation rows on the matrix cores.  Arithmetic as kx_gemm_w8_f32 (activations
 * rounded to fp16, one fp32 partial per 64-group times its fp32 scale).  Routing, residual and
 * leading dimensions as kx_gemm_w8_f32_ex; X is fp32 rows (_f32) or fp16 rows (_f16act, ldx in
 * halves).  split_workspace as for kx_gemm_w8_f16act_split_ws (tickets zero before first use,
 * left zero; NULL: reductions are never cut over workgroups, slow on narrow matrices).
 * KX_EUNSUPPORTED for T > 192, group_size != 64, M % 256 != 0 or unaligned operands: run
 * kx_gemm_w8_f32_ex instead. */
int kx_gemm_w8_rows_f32(const float* X, int64_t ldx, const int8_t* w, const float* scales,
                        int32_t group_size, float* y0, int64_t ld0, float* y1, int64_t ld1,
                        float* y2, int64_t ld2, int32_t k0, int32_t k1, const float* residual,
                        int64_t ldr, int32_t T, int32_t M, int32_t K, void* split_workspace,
                        size_t split_workspace_bytes, void* stream);
int kx_gemm_w8_rows_f16act(const void* X16, int64_t ldx, const int8_t* w, const float* scales,
                           int32_t group_size, float* y0, int64_t ld0, float* y1, int64_t ld1,
                           float* y2, int64_t ld2, int32_t k0, int32_t k1, const float* residual,
                           int64_t ldr, int32_t T, int32_t M, int32_t K, void* split_workspace,
                           size_t split_workspace_bytes, void* stream);
/* The few-rows product with what follows it in a layer done on the finished totals, in the same
 * launch (a weight row and its partner -- up row, or the other row of a rotation pair -- are
 * accumulated by the same lane):
 * _swiglu_: w13 = [W1; W3] stacked (2H x M); out16[t][h] = fp16(silu(gate) * up), saturating
 *   (feed_forward's two matmuls + the SwiGLU layer, llama3.cpp:761-776, cuda/swiglu_kernel.cu:4-22).
 *   H % 64 == 0.
 * _qkv_rope_: as kx_gemm_w8_f16act_qkv_rope above (attention_qkv + the RoPE layer,
 *   llama3.cpp:654-677; bias ahead of the rotation, matmul.cpp:76-79): q columns to q[t][ld_q], k
 *   and v columns to the given rows, q and k rotated for position pos0 + t.  head_size 32, 64 or
 *   128, q_dim and kv_dim multiples of 128.
 * Otherwise as kx_gemm_w8_rows_f16act (T <= 192, group size 64, M % 256 == 0);
 * KX_EUNSUPPORTED for other shapes: run the separate launches.
 * Preconditions the entries cannot check: the sin/cos tables hold at least pos0 + T rows of
 * head_size floats, and every leading dimension is at least the width of the rows it strides
 * (smaller ones are refused with KX_EINVAL). */
int kx_gemm_w8_rows_swiglu_f16act(const void* X16, int64_t ldx, const int8_t* w13,
                                  const float* scales13, int32_t group_size, void* out16,
                                  int64_t ld_out, int32_t T, int32_t M, int32_t H,
                                  void* split_workspace, size_t split_workspace_bytes, void* stream);
int kx_gemm_w8_rows_qkv_rope_f16act(const void* X16, int64_t ldx, const int8_t* w, const float* scales,
                                    int32_t group_size, float* q, int64_t ld_q, float* k, int64_t ld_k,
                                    float* v, int64_t ld_v, int32_t q_dim, int32_t kv_dim,
                                    int32_t head_size, int32_t rope_variant, const float* sin_cache,
                                    const float* cos_cache, int32_t pos0, const float* bias, int32_t T,
                                    int32_t M, void* split_workspace, size_t split_workspace_bytes,
                                    void* stream);
int kx_gemm_f32(const float* X, const float* w, float* Y, int32_t T, int32_t M, int32_t K,
                void* stream);

/* ---------------- attention ----------------
 * kx_mha_decode_f32 replaces mha_kernel_cu / multi_head_attention_kernel
 * (cuda/mha_kernel.cu:50-127) and mha_kernel (cpu/mha_kernel.cpp:5-61) behind MHAKernel
 * (kernels_interface.h:22-28): for each head h, softmax over t<=pos of
 * (q_h . K[layer][t][(h/kv_mul)*hs..]) / sqrt(hs), then the weighted sum of V rows.
 * `workspace` holds, in its first 16 KiB, one arrival counter per kv head (256 bytes apart), then the
 * split-KV partials; kx_attn_workspace_bytes() sizes it.  ZERO IT ONCE after allocating it: every
 * launch returns the counters to zero, so no per-call clearing is needed.  Launches that share a
 * workspace must not overlap in time (one stream, or one workspace per stream).  Head sizes other
 * than 32/64/128 fall back to a kernel that needs 16384 + head_num*seq_len*4 bytes instead.  The
 * reference's score tensor [head_num][seq_len] plays this role there (llama3.cpp:523-525); the
 * probabilities the reference leaves in it are not reproduced (nothing reads them,
 * llama3.cpp:724-733). */
size_t kx_attn_workspace_bytes(int32_t head_num, int32_t head_size);
/* Arrival tickets of a split workspace back to zero (its state after allocation + memset): the way
 * to recover a workspace after a launch that used it was aborted.  No reference counterpart. */
int kx_attn_workspace_reset(void* workspace, size_t workspace_bytes, void* stream);
int kx_mha_decode_f32(const float* query, const float* key_cache, const float* value_cache,
                      float* out, int32_t pos, int32_t head_num, int32_t kv_mul, int32_t head_size,
                      int32_t kv_dim, int32_t seq_len, int32_t layer_index, void* workspace,
                      size_t workspace_bytes, void* stream);

/* The same attention for a caller that has no zeroed workspace, only the reference's score tensor
 * ([head_num][seq_len] fp32, llama3.cpp:523-525, passed to every MHAKernel call,
 * kernels_interface.h:22-28): that tensor is allocated un-zeroed and keeps the last call's bytes.
 * Nothing is assumed about its contents: when it is large enough to serve as split workspace its
 * ticket words are cleared on `stream` ahead of the launch, otherwise each kv head runs as one
 * split; head sizes other than 32/64/128 use it as the [head_num][seq_len] score rows it is.
 * score may be NULL (one split per kv head; tiled head sizes only). */
int kx_mha_decode_scores_f32(const float* query, const float* key_cache, const float* value_cache,
                             float* out, int32_t pos, int32_t head_num, int32_t kv_mul,
                             int32_t head_size, int32_t kv_dim, int32_t seq_len, int32_t layer_index,
                             float* score, size_t score_bytes, void* stream);

/* kx_mha_decode_f32 for callers that cannot allocate device memory or hand over a workspace (code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart
 * in the reference.  The split workspace is a fixed array inside the library, one copy per GPU,
 * of kx_mha_decode_static_limit_bytes() bytes (16 KiB + 128 x 64 x 132 floats = 4.14 MiB): calls
 * whose kx_attn_workspace_bytes(head_num, head_size) -- or, for head sizes outside 32/64/128,
 * 16 KiB + head_num*seq_len*4 -- exceed it return KX_ELIMIT.  The call clears the array's ticket
 * words on `stream`, launches the same kernels as kx_mha_decode_f32 and gives the same result; it
 * calls no allocator.  All calls on one GPU share the array: calls that overlap in time on one
 * GPU (two streams) race. */
size_t kx_mha_decode_static_limit_bytes(void);
int kx_mha_decode_static_f32(const float* query, const float* key_cache, const float* value_cache,
                             float* out, int32_t pos, int32_t head_num, int32_t kv_mul,
                             int32_t head_size, int32_t kv_dim, int32_t seq_len, int32_t layer_index,
                             void* stream);

/* Same kernel with the position read from device memory (replayable under graph capture);
 * `max_pos` sizes the grid for the longest context this launch will ever see.  Needs a head
 * size the tiled kernel is built for (32/64/128) and kv_mul <= 16. */
int kx_mha_decode_dpos_f32(const float* query, const float* key_cache, const float* value_cache,
                           float* out, const int32_t* pos_dev, int32_t max_pos, int32_t head_num,
                           int32_t kv_mul, int32_t head_size, int32_t kv_dim, int32_t seq_len,
                           int32_t layer_index, void* workspace, size_t workspace_bytes,
                           void* stream);
/* kx_mha_decode_dpos_f32 with two hints from a caller that knows more than one launch.  No
 * counterpart in the reference.
 * rows_to_max_pos_readable (0 / 1): with 1 the caller vouches that rows 0..max_pos of the layer
 * may be READ whatever the position turns out to be -- they are allocated and hold finite numbers
 * (rows above the position get zero weight; a zero-filled or previously used cache qualifies).
 * The launch is then cut into splits and tiles for max_pos itself and the position read from
 * pos_dev only masks, so no cache address waits for it: one dependent memory round trip less
 * between the launch and the stream.  Every row up to max_pos is read: keep max_pos close above
 * the positions the launch will really see (model::LlamaModel captures its step for 256
 * positions at a time).  The position must not exceed max_pos, as for kx_mha_decode_dpos_f32.
 * next0 / next1: warm the caches for the launch behind this one.  No counterpart in
 * the reference (its launches start cold: mha_layer->forward, then wo_layer->forward,
 * llama3.cpp:733-743).  While the last workgroup per kv head merges the splits, HBM is idle and every other
 * workgroup is done: on their way out those read one dword of every 128-byte line of up to two
 * byte ranges -- `next0` / `next1`, cut into granules; granule c is what workgroup c of the NEXT
 * launch reads first (for kx_gemv_w8_f32_ex: kx_gemv_w8_block_rows() rows of the int8 matrix, and
 * of its scales), and it is read by a workgroup on XCD c mod 8, where workgroup c will run.  Only
 * the first part of every granule is read -- 7 MiB of `next0` in all, 40 % of it at most, and the
 * same share of `next1`: what arrives while the merge runs (more would make this launch end
 * later than the next gains).  The
 * result is that of kx_mha_decode_dpos_f32 bit for bit; ranges that are null, empty, or not whole
 * 128-byte lines from a 128-byte aligned start are ignored, and so is everything when the
 * context is short enough for one split per kv head. */
int kx_mha_decode_dpos_ex_f32(const float* query, const float* key_cache, const float* value_cache,
                                 float* out, const int32_t* pos_dev, int32_t max_pos,
                                 int32_t rows_to_max_pos_readable, int32_t head_num, int32_t kv_mul, int32_t head_size, int32_t kv_dim,
                                 int32_t seq_len, int32_t layer_index, void* workspace,
                                 size_t workspace_bytes, const void* next0, size_t next0_bytes,
                                 size_t next0_granule, const void* next1, size_t next1_bytes,
                                 size_t next1_granule, void* stream);
/* The same for a 16-bit cache (kx_mha_decode_kv16 with the two hints; `pos` is max_pos when pos_dev is given). */
int kx_mha_decode_kv16_ex(const float* query, const void* key_cache16, const void* value_cache16,
                             float* out, int32_t pos, const int32_t* pos_dev, int32_t rows_to_pos_readable, int32_t head_num,
                             int32_t kv_mul, int32_t head_size, int32_t kv_dim, int32_t seq_len,
                             int32_t layer_index, void* workspace, size_t workspace_bytes,
                             const void* next0, size_t next0_bytes, size_t next0_granule,
                             const void* next1, size_t next1_bytes, size_t next1_granule, void* stream);
/* Rows of a K x M int8 matrix that one workgroup of kx_gemv_w8_f32 / _ex (plain form) owns:
 * workgroup b reads rows [b * n, (b + 1) * n).  0 when the shape takes the generic kernel. */
int32_t kx_gemv_w8_block_rows(int32_t M, int32_t K, int32_t group_size);
/* 16-bit KV cache (not in the reference, whose cache is fp32: kv_cache_manager.cpp:6-40): the same
 * single-token attention over caches of IEEE half rows, same [layer][seq_len][kv_dim] layout.
 * K and V are widened to fp32 on load, so the only deviation from the fp32 cache is the one
 * rounding of each stored K/V element (2^-11 relative).  pos_dev non-null: the position is read
 * on the device and `pos` is the largest position the launch may see.  Head sizes 32/64/128. */
int kx_mha_decode_kv16(const float* query, const void* key_cache16, const void* value_cache16,
                       float* out, int32_t pos, const int32_t* pos_dev, int32_t head_num,
                       int32_t kv_mul, int32_t head_size, int32_t kv_dim, int32_t seq_len,
                       int32_t layer_index, void* workspace, size_t workspace_bytes, void* stream);

/* kx_flash_attention_f32 replaces flash_attention_kernel_cu (cuda/flash_attention_kernel.cu:
 * 214-303) behind FlashAttentionKernel (kernels_interface.h:49-54).  Semantics (the reference
 * kernel ignores pos and the layer/GQA mapping of its CPU twin): Tq query rows
 * [Tq][head_num][head_size]; row i is at absolute position pos-(Tq-1)+i and attends cache rows
 * 0..that position when causal, 0..pos otherwise; kv head = h / (head_num/kv_head_num).
 * Tq == 1 is kx_mha_decode_f32.  Tq > 1 runs the MFMA kernel (fp16 operands, fp32 softmax and
 * accumulation). */
/* kx_flash_attention_f32 with the output rows written as fp16 [Tq][head_num][head_size]
 * (Tq > 1 and head sizes 32/64/128 only). */
int kx_flash_attention_f16out(const float* query, const float* key_cache, const float* value_cache,
                              void* out16, int32_t Tq, int32_t head_num, int32_t kv_head_num,
                              int32_t head_size, int32_t seq_len, int32_t pos, int32_t layer_idx,
                              float softmax_scale, int32_t is_causal, void* workspace,
                              size_t workspace_bytes, void* stream);
/* kx_flash_attention_f32 / _f16out over a 16-bit KV cache (see kx_mha_decode_kv16).  Exactly one
 * of out (fp32 rows) and out16 (fp16 rows) is non-null; Tq > 1 needs the workspace of
 * kx_flash_attention_workspace_bytes() (the kernel works from per-head copies). */
int kx_flash_attention_kv16(const float* query, const void* key_cache16, const void* value_cache16,
                            float* out, void* out16, int32_t Tq, int32_t head_num,
                            int32_t kv_head_num, int32_t head_size, int32_t seq_len, int32_t pos,
                            int32_t layer_idx, float softmax_scale, int32_t is_causal,
                            void* workspace, size_t workspace_bytes, void* stream);
/* Workspace for Tq > 1 that lets the kernel work from fp16 copies of the layer's K/V rows 0..pos
 * (made once per call).  The first 16 KiB are left untouched, so the decode workspace, grown to
 * this size, can serve both.  With a smaller workspace each workgroup converts the fp32 rows it
 * stages (slower, same results up to fp16 rounding order). */
size_t kx_flash_attention_workspace_bytes(int32_t pos, int32_t kv_head_num, int32_t head_size);
int kx_flash_attention_f32(const float* query, const float* key_cache, const float* value_cache,
                           float* out, int32_t Tq, int32_t head_num, int32_t kv_head_num,
                           int32_t head_size, int32_t seq_len, int32_t pos, int32_t layer_idx,
                           float softmax_scale, int32_t is_causal, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---------------- support kernels (correct, not the graded path) ----------------
 * rmsnorm: rmsnorm_kernel_cu (cuda/rmsnorm_kernel.cu:62-108); eps is a runtime argument where
 * the reference picks it at compile time (cpu/rmsnorm_kernel.cpp:24-28).  `rows` independent
 * rows of `dim`. */
int kx_rmsnorm_f32(const float* in, const float* weight, float* out, int32_t dim, int32_t rows,
                   float eps, void* stream);
/* Rows of fp32 -> rows of fp16: dst16[r][c] = half(src[r][c]), c < cols (cols % 4 == 0; leading
 * dimensions in elements).  Used to file prompt K/V rows into a 16-bit KV cache. */
int kx_f32_to_f16_rows(const float* src, int64_t ld_src, void* dst16, int64_t ld_dst, int32_t cols,
                       int32_t rows, void* stream);
/* The same normalisation with fp16 output rows (input of kx_gemm_w8_f16act_ws). */
int kx_rmsnorm_f16out(const float* in, const float* weight, void* out16, int32_t dim, int32_t rows,
                      float eps, void* stream);
/* sin_cos_calc (cuda/rope_kernel.cu:38-49,84-95,130-136) with theta at run time. */
int kx_sin_cos_cache_f32(int32_t head_size, int32_t max_seq_len, float theta, float* sin_cache,
                         float* cos_cache, void* stream);
/* rope_kernel_cu (cuda/rope_kernel.cu:4-36,51-82,97-128): `variant` chooses the pairing the
 * reference fixes at compile time.  Rotates `rows` consecutive tokens: token r sits at position
 * pos+r, q row stride dim, k row stride k_stride (floats). */
int kx_rope_f32(int32_t variant, int32_t dim, int32_t kv_dim, int32_t head_size, float* q, float* k,
                int32_t k_stride, int32_t pos, int32_t rows, const float* sin_cache,
                const float* cos_cache, void* stream);
/* kx_rope_f32 with the q/k/v biases of a Qwen2-style attention block added first (the bias add
 * of MatmulLayer::forward, matmul.cpp:76-79, sits between the matmul and the RoPE layer:
 * qwen2.cpp attention_qkv).  bq [dim], bk / bv [kv_dim], each nullable; v rows (stride v_stride)
 * are touched only to receive bv. */
int kx_rope_bias_f32(int32_t variant, int32_t dim, int32_t kv_dim, int32_t head_size, float* q,
                     float* k, int32_t k_stride, float* v, int32_t v_stride, const float* bq,
                     const float* bk, const float* bv, int32_t pos, int32_t rows,
                     const float* sin_cache, const float* cos_cache, void* stream);
/* swiglu_kernel_cu (cuda/swiglu_kernel.cu:4-22): out = in1*sigmoid(in1)*in2. */
int kx_swiglu_f32(const float* in1, const float* in2, float* out, int32_t n, void* stream);
/* Row-batched form over a stacked [gate | up] buffer: out[t][h] = silu(in[t][h]) * in[t][half+h]. */
int kx_swiglu_rows_f32(const float* in, int64_t ld_in, int32_t half, float* out, int64_t ld_out,
                       int32_t rows, void* stream);
/* The same with fp16 result rows (ld_out in halves, saturating conversion): the form a GEMM
 * with fp16 activations takes next. */
int kx_swiglu_rows_f16out(const float* in, int64_t ld_in, int32_t half, void* out16, int64_t ld_out,
                          int32_t rows, void* stream);
/* add_kernel_cu (cuda/add_kernel.cu:4-12). */
int kx_add_f32(const float* a, const float* b, float* out, int32_t n, void* stream);
/* emb_kernel_cu (cuda/emb_kernel.cu:3-21): tokens on device. */
int kx_embedding_f32(const int32_t* tokens, int32_t n_tokens, const float* table, int32_t dim,
                     int32_t vocab_size, float* out, void* stream);
/* One-token gather with the token taken as history_dev[*pos_dev] (replayable under capture). */
int kx_embedding_dpos_f32(const int32_t* history_dev, const int32_t* pos_dev, const float* table,
                          int32_t dim, int32_t vocab_size, float* out, void* stream);
/* argmax_kernel_cu (cuda/argmax_kernel.cu:49-87): first maximum wins; result written to a
 * caller-provided device int32 (the reference allocates per call). */
int kx_argmax_f32(const float* logits, int32_t n, int32_t* out_index, void* stream);

/* The three pieces the reference's CPU attention is assembled from (cpu/mha_kernel.cpp:19-60),
 * behind ScaleKernel / SoftmaxInplaceKernel / ScaleSumKernel (kernels_interface.h:41-47,72-78; the
 * reference registers CPU versions only).  Not on the hot path -- kx_mha_decode_f32 does all of it
 * in one launch -- but a source written against those getters links and runs.
 *   scale:     x[i] *= scale                                  (cpu/scale_kernel.cpp:3-8)
 *   softmax:   x = exp(x - max x) / sum, in place, one vector  (cpu/softmax_kernel.cpp:4-15)
 *   scale_sum: out[d] += sum_{i<=pos} scale[i] * value[i*stride + d], d < size
 *                                                             (cpu/scale_sum_kernel.cpp:5-22) */
int kx_scale_f32(float scale, float* x, int32_t n, void* stream);
int kx_softmax_f32(float* x, int32_t n, void* stream);
int kx_scale_sum_f32(const float* value, const float* scale, float* out, int32_t pos, int32_t size,
                     int32_t stride, void* stream);
/* quantize_q80 of the reference's exporter (tools/export_llama.py:49-73) on the device: for every
 * group of group_size consecutive values scale = max|w| / 127 and q = round-half-even(w / scale),
 * byte for byte what the exporter writes (an all-zero group: scale 0, zeros).  Used at load time to
 * build the int8 classifier of a quant file whose classifier is tied to the fp32 embedding table --
 * the exporter writes no int8 copy then (export_llama.py:183-192) and the reference points its
 * int8 layer at the fp32 bytes (kuiper/source/model/llama3.cpp:273-286).  n % group_size == 0. */
int kx_quantize_q80_f32(const float* w, size_t n, int32_t group_size, int8_t* q, float* scales,
                        void* stream);

/* End of one decode step, on the device (post_processing, llama3.cpp:801-813, without the
 * host round trip): token = argmax(logits); history[pos+1] = token (when history != NULL and in
 * range); pos += 1.  One launch: every workgroup reduces a slice to a {value, index} pair, and
 * the workgroup that arrives last reduces the pairs and does the bookkeeping.  `workspace` of
 * kx_step_workspace_bytes() carries the pairs and the arrival counter: it must be ZERO before its
 * first use and is left ready for the next call by every call (calls that share a workspace
 * must not overlap). */
size_t kx_step_workspace_bytes(void);
int kx_decode_step_end(const float* logits, int32_t n, int32_t* token_dev, int32_t* pos_dev,
                       int32_t* history_dev, int32_t history_len, void* workspace,
                       size_t workspace_bytes, void* stream);
/* The same, and the embedding of the NEXT step in the same launch (the reference gathers it at the
 * start of the next forward, llama3.cpp:622-640): with write_history != 0 the picked token is
 * stored at history[pos+1]; otherwise (a prompt row) history keeps the token the caller put
 * there.  Either way x_out = table[history[pos+1]] (dim floats) when `table` is given and pos+1
 * is inside the history. */
int kx_decode_step_end_embed(const float* logits, int32_t n, int32_t* token_dev, int32_t* pos_dev,
                             int32_t* history_dev, int32_t history_len, int32_t write_history,
                             const float* table, int32_t dim, int32_t vocab_size, float* x_out,
                             void* workspace, size_t workspace_bytes, void* stream);
/* The classifier GEMV and kx_decode_step_end_embed as ONE launch (the reference runs cls_layer
 * forward and the argmax sampler as separate steps, llama3.cpp:155-175 / argmax_sampler.cpp:6-8):
 * `d` describes the product (mode KX_GEMV_PLAIN, no row routing, no residual; norm_w allowed) and
 * its logits are written to d->y0 as by kx_gemv_w8_fused_f32; every workgroup keeps the best of the
 * rows it finishes and the last one to arrive picks, does the bookkeeping and leaves the next
 * embedding row -- x_out may be d->x.  Same results as the two calls, bit for bit.  The workspace
 * is the step workspace (zero before first use).  KX_EUNSUPPORTED for shapes the fast GEMV form does
 * not take or with more workgroups than pair slots: run the two calls then. */
int kx_gemv_w8_classify_pick_f32(const kx_gemv_w8_desc* d, int32_t* token_dev, int32_t* pos_dev,
                                 int32_t* history_dev, int32_t history_len, int32_t write_history,
                                 const float* table, int32_t dim, int32_t vocab_size, float* x_out,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* The same step end for a classifier whose rows are split over tensor-parallel ranks (no
 * reference counterpart: the reference never shards, nccl_handler.cpp:60-80).  Each rank turns its
 * slice of the logits (`n` values, global index of the first = index_base) into one candidate
 * {value, global index as fp32}; the caller all-gathers the candidates ([ranks][2] floats) and
 * kx_decode_step_end_candidates picks the winner (largest value, lowest index on ties -- the first
 * maximum of the unsharded logits) and does the bookkeeping of kx_decode_step_end (the _embed
 * form: that of kx_decode_step_end_embed).  Indices must stay below 2^24.  `workspace`:
 * kx_step_workspace_bytes(), zero before its first use. */
int kx_argmax_candidate_f32(const float* logits, int32_t n, int32_t index_base, float* candidate,
                            void* workspace, size_t workspace_bytes, void* stream);
int kx_decode_step_end_candidates(const float* candidates, int32_t ranks, int32_t* token_dev,
                                  int32_t* pos_dev, int32_t* history_dev, int32_t history_len,
                                  void* stream);
int kx_decode_step_end_candidates_embed(const float* candidates, int32_t ranks, int32_t* token_dev,
                                        int32_t* pos_dev, int32_t* history_dev, int32_t history_len,
                                        int32_t write_history, const float* table, int32_t dim,
                                        int32_t vocab_size, float* x_out, void* stream);

/* Seeded synthetic tensors generated on the device (the benchmarks run on random-init weights;
 * the reference reads real checkpoints through mmap, model.cpp:93-115).  dst is a dense
 * [rows][cols] slice whose element (i, j) is element (row0+i, col0+j) of a global matrix with
 * leading dimension ld; the value depends only on (seed, global index), so tensor-parallel shards
 * are slices of one and the same tensor.  int8: uniform in [-127, 127]; f32: uniform in [lo, hi). */
/* Reads `bytes` (rounded down to 16) at `ptr` with `workgroups` workgroups of 256 threads and
 * keeps nothing: a prefetch of weights into the die-level cache, for a side stream that runs
 * beside a latency-bound launch.  `sink` is one int the kernel never writes in practice (it keeps
 * the loads alive).  No counterpart in the reference. */
int kx_touch_bytes(const void* ptr, size_t bytes, int32_t workgroups, int32_t* sink, void* stream);
int kx_fill_hash_i8(int8_t* dst, int32_t rows, int32_t cols, int64_t ld, int64_t row0, int64_t col0,
                    uint64_t seed, void* stream);
int kx_fill_hash_f32(float* dst, int32_t rows, int32_t cols, int64_t ld, int64_t row0,
                     int64_t col0, uint64_t seed, float lo, float hi, void* stream);

/* ---------------- tensor-parallel communicator ----------------
 * Replaces NcclHandler (kuiper/include/base/nccl_handler.h:29-58, source/base/nccl_handler.cpp:
 * 24-117).  One communicator per process/GPU over RCCL; the 128-byte unique id is produced on
 * rank 0 and handed to the other ranks by the caller (the step the reference leaves
 * unimplemented at nccl_handler.cpp:60-71). */
int kx_comm_unique_id(void* id128);
int kx_comm_init_rank(void** comm, const void* id128, int32_t world_size, int32_t rank);
int kx_comm_destroy(void* comm);
/* ncclAllReduce(sum, fp32, in place allowed) -- NcclHandler::AllReduce (nccl_handler.cpp:84-117) */
int kx_comm_allreduce_sum_f32(void* comm, const float* sendbuf, float* recvbuf, size_t count,
                              void* stream);
int kx_comm_broadcast_f32(void* comm, const float* sendbuf, float* recvbuf, size_t count,
                          int32_t root, void* stream);
int kx_comm_allgather_f32(void* comm, const float* sendbuf, float* recvbuf, size_t send_count,
                          void* stream);

/* ---------------- peer mailboxes: one-launch collectives for the token loop ----------------
 * The reference's only collective is NcclHandler::AllReduce (kuiper/source/base/nccl_handler.cpp:
 * 84-117), called behind every row-parallel matmul (kuiper/source/op/matmul.cpp:93-116).  These
 * entries carry the same sum for vectors of up to max_floats (a token's `dim` floats) in ONE kernel
 * on the caller's stream: every rank owns a mailbox in fine-grained device memory that the other
 * ranks of the node map through a 64-byte IPC handle; the kernel writes its vector into every
 * mailbox, publishes a sequence number behind it, waits for the other ranks' numbers and adds the
 * slots in rank order (bitwise the same sum on every rank).  Graph-capturable.  Set-up:
 * kx_peer_create on every rank, exchange the kx_peer_handle bytes, kx_peer_connect with all of
 * them ordered by rank, then a host-side barrier before the first collective.  A wait that gives
 * up after 4 s finishes the launch and raises the flag kx_peer_timed_out reads; that rank then stops
 * posting and marks itself dead in every mailbox, and every rank that waits on it raises its own flag
 * within one call.  Results of a group are to be trusted only while NO rank's flag is up: a caller
 * that checks one rank's flag checks them all (bench.py max-reduces them). */
int kx_peer_create(void** group, int32_t world_size, int32_t rank, size_t max_floats);
int kx_peer_handle(void* group, void* handle64);
int kx_peer_connect(void* group, const void* handles);
/* in place; count <= max_floats (KX_ELIMIT beyond), buf 16-byte aligned */
int kx_peer_allreduce_sum_f32(void* group, float* buf, size_t count, void* stream);
/* buf holds world_size x count_per_rank floats, this rank's part already in place (any 4-byte
 * alignment: parts that do not start on 16 bytes are moved word by word) */
int kx_peer_allgather_f32(void* group, float* buf, size_t count_per_rank, void* stream);
int kx_peer_timed_out(void* group, int32_t* flag);
int kx_peer_destroy(void* group);

#ifdef __cplusplus
}
#endif
#endif /* KX_HIP_H_ */
