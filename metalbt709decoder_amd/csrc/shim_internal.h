// Internals shared by the translation units of the C-ABI shim (shim_*.cpp): the objects behind the opaque handles and the
// helpers more than one of the files needs.  Not part of the boundary (include/bt709hip.h, include/bt709hip_ext.h).
#pragma once
#include "../../include/bt709hip_ext.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <vector>

#include "bt709_constants.h"
#include "bt709_kernels.h"
#include "transfer_tables.h"

using namespace bt709;

struct EncoderTables {  // device copies for one (input gamma, output gamma) pair
  void *d_per_byte = nullptr;
  void *d_from_linear = nullptr;
  uint32_t from_linear_bytes = 0;
  uint32_t from_linear_n = 0;
  float split = 0.0f, coarse_scale = 1.0f;
  uint32_t coarse_offset = 0;
};

struct bt709hip_context {
  int device = 0;
  hipDeviceProp_t props;
  hipStream_t default_stream = nullptr;
  int grid_mult = 2;    // BT709HIP_CTX_OPT_GRID_MULT
  int grid_blocks = 0;  // workgroups a general-path launch aims for (all frames together)
  int encode_row_pairs = 0, encode_threads = 0;  // BT709HIP_CTX_OPT_ENCODE_*: 0 = sized per launch
  int xcd_bands = 1;                             // BT709HIP_CTX_OPT_XCD_BANDS: XCD-aware work map of batched encoder launches
  int streaming_tries = 4;                       // BT709HIP_CTX_OPT_STREAMING_TRIES: placement candidates for buffers of 256 MB and more that the library allocates itself
  std::mutex encoder_mutex;
  EncoderTables encoders[3][3];  // [input gamma][output gamma], built on first use
  // bt709hip_render_scaled (pass 2 alone): built on first use under encoder_mutex
  void *d_render_encode = nullptr, *d_render_lin = nullptr;
  uint32_t render_encode_bytes = 0, render_encode_log_first = 0;
  float render_encode_log_add = 0.0f;
  // decoders of this context with BT709HIP_OPT_COALESCE on: every entry point that takes a stream issues their queued
  // frames for that stream first (flush_stream), so the stream keeps its order
  std::atomic<int> n_coalescing{0};
  std::mutex coalescing_mutex;
  std::vector<bt709hip_decoder *> coalescing;
};

// BT709HIP_OPT_COALESCE: frames validated and queued for one stream, not yet launched (include/bt709hip_ext.h, COALESCING SUBMIT)
struct PendingQueue {
  hipStream_t stream = nullptr;
  bool with_alphas = false;  // the calls passed alpha descriptors
  int64_t oldest_us = 0;     // steady-clock time at which the oldest frame it holds was queued (BT709HIP_OPT_COALESCE_MAX_AGE_US)
  std::vector<bt709hip_frame> frames, alphas;
  std::vector<bt709hip_surface> outs;
};

struct bt709hip_decoder {
  bt709hip_context *ctx = nullptr;
  int gamma = BT709HIP_GAMMA_APPLE;
  int has_alpha = 0;
  // Options: atomics, because a thread may change one while others are inside a decode (tests/native/shim_stress.cpp does, under
  // TSan); a call reads each option once and runs with what it read.
  std::atomic<uint32_t> alpha_fill{0xFF};
  std::atomic<bool> nontemporal{true};  // BT709HIP_OPT_NONTEMPORAL
  std::atomic<int> half_rep{-1};        // BT709HIP_OPT_HALF_KERNEL: persistent conflict-free rescale kernel: -1 = when the launch is large enough, 0 never, 1 always
  std::atomic<int> half_workgroups{0};  // BT709HIP_OPT_HALF_WORKGROUPS: 0 = one per compute unit
  std::atomic<int> half_lds_kb{0};      // BT709HIP_OPT_HALF_LDS_KB: 0 = all 160
  std::atomic<int> xcd_bands{1};        // BT709HIP_OPT_XCD_BANDS: XCD-aware work map of the batched 1:1 kernels (frames a multiple of 8)
  std::atomic<int> coalesce{0};         // BT709HIP_OPT_COALESCE: 0 off, else frames gathered per launch (2..32)
  std::atomic<int> coalesce_max_age_us{0};  // BT709HIP_OPT_COALESCE_MAX_AGE_US: 0 = no age limit
  std::mutex queue_mutex;   // guards queues
  std::vector<PendingQueue> queues;  // one per stream that has (had) queued frames
  std::mutex setup_mutex;
  std::atomic<bool> ready{false};  // release-stored after the tables below are published, acquire-loaded by every decode
  // device copies (transfer_tables.h)
  uint32_t table_n = 0;            // bucket count N of the decoder's gamma
  float unit1_magic = 0.0f;        // index function of d_table_unit (DecodeParams::unit1_*): uniform or log-bucket form
  uint32_t unit1_first = 0, unit1_shift = 0;
  void *d_table_unit = nullptr;    // TransferBucket[N + 1] (decode kernels)
  uint32_t table_unit_bytes = 0;
  void *d_table_linear = nullptr;  // TransferBucketLinear[N + 1] (rescale kernels, decode side)
  uint32_t table_linear_bytes = 0;
  void *d_encode = nullptr;        // the sRGB-encode composite's log-bucket TransferBucket[] (rescale kernels, encode side)
  uint32_t encode_bytes = 0;
  float encode_log_add = 0.0f;
  uint32_t encode_log_first = 0;
  void *d_encode_u = nullptr;      // the same composite as a UniformTable (persistent 2:1 kernel)
  uint32_t encode_u_bytes = 0, encode_u_n = 0;
  // RGBA16F targets: threshold table of the half-float composite (transfer_tables.h HalfTable), built on
  // first use under setup_mutex; half.table_bytes == 0 with half_ready: the gamma has no curve
  bool half_ready = false;  // under setup_mutex
  HalfParams half = {};
};

struct bt709hip_pool {
  struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *h_in = nullptr, *h_out = nullptr;  // pinned; h_in = Y, CbCr (and the alpha plane behind them)
    uint8_t *d_in = nullptr, *d_out = nullptr;
    bool busy = false;       // submitted, not yet waited for
    bool acquired = false;   // handed out, not yet submitted
  };
  bt709hip_decoder *dec = nullptr;
  int width = 0, height = 0;
  size_t in_bytes = 0, out_bytes = 0;
  std::vector<Slot> slots;
  size_t next = 0;
};

namespace bt709shim __attribute__((visibility("hidden"))) {

// this thread's last launched kernel / last failing HIP call (shim_core.cpp owns the thread-locals)
void set_kernel_name(const char *name);  // bt709hip_last_kernel_name
const char *kernel_name();
hipError_t last_hip_error();             // bt709hip_last_hip_error

int hip_fail(hipError_t e);  // records e for bt709hip_last_hip_error, clears the runtime's sticky copy -> BT709HIP_ERR_HIP

#define HIP_TRY(expr)                          \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return hip_fail(_e); \
  } while (0)

int bind(const bt709hip_context *ctx);  // hipSetDevice of the context's device

inline hipStream_t pick(const bt709hip_context *ctx, void *stream) {
  return stream ? static_cast<hipStream_t>(stream) : ctx->default_stream;
}

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

inline int clamp_int(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// gridDim.y / .z limit of a HIP launch: the row-pair dimension of every kernel lives there
constexpr int kMaxGridYZ = 65535;

// shim_core.cpp
int upload_table(const void *src, size_t bytes, void **dst);
bool capturing(hipStream_t s);
int finish_launch(hipStream_t s, int wait_until_completed);

// shim_decode.cpp
int required_transfer(int gamma);
int validate(const bt709hip_decoder *dec, const bt709hip_frame *f, const bt709hip_frame *a, const bt709hip_surface *o, int out_w,
             int out_h, int render_w, int render_h);
int ensure_setup(bt709hip_decoder *dec, void *stream);
int ensure_half_table(bt709hip_decoder *dec, void *stream);
void set_tables(DecodeParams *p, const bt709hip_decoder *dec);
int64_t byte_step(const void *a, const void *b);
bool evenly_spaced(int count, const bt709hip_frame *frames, const bt709hip_frame *alphas, const bt709hip_surface *outs);
uint32_t grid_x_for(const bt709hip_context *ctx, uint32_t rows, int frames);
enum class OutShape { kSame, kHalf, kAny };  // output size relative to the frame: pass 1 / exact 2:1 / view-fit

struct BatchInfo {
  bool uniform = false;    // frames evenly spaced in memory (no pointer table needed)
  uint32_t in_align = 16;  // largest power of two <= 16 dividing every input plane pointer and pitch
  uint32_t out_align = 16; // same for the outputs
  int format = BT709HIP_FORMAT_BGRA8_SRGB;
};
// validation of a batch in the reference's order + the launch parameters it implies
int gather_batch(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                 const bt709hip_surface *outs, OutShape shape, void *stream, DecodeParams *p, BatchInfo *info);
int decode_batch_now(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                     const bt709hip_surface *outs, void *stream, int wait_until_completed);

// shim_convert.cpp
int encoder_tables(bt709hip_context *ctx, int input_gamma, int output_gamma, hipStream_t s);

// shim_coalesce.cpp (BT709HIP_OPT_COALESCE)
int issue_queue(bt709hip_decoder *dec, PendingQueue &q);
int flush_decoder(bt709hip_decoder *dec, hipStream_t s, bool all, bool aged_only = false);
int flush_stream(bt709hip_context *ctx, hipStream_t s, const bt709hip_decoder *skip = nullptr);
int set_coalescing(bt709hip_decoder *dec, int n);
int coalescing_submit(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                      const bt709hip_surface *outs, void *stream, int wait_until_completed);

// every entry point that takes a stream issues what coalescing decoders have queued for it first
#define FLUSH_STREAM(ctx, stream)                                                        \
  do {                                                                                   \
    if ((ctx) != nullptr)                                                                \
      if (int _rc = flush_stream((ctx), pick((ctx), (stream)))) return _rc;              \
  } while (0)

}  // namespace bt709shim

using namespace bt709shim;
