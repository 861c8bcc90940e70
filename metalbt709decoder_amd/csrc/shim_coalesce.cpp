// C-ABI shim, the coalescing submit (include/bt709hip_ext.h, BT709HIP_OPT_COALESCE): a decoder that keeps the reference's
// one -decodeBT709: call per frame validates each call at once but gathers the frames of a stream into one launch.
#include "shim_internal.h"

namespace bt709shim __attribute__((visibility("hidden"))) {

// Launches what `q` holds (dec->queue_mutex held).  The queue is emptied first: a failed launch is reported once, to the
// call that issued it, and never re-issued.
int issue_queue(bt709hip_decoder *dec, PendingQueue &q) {
  if (q.frames.empty()) return BT709HIP_OK;
  std::vector<bt709hip_frame> frames, alphas;
  std::vector<bt709hip_surface> outs;
  frames.swap(q.frames);
  alphas.swap(q.alphas);
  outs.swap(q.outs);
  return decode_batch_now(dec, static_cast<int>(frames.size()), frames.data(), q.with_alphas ? alphas.data() : nullptr, outs.data(),
                          q.stream, 0);
}

namespace {
int64_t now_us() {
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// one decoder: the queue of stream `s`, or every queue (all = true); aged_only: only queues older than the decoder's age limit
int flush_decoder(bt709hip_decoder *dec, hipStream_t s, bool all, bool aged_only) {
  std::lock_guard<std::mutex> lock(dec->queue_mutex);
  int rc = BT709HIP_OK;
  const int64_t limit = aged_only ? now_us() - dec->coalesce_max_age_us : 0;
  for (PendingQueue &q : dec->queues) {
    if (q.frames.empty()) continue;
    const bool aged = dec->coalesce_max_age_us > 0 && q.oldest_us <= limit;
    if (aged_only ? (aged || q.stream == s) : (all || q.stream == s))
      if (int e = issue_queue(dec, q)) rc = rc ? rc : e;
  }
  return rc;
}

// Every coalescing decoder of `ctx` except `skip`: called by each entry point that takes a stream, before it touches the stream
// -- the queue of THAT stream goes out (stream order), and so does any queue of any stream that has outlived its decoder's
// BT709HIP_OPT_COALESCE_MAX_AGE_US (nothing here runs on a timer: an idle caller's frames wait for the context's next call).
// coalescing_mutex is held across the loop (lock order: coalescing_mutex, then a decoder's queue_mutex): a decoder that is being
// destroyed leaves the list under the same mutex (set_coalescing), so none of the pointers can dangle.
int flush_stream(bt709hip_context *ctx, hipStream_t s, const bt709hip_decoder *skip) {
  if (ctx == nullptr || ctx->n_coalescing.load(std::memory_order_acquire) == 0) return BT709HIP_OK;
  // While the CALLING stream records a graph only its own queue goes out: an aged queue of another stream may need a first-use
  // table (an allocation + blocking copies: illegal on a thread that is capturing) and has nothing to do with this recording;
  // it waits for the context's next call outside a capture.
  const bool aged_too = !capturing(s);
  // SERIALISATION: coalescing_mutex is held across the launches the loop issues, so while at least one decoder of the context
  // coalesces, the context's stream-taking entry points run one at a time for the length of those (asynchronous, ~10 us) launch
  // calls.  Contexts without a coalescing decoder never take the mutex (the counter above).
  std::lock_guard<std::mutex> lock(ctx->coalescing_mutex);
  int rc = BT709HIP_OK;
  for (bt709hip_decoder *d : ctx->coalescing)
    if (d != skip)
      if (int e = flush_decoder(d, s, false, aged_too)) rc = rc ? rc : e;
  return rc;
}

// Turns the coalescing submit of `dec` on (n > 1), off (0) or changes its count.  Order matters when other threads are inside a
// decode: ON registers the decoder with its context BEFORE the count becomes visible (a frame queued from then on is seen by
// every flush_stream), OFF issues what is queued and clears the count under the queue_mutex -- a submit that is waiting for that
// mutex re-reads the count behind it and launches instead of queueing -- and only then leaves the context's list.  Lock order:
// the context's coalescing_mutex and a decoder's queue_mutex are never held together here.  Returns the flush's status.
int set_coalescing(bt709hip_decoder *dec, int n) {
  bt709hip_context *ctx = dec->ctx;
  const bool now = n > 1;
  if (now && ctx != nullptr) {
    std::lock_guard<std::mutex> lock(ctx->coalescing_mutex);
    if (std::find(ctx->coalescing.begin(), ctx->coalescing.end(), dec) == ctx->coalescing.end()) ctx->coalescing.push_back(dec);
    ctx->n_coalescing.store(static_cast<int>(ctx->coalescing.size()), std::memory_order_release);
  }
  int rc = BT709HIP_OK;
  {
    std::lock_guard<std::mutex> lock(dec->queue_mutex);
    if (!now)  // off: nothing may stay queued behind the switch
      for (PendingQueue &q : dec->queues)
        if (int e = issue_queue(dec, q)) rc = rc ? rc : e;
    dec->coalesce.store(now ? n : 0);
  }
  if (!now && ctx != nullptr) {
    std::lock_guard<std::mutex> lock(ctx->coalescing_mutex);
    auto it = std::find(ctx->coalescing.begin(), ctx->coalescing.end(), dec);
    if (it != ctx->coalescing.end()) ctx->coalescing.erase(it);
    ctx->n_coalescing.store(static_cast<int>(ctx->coalescing.size()), std::memory_order_release);
  }
  return rc;
}

namespace {
bool same_shape(const bt709hip_frame &a, const bt709hip_frame &b) {
  return a.width == b.width && a.height == b.height && a.y_stride == b.y_stride && a.cbcr_stride == b.cbcr_stride;
}
}  // namespace

// BT709HIP_OPT_COALESCE (include/bt709hip_ext.h, COALESCING SUBMIT): validate now, launch later.
int coalescing_submit(bt709hip_decoder *dec, int count, const bt709hip_frame *frames, const bt709hip_frame *alphas,
                      const bt709hip_surface *outs, void *stream, int wait_until_completed) {
  if (dec->ctx == nullptr) return decode_batch_now(dec, count, frames, alphas, outs, stream, wait_until_completed);
  hipStream_t s = pick(dec->ctx, stream);
  // what the context's OTHER coalescing decoders have queued for this stream was submitted before this call: it goes first.
  // Done before this decoder's own queue_mutex is taken (lock order: the context's coalescing_mutex, then a queue_mutex).
  if (int rc = bind(dec->ctx)) return rc;
  if (int rc = flush_stream(dec->ctx, s, dec)) return rc;
  std::lock_guard<std::mutex> lock(dec->queue_mutex);
  PendingQueue *q = nullptr;
  if (dec->coalesce_max_age_us > 0 && !capturing(s)) {  // this decoder's queues of OTHER streams that have waited too long (never from inside a capture)
    const int64_t limit = now_us() - dec->coalesce_max_age_us;
    for (PendingQueue &c : dec->queues)
      if (c.stream != s && !c.frames.empty() && c.oldest_us <= limit)
        if (int rc = issue_queue(dec, c)) return rc;
  }
  for (PendingQueue &c : dec->queues)
    if (c.stream == s) q = &c;
  const int n = dec->coalesce.load();  // read ONCE, behind the mutex: set_coalescing changes it under the same mutex
  const bool eligible = n > 1 && wait_until_completed == 0 && count >= 1 && count < n && frames != nullptr && outs != nullptr;
  if (!eligible) {  // in stream order: what is queued goes first
    if (q != nullptr)
      if (int rc = issue_queue(dec, *q)) return rc;
    return decode_batch_now(dec, count, frames, alphas, outs, stream, wait_until_completed);
  }
  {  // the call's own status: everything -decodeBT709: checks, now
    DecodeParams p;
    BatchInfo info;
    if (int rc = gather_batch(dec, count, frames, alphas, outs, OutShape::kSame, stream, &p, &info)) return rc;
    if (p.width == 0) return BT709HIP_OK;  // empty frames: nothing to launch
  }
  if (q == nullptr) {
    dec->queues.emplace_back();
    q = &dec->queues.back();
    q->stream = s;
  }
  if (!q->frames.empty()) {
    const bool fits = q->frames.size() + static_cast<size_t>(count) <= static_cast<size_t>(n) &&
                      same_shape(q->frames[0], frames[0]) && q->frames[0].transfer == frames[0].transfer &&
                      q->outs[0].stride == outs[0].stride && q->outs[0].format == outs[0].format &&
                      q->with_alphas == (alphas != nullptr) && (alphas == nullptr || q->alphas[0].y_stride == alphas[0].y_stride);
    if (!fits)
      if (int rc = issue_queue(dec, *q)) return rc;
  }
  q->with_alphas = alphas != nullptr;
  if (q->frames.empty()) q->oldest_us = now_us();
  q->frames.insert(q->frames.end(), frames, frames + count);
  if (alphas != nullptr) q->alphas.insert(q->alphas.end(), alphas, alphas + count);
  q->outs.insert(q->outs.end(), outs, outs + count);
  set_kernel_name("(queued: coalescing submit)");
  if (q->frames.size() >= static_cast<size_t>(n)) return issue_queue(dec, *q);
  return BT709HIP_OK;
}

}  // namespace bt709shim

extern "C" {

int bt709hip_decoder_flush(bt709hip_decoder *dec, void *stream) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (dec->ctx == nullptr) return BT709HIP_OK;  // never had a stream to queue on
  if (int rc = bind(dec->ctx)) return rc;
  return flush_decoder(dec, pick(dec->ctx, stream), false);
}

int bt709hip_decoder_flush_all(bt709hip_decoder *dec) {
  if (dec == nullptr) return BT709HIP_ERR_INVALID_ARG;
  if (dec->ctx == nullptr) return BT709HIP_OK;
  if (int rc = bind(dec->ctx)) return rc;
  return flush_decoder(dec, nullptr, true);
}

}  // extern "C"
