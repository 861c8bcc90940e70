// TESTS ONLY: what the tests ask the fake HIP runtime (hip/hip_runtime.h beside this file).  Plain C so that ctypes can call it.
#pragma once
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* One entry per operation the shim put on a stream, in issue order (global sequence number = index). */
typedef struct {
  uint64_t seq;
  void *stream;        /* the hipStream_t */
  int32_t device;
  int32_t frames;      /* kernels: frames / pictures of the launch; 0 otherwise */
  char op[48];         /* "kernel:decode_nv12_quads<nt>", "memcpy", "memcpy2d", "memset", "event_record", "wait_event", "graph_launch" */
  const void *first_in, *first_out; /* kernels: first frame's input / output pointer; copies: src / dst */
} fake_hip_op;
void fake_hip_reset(void);                 /* forget the log, failure injections and rates; devices keep their allocations */
void fake_hip_set_device_count(int n);     /* default 2 */
void fake_hip_set_device_memory(uint64_t total_bytes); /* per device; default 288e9 */
uint64_t fake_hip_log_size(void);          /* operations issued so far (the log keeps the first 1 << 20) */
int fake_hip_log_get(uint64_t index, fake_hip_op *out);
uint64_t fake_hip_allocated(int device);   /* device bytes alive */
uint64_t fake_hip_allocations(int device); /* device allocations alive */
uint64_t fake_hip_host_allocations(void);  /* pinned host allocations alive */
uint64_t fake_hip_live_streams(void);
uint64_t fake_hip_live_events(void);
void fake_hip_fail_malloc_at(int64_t nth); /* the nth hipMalloc from now (1 = the next) returns hipErrorOutOfMemory; 0 = none */
void fake_hip_fail_launch_at(int64_t nth); /* the nth kernel launch from now leaves hipErrorInvalidValue as the thread's last error */
/* decode launches that write into [ptr, ptr + bytes) stream at this rate (GB/s) on the fake device clock; default 6000 */
void fake_hip_set_output_rate(const void *ptr, uint64_t bytes, double GBps);
void fake_hip_set_rate_by_allocation_order(const double *GBps, int n); /* the k-th LARGE (>= 64 MiB) hipMalloc from now gets GBps[k % n] */
#ifdef __cplusplus
}
#endif
