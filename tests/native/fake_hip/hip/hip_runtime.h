// TESTS ONLY -- a fake HIP runtime for the CPU build of the host shim (tests/native/fake_hip/fake_hip.cpp implements it).
//
// GPU sanitizers do not exist on this pool, so the 2 000+ lines of host code that hold mutexes, thread-locals and per-stream
// queues (metalbt709decoder_amd/csrc/shim_*.cpp, bt709_ring.cpp) are compiled with g++ against THIS header instead of ROCm's and
// run under ASan / UBSan / TSan (tools/sanitize.sh, tests/test_fake_hip.py).  Device memory is host memory (large slabs are
// address reservations that are never touched), a stream is a FIFO run by a worker thread, a kernel launch is a log entry plus
// a tick of a fake device clock, events are stamps of that clock.  Only the API subset the shim uses exists.  Nothing under
// tests/native/fake_hip/ is part of the product, is shipped, or is loaded by it.
#pragma once

#include <cstddef>
#include <cstdint>

typedef enum hipError_t {
  hipSuccess = 0,
  hipErrorInvalidValue = 1,
  hipErrorOutOfMemory = 2,
  hipErrorNoDevice = 100,
  hipErrorInvalidDevice = 101,
  hipErrorInvalidResourceHandle = 400,
  hipErrorNotReady = 600,
  hipErrorStreamCaptureUnsupported = 900,
} hipError_t;

struct fakeStream;
struct fakeEvent;
struct fakeGraph;
typedef fakeStream *hipStream_t;
typedef fakeEvent *hipEvent_t;
typedef fakeGraph *hipGraph_t;
typedef fakeGraph *hipGraphExec_t;
typedef void *hipGraphNode_t;

struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  size_t totalGlobalMem;
  size_t sharedMemPerBlock;
  int warpSize;
  int clockRate;
  int memoryClockRate;
  int memoryBusWidth;
  int l2CacheSize;
  int multiProcessorCount;
  int pciDomainID, pciBusID, pciDeviceID;
};
struct hipUUID {
  char bytes[16];
};

enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum { hipHostMallocDefault = 0 };
typedef enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
typedef enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 } hipStreamCaptureStatus;
typedef enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 } hipStreamCaptureMode;

hipError_t hipGetDeviceCount(int *count);
hipError_t hipSetDevice(int device);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int device);
hipError_t hipDeviceGetPCIBusId(char *pciBusId, int len, int device);
hipError_t hipDeviceGetUuid(hipUUID *uuid, int device);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipGetLastError(void);
const char *hipGetErrorString(hipError_t e);

hipError_t hipMalloc(void **ptr, size_t bytes);
hipError_t hipFree(void *ptr);
hipError_t hipHostMalloc(void **ptr, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *ptr);
typedef enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 } hipMemoryType;
typedef struct hipPointerAttribute_t {
  hipMemoryType type;
} hipPointerAttribute_t;
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *attributes, const void *ptr);
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t stream);
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                            hipStream_t stream);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t stream);

hipError_t hipStreamCreateWithFlags(hipStream_t *stream, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *stream, unsigned flags, int priority);
hipError_t hipStreamDestroy(hipStream_t stream);
hipError_t hipStreamSynchronize(hipStream_t stream);
hipError_t hipStreamWaitEvent(hipStream_t stream, hipEvent_t event, unsigned flags);
hipError_t hipStreamIsCapturing(hipStream_t stream, hipStreamCaptureStatus *status);
hipError_t hipStreamBeginCapture(hipStream_t stream, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t stream, hipGraph_t *graph);

hipError_t hipEventCreate(hipEvent_t *event);
constexpr unsigned hipEventDisableTiming = 2;
hipError_t hipEventCreateWithFlags(hipEvent_t *event, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t event);
hipError_t hipEventRecord(hipEvent_t event, hipStream_t stream);
hipError_t hipEventSynchronize(hipEvent_t event);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t start, hipEvent_t stop);

hipError_t hipGraphInstantiate(hipGraphExec_t *exec, hipGraph_t graph, hipGraphNode_t *error_node, char *log, size_t log_bytes);
hipError_t hipGraphDestroy(hipGraph_t graph);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t stream);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
