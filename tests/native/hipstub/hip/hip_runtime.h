// TEST INFRASTRUCTURE: a CPU stand-in for the part of the HIP runtime that mitoflex_amd/csrc/mf_devingest.cpp uses, so that the
// device-ingest ORCHESTRATION (producer, uploader, consumers, writers, ring, text-buffer pool, carry hand-off) can run in the CPU test
// suite, plain and under ThreadSanitizer (tests/native/ingest_check.cpp).  Nothing of the product includes this file: it is found in
// place of <hip/hip_runtime.h> only when a test is compiled with -Itests/native/hipstub.
//
// Streams are real in-order queues, each drained by a thread of its own; events are recorded and waited for in stream order; copies and
// the stand-in "kernels" (tests/native/ingest_stub.cpp) run on the stream's thread.  So a missing dependency between streams, or a
// buffer that goes back to a pool while a stream still works on it, is a data race ThreadSanitizer can see -- not only the races
// between host threads.  "Device memory" is host memory.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <functional>

typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
struct StubStream; typedef StubStream *hipStream_t;
struct StubEvent; typedef StubEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum : unsigned { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocCoherent = 0x40000000, hipHostRegisterPortable = 1, hipHostRegisterReadOnly = 8 };
struct hipDeviceProp_t { int multiProcessorCount; char gcnArchName[256]; char name[256]; };

const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipInit(unsigned flags);
hipError_t hipGetDeviceCount(int *n);          // STUB_DEVICES of the environment (default 1)
hipError_t hipGetDevice(int *d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi);
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes);
hipError_t hipMalloc(void **p, size_t bytes);          // (STUB_FAIL_MALLOC_AT=n: the n-th allocation of the process fails as if the device were full)
template <class T> inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc(reinterpret_cast<void **>(p), bytes); }
hipError_t hipFree(void *p);                           // like the runtime's: waits for every stream first
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
template <class T> inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned flags) { return hipHostMalloc(reinterpret_cast<void **>(p), bytes, flags); }
hipError_t hipHostFree(void *p);
hipError_t hipHostRegister(void *p, size_t bytes, unsigned flags);          // (STUB_NO_REGISTER=1: fails, as on a file system whose pages cannot be pinned)
hipError_t hipHostUnregister(void *p);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int priority);
hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t words, const uint32_t *mask);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind);
hipError_t hipMemsetAsync(void *dst, int value, size_t n, hipStream_t s);

// for the stand-in kernels: run f on stream s, in order (s == nullptr: after everything else, on the caller's thread)
void stub_enqueue(hipStream_t s, std::function<void()> f);
// counters a test can look at
uint64_t stub_streams_made();
uint64_t stub_bytes_allocated_now();
uint64_t stub_bytes_allocated_peak();
