/*
 * Minimal stand-in for NVIDIA's <cupti.h>, TEST INFRASTRUCTURE ONLY.
 *
 * The reference's native profiler (cupti_src/CuptiProfiler.{h,cpp}) cannot be built in a ROCm image
 * because libcupti / cupti.h do not exist here.  This header declares just the handful of CUPTI
 * names that file uses, so that the reference's OWN, UNMODIFIED sources (compiled from where they
 * lie under /root/reference -- they are never copied into this repo) can be linked against the fake
 * activity feed implemented in ref_driver.cpp.  That gives us the real computeStats(),
 * CircularBuffer and bufferCompleted() record handling as a ground truth for oracle/ and for the
 * golden vectors in tests/golden/.  None of this is product code.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#define CUPTIAPI

typedef enum {
    CUPTI_SUCCESS = 0,
    CUPTI_ERROR_MAX_LIMIT_REACHED = 12,
    CUPTI_ERROR_UNKNOWN = 999
} CUptiResult;

typedef enum {
    CUPTI_ACTIVITY_KIND_INVALID = 0,
    CUPTI_ACTIVITY_KIND_MEMCPY = 1,
    CUPTI_ACTIVITY_KIND_CONCURRENT_KERNEL = 10
} CUpti_ActivityKind;

typedef struct CUctx_st *CUcontext;

typedef struct {
    CUpti_ActivityKind kind;
} CUpti_Activity;

typedef struct {
    CUpti_ActivityKind kind;
    uint64_t start;
    uint64_t end;
    int32_t gridX, gridY, gridZ;
    int32_t blockX, blockY, blockZ;
    const char *name;
} CUpti_ActivityKernel4;

typedef void (*CUpti_BuffersCallbackRequestFunc)(uint8_t **buffer, size_t *size, size_t *maxNumRecords);
typedef void (*CUpti_BuffersCallbackCompleteFunc)(CUcontext ctx, uint32_t streamId, uint8_t *buffer,
                                                  size_t size, size_t validSize);

#ifdef __cplusplus
extern "C" {
#endif
CUptiResult cuptiGetResultString(CUptiResult result, const char **str);
CUptiResult cuptiActivityRegisterCallbacks(CUpti_BuffersCallbackRequestFunc req,
                                           CUpti_BuffersCallbackCompleteFunc done);
CUptiResult cuptiFinalize(void);
CUptiResult cuptiActivityEnable(CUpti_ActivityKind kind);
CUptiResult cuptiActivityDisable(CUpti_ActivityKind kind);
CUptiResult cuptiActivityFlushAll(uint32_t flag);
CUptiResult cuptiActivityGetNextRecord(uint8_t *buffer, size_t validBufferSizeBytes, CUpti_Activity **record);
#ifdef __cplusplus
}
#endif
