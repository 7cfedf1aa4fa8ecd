/*
 * trace.h -- optional instrumentation of the host side of the extraction path.
 *
 *   POPSIFT_ROCTX       (make ROCTX=1): roctx ranges around submit / pyramid / keypoint stages / wait / fetch and, in
 *                       the C++ layer, around every job -- what the reference's NVTX ranges give under nvprof
 *                       (src/popsift/popsift.h:20-25, common/debug_macros.h); view with rocprofv3 --marker-trace.
 *   POPSIFT_SYNC_CHECK  (make SYNC_CHECK=1): after every kernel launch of the per-image sequence, synchronise the
 *                       stream and check the error state, so that a faulting kernel is reported at its launch site --
 *                       the reference's POP_SYNC_CHK (src/popsift/common/debug_macros.h:25-29).
 * Neither is compiled into the product library.
 */
#pragma once

#ifdef POPSIFT_ROCTX
#include <rocprofiler-sdk-roctx/roctx.h>
namespace popsift_hip {
struct TraceRange {
    explicit TraceRange(const char* name) { roctxRangePushA(name); }
    ~TraceRange() { roctxRangePop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};
}  // namespace popsift_hip
#define POPSIFT_RANGE_CAT2(a, b) a##b
#define POPSIFT_RANGE_CAT(a, b) POPSIFT_RANGE_CAT2(a, b)
#define POPSIFT_RANGE(name) popsift_hip::TraceRange POPSIFT_RANGE_CAT(popsift_range_, __LINE__)(name)
#else
#define POPSIFT_RANGE(name) \
    do {                    \
    } while (0)
#endif
