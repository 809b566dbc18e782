/*
 * ring_driver.cpp -- the one piece of the reference's native code that compiles here from its own file alone.
 *
 * TEST INFRASTRUCTURE ONLY.  The reference's CircularBuffer.h (cupti_src/CircularBuffer.h:22-70) is header-only and needs
 * nothing but <vector>: this translation unit #includes it from where it lies under /root/reference (include path given by
 * oracle/Makefile; nothing is copied) and exposes one C entry point.  No stand-in header, no fake library: unlike
 * ref_driver.cpp (which needs the stand-in cupti.h next to it to get CuptiProfiler.cpp through the compiler), this is a
 * reference build in the strict sense.  Output: oracle/_ref/libnvrx_ring_ref.so (git-ignored, shipped to the GPU box).
 */
#include <cstddef>
#include <vector>

#include "CircularBuffer.h"

extern "C" int ref_ring_run(const float *vals, int n, int capacity, float *out) {
    CircularBuffer<float> cb((size_t)capacity);
    for (int i = 0; i < n; i++) cb.push_back(vals[i]);
    std::vector<float> lin = cb.linearize();
    for (size_t i = 0; i < lin.size(); i++) out[i] = lin[i];
    return (int)lin.size();
}
