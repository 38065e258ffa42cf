// Experiment switches.  The shipped library reads four environment variables -- COMPEG_TRACE, COMPEG_TRACE_BATCH,
// COMPEG_VERBOSE (stderr lines about what it does) and COMPEG_SCAN_THREADS (thread count of the host scan
// preprocessor) -- and nothing else: every knob that selects another kernel, another workgroup shape or a knock-out
// arm of a kernel body (CG_EXP) exists only in builds with -DCOMPEG_LAB (`make -C compeg_amd/csrc lab` ->
// compeg_amd/libcompeg_hip_lab.so, tools/build_variant.sh), which the A/B tools and one GPU test load through
// COMPEG_LIB.  tests/test_code_objects.py checks the shipped library's strings for strays.
#pragma once

#include <cstdlib>

namespace compeg {

inline const char *lab_env(const char *name)
{
#if defined(COMPEG_LAB)
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

} // namespace compeg
