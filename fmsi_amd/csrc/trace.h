// rocTX ranges (tracing): named host-side ranges around the stages of the CLI and of the library's one-off
// builders, visible in `rocprofv3 --marker-trace --kernel-trace` next to the kernels they enclose.  The rocTX library is
// dlopen()ed on first use (librocprofiler-sdk-roctx.so, else libroctx64.so); without it the ranges cost one branch.
#pragma once

namespace fmsi {

void trace_push(const char* name);
void trace_pop();

struct TraceRange {
    explicit TraceRange(const char* name) { trace_push(name); }
    ~TraceRange() { trace_pop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

}  // namespace fmsi
