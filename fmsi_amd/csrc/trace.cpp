#include "trace.h"

#include <dlfcn.h>

#include <mutex>

namespace fmsi {
namespace {

using PushFn = int (*)(const char*);
using PopFn = int (*)();
PushFn g_push = nullptr;
PopFn g_pop = nullptr;
std::once_flag g_once;

void load() {
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        void* h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
        if (!h) continue;
        g_push = reinterpret_cast<PushFn>(dlsym(h, "roctxRangePushA"));
        g_pop = reinterpret_cast<PopFn>(dlsym(h, "roctxRangePop"));
        if (g_push && g_pop) return;
        g_push = nullptr;
        g_pop = nullptr;
    }
}

}  // namespace

void trace_push(const char* name) {
    std::call_once(g_once, load);
    if (g_push) (void)g_push(name);
}

void trace_pop() {
    if (g_pop) (void)g_pop();
}

}  // namespace fmsi
