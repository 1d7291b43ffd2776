#pragma once

#include <cstdint>
#include <memory>
#include <new>
#include <utility>
#include <vector>

#include "../../include/fmsi_hip.h"

namespace fmsi {

// (the 1.4 GB of blocks at human scale are written once, every dword, by build_layout: no clearing pass first)
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U> void construct(U*) {}
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};

struct HostLayout {
    std::vector<uint32_t, NoInitAlloc<uint32_t>> blocks;   // n_blocks * 16 dwords, see common.h
    std::vector<uint32_t> mrank;    // n_blocks + 1 cumulative mask popcounts
    uint64_t n_blocks = 0;
};

// Returns 0 or a negative FMSI_HIP_E* code (message via set_last_error).
int build_layout(const fmsi_index_view& view, HostLayout& out);

}  // namespace fmsi
