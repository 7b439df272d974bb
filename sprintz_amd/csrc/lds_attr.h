// lds_attr.h -- the 150 KB dynamic-LDS attribute, set once per (kernel instantiation, device).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace sprintz {

// Kernels whose dynamic LDS carve can exceed 48 KB get the 150 KB maximum ONCE per (kernel instantiation, device) instead of a
// hipFuncSetAttribute on every launch (the single-call hot path; concurrent callers with different sizes raced on the attribute).
// Lock-free: a small open-addressed set of (kernel, device) keys; a lost race just sets the same value twice.
inline hipError_t ensure_max_dynamic_lds(const void* kernel)
{
    constexpr int kMaxLds = 150 * 1024;
    static std::atomic<uintptr_t> seen[256];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uintptr_t key = (uintptr_t)kernel * 64u + (uintptr_t)(dev & 63) + 1u;      // != 0
    for (unsigned h = (unsigned)((key >> 4) * 2654435761u) & 255u, n = 0; n < 256; h = (h + 1) & 255u, n++) {
        const uintptr_t v = seen[h].load(std::memory_order_acquire);
        if (v == key) return hipSuccess;
        if (v == 0) {
            const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
            if (e != hipSuccess) return e;
            uintptr_t expect = 0;
            (void)seen[h].compare_exchange_strong(expect, key, std::memory_order_release);
            return hipSuccess;
        }
    }
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);   // table full (never: a few dozen instantiations)
}

}  // namespace sprintz
