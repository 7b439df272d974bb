// lds_attr.h -- the 150 KB dynamic-LDS attribute, set once per (kernel instantiation, device).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace sprintz {

// Kernels whose dynamic LDS carve can exceed 48 KB get the 150 KB maximum ONCE per (kernel instantiation, device) instead of a
// hipFuncSetAttribute on every launch (the single-call hot path; concurrent callers with different sizes raced on the attribute).
// Lock-free: a small open-addressed set of (kernel, device) keys; a lost race just sets the same value twice.
inline std::atomic<uintptr_t>* lds_attr_table()
{
    static std::atomic<uintptr_t> seen[256];
    return seen;
}
inline uintptr_t lds_attr_key(const void* kernel)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (uintptr_t)kernel * 64u + (uintptr_t)(dev & 63) + 1u;                      // != 0
}
// (kernel stubs are 16-byte aligned: the multiplicative hash takes its 8 bits from the TOP of the 32-bit product)
inline unsigned lds_attr_slot(uintptr_t key) { return ((uint32_t)(key >> 4) * 2654435761u) >> 24; }

inline hipError_t ensure_max_dynamic_lds(const void* kernel)
{
    constexpr int kMaxLds = 150 * 1024;
    std::atomic<uintptr_t>* const seen = lds_attr_table();
    const uintptr_t key = lds_attr_key(kernel);
    for (unsigned h = lds_attr_slot(key), n = 0; n < 256; h = (h + 1) & 255u, n++) {
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

// A launch that asked for more than 48 KB failed although the table says the attribute is set: the device was reset behind the
// table's back (hipDeviceReset drops function attributes).  Set it again -- the entry itself stays right -- and let the caller retry once.
inline hipError_t refresh_max_dynamic_lds(const void* kernel)
{
    (void)hipGetLastError();
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

// launch with a dynamic-LDS carve that may exceed 48 KB: attribute once per (kernel, device), one retry after a lost attribute
template <typename K, typename... Args>
inline hipError_t launch_with_lds(K kernel, unsigned grid, unsigned block, size_t shmem, hipStream_t st, const Args&... args)
{
    const void* const fn = reinterpret_cast<const void*>(kernel);
    if (shmem > 48 * 1024) {
        const hipError_t e = ensure_max_dynamic_lds(fn);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, st, args...);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess && shmem > 48 * 1024 && refresh_max_dynamic_lds(fn) == hipSuccess) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, st, args...);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace sprintz
