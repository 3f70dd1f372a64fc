// pt_pinned.h — std::vector over page-locked host memory. Host code only.
#pragma once
#include <cstddef>
#include <new>
#include <vector>

#include <hip/hip_runtime_api.h>

namespace pt {

// Page-locked host memory for what the host reads back while the device works (the round counters it looks at to see
// whether a batch is done, the refinement's totals): pageable copies are staged by the runtime at a few GB/s, pinned ones run at link
// speed and are truly asynchronous. Default construction is a no-op so that resize() does not zero-fill.
template <class T>
struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U>
    PinnedAlloc(const PinnedAlloc<U>&) {}
    T* allocate(size_t n) {
        void* p = nullptr;
        if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) { (void)hipHostFree(p); }
    template <class U>
    void construct(U*) noexcept {}
    template <class U, class A, class... Rest>
    void construct(U* p, A&& a, Rest&&... rest) {
        ::new ((void*)p) U(static_cast<A&&>(a), static_cast<Rest&&>(rest)...);
    }
    template <class U>
    bool operator==(const PinnedAlloc<U>&) const { return true; }
    template <class U>
    bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <class T>
using PinnedVec = std::vector<T, PinnedAlloc<T>>;

}  // namespace pt
