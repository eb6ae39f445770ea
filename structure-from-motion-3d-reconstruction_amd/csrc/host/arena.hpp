// arena.hpp — bump allocator for the pipeline's std::unordered_map nodes.
//
// The reference keeps track/keyframe/map bookkeeping in std::unordered_map (T:766-798, 1695) and ITERATES
// those maps where order matters (H2 in SURVEY.md), so sfmx must use the same container with the same
// hash, insertion sequence and rehash policy.  The allocator is not part of that contract: it changes
// neither bucket counts nor node order.  Packing the nodes contiguously removes one malloc per inserted
// element and most of the cache misses of the per-keyframe full-map walks (T:1804, T:871).
// Memory is released when the arena dies (end of the run); deallocate() is a no-op.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <functional>
#include <memory>
#include <new>
#include <unordered_map>
#include <vector>

namespace sfmx_host {

class Arena {
 public:
  Arena() = default;
  Arena(const Arena&) = delete;
  Arena& operator=(const Arena&) = delete;
  ~Arena() { for (void* p : chunks_) std::free(p); }
  void* alloc(std::size_t bytes, std::size_t align) {
    std::size_t cur = (off_ + align - 1) & ~(align - 1);
    if (chunks_.empty() || cur + bytes > cap_) {
      const std::size_t want = bytes + align > kChunk ? bytes + align : kChunk;
      void* p = std::malloc(want);
      if (!p) throw std::bad_alloc();
      chunks_.push_back(p);
      base_ = static_cast<char*>(p);
      cap_ = want;
      cur = (reinterpret_cast<std::uintptr_t>(base_) % align) ? align - reinterpret_cast<std::uintptr_t>(base_) % align : 0;
    }
    off_ = cur + bytes;
    return base_ + cur;
  }

 private:
  static constexpr std::size_t kChunk = 1u << 20;
  std::vector<void*> chunks_;
  char* base_ = nullptr;
  std::size_t cap_ = 0, off_ = 0;
};

template <class T>
struct ArenaAlloc {
  using value_type = T;
  Arena* arena = nullptr;  // null => plain operator new/delete
  ArenaAlloc() = default;
  explicit ArenaAlloc(Arena* a) : arena(a) {}
  template <class U> ArenaAlloc(const ArenaAlloc<U>& o) : arena(o.arena) {}
  T* allocate(std::size_t n) {
    if (!arena) return static_cast<T*>(::operator new(n * sizeof(T)));
    return static_cast<T*>(arena->alloc(n * sizeof(T), alignof(T) < 8 ? 8 : alignof(T)));
  }
  void deallocate(T* p, std::size_t) noexcept {
    if (!arena) ::operator delete(p);
  }
  template <class U> bool operator==(const ArenaAlloc<U>& o) const { return arena == o.arena; }
  template <class U> bool operator!=(const ArenaAlloc<U>& o) const { return arena != o.arena; }
};

// same hash / equality / policy as std::unordered_map<K,V>; only the allocator differs
template <class K, class V>
using ArenaMap = std::unordered_map<K, V, std::hash<K>, std::equal_to<K>, ArenaAlloc<std::pair<const K, V>>>;

}  // namespace sfmx_host
