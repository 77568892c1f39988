// Internal helpers shared by the libsfem_hip translation units (not part of
// the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "sfem.h"

namespace sfem {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(sfem_stream_t s) {
  return reinterpret_cast<hipStream_t>(s);
}

#define SFEM_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ::sfem::set_error(__VA_ARGS__);    \
      return SFEM_EINVAL;                \
    }                                    \
  } while (0)

#define SFEM_HIP(call)                                                   \
  do {                                                                   \
    hipError_t err__ = (call);                                           \
    if (err__ != hipSuccess) {                                           \
      ::sfem::set_error("%s failed: %s", #call, hipGetErrorString(err__)); \
      return SFEM_EHIP;                                                  \
    }                                                                    \
  } while (0)

#define SFEM_LAUNCH_CHECK()                                              \
  do {                                                                   \
    hipError_t err__ = hipGetLastError();                                \
    if (err__ != hipSuccess) {                                           \
      ::sfem::set_error("kernel launch failed: %s",                      \
                        hipGetErrorString(err__));                       \
      return SFEM_EHIP;                                                  \
    }                                                                    \
  } while (0)

// Streaming kernels: grid-stride over at most 128 workgroups per CU.  Measured
// on MI355X (90 M doubles): y = a x + b y reaches 4.7 TB/s with 16 workgroups
// per CU, 5.3 with 64, 5.5 with 128 and no more beyond.
inline unsigned stream_grid(int64_t work_items, int block) {
  int64_t blocks = (work_items + block - 1) / block;
  const int64_t cap = 256 * 128;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

// A vector of this many bytes does not survive in the 256 MB Infinity Cache
// between two kernels: streaming kernels then use non-temporal accesses.
inline bool streams_past_caches(int64_t count, size_t elem_size) {
  return (size_t)count * elem_size >= ((size_t)256 << 20);
}

// Kernels that end in one atomic per workgroup on ONE address (dot products,
// the fused r.r): 16 workgroups per CU.  More helps the streaming part of a
// 90 M-element vector by 1.5 % but the serialised atomics then dominate short
// launches (11 M elements: +35 % with 64 per CU).
inline unsigned reduce_grid(int64_t work_items, int block) {
  int64_t blocks = (work_items + block - 1) / block;
  const int64_t cap = 256 * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

}  // namespace sfem
