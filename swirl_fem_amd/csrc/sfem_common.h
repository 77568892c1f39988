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

// Streaming kernels: enough workgroups to fill 256 CUs several times over,
// grid-stride the rest.
inline unsigned stream_grid(int64_t work_items, int block) {
  int64_t blocks = (work_items + block - 1) / block;
  const int64_t cap = 256 * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

}  // namespace sfem
