// Test-only: lets the device-side inline arithmetic of falcon-r1cs_amd/csrc/frw_fr.h / frw_fr29.h compile as plain host
// C++ (g++), so that the CPU test-suite can exercise it against Python integers (tests/test_fr29_host.py).  Never on an
// include path of the product.
#pragma once
#include <cstdint>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
