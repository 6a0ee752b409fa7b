// wide_node.h -- the 64-byte record of the 4-wide quantized tree k_trace_w4 walks (layout: wide_bvh.cpp, "Record").  Plain header: host and device code share it.
#pragma once
#include <stdint.h>

#ifndef RT_LEAF_BIT
#define RT_LEAF_BIT 0x80000000u
#define RT_EMPTY_REF 0xFFFFFFFFu
#endif

struct WideNode { float ox, oy, oz; uint32_t meta; uint32_t lo[3]; uint32_t hi[3]; uint32_t ref[4]; uint32_t order; uint32_t pad; };
static_assert(sizeof(WideNode) == 64, "wide node record");
