// Open-addressing hash table in HBM: uint64 keys | int32 values, struct-of-arrays, linear probing,
// load factor <= 0.5 (capacity = pow2 >= 2n, >= 1024).  Shared by hash.hip / kmap.hip / stvoxel.hip.
#pragma once
#include "common.h"

struct LtTable {
  uint64_t* keys;
  int* vals;
  uint32_t mask;
};

static inline int64_t lt_table_capacity_for(int64_t n) {
  int64_t want = 2 * (n > 0 ? n : 1);
  if (want < 1024) want = 1024;
  return lt_pow2_ge(want);
}
static inline int64_t lt_table_bytes_for(int64_t n) {
  const int64_t cap = lt_table_capacity_for(n);
  return lt_align_up(cap * 8, 256) + lt_align_up(cap * 4, 256);
}
static inline LtTable lt_table_view(void* mem, int64_t n) {
  const int64_t cap = lt_table_capacity_for(n);
  LtTable t;
  t.keys = (uint64_t*)mem;
  t.vals = (int*)((char*)mem + lt_align_up(cap * 8, 256));
  t.mask = (uint32_t)(cap - 1);
  return t;
}

int lt_table_clear(const LtTable& t, hipStream_t stream);  // keys=EMPTY, vals=0x7f7f7f7f (hash.hip)

// insert key; duplicates keep the smallest value.  Probe loop is bounded by the capacity so every wave exits
// (at load factor <= 0.5 it ends after a handful of probes).
__device__ __forceinline__ void lt_table_insert_min(const LtTable& t, uint64_t key, int val) {
  uint32_t s = lt_slot_mix(key) & t.mask;
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    const unsigned long long prev =
        atomicCAS((unsigned long long*)&t.keys[s], (unsigned long long)LT_EMPTY_KEY, (unsigned long long)key);
    if (prev == LT_EMPTY_KEY || prev == key) {
      atomicMin(&t.vals[s], val);
      return;
    }
    s = (s + 1) & t.mask;
  }
}

// slot of key or -1
__device__ __forceinline__ int lt_table_find(const LtTable& t, uint64_t key) {
  uint32_t s = lt_slot_mix(key) & t.mask;
  for (uint32_t probe = 0; probe <= t.mask; ++probe) {
    const uint64_t k = t.keys[s];
    if (k == key) return (int)s;
    if (k == LT_EMPTY_KEY) return -1;
    s = (s + 1) & t.mask;
  }
  return -1;
}
