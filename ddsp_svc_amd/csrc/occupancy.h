// How many workgroups of a kernel the CURRENT device holds at once (CUs x resident workgroups per CU): the launchers that
// size one round of exactly resident workgroups (loss_czt.hip, ir_czt.hip) ask once per (device, kernel) and keep the answer
// in a table of atomics -- several host threads, several devices per process.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace ddsp {

constexpr int kMaxDevices = 64;

struct ResidentCache {
  std::atomic<int> per_device[kMaxDevices];
};

template <class K>
static int resident_workgroups(K kernel, int threads, ResidentCache& cache) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  std::atomic<int>* slot = dev < kMaxDevices ? &cache.per_device[dev] : nullptr;
  if (slot) {
    const int v = slot->load(std::memory_order_relaxed);
    if (v > 0) return v;
  }
  int cus = 0, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  const int v = cus * per_cu;
  if (slot) slot->store(v, std::memory_order_relaxed);          // every thread computes the same value: a benign double fill
  return v;
}

}  // namespace ddsp
