// csrc/vit_device.hpp -- what a launcher may know about "the device": always the CURRENT one, asked per call.
// Engines of several devices share a process (host/vit_facade.c, one host thread per device), so nothing about a device
// is cached in a process-wide variable: hipGetDevice + hipDeviceGetAttribute are host-side table look-ups.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace vitdev {

constexpr int MAX_DEVICES = 64;

// compute units of the current device (0 on error) and its ordinal
inline int current_cus(int *ordinal = nullptr) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (ordinal) *ordinal = dev;
    return cus;
}

// A kernel attribute belongs to one device's copy of the kernel: remember per ordinal that it was set (setting it twice from two
// threads is harmless, so a relaxed flag is enough).
struct PerDeviceOnce {
    std::atomic<bool> done[MAX_DEVICES];
    bool is_done(int dev) const { return dev >= 0 && dev < MAX_DEVICES && done[dev].load(std::memory_order_relaxed); }
    void set(int dev) { if (dev >= 0 && dev < MAX_DEVICES) done[dev].store(true, std::memory_order_relaxed); }
};

}  // namespace vitdev
