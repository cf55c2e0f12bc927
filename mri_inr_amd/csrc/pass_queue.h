// Host-side bookkeeping of the persistent trunks' pass queue (plain C++, no HIP: unit-tested on the CPU by
// tests/test_pass_queue.py).
//
// Device side: workgroup g starts with pass g; every executed pass performs exactly one atomicAdd(counter, 1)
// and takes  next = (old - base) + grid  as its next pass, so a launch of n passes advances the counter by
// exactly n.  The counter is never reset between launches: the host hands each launch the value the counter
// will hold when it starts (`base`), arithmetic modulo 2^32.
//
// The host value may only move once the launch is known to have been accepted: a launch that fails after the
// host has advanced would leave `base` ahead of the device counter for the life of the handle, and every later
// launch would compute negative pass ids.  Hence begin() / commit() / abort(): begin() hands out the base
// without moving it, commit() advances it after hipGetLastError() == hipSuccess, abort() forgets the claim.
#pragma once
#include <cstdint>

namespace msiren {

struct PassQueue {
    unsigned base = 0;      // value of the device counter when the next launch starts
    unsigned claimed = 0;   // passes of the launch between begin() and commit()/abort()
    bool open = false;

    unsigned begin(int64_t npasses) {
        claimed = (unsigned)npasses;
        open = true;
        return base;
    }
    void commit() {
        if (open) base += claimed;
        open = false;
    }
    void abort() { open = false; }
    // after a launch whose pass count only the device knows (the counter is memset to `value` behind it)
    void reset(unsigned value = 0) {
        base = value;
        open = false;
    }
    // the pass a workgroup takes after an atomicAdd that returned `old` (device arithmetic, restated for tests)
    static int next_pass(unsigned old, unsigned base, int grid) { return (int)(old - base) + grid; }
};

}  // namespace msiren
