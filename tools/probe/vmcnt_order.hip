// What hipcc itself assumes about vmcnt on gfx950: a load, then a younger STORE, then the first use of the
// loaded value.  If stores could retire out of order against older loads, the wait in front of the use would
// have to be vmcnt(0); hipcc emits vmcnt(1) -- its gfx9 model (SIInsertWaitcnts: loads and stores are one event
// type on one in-order counter when the target has no separate store counter) leaves the younger store
// outstanding.  Every compiler-scheduled kernel relies on this.  tests/test_asm_audit.py compiles this file and
// checks the count.
#include <hip/hip_runtime.h>
extern "C" __global__ void vmcnt_order_probe(const float *a, float *b, float *c)
{
    const int i = threadIdx.x;
    const float v = a[i];                 // load  (older)
    b[i] = 1.0f;                          // store (younger, independent)
    c[i] = v * 2.0f;                      // first use of the load
}
