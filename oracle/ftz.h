/* oracle/ftz.h -- test infrastructure (see oracle/__init__.py).
 * TensorFlow-1.15's CPU kernels run with denormals flushed to zero (port::ScopedFlushDenormal in its thread pools).
 * The reference's own checkpoints show it: beta1_power is exactly 0.0 after 10 001 multiplications by 0.9f (with
 * gradual underflow the product sticks at 4 ulp = 5.6e-45), and thousands of Adam m slots rest in
 * [0.9, 1.0) x 1.1755e-37 -- where (g - m) * (1 - beta1) flushes to zero and m stops decaying
 * (tests/test_ckpt_pins.py).  The restatement therefore runs its updates in the same mode (x86 MXCSR FTZ + DAZ). */
#ifndef ORACLE_FTZ_H
#define ORACLE_FTZ_H
#if defined(__x86_64__) || defined(__i386__)
#include <xmmintrin.h>
static inline unsigned oracle_ftz_on(void) {
    const unsigned old = _mm_getcsr();
    _mm_setcsr(old | 0x8040u);          /* FTZ (bit 15) | DAZ (bit 6) */
    return old;
}
static inline void oracle_ftz_restore(unsigned old) { _mm_setcsr(old); }
#else
static inline unsigned oracle_ftz_on(void) { return 0; }
static inline void oracle_ftz_restore(unsigned old) { (void)old; }
#endif
#endif
