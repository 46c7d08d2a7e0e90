// rtx_pow.h — the specular factor `reflect_dot_eye.powf(shininess)` (material.rs:76-77).
//
// The device library's f64 pow is ~250 VALU instructions (double-double log and exp) and was 19 % of all
// vector instructions of the glass/mirror workload.  Scene files give shininess as a small non-negative
// integer (10 ... 300), for which x^n by binary powering needs <= 2*log2(n) multiplications.  Its result
// differs from a correctly rounded pow by at most ~n * 2^-53 relative (3e-14 for n = 300); the factor only
// scales a colour term <= specular * intensity, so the effect on an 8-bit channel is < 1e-11 LSB, far inside
// the +-1 LSB bar (the reference's own powf — libm — is not bit-identical across platforms either).
// Everything else (non-integer, negative or huge exponents) takes the library pow.
// Tested against the host in tests/test_gpu_parity.py::test_device_math_bits (op 7, tolerance stated there).
#ifndef RTX_POW_H
#define RTX_POW_H

// (out of line on purpose: inlined, the library pow's ~17 polynomial constants were hoisted to the kernel's prologue and
// spilled there by every wave — 34 spilled registers for a path scene files never take)
__device__ __attribute__((noinline)) double rtx_lib_pow(double x, double y) { return pow(x, y); }

__device__ __forceinline__ double rtx_spec_pow(double x, double y) {
    double r;
    if (__builtin_trunc(y) == y && y >= 0.0 && y <= 4096.0) {
        unsigned n = (unsigned)(int)y;
        double b = x;
        r = 1.0;
        while (n) {
            if (n & 1u) r *= b;
            n >>= 1;
            if (n) b *= b;
        }
    } else {
        r = rtx_lib_pow(x, y);
    }
    return r;
}

#endif
