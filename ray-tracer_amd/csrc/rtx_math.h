// rtx_math.h — correctly rounded f64 division with a SHARED reciprocal (device only).
//
// The compiler expands every f64 `/` into ~38 VALU instructions (v_div_scale x2, v_rcp_f64, two
// Newton steps, quotient, residual, v_div_fmas, v_div_fixup).  The reference divides several
// numerators by the same denominator all over the hot path (Vector3::normalize = 3 quotients by one
// norm, the two sphere roots by 2a, both slab distances of an axis by one direction component, the
// Projective3 point division by w33).  The Newton-refined reciprocal depends on the denominator only,
// so it is computed once and each extra quotient costs 3 instructions.
//
// Bit-exactness: for operands in the "middle" exponent range (denominator 2^-300 <= |d| < 2^301,
// numerator zero or in the same range) v_div_scale does not rescale and v_div_fixup passes the
// quotient through, so the sequence below IS the compiler's expansion with the common part hoisted —
// the quotients are the same correctly rounded IEEE-754 results, which is what Rust's `/` produces.
// Anything outside that range (denormals, infinities, NaN, huge exponent gaps) takes the plain `/`.
// tests/test_gpu_parity.py::test_device_math_bits checks both paths against the host's division.
#ifndef RTX_MATH_H
#define RTX_MATH_H
#include <stdint.h>

#define RTX_DI __device__ __forceinline__

// biased exponent in [723, 1323]: finite, normal, 2^-300 <= |x| < 2^301
RTX_DI bool rtx_exp_mid(double x) {
    const uint32_t hi = (uint32_t)(__builtin_bit_cast(uint64_t, x) >> 32);
    return (((hi >> 20) & 0x7ffu) - 723u) <= 600u;
}
RTX_DI bool rtx_num_ok(double x) { return rtx_exp_mid(x) || x == 0.0; }

// 1/d after the two Newton steps of the compiler's own f64 division expansion
RTX_DI double rtx_recip(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}
// n / d given r = rtx_recip(d): quotient estimate, exact residual, final correction
RTX_DI double rtx_quot(double n, double d, double r) {
    const double q = n * r;
    const double e = __builtin_fma(-d, q, n);
    // a zero numerator keeps n * r: the correction step would turn -0 / (d > 0) into +0
    return n == 0.0 ? q : __builtin_fma(e, r, q);
}

RTX_DI double rtx_div1(double n, double d) {
    double q;
    if (rtx_exp_mid(d) && rtx_num_ok(n)) q = rtx_quot(n, d, rtx_recip(d));
    else q = n / d;
    return q;
}
RTX_DI void rtx_div2(double n0, double n1, double d, double& q0, double& q1) {
    if (rtx_exp_mid(d) && rtx_num_ok(n0) && rtx_num_ok(n1)) {
        const double r = rtx_recip(d);
        q0 = rtx_quot(n0, d, r); q1 = rtx_quot(n1, d, r);
    } else {
        q0 = n0 / d; q1 = n1 / d;
    }
}
RTX_DI void rtx_div3(double n0, double n1, double n2, double d, double& q0, double& q1, double& q2) {
    if (rtx_exp_mid(d) && rtx_num_ok(n0) && rtx_num_ok(n1) && rtx_num_ok(n2)) {
        const double r = rtx_recip(d);
        q0 = rtx_quot(n0, d, r); q1 = rtx_quot(n1, d, r); q2 = rtx_quot(n2, d, r);
    } else {
        q0 = n0 / d; q1 = n1 / d; q2 = n2 / d;
    }
}

#endif
