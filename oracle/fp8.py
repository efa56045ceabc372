"""ORACLE (test infrastructure only) -- the operand quantisation of the opt-in fp8 attention forward (csrc/attention8.hip).

There is no reference code for this mode: the reference computes attention in fp32 (multihead_custom_attention.py:386-447),
BASELINE.json configs[4] only names "fp8 MFMA attention".  What is restated here is the published OCP FP8 E4M3 ("e4m3fn")
encoding -- 1 sign, 4 exponent (bias 7), 3 mantissa bits, subnormals at exponent field 0 (step 2^-9), no infinities, largest
finite value 448 -- with round-to-nearest-even, and the scale rule of the kernel (power-of-two scales from per-(sample, head)
maxima).  Pinned on CPU against torch's own float8_e4m3fn conversion (tests/test_oracle_golden.py).
"""
import numpy as np

E4M3_MAX = 448.0


def e4m3_bytes(x):
    """float32 array -> uint8 e4m3fn codes, round to nearest even, magnitudes above 448 saturate to 448 (the kernel clamps
    before converting)."""
    x = np.asarray(x, dtype=np.float32)
    sign = np.signbit(x).astype(np.uint8) << 7
    a = np.minimum(np.abs(x).astype(np.float64), E4M3_MAX)
    # exponent of the binade (normal range 2^-6 .. 2^8); everything below 2^-6 shares the subnormal step 2^-9
    _, ex = np.frexp(a)                              # a = f 2^ex with f in [0.5, 1): exact, unlike floor(log2(a))
    e = np.clip(ex.astype(np.int64) - 1, -6, 8)
    step = np.exp2(e - 3.0)
    q = np.rint(a / step)                            # np.rint rounds half to even; q is the integer significand (0 .. 16)
    # a significand of 16 means the value rounded up into the next binade
    up = q >= 16
    e = np.where(up, e + 1, e)
    q = np.where(up, 8, q)
    sub = q < 8                                      # subnormal (only possible in the lowest binade) or zero
    expf = np.where(sub, 0, e + 7).astype(np.int64)
    mant = np.where(sub, q, q - 8).astype(np.int64)
    code = (expf << 3) | mant
    code = np.minimum(code, 0x7E)                    # 0x7E = 448; 0x7F is NaN in e4m3fn
    return (sign | code.astype(np.uint8)).astype(np.uint8)


def e4m3_values(codes):
    """uint8 e4m3fn codes -> float32 values (0x7F / 0xFF = NaN)."""
    c = np.asarray(codes, dtype=np.uint8).astype(np.int64)
    s = np.where(c & 0x80, -1.0, 1.0)
    e = (c >> 3) & 0xF
    m = c & 7
    v = np.where(e == 0, m * 2.0 ** -9, (8 + m) * np.exp2(e - 10.0))
    v = np.where((c & 0x7F) == 0x7F, np.nan, v)
    return (s * v).astype(np.float32)


def frexp_exponent(amax):
    """e with amax = f 2^e, f in [0.5, 1); 0 for amax == 0 (as exponent_of in attention8.hip)."""
    amax = np.asarray(amax, dtype=np.float32)
    _, e = np.frexp(amax)
    return np.where(amax > 0, e, 0).astype(np.int64)


def attention_scales(amax_k, amax_q, amax_v):
    """(ek, ev) of attention8.hip:scales_of -- k8 = k 2^ek, q8 = q 2^-ek with ek = floor((e_q - e_k) / 2); v8 = v 2^ev with
    ev = 8 - e_v (largest |v8| in [128, 256))."""
    e_k, e_q, e_v = frexp_exponent(amax_k), frexp_exponent(amax_q), frexp_exponent(amax_v)
    return np.floor_divide(e_q - e_k, 2), 8 - e_v
