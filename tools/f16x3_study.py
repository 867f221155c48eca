#!/usr/bin/env python3
"""CPU study (numpy, no GPU): would a TWO-piece fp16 operand split with a scaled residual -- three matrix products instead of the
six of the bf16x6 mode -- still be fp32-wide?

  x  =  h + l' * 2^-12 + e,     h = fp16_rne(x),   l' = fp16_rne((x - h) * 2^12),   |e| <= 2^-23 |x|   (two roundings to nearest:
                                                                                  11 + 11 bits and one bit from each sign)
  A W  ~=  Ah Wh  +  2^-12 (Ah Wl' + Al' Wh)          (two fp32 accumulators; the dropped Al' Wl' term is 2^-24 relative)

against   bf16x6:  x = h + m + l (bf16, rne),  A W ~= AhWh + AhWm + AmWh + AhWl + AlWh + AmWm   (one accumulator, the product mode)
and       plain fp32 (numpy float32 matmul) -- all three measured against the float64 product of the SAME fp32 operands.

Operands: the shapes and magnitudes of the hot path: post-ReLU activations of a default-initialised 256-wide layer,
weights U(+-1/16); a "gradient" case with a log-normal spread over eight decades (what dY looks like) to see the fp16 range at work;
and an adversarial tiny-value case.  Output: max and rms error relative to rms(|exact|) per case.  Round-4 note: profiles/r04_f16x3_study.md.
"""
import numpy as np


def bf16_rne(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split_bf16x3(x):
    h = bf16_rne(x); r = (x - h).astype(np.float32)
    m = bf16_rne(r); r2 = (r - m).astype(np.float32)
    return h, m, bf16_rne(r2)


def split_f16x2(x, S=12):
    h = x.astype(np.float16).astype(np.float32)           # numpy converts with round-to-nearest-even, subnormals kept
    r = (x - h).astype(np.float32)                        # exact in fp32
    l = (r * np.float32(2.0 ** S)).astype(np.float16).astype(np.float32)
    return h, l


def mm32(a, b):   # fp32 accumulate (the order differs from the MFMA's; the error level does not)
    return a.astype(np.float32) @ b.astype(np.float32)


def prod_bf16x6(A, W):
    Ah, Am, Al = split_bf16x3(A); Wh, Wm, Wl = split_bf16x3(W)
    acc = mm32(Al, Wh); acc = acc + mm32(Ah, Wl); acc = acc + mm32(Am, Wm); acc = acc + mm32(Am, Wh); acc = acc + mm32(Ah, Wm)
    return acc + mm32(Ah, Wh)


def prod_f16x3(A, W, S=12):
    Ah, Al = split_f16x2(A, S); Wh, Wl = split_f16x2(W, S)
    acc2 = mm32(Ah, Wl) + mm32(Al, Wh)
    return mm32(Ah, Wh) + acc2 * np.float32(2.0 ** -S)


def prod_f16x3_unscaled(A, W):   # what NOT to do: the residual straight to fp16 (subnormal for |x| < 0.25)
    return prod_f16x3(A, W, 0)


def report(name, A, W):
    exact = A.astype(np.float64) @ W.astype(np.float64)
    scale = np.sqrt(np.mean(exact ** 2))
    rows = []
    for label, fn in (('fp32 matmul', mm32), ('bf16x6', prod_bf16x6), ('f16x3 scaled residual', prod_f16x3), ('f16x3 unscaled', prod_f16x3_unscaled)):
        d = fn(A, W).astype(np.float64) - exact
        rows.append((label, np.max(np.abs(d)) / scale, np.sqrt(np.mean(d ** 2)) / scale))
    print('%s   (rms |exact| = %.3e, inf / nan in f16x3: %s)' % (name, scale, not np.isfinite(prod_f16x3(A, W)).all()))
    for label, mx, rms in rows:
        print('    %-24s max %.3e   rms %.3e' % (label, mx, rms))
    return rows


def main():
    rng = np.random.default_rng(0)
    P, K, N = 4096, 256, 256
    W = rng.uniform(-1 / 16, 1 / 16, (K, N)).astype(np.float32)
    X0 = rng.uniform(-1, 1, (P, K)).astype(np.float32)
    A = np.maximum(X0 @ rng.uniform(-1 / 16, 1 / 16, (K, K)).astype(np.float32), 0).astype(np.float32)   # post-ReLU activations
    report('hidden layer: relu activations [4096,256] x weights U(+-1/16)', A, W)
    G = (rng.standard_normal((P, K)) * np.exp(rng.normal(-14, 3.0, (P, K)))).astype(np.float32)           # ~1e-6 with a wide spread
    report('dX-like: gradients, log-normal spread (median 8e-7, 1e-10 .. 1e-2) x weights', G, W)
    report('dW-like: gradients^T x activations (K = 4096 points)', G.T.copy(), A)
    T = (rng.standard_normal((P, K)) * 1e-9).astype(np.float32)
    report('tiny operands 1e-9 (fp16 flushes to its subnormal grid: absolute, not relative, accuracy)', T, W)
    Bg = (A * np.float32(3000)).astype(np.float32)
    report('large activations (x 3000, max %.0f; fp16 overflows above 65504)' % Bg.max(), Bg, W)


if __name__ == '__main__':
    main()
