"""Reduced-precision convolution semantics of the MI355X path (TEST INFRASTRUCTURE ONLY; beyond the reference, which
computes in floatX=float32 -- experiment.5.sh:5 -- so there is nothing in /root/reference to pin this on).

BASELINE configs 4 / 5 ask for bf16 / fp16 matrix-core arithmetic: the HIP kernels (csrc/conv_lp.hip) multiply operands
rounded to bf16 (or fp16) with round-to-nearest-even exactly and accumulate in float32; BatchNorm statistics, losses,
master weights and the optimiser stay float32.  This module states that rule in numpy so the kernels can be checked
bit-tightly: conv(round(x), round(W)) in float64.

Round 3: an operand is rounded ONCE, AT ITS PRODUCER, and stored as a "q tensor" (include/ghm.h) that every consumer
reads -- forward convolution, data gradient and weight gradient see the identical rounded values.  Rounding is a
function of the fp32 value alone, so this is the same arithmetic as rounding at each consumer: ROUND[dtype](t) below is
both "what the kernel's operand is" and "what the producer's q epilogue must have stored" (tests: q == ROUND(fp32 result)
bit for bit).  Where the fp32 copy of a tensor is no longer written at all (its only readers are low-precision
products), nothing changes for this restatement: those readers never saw anything but the rounded value.
"""
import numpy as np

from . import ops


def round_bf16(a):
    """float32 -> nearest bfloat16 (ties to even), returned as float32"""
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    out = u.astype(np.uint32).view(np.float32).reshape(a.shape)
    return np.where(np.isfinite(a), out, a)


def round_f16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


ROUND = {'bf16': round_bf16, 'f16': round_f16, 'f32': lambda a: np.asarray(a, np.float32)}


def conv2d_fwd(x, W, b, stride, pad, dtype):
    r = ROUND[dtype]
    return ops.conv2d_fwd(r(x).astype(np.float64), r(W).astype(np.float64),
                          None if b is None else np.asarray(b, np.float64), stride, pad)


def conv2d_vjp(x, W, dy, stride, pad, dtype):
    """(dx, dW, db) with every matrix-core operand rounded: dx = conv^T(round(dy), round(W)),
    dW = corr(round(x), round(dy)); db is a plain float32 sum (not a matrix-core product)."""
    r = ROUND[dtype]
    x64, W64, dy64 = (r(v).astype(np.float64) for v in (x, W, dy))
    dx, dW, _ = ops.conv2d_vjp(x64, W64, dy64, stride, pad)
    db = np.asarray(dy, np.float64).sum(axis=(0, 2, 3))
    return dx, dW, db


def split_bf16x3(a):
    """float32 -> three bfloat16 pieces (returned as float32 arrays) with p0 + p1 + p2 == a EXACTLY: the operand form of
    the split-fp32 kernels (csrc/conv_split.hip; beyond the reference, which computes in floatX=float32).
    p0 = bf16(a), p1 = bf16(a - p0), p2 = a - p0 - p1: both subtractions are exact in float32 (the residual of a
    round-to-nearest has fewer significant bits than its operand), and p2 has at most 8 significant bits."""
    a = np.ascontiguousarray(a, np.float32)
    p0 = round_bf16(a)
    r1 = (a - p0).astype(np.float32)
    p1 = round_bf16(r1)
    p2 = (r1 - p1).astype(np.float32)
    return p0, p1, p2


def split_product_terms():
    """the (operand-A piece, operand-B piece) pairs the kernels multiply: i + j <= 2; the three they drop are each below
    2^-24 |a b| (piece i is at most 2^-8i of the value, up to rounding)"""
    return [(i, j) for i in range(3) for j in range(3) if i + j <= 2]


def split_bf16x2(a):
    """float32 -> TWO bfloat16 pieces (as float32 arrays): p0 = bf16(a), p1 = bf16(a - p0), both round-to-nearest-even -- the
    operand form of the 'bf16x2' mode (csrc/conv_split.hip with pieces = 2; BASELINE config 4's arithmetic).  p0 + p1 carries
    16-17 significant bits: |a - p0 - p1| <= 2^-17 |a|."""
    a = np.ascontiguousarray(a, np.float32)
    p0 = round_bf16(a)
    p1 = round_bf16((a - p0).astype(np.float32))
    return p0, p1


def split2_product_terms():
    """the (operand-A piece, operand-B piece) pairs of the 'bf16x2' kernels: i + j <= 1 -- x0 w0 + x1 w0 + x0 w1; the dropped
    x1 w1 is 2^-16 of the product, the same order as the operands' own truncation"""
    return [(0, 0), (1, 0), (0, 1)]


def conv2d_fwd_x2(x, W, b, stride, pad):
    """what the 'bf16x2' forward kernel computes, in float64: the three piece products of the two-piece operands"""
    px, pw = split_bf16x2(x), split_bf16x2(W)
    out = 0.0
    for i, j in split2_product_terms():
        out = out + ops.conv2d_fwd(px[i].astype(np.float64), pw[j].astype(np.float64), np.zeros(W.shape[0]), stride, pad)
    return out + np.asarray(b, np.float64)[None, :, None, None]


def conv2d_vjp_x2(x, W, dy, stride, pad):
    """-> (dx, dW) of the 'bf16x2' data- and weight-gradient kernels: the same three piece products of (dy, W) and (x, dy)"""
    px, pw, pd = split_bf16x2(x), split_bf16x2(W), split_bf16x2(dy)
    dx = dW = 0.0
    for i, j in split2_product_terms():
        dx = dx + ops.conv2d_vjp(px[0].astype(np.float64), pw[j].astype(np.float64), pd[i].astype(np.float64), stride, pad)[0]
        dW = dW + ops.conv2d_vjp(px[i].astype(np.float64), pw[0].astype(np.float64), pd[j].astype(np.float64), stride, pad)[1]
    return dx, dW
