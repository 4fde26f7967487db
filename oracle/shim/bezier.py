"""TEST INFRASTRUCTURE — container-only stand-in for the un-vendored `bezier` package
(unpinned dependency of the reference, call sites DAM:616-618, 649-651, 684-686).

Restates the package's published evaluation scheme for `Curve.evaluate_multi`
(barycentric Horner form: result = ((l1*n0 + C(3,1) l2 n1) l1 + C(3,2) l2^2 n2) l1 + l2^3 n3
with l1 = 1-s, l2 = s), computed in float64 like the package does; the reference casts the
result to fp32 right after (DAM:619), so any correct float64 evaluator agrees after the cast
up to rounding ties.  Used only by oracle/gen_golden.py.
"""
import numpy as np


class Curve(object):
    def __init__(self, nodes, degree):
        self.nodes = np.asarray(nodes, dtype=np.float64)
        self.degree = int(degree)
        assert self.nodes.shape[1] == self.degree + 1

    def evaluate_multi(self, s_vals):
        s = np.asarray(s_vals, dtype=np.float64)
        lambda1 = 1.0 - s
        lambda2 = s
        nodes = self.nodes
        degree = self.degree
        result = np.outer(nodes[:, 0], lambda1)
        binom_val = 1.0
        lambda2_pow = np.ones_like(s)
        for index in range(1, degree):
            lambda2_pow = lambda2_pow * lambda2
            binom_val = (binom_val * (degree - index + 1)) / index
            result = result + np.outer(nodes[:, index], binom_val * lambda2_pow)
            result = result * lambda1[np.newaxis, :]
        result = result + np.outer(nodes[:, degree], lambda2 * lambda2_pow)
        return result
