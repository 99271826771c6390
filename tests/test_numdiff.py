"""Finite differences used in place of numdifftools (expr.py:67, 108 of the reference).
The reference's tests pin numeric derivatives to rtol 1e-5 / atol 1e-8 of the analytic
ones (tests/sco_osqp/test_expr.py:71-78, 151-211); this scheme is far inside that."""
import numpy as np

from oracle import arm_family as af
from oracle import sco_ref as sr
from sco_py_amd import numdiff


def test_jacobian_of_smooth_functions():
    f = lambda x: np.array([np.sin(x[0]) * x[1], np.exp(0.3 * x[0]) + x[1] ** 3, x[0] * x[1]])
    x = np.array([0.7, -1.3])
    J = numdiff.jacobian(f, x)
    Ja = np.array([[np.cos(x[0]) * x[1], np.sin(x[0])], [0.3 * np.exp(0.3 * x[0]), 3 * x[1] ** 2], [x[1], x[0]]])
    assert np.abs(J - Ja).max() < 1e-11


def test_hessian_of_smooth_function():
    f = lambda x: np.array([np.log(1 + x[0] ** 2) - x[1] + x[0] * x[1] ** 2])
    x = np.array([2.0, 2.0])
    H = numdiff.hessian(f, x)
    d2 = (2 * (1 + x[0] ** 2) - 4 * x[0] ** 2) / (1 + x[0] ** 2) ** 2
    Ha = np.array([[d2, 2 * x[1]], [2 * x[1], 2 * x[0]]])
    assert np.abs(H - Ha).max() < 1e-8 and np.array_equal(H, H.T)


def test_host_and_oracle_schemes_are_the_same_formula():
    pr = af.make_problem(3, d=4, T=3, K=3, O=2)
    f = lambda th: af.arm_dist(th, pr["link_len"], pr["point_link"], pr["point_frac"], pr["obstacles"])
    th = pr["x0"][:4]
    assert np.array_equal(numdiff.jacobian(f, th), sr.fd_jacobian(f, th))
    assert np.abs(numdiff.jacobian(f, th) - af.arm_dist_jac(th, pr["link_len"], pr["point_link"], pr["point_frac"],
                                                            pr["obstacles"])).max() < 1e-11
