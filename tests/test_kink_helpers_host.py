"""The test yardsticks themselves (tests/helpers.py: kink_mask, lmi_gradient_bound) on sets small enough to check by
hand.  No GPU."""
import numpy as np
import torch

from helpers import kink_mask, lmi_gradient_bound
from oracle import rayen_oracle as oracle
from rayen_amd import constraints


def _square():
    """|y_i| <= 1 in R^2 around y0 = 0: kappa(v) = max |v_i|."""
    A = np.vstack((np.eye(2), -np.eye(2)))
    return constraints.ConvexConstraints(lc=constraints.LinearConstraint(A, np.ones((4, 1)), None, None), y0=np.zeros((2, 1)))


def test_kinks_of_the_unit_square():
    cs = _square()
    v = torch.tensor([[2.0, 0.5], [2.0, 2.0], [2.0, 2.0 - 1e-6], [1.0, 0.2], [1e-9, 0.0], [0.5, 0.1], [0.0, 0.0]]).reshape(7, 2, 1)
    got = kink_mask(oracle, cs, v, 1e-4)
    #        plain   tie    near tie  kappa=1  tiny   interior  zero
    assert got.tolist() == [False, True, True, True, False, False, False]
    assert kink_mask(oracle, cs, v, 1e-8).tolist() == [False, True, False, True, False, False, False]


def test_lmi_bound_follows_the_eigen_gap():
    """diag(1 - y_1, 1 - y_2) >= 0, i.e. F_1 = -e1 e1', F_2 = -e2 e2', F_0 = I: the pencil matrix along v is diag(v):
    gap = |v_1 - v_2|, spectral radius max |v_i|."""
    F = [-np.diag([1.0, 0.0]), -np.diag([0.0, 1.0]), np.eye(2)]
    cs = constraints.ConvexConstraints(lmic=constraints.LMIConstraint(F), y0=np.zeros((2, 1)))
    v = torch.tensor([[2.0, 1.0], [2.0, 1.999], [-1.0, -2.0]]).reshape(3, 2, 1)
    bound = lmi_gradient_bound(oracle, cs, v, 6e-8)
    # sample 0: radius 2 / gap 1; sample 1: radius 2 / gap 1e-3; sample 2: both eigenvalues negative, the LMI is not on top
    assert np.isclose(bound[0], 4 * 2 * 6e-8 * 2.0 / 1.0, rtol=1e-6)
    assert np.isclose(bound[1], 4 * 2 * 6e-8 * 2.0 / 1e-3, rtol=1e-3)
    assert bound[2] == 0.0 or bound[2] < 1e-5
    km = kink_mask(oracle, cs, v, 1e-4)
    assert km.tolist() == [False, True, False]          # gap 1e-3 <= 10 x 1e-4 x radius: the eigenvector is undefined to 1e-3
