#!/usr/bin/env python
"""Minimal end-to-end use of the layer on an MI355X: the README example of the reference (a polytope, a
sphere-like quadratic, a second-order cone and an LMI in R^3), a small network, one optimisation run.

    python examples/quickstart.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rayen_amd import constraint_module, constraints  # noqa: E402  (or: from rayen import ...)


def main():
    # linear:  0 <= y <= 1 (a cube);  quadratic: ||y - c||^2 <= 0.8^2;  SOC: ||(y1, y2)|| <= y3 + 0.9;
    # LMI: [[y1 + 1, y2], [y2, y3 + 1]] >= 0
    A1 = np.concatenate((np.eye(3), -np.eye(3)), axis=0)
    b1 = np.array([[1.0], [1.0], [1.0], [0.0], [0.0], [0.0]])
    lc = constraints.LinearConstraint(A1, b1, None, None)
    c = np.array([[0.5], [0.5], [0.5]])
    qc = constraints.ConvexQuadraticConstraint(2.0 * np.eye(3), -2.0 * c, c.T @ c - 0.64)
    soc = constraints.SOCConstraint(np.array([[1.0, 0, 0], [0, 1.0, 0]]), np.zeros((2, 1)),
                                    np.array([[0.0], [0.0], [1.0]]), np.array([[0.9]]))
    F = [np.array([[1.0, 0], [0, 0]]), np.array([[0, 1.0], [1.0, 0]]), np.array([[0, 0], [0, 1.0]]), np.eye(2)]
    lmic = constraints.LMIConstraint(F)
    cs = constraints.ConvexConstraints(lc=lc, qcs=[qc], socs=[soc], lmic=lmic, y0=c)

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(2, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                                constraint_module.ConstraintModule(cs, input_dim=64, create_map=True)).cuda()
    x = torch.randn(4096, 2, device="cuda")
    target = torch.tensor([2.0, 0.3, 0.1], device="cuda").view(1, 3, 1)   # outside the set
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    for step in range(200):
        opt.zero_grad()
        y = model(x)                                  # [4096, 3, 1], every row inside the set
        loss = ((y - target) ** 2).mean()
        loss.backward()
        opt.step()
        if step % 50 == 0 or step == 199:
            viol = cs.getMaxViolation(y.detach()[:, :, 0].double().cpu().numpy())
            print(f"step {step:3d}  loss {loss.item():.5f}  max constraint violation {viol:.2e}")
    assert viol < 1e-5


if __name__ == "__main__":
    main()
