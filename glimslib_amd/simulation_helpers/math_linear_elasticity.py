"""
Constitutive helpers of glimslib/simulation_helpers/math_linear_elasticity.py:6-33 on numpy arrays.
Tensor arguments are arrays whose last two axes are the (dim x dim) tensor; displacement gradients are
[..., dim, dim] with grad[..., a, b] = d u_a / d x_b.
"""
import numpy as np


def compute_mu(young_modulus, poisson_ratio):
    return young_modulus / (2.0 * (1.0 + poisson_ratio))


def compute_lambda(young_modulus, poisson_ratio):
    return young_modulus * poisson_ratio / ((1.0 + poisson_ratio) * (1.0 - 2.0 * poisson_ratio))


def compute_strain(grad_u):
    grad_u = np.asarray(grad_u)
    return 0.5 * (grad_u + np.swapaxes(grad_u, -1, -2))


def compute_stress(grad_u, mu, lmbda):
    eps = compute_strain(grad_u)
    d = eps.shape[-1]
    tr = np.trace(eps, axis1=-2, axis2=-1)
    return 2.0 * np.asarray(mu)[..., None, None] * eps + (np.asarray(lmbda) * tr)[..., None, None] * np.eye(d)


def compute_pressure_from_stress_tensor(stress_tensor):
    return 1.0 / 3.0 * np.trace(stress_tensor, axis1=-2, axis2=-1)


def compute_growth_induced_strain(conc_field, coupling_constant, dim):
    return (np.asarray(conc_field) * coupling_constant)[..., None, None] * np.eye(dim)


def compute_total_jacobian(grad_u):
    d = np.asarray(grad_u).shape[-1]
    return np.linalg.det(np.eye(d) + grad_u)


def compute_growth_induced_jacobian(growth_induced_strain, dim):
    return np.linalg.det(np.eye(dim) + growth_induced_strain)


def compute_deviatoric_stress_tensor(stress_tensor, dim):
    tr = np.trace(stress_tensor, axis1=-2, axis2=-1)
    return stress_tensor - (1.0 / 3.0) * tr[..., None, None] * np.eye(dim)


def compute_van_mises_stress(stress_tensor, dim):
    dev = compute_deviatoric_stress_tensor(stress_tensor, dim)
    return np.sqrt(1.5 * np.einsum('...ab,...ab->...', dev, dev))
