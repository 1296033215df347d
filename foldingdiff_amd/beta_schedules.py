"""
Variance schedules and the derived per-timestep tables, computed ONCE on the
host in float32 with the same torch op order as the reference
(foldingdiff/beta_schedules.py:20-78) so the values are bit-identical, then
uploaded to the GPU where the sampling loop indexes them by the device-resident
step counter (csrc/rowwise.hip ``head_update_kernel``).
"""
from typing import Dict, Literal, get_args

import torch
import torch.nn.functional as F

SCHEDULES = Literal["linear", "cosine", "quadratic"]


def cosine_beta_schedule(timesteps: int, s: float = 8e-3) -> torch.Tensor:
    """Nichol & Dhariwal cosine schedule, clipped to [1e-4, 0.9999]."""
    grid = torch.linspace(0, timesteps, timesteps + 1)
    abar = torch.cos(((grid / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.clip(1 - (abar[1:] / abar[:-1]), 0.0001, 0.9999)


def linear_beta_schedule(timesteps: int, beta_start=1e-4, beta_end=0.02) -> torch.Tensor:
    return torch.linspace(beta_start, beta_end, timesteps)


def quadratic_beta_schedule(timesteps: int, beta_start=1e-4, beta_end=0.02) -> torch.Tensor:
    ramp = torch.linspace(-6, 6, timesteps)
    return torch.sigmoid(ramp) * (beta_end - beta_start) + beta_start


_BUILDERS = {"cosine": cosine_beta_schedule, "linear": linear_beta_schedule, "quadratic": quadratic_beta_schedule}


def get_variance_schedule(keyword: SCHEDULES, timesteps: int, **kwargs) -> torch.Tensor:
    if keyword not in _BUILDERS:
        raise ValueError(f"Unrecognized variance schedule: {keyword}")
    return _BUILDERS[keyword](timesteps, **kwargs)


def compute_alphas(betas: torch.Tensor) -> Dict[str, torch.Tensor]:
    alphas = 1.0 - betas
    abar = torch.cumprod(alphas, dim=0)
    abar_prev = F.pad(abar[:-1], (1, 0), value=1.0)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": abar,
        "sqrt_alphas_cumprod": torch.sqrt(abar),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - abar),
        "posterior_variance": betas * (1.0 - abar_prev) / (1.0 - abar),
    }


def step_coefficients(betas: torch.Tensor) -> torch.Tensor:
    """The four scalars p_sample reads per step (foldingdiff/sampling.py:41-72), as
    one float32 [4, T] table for fd_finalize:
        row 0  1 / sqrt(alpha_t)            row 2  sqrt(1 - alphabar_t)
        row 1  beta_t                       row 3  sqrt(posterior_variance_t)
    sqrt(posterior_variance_0) is exactly 0, so t = 0 needs no special table entry
    (the kernel still branches on t > 0 like the reference does)."""
    betas = betas.detach().to(dtype=torch.float32, device="cpu")
    terms = compute_alphas(betas)
    return torch.stack(
        [
            1.0 / torch.sqrt(terms["alphas"]),
            betas,
            terms["sqrt_one_minus_alphas_cumprod"],
            torch.sqrt(terms["posterior_variance"]),
        ]
    ).contiguous()
