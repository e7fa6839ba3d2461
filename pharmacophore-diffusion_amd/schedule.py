"""Noise schedule and per-step scalar algebra of the diffusion wrapper (host logic, tiny B-length
vector work that the reference also does with torch ops): pharmacoforge/models/pharmacodiff.py
:140-160 (sigma, alpha, sigma_and_alpha_t_given_s), :387-420 (per-step terms), :582-668 (schedules)."""
from typing import Dict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def cosine_beta_schedule(timesteps, s=0.008, raise_to_power: float = 1):
    """pharmacodiff.py:582-599 (kept for API completeness; the model hard-wires 'polynomial_2')."""
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    ac = np.cumprod(1. - betas, axis=0)
    return np.power(ac, raise_to_power) if raise_to_power != 1 else ac


def clip_noise_schedule(alphas2, clip_value=0.001):
    """pharmacodiff.py:602-615: clip alpha_t / alpha_{t-1} for sampling stability."""
    a2 = np.concatenate([np.ones(1), alphas2], axis=0)
    step = np.clip(a2[1:] / a2[:-1], a_min=clip_value, a_max=1.)
    return np.cumprod(step, axis=0)


def polynomial_schedule(timesteps: int, s=1e-4, power=3.):
    """pharmacodiff.py:618-632: alpha^2 = (1 - (x/steps)^power)^2, clipped, then squeezed by s."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    a2 = clip_noise_schedule((1 - np.power(x / steps, power)) ** 2, clip_value=0.001)
    return (1 - 2 * s) * a2 + s


class PredefinedNoiseSchedule(nn.Module):
    """Lookup table gamma[t] = -(log alpha_t^2 - log sigma_t^2)  (pharmacodiff.py:636-668)."""

    def __init__(self, noise_schedule, timesteps, precision):
        super().__init__()
        self.timesteps = timesteps
        if noise_schedule == 'cosine':
            alphas2 = cosine_beta_schedule(timesteps)
        elif 'polynomial' in noise_schedule:
            splits = noise_schedule.split('_')
            assert len(splits) == 2
            alphas2 = polynomial_schedule(timesteps, s=precision, power=float(splits[1]))
        else:
            raise ValueError(noise_schedule)
        sigmas2 = 1 - alphas2
        self.gamma = nn.Parameter(torch.from_numpy(-(np.log(alphas2) - np.log(sigmas2))).float(), requires_grad=False)

    def forward(self, t):
        return self.gamma[torch.round(t * self.timesteps).long()]


def sigma(gamma):
    return torch.sqrt(torch.sigmoid(gamma))


def alpha(gamma):
    return torch.sqrt(torch.sigmoid(-gamma))


def sigma_and_alpha_t_given_s(gamma_t, gamma_s):
    sigma2_t_given_s = -torch.expm1(F.softplus(gamma_s) - F.softplus(gamma_t))
    log_alpha2_t = F.logsigmoid(-gamma_t)
    log_alpha2_s = F.logsigmoid(-gamma_s)
    alpha_t_given_s = torch.exp(0.5 * (log_alpha2_t - log_alpha2_s))
    alpha_s = torch.exp(0.5 * log_alpha2_s)
    return sigma2_t_given_s, torch.sqrt(sigma2_t_given_s), alpha_t_given_s, alpha_s


def step_coefficients(gamma_table: torch.Tensor, timesteps: int) -> Dict[str, torch.Tensor]:
    """Everything sample_p_zs_given_zt derives from (s, t=s+1) for all s at once, in fp32 on the host,
    with the reference's op order (pharmacodiff.py:387-400, 413-420).  Index = s."""
    g = gamma_table.detach().float().cpu()
    s_int = torch.arange(timesteps)
    s = s_int.float() / timesteps
    t = (s_int + 1).float() / timesteps
    g_s = g[torch.round(s * timesteps).long()]
    g_t = g[torch.round(t * timesteps).long()]
    s2_ts, s_ts, a_ts, a_s = sigma_and_alpha_t_given_s(g_t, g_s)
    sig_s, sig_t = sigma(g_s), sigma(g_t)
    return {"t": t, "s": s, "alpha_t_given_s": a_ts, "var_terms": s2_ts / a_ts / sig_t, "sigma": s_ts * sig_s / sig_t,
            "ep_zt": a_ts * (sig_s ** 2) / (sig_t ** 2), "ep_pred": a_s * s2_ts / (sig_t ** 2)}
