"""Noise schedule tables (src/mnist.py:23-33, duplicated at src/shakespeare.py:25-35).

Computed with the same torch CPU calls as the reference, so they are
bit-identical to its module globals, then uploaded once per device together
with the three per-step coefficients p_sample derives from them
(src/mnist.py:169-171,179), computed with p_sample's own op order."""
from typing import Dict

import torch

TIMESTEPS = 1000


def linear_beta_schedule(timesteps: int, start=1e-4, end=2e-2) -> torch.Tensor:
    """Linear schedule from Ho et al. 2020 (src/mnist.py:23-25)."""
    return torch.linspace(start, end, timesteps)


def make_tables(timesteps: int = TIMESTEPS) -> Dict[str, torch.Tensor]:
    betas = linear_beta_schedule(timesteps)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    sqrt_alphas_cumprod = torch.sqrt(alphas_cumprod)
    sqrt_one_minus_alphas_cumprod = torch.sqrt(1.0 - alphas_cumprod)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": alphas_cumprod,
        "sqrt_alphas_cumprod": sqrt_alphas_cumprod,
        "sqrt_one_minus_alphas_cumprod": sqrt_one_minus_alphas_cumprod,
        # p_sample's per-step scalars, same fp32 ops as src/mnist.py:171,174,179
        "sqrt_recip_alphas": 1.0 / torch.sqrt(alphas),
        "eps_coef": betas / sqrt_one_minus_alphas_cumprod,
        "sigma": torch.sqrt(betas),
    }


BASE_KEYS = ("betas", "alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")
DERIVED_KEYS = ("sqrt_recip_alphas", "eps_coef", "sigma")

_CPU_TABLES = None
_DEV_TABLES: Dict[str, Dict[str, torch.Tensor]] = {}
_GENERATION = 0


def schedule_generation() -> int:
    """Bumped by set_tables(): anything that captured device tables (hipGraph samplers) keys on it."""
    return _GENERATION


def cpu_tables() -> Dict[str, torch.Tensor]:
    global _CPU_TABLES
    if _CPU_TABLES is None:
        _CPU_TABLES = make_tables()
    return _CPU_TABLES


def device_tables(device) -> Dict[str, torch.Tensor]:
    """Device-resident copies (the reference moves its globals to the device
    only inside __main__, src/mnist.py:228-231; here each device gets its own)."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = str(dev)
    if key not in _DEV_TABLES:
        _DEV_TABLES[key] = {k: v.to(dev).contiguous() for k, v in cpu_tables().items()}
    return _DEV_TABLES[key]


def set_tables(tables: Dict[str, torch.Tensor] = None) -> None:
    """Pin the schedule to externally supplied tables (None = recompute locally).

    Why this exists: `torch.sqrt` on CPU goes through MKL, whose result differs
    by 1 ulp between hosts (measured: Intel Xeon vs AMD EPYC 9575F, 179 of 1000
    entries of sqrt_alphas_cumprod), so the reference's module-level tables are
    host-dependent.  To reproduce another host's run bit-for-bit (or a golden
    fixture), install that host's tables here.  The five base tables are
    required; the three derived ones are recomputed from them when absent.
    Updates the CPU tensors in place (module globals that alias them follow)
    and drops the per-device copies."""
    cur = cpu_tables()
    new = make_tables() if tables is None else {k: torch.as_tensor(v).detach().clone().float() for k, v in tables.items()}
    if tables is not None:
        missing = [k for k in BASE_KEYS if k not in new]
        if missing:
            raise ValueError(f"set_tables: missing {missing}")
        if "sqrt_recip_alphas" not in new:
            new["sqrt_recip_alphas"] = 1.0 / torch.sqrt(new["alphas"])
        if "eps_coef" not in new:
            new["eps_coef"] = new["betas"] / new["sqrt_one_minus_alphas_cumprod"]
        if "sigma" not in new:
            new["sigma"] = torch.sqrt(new["betas"])
    for k in BASE_KEYS + DERIVED_KEYS:
        if new[k].shape != cur[k].shape:
            raise ValueError(f"set_tables: {k} has shape {tuple(new[k].shape)}, expected {tuple(cur[k].shape)}")
        cur[k].copy_(new[k])
    _DEV_TABLES.clear()
    global _GENERATION
    _GENERATION += 1
