"""`python -m src.mnist --train|--sample` — alias of tinydiffusionmodels_amd.mnist."""
from tinydiffusionmodels_amd.mnist import *  # noqa: F401,F403
from tinydiffusionmodels_amd.mnist import main, timesteps, betas, alphas, alphas_cumprod  # noqa: F401
from tinydiffusionmodels_amd.mnist import sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod  # noqa: F401

if __name__ == "__main__":
    main()
