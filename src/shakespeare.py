"""`src.shakespeare` — alias of tinydiffusionmodels_amd.shakespeare (denoiser path native on MI355X)."""
from tinydiffusionmodels_amd.shakespeare import *  # noqa: F401,F403
from tinydiffusionmodels_amd.shakespeare import T, betas, alphas, alphas_cumprod  # noqa: F401
from tinydiffusionmodels_amd.shakespeare import sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod  # noqa: F401


if __name__ == "__main__":
    from tinydiffusionmodels_amd.shakespeare import main
    main()
