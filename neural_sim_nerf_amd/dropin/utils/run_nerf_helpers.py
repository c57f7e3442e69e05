"""Drop-in for the reference's utils/run_nerf_helpers.py; see run_nerf_noscale.py next to this file."""
from neural_sim_nerf_amd.run_nerf_helpers import *  # noqa: F401,F403
from neural_sim_nerf_amd.run_nerf_helpers import (NeRF, Embedder, get_embedder, get_rays, ndc_rays, sample_pdf,  # noqa: F401
                                                  to8b, img2mse, mse2psnr, np, torch, nn)
