"""Drop-in module: put `<repo>` and `<repo>/neural_sim_nerf_amd/dropin` in front of the reference's
`optimization/` on PYTHONPATH and `from utils.run_nerf_noscale import *` (neural_sim_main.py:35) binds the
MI355X-native implementation.  `utils` is a namespace package on both sides (no __init__.py), so every other
`utils.*` module (load_LINEMOD_noscale, gumble, the vendored detectron2 pieces) still comes from the reference."""
from neural_sim_nerf_amd.run_nerf_noscale import *  # noqa: F401,F403
from neural_sim_nerf_amd.run_nerf_noscale import (create_nerf, render, render_path, render_path_grad, run_network,  # noqa: F401
                                                  batchify, to8b, device, NeRF, get_embedder, get_rays, sample_pdf,
                                                  img2mse, mse2psnr, np, torch, os, time)
