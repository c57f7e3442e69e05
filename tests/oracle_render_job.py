"""One worker of conftest.oracle_render_parallel: renders the rays of <dir>/job<i>.npz with the oracle (torch-CPU ops, <threads>
threads) and writes what the census reads to <dir>/out<i>.npz.  TEST INFRASTRUCTURE (the oracle is the checker, never the product)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as O  # noqa: E402


def main():
    tmp, i, threads = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    nets = np.load(os.path.join(tmp, "nets.npz"))
    sd_c = {k[2:]: nets[k] for k in nets.files if k.startswith("c.")}
    sd_f = {k[2:]: nets[k] for k in nets.files if k.startswith("f.")}
    j = np.load(os.path.join(tmp, "job%d.npz" % i))
    O.set_backend("torch")
    torch.set_num_threads(threads)
    ref = O.render(sd_c, sd_f, 400, 400, j["K"].tolist(), rays=(j["ro"], j["rd"]), near=float(j["near"]), far=float(j["far"]), chunk=8192,
                   extras=True)
    ref = {k: v for k, v in ref.items() if k not in ("raw", "weights", "cdf")}       # 3 KB per ray the census does not read
    ref["sigma0_last"] = ref.pop("raw0")[:, -1, 3].copy()                            # what census_ref keeps of the coarse raw
    np.savez(os.path.join(tmp, "out%d.npz" % i), **ref)


if __name__ == "__main__":
    main()
