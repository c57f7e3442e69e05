"""pytest configuration: `gpu` marker + shared fixtures (golden vectors, oracle import)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def oracle():
    import nerf_oracle
    return nerf_oracle


@pytest.fixture(scope="session")
def synth_nets(oracle):
    """(coarse, fine) synthetic state dicts for the seed the golden files were generated with."""
    seed = int(load_golden("g3_mlp")["seed"])
    sd_c = oracle.synth_weights(seed)
    return sd_c, oracle.synth_weights(seed + 1000, fine_of=sd_c)


def trained_pair(g):
    """(coarse, fine) state dicts of tests/golden/g26_trained.npz: a pair TRAINED by the reference's own code (oracle/train_g26.py)"""
    sd_c = {k[2:]: np.asarray(g[k], np.float32) for k in g.files if k.startswith("c.")}
    sd_f = {k[2:]: np.asarray(g[k], np.float32) for k in g.files if k.startswith("f.")}
    return sd_c, sd_f


def assert_close(a, b, atol=0.0, rtol=0.0, what=""):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    with np.errstate(invalid="ignore"):
        err = np.abs(a.astype(np.float64) - b.astype(np.float64))
        ok = both_nan | (err <= atol + rtol * np.abs(b.astype(np.float64)))
    assert ok.all(), "%s: %d/%d out of tolerance, max abs err %.3e" % (
        what, (~ok).sum(), ok.size, np.nanmax(np.where(both_nan, 0, err)))


def census_ref(g):
    """The reference's own end-to-end quantities held by a golden file (g6 / g11 / g13) in the form oracle/census.py
    takes as `ref`: pixels, the sigma of the last coarse sample, the coarse weights its sample_pdf saw, its indices and
    samples -- every one produced by running the reference (oracle/gen_golden.py)."""
    flat = lambda a, t: np.asarray(a).reshape((-1,) + np.asarray(a).shape[np.asarray(a).ndim - t:])
    return dict(rgb_map=flat(g["rgb"], 1), acc_map=flat(g["acc"], 0), disp_map=flat(g["disp"], 0),
                rgb0=flat(g["rgb0"], 1), acc0=flat(g["acc0"], 0), sigma0_last=g["sigma0_last"],
                pdf_weights=g["pdf_weights"], inds=g["inds"], z_samples=g["z_samples"])
