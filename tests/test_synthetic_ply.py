"""CPU tests of the host-side helpers: synthetic generators and the PLY subset."""
import numpy as np


def test_uniform_cloud_is_deterministic_and_in_range(pkg):
    a = pkg.synthetic.uniform_cloud(10000, 42)
    b = pkg.synthetic.uniform_cloud(10000, 42)
    assert a.dtype == np.float32 and a.shape == (10000, 3)
    assert np.array_equal(a, b)
    assert a.min() >= 0.0 and a.max() < 1.0
    assert not np.array_equal(a, pkg.synthetic.uniform_cloud(10000, 43))
    # every value is a multiple of 2^-24
    assert np.array_equal(a * np.float32(2 ** 24), np.floor(a * np.float32(2 ** 24)))


def test_clustered_cloud(pkg):
    c = pkg.synthetic.clustered_cloud(20000, seed=44)
    assert c.shape == (20000, 3) and c.min() >= 0.0 and c.max() < 1.0
    assert np.array_equal(c, pkg.synthetic.clustered_cloud(20000, seed=44))
    # clustered: far denser than uniform at small scale
    h, _ = np.histogramdd(c, bins=16, range=[(0, 1)] * 3)
    assert h.max() > 20 * 20000 / 16 ** 3


def test_ply_roundtrip(pkg, tmp_path):
    rng = np.random.default_rng(3)
    p = rng.random((100, 3), dtype=np.float32)
    n = rng.random((100, 3), dtype=np.float32)
    for fmt in ("ascii", "binary_little_endian", "binary_big_endian"):
        f = tmp_path / (fmt + ".ply")
        pkg.ply.write_ply(str(f), p, n, fmt)
        rp, rn = pkg.ply.read_ply(str(f))
        assert np.allclose(rp, p, atol=1e-6) and np.allclose(rn, n, atol=1e-6)
        if fmt != "ascii":
            assert np.array_equal(rp, p)
    rp, rn = pkg.ply.read_ply(str(tmp_path / "missing.ply"))
    assert rp.shape == (0, 3) and rn.shape == (0, 3)


def test_bunny_fixture(bunny):
    assert bunny.dtype == np.float32 and np.isfinite(bunny).all()
