"""The raw side channel of the stage scripts' file I/O (stages/stage_io.py) -- host logic, no GPU."""
import importlib.util
import os
import pickle
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(raw: bool):
    if raw:
        os.environ["ORIP_RAW_NPY"] = "1"
    else:
        os.environ.pop("ORIP_RAW_NPY", None)
    spec = importlib.util.spec_from_file_location("stage_io_t", os.path.join(ROOT, "omnirevolve-image-processor_amd", "stages", "stage_io.py"))
    m = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(m)
    finally:
        os.environ.pop("ORIP_RAW_NPY", None)
    return m


def test_png_only_by_default(tmp_path):
    io = _load(False)
    g = (np.arange(35 * 20).reshape(35, 20) % 251).astype(np.uint8)
    io.write_png(str(tmp_path / "g.png"), g)
    assert not (tmp_path / "g.png.npy").exists()
    assert np.array_equal(io.read_gray(str(tmp_path / "g.png")), g)


def test_raw_rasters_and_lists(tmp_path):
    io = _load(True)
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, (40, 30), dtype=np.uint8)
    c = rng.integers(0, 256, (40, 30, 3), dtype=np.uint8)
    io.write_png(str(tmp_path / "g.png"), g); io.write_png(str(tmp_path / "c.png"), c)
    assert np.array_equal(np.load(tmp_path / "g.png.npy"), g) and np.array_equal(np.load(tmp_path / "c.png.npy"), c)
    assert np.array_equal(io.read_gray(str(tmp_path / "g.png")), g) and np.array_equal(io.read_bgr(str(tmp_path / "c.png")), c)
    os.remove(tmp_path / "g.png"); os.remove(tmp_path / "c.png")                 # the raw files alone are enough
    assert io.exists(str(tmp_path / "g.png"))
    assert np.array_equal(io.read_gray(str(tmp_path / "g.png")), g) and np.array_equal(io.read_bgr(str(tmp_path / "c.png")), c)

    polys = [rng.integers(0, 1000, (n, 1, 2)).astype(np.int32) for n in (2, 7, 1, 30)]
    io.save_pickle(str(tmp_path / "p.pkl"), polys)
    with open(tmp_path / "p.pkl", "rb") as fh:                                   # the reference's own format is still written
        ref = pickle.load(fh)
    assert all(np.array_equal(a, b) for a, b in zip(ref, polys))
    os.remove(tmp_path / "p.pkl")
    got = io.load_pickle(str(tmp_path / "p.pkl"))
    assert len(got) == len(polys) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got, polys))
    io.save_pickle(str(tmp_path / "e.pkl"), [])
    os.remove(tmp_path / "e.pkl")
    assert io.load_pickle(str(tmp_path / "e.pkl")) == []
    io.save_pickle(str(tmp_path / "t.pkl"), [(1, 2), (3, 4)])                   # tap lists / op dicts stay pickles only
    assert not (tmp_path / "t.pkl.npz").exists() and io.load_pickle(str(tmp_path / "t.pkl")) == [(1, 2), (3, 4)]


def test_stale_raw_file_is_ignored(tmp_path):
    """a reference stage (or anything else) rewrote the PNG after the raw file: the PNG wins"""
    io = _load(True)
    a = np.full((8, 8), 10, np.uint8); b = np.full((8, 8), 200, np.uint8)
    io.write_png(str(tmp_path / "m.png"), a)
    t = time.time() + 5
    io_plain = _load(False)
    io_plain.write_png(str(tmp_path / "m.png"), b)
    os.utime(tmp_path / "m.png", (t, t))
    assert np.array_equal(io.read_gray(str(tmp_path / "m.png")), b)
