"""CPU: the synthetic sweep generator (r-pcc_amd/synth.py).  The committed golden fixtures were made from its default scene, so the
generator must keep producing exactly those points -- whatever options it grows (round 4: the adversarial scenes of bench.py --scene)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rpcc_amd  # noqa: E402,F401
from rpcc_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


@pytest.mark.parametrize("case,geom,fid", [("synth_vlp16", "VelodyneVLP16", 2), ("synth_32E", "Velodyne32E", 1)])
def test_default_scene_reproduces_the_golden_input(case, geom, fid):
    g = orc.GEOMS[geom]
    want = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))["xyz"]
    got = synth.make_frame(fid, g["H"], g["W"], vmax_deg=g["vmax_deg"], vmin_deg=g["vmin_deg"]).numpy()
    assert got.dtype == np.float32 and got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("scene", [s for s in synth.SCENES if s != "default"])
def test_adversarial_scenes_are_what_they_say(scene):
    """shell: every return at 30 m; noise: ranges spread over 2 .. 80 m with no spatial coherence; corridor: nothing farther
    than 1.5 m to the side.  Deterministic per frame id, different between ids."""
    a = synth.make_frame(7, 16, 1800, vmax_deg=15.0, vmin_deg=-15.0, scene=scene).numpy()
    b = synth.make_frame(7, 16, 1800, vmax_deg=15.0, vmin_deg=-15.0, scene=scene).numpy()
    c = synth.make_frame(8, 16, 1800, vmax_deg=15.0, vmin_deg=-15.0, scene=scene).numpy()
    assert np.array_equal(a, b) and a.shape != c.shape or not np.array_equal(a[:100], c[:100])
    r = np.linalg.norm(a.astype(np.float64), axis=1)
    assert a.shape[0] > 0.8 * 16 * 1800 and np.isfinite(a).all()
    if scene == "shell":
        assert abs(r - 30.0).max() < 0.1
    elif scene == "noise":
        assert r.min() < 3.0 and r.max() > 79.0 and 35.0 < r.mean() < 47.0
    else:
        assert np.abs(a[:, 1]).max() < 1.6 and r.max() > 20.0
