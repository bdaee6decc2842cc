"""The synthetic asset generator must be bit-reproducible (goldens record its fingerprint)."""
import numpy as np

from keypoints2body_amd import synthetic
from tests import helpers as H


def test_body_model_shapes_and_fingerprint():
    c = H.body_consts()
    assert c.v_template.shape == (6890, 3) and c.shapedirs.shape == (6890, 3, 10)
    assert c.posedirs.shape == (207, 20670) and c.J_regressor.shape == (24, 6890)
    assert c.lbs_weights.shape == (6890, 24) and c.extra_vertex_ids.shape == (21,)
    assert list(c.parents) == [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
    assert c.fingerprint() == int(H.load_case("amass_zero_init")["model_fingerprint"])
    assert np.allclose(c.lbs_weights.sum(1), 1.0, atol=1e-6) and np.allclose(c.J_regressor.sum(1), 1.0, atol=1e-5)
    assert (c.lbs_weights > 0).sum(1).max() <= 4


def test_generator_is_seeded():
    a, b = synthetic.make_poses(5, seed=1), synthetic.make_poses(5, seed=1)
    assert np.array_equal(a.body_pose, b.body_pose)
    assert not np.array_equal(a.body_pose, synthetic.make_poses(5, seed=2).body_pose)
    n = synthetic.normalish(9, (20000,))
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1.0) < 0.03
