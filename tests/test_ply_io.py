"""PLY layout of the reference's save_ply / load_ply (scene/gaussian_model.py:153-250)."""
import numpy as np
import torch

from splatfields_amd import ply_io


def test_attribute_order_matches_construct_list_of_attributes():
    # scene/gaussian_model.py:153-165 with max_sh_degree = 3: 3 dc + 45 rest
    names = ply_io.attribute_names(45)
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9] == "f_rest_0" and names[53] == "f_rest_44" and names[54] == "opacity"
    assert names[55:58] == ["scale_0", "scale_1", "scale_2"] and names[58:] == ["rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(names) == 62


def test_round_trip_and_activations(tmp_path):
    g = torch.Generator().manual_seed(0)
    n = 37
    xyz = torch.randn(n, 3, generator=g)
    dc, rest = torch.randn(n, 1, 3, generator=g), torch.randn(n, 15, 3, generator=g)
    op, sc, rot = torch.randn(n, 1, generator=g), torch.randn(n, 3, generator=g) - 3, torch.randn(n, 4, generator=g)
    path = str(tmp_path / "point_cloud.ply")
    ply_io.save_ply(path, xyz, dc, rest, op, sc, rot)
    head = open(path, "rb").read(200).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 37\nproperty float x\n")
    raw = ply_io.load_ply(path)
    np.testing.assert_array_equal(raw["xyz"], xyz.numpy())
    np.testing.assert_array_equal(raw["features_dc"], dc.numpy())
    np.testing.assert_array_equal(raw["features_rest"], rest.numpy())
    # channel-major on disk: f_rest_{c*15+k} = rest[:, k, c]  (features.transpose(1,2).flatten(1))
    data = np.frombuffer(open(path, "rb").read()[-n * 62 * 4:], dtype="<f4").reshape(n, 62)
    assert data[5, 9 + 1 * 15 + 4] == rest[5, 4, 1].item()
    inp = ply_io.load_rasterizer_inputs(path)
    assert inp["shs"].shape == (n, 16, 3) and torch.equal(inp["shs"][:, :1], dc) and torch.equal(inp["shs"][:, 1:], rest)
    assert torch.allclose(inp["opacities"], torch.sigmoid(op)) and torch.allclose(inp["scales"], torch.exp(sc))
    assert torch.allclose(inp["rotations"].norm(dim=1), torch.ones(n), atol=1e-6)
