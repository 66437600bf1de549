"""Host check of the quad-reach geometry of the backward blend (splatfields_amd/csrc/quadmask.h): compiled for the CPU
with g++ and compared with a brute-force evaluation of the kernels' per-pixel alpha >= 1/255 test (tests/native/
quadmask_check.cpp).  A quad the mask leaves out although one of its pixels blends would silently drop gradient terms."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quad_mask_is_a_tight_superset_of_the_blended_pixels(tmp_path):
    exe = tmp_path / "quadmask_check"
    subprocess.run(["g++", "-O2", "-o", str(exe), os.path.join(ROOT, "tests", "native", "quadmask_check.cpp")], check=True)
    out = subprocess.run([str(exe), "200000"], check=True, capture_output=True, text=True).stdout.split()
    cases, missed, mask_quads, true_quads, nonempty = map(int, out)
    assert cases == 200000 and missed == 0
    assert nonempty > 0.5 * cases                      # the random splats do reach the tile
    assert mask_quads <= 1.05 * true_quads             # conservative, but by no more than 5 % of the quads


def test_reach_mask_packing_of_the_tile_rectangles(tmp_path):
    """Geom::rect carries the small rectangles' reach mask in the top nibbles of its four 12-bit tile coordinates
    (csrc/common.h, NOTEBOOK.md section 4 item 29): host build of the three helpers, round trip over a million random cases."""
    exe = tmp_path / "rectmask_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-value", "-o", str(exe),
                    os.path.join(ROOT, "tests", "native", "rectmask_check.hip")], check=True, capture_output=True)
    cases, bad = map(int, subprocess.run([str(exe), "1000000"], check=True, capture_output=True, text=True).stdout.split())
    assert cases == 1000000 and bad == 0
