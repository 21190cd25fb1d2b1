"""ADVISORY check against the only render the reference tree itself holds: /root/reference/output.png, a 200 x 200
8-bit image of input.txt in BDPT mode with an unseeded generator and unknown sample counts (SURVEY F10, Appendix C:
8-bit RMSE 20.2 against a 16-spp cpu_bdpt render, 58.5 against the PT estimator).  It is committed as an array
(tests/golden/reference_output_png_200x200_rgb8.npz, read from the PNG with PIL in the build container; data, not code).
It cannot pin anything bit for bit -- different random streams, and the file came from the reference's CUDA BDPT
kernel or its CPU renderer, which differ from each other -- so the test is statistical: the tone-mapped GPU
render of the cpu_bdpt estimator at high spp agrees with it at the level the survey measured, block means agree far
more tightly than single pixels, and the PT estimator (a different estimator) is clearly further away.
This is the one check in the suite whose expected values were not produced by this repository."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _rmse8(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt((d * d).mean()))


def _block_means(img, n=8):
    H, W, _ = img.shape
    return np.asarray(img, np.float64)[:H // n * n, :W // n * n].reshape(n, H // n, n, W // n, 3).mean(axis=(1, 3))


def test_bdpt_render_agrees_with_the_reference_output_png(hpt, sio):
    ref8 = np.load(os.path.join(GOLDEN, "reference_output_png_200x200_rgb8.npz"))["rgb8"]
    assert ref8.shape == (200, 200, 3)
    sc = sio.load_scene(os.path.join(GOLDEN, "scenes", "input.txt"))
    assert tuple(sc.resolution) == (200, 200)
    L, sp, tr = sio.flatten_for_pt(sc)
    cam = sio.camera_for(sc, 200, 200, 50.0)              # the CLI renders with fov 50 (src/main_cli.cpp:158)
    # run_cpu_bdpt traces its nl * spl light subpaths once per frame and connects every eye vertex of every pixel to them,
    # so one render carries image-wide, heavy-tailed noise that more samples per pixel do not remove (measured here: two
    # single renders at spl 8 differ by 25 8-bit RMSE whatever the spp, at spl 64 by 9).  The expectation is approached
    # with spl 64 and the mean of the tone-mapped images of renders with different seeds (two disjoint sets of 12).
    def mean_tonemapped(scene, seeds):
        acc = np.zeros((200, 200, 3), np.float64)
        for s in seeds:
            acc += hpt.tonemap(scene.render_bdpt(cam, 200, 200, 4, 4, 16, 64, hpt.make_params(seed=s)))
        return acc / len(seeds)

    with hpt.Scene(L, sp, tr) as scene:
        scene.set_groups(*sio.object_order(sc))
        b8 = mean_tonemapped(scene, range(100, 112))
        b8b = mean_tonemapped(scene, range(200, 212))
        pt = scene.render_pt(cam, 200, 200, 4, 256, hpt.make_params(seed=20))
    p8 = hpt.tonemap(pt)
    e_bdpt, e_pt, e_self = _rmse8(b8, ref8), _rmse8(p8, ref8), _rmse8(b8, b8b)
    blocks = np.abs(_block_means(b8) - _block_means(ref8))
    print("8-bit RMSE vs output.png: bdpt %.2f, pt %.2f; two disjoint bdpt averages %.2f; block means: mean |d| %.2f max %.2f"
          % (e_bdpt, e_pt, e_self, blocks.mean(), blocks.max()))
    # calibration run (24 renders per set): 12.7 / 52.1 / 2.7 / 5.1 (max 22.6)
    assert e_self < 6.0                                     # our side is converged below the comparison's noise
    assert e_bdpt < 18.0                                    # the survey measured 20.2 with one 16-spp render on our side
    assert e_pt > 2.5 * e_bdpt                              # attribution: output.png is a BDPT-estimator image, not PT
    assert blocks.mean() < 8.0                              # 25 x 25-pixel block means, 8-bit units
