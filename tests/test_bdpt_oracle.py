"""CPU tests of the cpu_bdpt-estimator oracle (oracle/bdpt_oracle.cpp, restating
src/cpu_bdpt.cpp:30-488 and the CPU scene model of src/object.cpp)."""
import os

import numpy as np

from conftest import GOLDEN, rmse


def _input(sio, oracle_mod, raw_light_dirs):
    sc = sio.load_scene(os.path.join(GOLDEN, "scenes", "input.txt"))
    L, sp, tr = sio.flatten_for_pt(sc)
    if raw_light_dirs:                       # run_cpu_bdpt receives the lights as parsed (src/main_cli.cpp:131-140)
        for i, l in enumerate(sc.lights):
            L[i]["dir"] = l["dir"]
    return sc, L, sp, tr, sio.object_order(sc)


def test_advisory_replay_of_survey_cpu_bdpt_image(oracle_mod, sio):
    """ADVISORY pin.  survey_cpu_bdpt_input_256x256_4spp_spl8_rows0-63.f32 holds the first 64 rows
    of the image the survey stage got from the reference's own run_cpu_bdpt (input.txt, 256x256,
    4 spp, spl 8, depth 4/4, OMP_NUM_THREADS=1; built there with stand-in glm/CUDA headers, which
    this round may not do -- see tests/golden/README.md).  Replaying the reference's two
    std::mt19937 streams (1337 / 9999) on one thread must reproduce it bit for bit; the whole
    256x256 image was checked once the same way (max-abs 0, mean 0.16336238 = SURVEY Appendix C)."""
    sc, L, sp, tr, order = _input(sio, oracle_mod, raw_light_dirs=True)
    ref = np.fromfile(os.path.join(GOLDEN, "survey_cpu_bdpt_input_256x256_4spp_spl8_rows0-63.f32"), np.float32).reshape(64, 256, 3)
    rows = 32
    img, st = oracle_mod.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, 256, 256, 4, 4, 4, 8,
                                     rng_mode=1, threads=1, rows=rows)
    assert rmse(img[:rows], ref[:rows]) < 1e-5
    assert np.array_equal(img[:rows], ref[:rows])
    assert st["shadow_rays"] > 100 * st["samples"]          # ~200 connection shadow rays per sample


def test_counter_mode_is_thread_count_independent(oracle_mod, sio):
    sc, L, sp, tr, order = _input(sio, oracle_mod, raw_light_dirs=False)
    a, _ = oracle_mod.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, 32, 24, spp=2, spl=4, seed=5, threads=1)
    b, _ = oracle_mod.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, 32, 24, spp=2, spl=4, seed=5, threads=4)
    w, _ = oracle_mod.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, 32, 24, spp=2, spl=4, seed=5, window=(4, 2, 20, 18))
    assert np.array_equal(a, b)
    assert np.array_equal(w[2:18, 4:20], a[2:18, 4:20]) and not w[:2].any()
    assert np.isfinite(a).all() and a.min() >= 0 and a.mean() > 0.01


def test_bdpt_and_pt_are_different_estimators(oracle_mod, sio):
    # directly seen light balls are `illum` under cpu_bdpt and illum/(4 pi r^2) under the PT kernel
    sc, L, sp, tr, order = _input(sio, oracle_mod, raw_light_dirs=False)
    cam = sio.camera_for(sc, 64, 64, sc.fov)
    bd, _ = oracle_mod.bdpt_render(L, sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, 64, 64, spp=8, spl=8, seed=2)
    pt, _ = oracle_mod.pt_render(L, sp, tr, cam, 64, 64, 4, 8, seed=2)
    assert bd.max() <= 1.0 + 15 * 4 * 32 and pt.max() > 5.0
    assert rmse(bd, pt) > 0.5


def test_no_lights_returns_untouched_image(oracle_mod, sio):
    from path_tracing_amd.layouts import LIGHT
    sc, L, sp, tr, order = _input(sio, oracle_mod, raw_light_dirs=False)
    img, st = oracle_mod.bdpt_render(np.zeros(0, LIGHT), sp, tr, order, sc.eye, sc.look_at, sc.view_up, sc.fov, 8, 8, spp=1, spl=1)
    assert not img.any() and st["samples"] == 0              # src/cpu_bdpt.cpp:178
