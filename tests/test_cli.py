"""The `rtx` command line (host/main.cpp = src/main.rs:63-175 on top of the C ABI): the reference's JSON job
schema in, P3 PPM files out (util/export.rs:5-18, animator.rs:33-39 frame naming)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTX = os.path.join(ROOT, "rust-raytracer_amd", "rtx")

# the reference's sample-config.json (values verbatim), reduced in size by the tests below
SAMPLE = {"name": "cornell", "size": [512, 512], "fov": 30.0, "reflect_depth": 3, "refract_depth": 6, "shadow_samples": 64,
          "gloss_samples": 8, "pixel_samples": 2, "output_file": "test", "animating": False, "fps": 25.0,
          "time_slice": [0.0, 10.0], "starting_frame_number": 0}


def _run(tmp_path, cfg, expect_ok=True):
    path = tmp_path / "job.json"
    path.write_text(json.dumps(cfg))
    p = subprocess.run([RTX, str(path)], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    if expect_ok:
        assert p.returncode == 0, p.stderr + p.stdout
    return p


def _read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int64).reshape(h, w, 3)


def test_cli_usage_and_config_errors(tmp_path):
    """main.rs:51-61, 66-111: usage line, unreadable file, a missing field and an unknown scene all exit with 1."""
    assert os.path.exists(RTX), "run __graft_entry__.build()"
    p = subprocess.run([RTX], capture_output=True, text=True)
    assert p.returncode == 1 and "Usage" in p.stderr
    p = subprocess.run([RTX, str(tmp_path / "nope.json")], capture_output=True, text=True)
    assert p.returncode == 1
    bad = dict(SAMPLE)
    del bad["fov"]
    p = _run(tmp_path, bad, expect_ok=False)
    assert p.returncode == 1 and "fov" in p.stderr
    p = _run(tmp_path, dict(SAMPLE, name="box"), expect_ok=False)     # tools/conf/box.json: no longer in the registry
    assert p.returncode == 1 and "box" in p.stderr


def test_cli_fails_loudly_without_a_device(tmp_path):
    """No CPU fallback: on a host without a HIP device the render must fail with an error, not produce an image."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this host has a GPU")
    p = _run(tmp_path, dict(SAMPLE, size=[64, 64], shadow_samples=1, gloss_samples=1, pixel_samples=1), expect_ok=False)
    assert p.returncode == 1 and not (tmp_path / "test.ppm").exists()


@pytest.mark.gpu
def test_cli_still_frame_matches_the_library(rtx, tmp_path):
    """sample-config.json's scene through the CLI = the same frame through the Python binding of the C ABI."""
    cfg = dict(SAMPLE, size=[96, 64], shadow_samples=4, gloss_samples=2, seed=5)
    p = _run(tmp_path, cfg)
    assert "Render done" in p.stdout
    img = _read_ppm(tmp_path / "test.ppm")
    named = rtx.host.NamedScene("cornell")
    sc = rtx.Scene(named)
    rgba, _, st = sc.render(named.camera(96, 64, 30.0),
                            rtx.options(reflect_depth=3, refract_depth=6, shadow_samples=4, gloss_samples=2, pixel_samples=2,
                                        rng_mode=1, seed=5), want_ids=False)
    sc.close()
    assert img.shape == (64, 96, 3)
    d = np.abs(img - rgba[..., :3].astype(np.int64))
    assert (d <= 1).all() and (d != 0).mean() < 1e-3      # identical up to the order of f64 atomic adds


@pytest.mark.gpu
def test_cli_animation_writes_numbered_frames(rtx, tmp_path):
    """animator.rs:21-56: floor(fps * (to - from)) frames named {output_file}{:06}.ppm from starting_frame_number,
    frame k at time from + k / fps; the first frame of easing's animation camera is its still camera."""
    cfg = dict(SAMPLE, name="easing", size=[64, 40], shadow_samples=1, gloss_samples=1, pixel_samples=1, animating=True,
               fps=2.0, time_slice=[0.0, 1.5], starting_frame_number=7, output_file="anim", rng_mode="center")
    _run(tmp_path, cfg)
    frames = sorted(f for f in os.listdir(tmp_path) if f.startswith("anim"))
    assert frames == ["anim000007.ppm", "anim000008.ppm", "anim000009.ppm"]
    first = _read_ppm(tmp_path / frames[0])
    named = rtx.host.NamedScene("easing")
    sc = rtx.Scene(named)
    rgba, _, _ = sc.render(named.camera(64, 40, 30.0, time=0.0),
                           rtx.options(reflect_depth=3, refract_depth=6, shadow_samples=1, gloss_samples=1, pixel_samples=1, rng_mode=0),
                           want_ids=False)
    sc.close()
    assert (np.abs(first - rgba[..., :3].astype(np.int64)) <= 1).all()
    assert not np.array_equal(first, _read_ppm(tmp_path / frames[2]))   # the camera has moved


@pytest.mark.gpu
def test_cli_sphere_job_at_config_1_against_the_oracle(rtx, orc, tmp_path):
    """BASELINE config 1 exactly as stated (main.rs:34-49 schema): the sphere scene at 512x512, 1 spp, 1 shadow sample, depth 4,
    written in the reference's JSON job format, through the `rtx` CLI; the PPM it writes against the oracle's frame
    (deterministic mode: every pixel within one 8-bit step on >= 99.9 %)."""
    cfg = dict(SAMPLE, name="sphere", size=[512, 512], fov=30.0, reflect_depth=4, refract_depth=4, shadow_samples=1,
               gloss_samples=1, pixel_samples=1, output_file="sphere", rng_mode="center")
    p = _run(tmp_path, cfg)
    assert "Render done" in p.stdout
    img = _read_ppm(tmp_path / "sphere.ppm")
    assert img.shape == (512, 512, 3)
    named = rtx.host.NamedScene("sphere")
    opt = rtx.options(reflect_depth=4, refract_depth=4, shadow_samples=1, gloss_samples=1, pixel_samples=1, rng_mode=0)
    orgba, _, _ = orc.Scene(named).render(named.camera(512, 512, 30.0), opt, threads=16)
    d = np.abs(img - orgba[..., :3].astype(np.int64)).max(axis=-1)
    assert (d <= 1).mean() >= 0.999, float((d <= 1).mean())
    assert (d == 0).mean() >= 0.99
