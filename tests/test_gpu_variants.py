"""Forced-failure builds of the HIP library (rust-raytracer_amd/csrc/Makefile) against the oracle: paths that practically never run
in the product build.  Each build is loaded by a child process (the library is chosen at import time, RTX_LIB)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {
    # traverse_bvh / traverse_bvh_dealt with a 3-entry stack: most closest-hit rays and many feelers overflow it and are answered by the all-f64 walks
    # (k_closest_retry, the shadow retry list of the any-hit walk and of the two-phase product; the build's retry lists hold 2^14
    # feelers instead of 2^20, so a 1.5 M-sample frame overflows them and is rendered again with exact shadows); a 1-entry list of distinct hit prims: phase 2 after every hit, and a second
    # hit of a feeler in one round waits in the pair list and is tested again
    "librtx_smallstack.so": "-DRTX_BVH_STACK=3 -DRTX_BVH_HITS=1 -DRTX_RETRY_CAP=16384u",
    # k_shadow_coop: every 7th tentative hit fails its deferred f64 entry-bbox confirmation and the feeler goes to the retry list
    "librtx_failconfirm.so": "-DRTX_DEBUG_FAIL_CONFIRM=7",
}


@pytest.mark.gpu
@pytest.mark.parametrize("lib", sorted(VARIANTS))
def test_forced_failure_build_matches_the_oracle(lib):
    path = os.path.join(ROOT, "rust-raytracer_amd", lib)
    assert os.path.exists(path), "%s is missing: make -C rust-raytracer_amd/csrc (flags: %s)" % (lib, VARIANTS[lib])
    env = dict(os.environ, RTX_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_check.py")], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-2000:], r.stderr[-2000:])


def test_variant_builds_are_declared_in_the_makefile():
    mk = open(os.path.join(ROOT, "rust-raytracer_amd", "csrc", "Makefile")).read()
    for lib, flags in VARIANTS.items():
        assert lib in mk and flags in mk, lib
