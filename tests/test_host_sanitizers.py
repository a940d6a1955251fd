"""The host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: no GPU sanitizer exists on this pool).

deep-space-ray-tracer_amd/host/*.cpp -- OBJ / MTL / world loaders, PNM / PNG / JPEG / BMP / TGA decoders, PPM / PNG writers, the median and SAH builders, the
second-tree preparation, the pose reader -- has no HIP in it, so it links into a plain g++ program (tests/host_sanitize_driver.cpp) that drives it through the C ABI
over every committed asset and over damaged copies of them (truncations, flipped bytes).  Intact files must be accepted; damaged ones may be refused or decoded,
but no call may touch memory it does not own or execute undefined behaviour: any sanitizer report fails the test."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "deep-space-ray-tracer_amd", "host")
ASSETS = os.path.join(ROOT, "tests", "golden", "assets")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    out = tmp_path_factory.mktemp("san")
    exe = str(out / "host_sanitize_driver")
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp"))) + [os.path.join(ROOT, "tests", "host_sanitize_driver.cpp")]
    flags = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    objs = []
    procs = []
    for s in srcs:                                   # eight small translation units, compiled side by side
        o = str(out / (os.path.basename(s) + ".o"))
        objs.append(o)
        procs.append((s, subprocess.Popen(["g++", *flags, "-c", s, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        log, _ = p.communicate()
        assert p.returncode == 0, f"{s}:\n{log}"
    r = subprocess.run(["g++", *flags, *objs, "-o", exe, "-lz", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe, str(out)


def test_host_code_is_clean_under_asan_and_ubsan(driver):
    exe, scratch = driver
    images = []
    for p in sorted(glob.glob(os.path.join(ASSETS, "images", "*")) + glob.glob(os.path.join(ASSETS, "jpeg", "*")) + [os.path.join(ASSETS, "checker.ppm"), os.path.join(ASSETS, "stripes.png")]):
        name = os.path.basename(p)
        if not os.path.isfile(p) or name.endswith((".json", ".txt", ".py")):
            continue
        images.append(p if "refused" in name or "unsupported" in name or "corrupt" in name or "truncated" in name else "+" + p)
    assert len(images) > 60
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, ASSETS, scratch, *images], capture_output=True, text=True, env=env, timeout=900, cwd=ASSETS)       # (the world files name their OBJ files relative to the working directory)
    tail = (r.stdout + r.stderr)[-6000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, tail
    assert r.returncode == 0 and "host sanitize driver: ok" in r.stdout, tail
