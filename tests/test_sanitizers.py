"""AddressSanitizer + UndefinedBehaviorSanitizer over the plain CPU code on either side of the HIP path -- the C++20 host loader
(host/scene_loader.cpp, host/image_io.cpp) and the C oracle (oracle/pt_oracle.c, test infrastructure) -- SURVEY.md section 5 lists such builds
among the reference-side hygiene a rebuild should have.  tests/san/san_main.cpp drives them: the reference's scene whole and in 256-byte chunks,
malformed and ragged OBJ texts, both closest-hit modes on 4096 rays (null and axis-parallel directions among them), small frames in every mode
(LBVH, brute force, rectangle, NEE, instances).  No GPU involved."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_loader_and_oracle_under_asan_and_ubsan(tmp_path):
    gxx, gcc = shutil.which("g++"), shutil.which("gcc")
    if not gxx or not gcc:
        pytest.skip("no gcc / g++")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    orc_o = tmp_path / "pt_oracle.o"
    subprocess.check_call([gcc, "-std=c11", "-ffp-contract=off", *san, "-c", os.path.join(REPO, "oracle", "pt_oracle.c"), "-o", str(orc_o)])
    exe = tmp_path / "san_main"
    host = os.path.join(REPO, "single-file-vulkan-pathtracing_amd", "host")
    subprocess.check_call([gxx, "-std=c++20", "-pthread", *san, os.path.join(REPO, "tests", "san", "san_main.cpp"), os.path.join(host, "scene_loader.cpp"),
                           os.path.join(host, "image_io.cpp"), str(orc_o), "-lm", "-o", str(exe)])
    scratch = tmp_path / "objs"
    scratch.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([str(exe), os.path.join(REPO, "assets", "CornellBox-Original.obj"), str(scratch)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "0 check(s) failed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout, r.stderr[-4000:])
