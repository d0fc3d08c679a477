"""-m gpu: the reference's OWN clients, compiled unmodified from /root/reference against this
repository's libedlib.so (tools/build_ref_clients.sh -> oracle/_ref/, prebuilt, travels):
test/runTests.cpp (600 random differential tests vs its O(mn) DP + 19 specific tests,
SURVEY.md §4), apps/aligner (meson `aligner` test: NW score 17) and apps/hello-world."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def _need(name):
    path = os.path.join(REFDIR, name)
    if not os.path.exists(path):
        pytest.skip("%s was not prebuilt (needs /root/reference at build time)" % name)
    return path


def test_hello_world():
    out = subprocess.run([_need("hello_amd")], capture_output=True, text=True, timeout=120)
    assert "edit_distance('hello', 'world!') = 5" in out.stdout, out.stdout + out.stderr


def test_reference_test_driver_passes_against_this_library():
    out = subprocess.run([_need("runTests_amd")], capture_output=True, text=True, timeout=1500)
    txt = re.sub(r"\x1b\[[0-9;]*m", "", out.stdout)
    assert out.returncode == 0, txt[-2000:] + out.stderr[-500:]
    assert len(re.findall(r"100/100 random tests passed", txt)) == 6, txt[-2000:]
    assert "All specific tests passed" in txt


def test_reference_cli_on_its_test_data():
    exe = _need("aligner_amd")
    out = subprocess.run([exe, "-m", "NW", os.path.join(REFDIR, "aligner_query.fasta"),
                          os.path.join(REFDIR, "aligner_target.fasta")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.search(r"#0: 17\b", out.stdout), out.stdout       # NW score 17
