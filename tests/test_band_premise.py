"""The premise of a two-reads-per-lane kernel on 16-row words, pinned (the full-size numbers are
profiles/r06_band_premise.json from tools/band_premise.py): against unrelated sequence the score 16 rows down hovers around 5-6
in the minimum over a wave, so at pass 1's threshold of 6 a 16-row band would have to grow at nearly every checkpoint, while
the 32-row band the kernel has stays one word at > 99.8 % of them.  A small sample of the same computation."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sixteen_row_words_cannot_hold_a_wave_at_threshold_six():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "band_premise.py"), "--reads", "1280", "--columns", "12000"],
                         capture_output=True, text=True, check=True).stdout
    r = json.loads(out)["rows"]
    assert r["16"]["wave 128, k = 6, checkpoint every 1"] < 0.2
    assert r["16"]["wave 64, k = 6, checkpoint every 4"] < 0.05
    assert r["32"]["wave 64, k = 6, checkpoint every 4"] > 0.99


def test_the_recorded_full_size_run_says_the_same():
    path = os.path.join(ROOT, "profiles", "r06_band_premise.json")
    r = json.load(open(path))["rows"]
    assert r["16"]["wave 128, k = 6, checkpoint every 1"] < 0.1 and r["32"]["wave 128, k = 6, checkpoint every 4"] > 0.99
